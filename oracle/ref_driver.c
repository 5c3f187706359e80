/* oracle/ref_driver.c - TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Plain-C entry points (ctypes-friendly: pointers, ints, doubles, C strings) around the REFERENCE's
 * own library API, compiled together with the reference sources by oracle/ref_build.py into
 * oracle/_ref/libgstref.so.  Every function here only marshals arguments into the reference's
 * types and calls the reference:
 *   gst_video_converter_new / _frame      (gst-libs/gst/video/video-converter.h:291-316)
 *   BlendFunction / FillCheckerFunction   (gst/compositor/blend.h:50-52, blend.c:2024)
 *   gst_audio_resampler_new / _resample   (gst-libs/gst/audio/audio-resampler.h:222-256)
 */
#include <gst/gst.h>
#include <gst/video/video.h>
#include <gst/audio/audio.h>
#include <string.h>
#include "blend.h"

static gsize ref_inited = 0;

int
ref_init (void)
{
  if (g_once_init_enter (&ref_inited)) {
    gst_init (NULL, NULL);
    /* make the enum GTypes known to gst_structure_from_string() */
    g_type_class_ref (gst_video_resampler_method_get_type ());
    g_type_class_ref (gst_video_dither_method_get_type ());
    g_type_class_ref (gst_video_chroma_method_get_type ());
    g_type_class_ref (gst_video_alpha_mode_get_type ());
    g_type_class_ref (gst_video_chroma_mode_get_type ());
    g_type_class_ref (gst_video_matrix_mode_get_type ());
    g_type_class_ref (gst_video_gamma_mode_get_type ());
    g_type_class_ref (gst_video_primaries_mode_get_type ());
    g_type_class_ref (gst_audio_resampler_filter_mode_get_type ());
    g_type_class_ref (gst_audio_dither_method_get_type ());
    g_type_class_ref (gst_audio_noise_shaping_method_get_type ());
    g_type_class_ref (gst_audio_resampler_filter_interpolation_get_type ());
    gst_compositor_init_blend ();
    g_once_init_leave (&ref_inited, 1);
  }
  return 0;
}

/* Fill a GstVideoInfo.  colorimetry / chroma_site may be NULL or "" (keep the defaults that
 * gst_video_info_set_format picks, video-info.c:155-225).  stride/offset may be NULL (defaults). */
static gboolean
make_info (GstVideoInfo * info, const char *format, int w, int h, const char *colorimetry,
    const char *chroma_site, const int *stride, const gsize * offset)
{
  /* go through caps exactly as the elements do (gst_video_info_from_caps, video-info.c:452-600),
   * so that absent colorimetry / chroma-site get the reference's negotiated defaults */
  GstCaps *caps;
  gboolean ok;
  int i;
  if (gst_video_format_from_string (format) == GST_VIDEO_FORMAT_UNKNOWN)
    return FALSE;
  caps = gst_caps_new_simple ("video/x-raw", "format", G_TYPE_STRING, format,
      "width", G_TYPE_INT, w, "height", G_TYPE_INT, h, "framerate", GST_TYPE_FRACTION, 30, 1, NULL);
  if (colorimetry && *colorimetry)
    gst_caps_set_simple (caps, "colorimetry", G_TYPE_STRING, colorimetry, NULL);
  if (chroma_site && *chroma_site)
    gst_caps_set_simple (caps, "chroma-site", G_TYPE_STRING, chroma_site, NULL);
  gst_video_info_init (info);
  ok = gst_video_info_from_caps (info, caps);
  gst_caps_unref (caps);
  if (!ok)
    return FALSE;
  if (stride && offset) {
    for (i = 0; i < (int) GST_VIDEO_INFO_N_PLANES (info); i++) {
      info->stride[i] = stride[i];
      info->offset[i] = offset[i];
    }
  }
  return TRUE;
}

/* Query the default layout: returns n_planes, fills stride[4]/offset[4]/size; also reports the
 * default colorimetry string and chroma-site the reference assigns. */
int
ref_video_info (const char *format, int w, int h, int *stride, gsize * offset, gsize * size,
    char *colorimetry_out, int colorimetry_len, char *chroma_out, int chroma_len)
{
  GstVideoInfo info;
  int i;
  ref_init ();
  if (!make_info (&info, format, w, h, NULL, NULL, NULL, NULL))
    return -1;
  for (i = 0; i < 4; i++) {
    stride[i] = i < (int) GST_VIDEO_INFO_N_PLANES (&info) ? info.stride[i] : 0;
    offset[i] = i < (int) GST_VIDEO_INFO_N_PLANES (&info) ? info.offset[i] : 0;
  }
  *size = info.size;
  if (colorimetry_out) {
    gchar *c = gst_video_colorimetry_to_string (&info.colorimetry);
    g_strlcpy (colorimetry_out, c ? c : "", colorimetry_len);
    g_free (c);
  }
  if (chroma_out) {
    gchar *c = gst_video_chroma_site_to_string (info.chroma_site);
    g_strlcpy (chroma_out, c ? c : "", chroma_len);
    g_free (c);
  }
  return GST_VIDEO_INFO_N_PLANES (&info);
}

/* The reference's own format table entry (video-format.c): flags (GstVideoFormatFlags), the unpack format's name, bits of the unpack
 * format (8 or 16), the chroma subsampling shifts.  For tests that split a conversion into the chain's stages. */
int
ref_video_format_props (const char *format, int *flags, char *unpack_out, int unpack_len, int *unpack_bits, int *w_sub, int *h_sub)
{
  const GstVideoFormatInfo *fi, *ui;
  GstVideoFormat f;
  ref_init ();
  f = gst_video_format_from_string (format);
  if (f == GST_VIDEO_FORMAT_UNKNOWN)
    return -1;
  fi = gst_video_format_get_info (f);
  ui = gst_video_format_get_info (fi->unpack_format);
  *flags = (int) fi->flags;
  g_strlcpy (unpack_out, ui->name, unpack_len);
  *unpack_bits = ui->depth[0];
  *w_sub = fi->n_components > 1 ? fi->w_sub[1] : 0;
  *h_sub = fi->n_components > 1 ? fi->h_sub[1] : 0;
  return 0;
}

/* The vertical (or horizontal) scaler gst_video_converter_new would make for in -> out lines under `config` (chain_vscale, video-converter.c:1653:
 * method and taps from the config, which also carries the resampler options): the first source line and the number of taps of every output
 * line (gst_video_scaler_get_coeff, video-scaler.c:300).  For tests that need the ORDER in which a scaler asks for lines. */
int
ref_video_scaler_windows (const char *config, int in_size, int out_size, int *offsets, int *n_taps)
{
  GstStructure *cfg = NULL;
  GstVideoResamplerMethod method = GST_VIDEO_RESAMPLER_METHOD_CUBIC;
  guint taps = 0, n = 0, off = 0;
  GstVideoScaler *sc;
  int j;
  ref_init ();
  if (config && *config)
    cfg = gst_structure_from_string (config, NULL);
  if (!cfg)
    cfg = gst_structure_new_empty ("GstVideoConverter");
  gst_structure_get_enum (cfg, GST_VIDEO_CONVERTER_OPT_RESAMPLER_METHOD, GST_TYPE_VIDEO_RESAMPLER_METHOD, (gint *) & method);
  gst_structure_get_uint (cfg, GST_VIDEO_CONVERTER_OPT_RESAMPLER_TAPS, &taps);
  sc = gst_video_scaler_new (method, GST_VIDEO_SCALER_FLAG_NONE, taps, in_size, out_size, cfg);
  for (j = 0; j < out_size; j++) {
    gst_video_scaler_get_coeff (sc, j, &off, &n);
    offsets[j] = (int) off;
  }
  *n_taps = (int) n;
  gst_video_scaler_free (sc);
  gst_structure_free (cfg);
  return 0;
}

static GstBuffer *
wrap_frame (GstVideoInfo * info, guint8 * data, gsize size, gboolean writable)
{
  GstBuffer *buf = gst_buffer_new_wrapped_full (writable ? 0 : GST_MEMORY_FLAG_READONLY,
      data, size, 0, size, NULL, NULL);
  gst_buffer_add_video_meta_full (buf, GST_VIDEO_FRAME_FLAG_NONE, GST_VIDEO_INFO_FORMAT (info),
      GST_VIDEO_INFO_WIDTH (info), GST_VIDEO_INFO_HEIGHT (info), GST_VIDEO_INFO_N_PLANES (info),
      info->offset, info->stride);
  return buf;
}

typedef struct
{
  GstVideoInfo in_info, out_info;
  GstVideoConverter *conv;
} RefConverter;

/* config: serialized GstStructure ("GstVideoConverter, GstVideoConverter.threads=(uint)1, ...")
 * or NULL for the library defaults (video-converter.c:778-796). */
void *ref_video_converter_new_interlaced (const char *in_format, int in_w, int in_h, const char *in_colorimetry,
    const char *in_chroma_site, const int *in_stride, const gsize * in_offset,
    const char *out_format, int out_w, int out_h, const char *out_colorimetry,
    const char *out_chroma_site, const int *out_stride, const gsize * out_offset,
    const char *config, int interlace_mode);

void *
ref_video_converter_new (const char *in_format, int in_w, int in_h, const char *in_colorimetry,
    const char *in_chroma_site, const int *in_stride, const gsize * in_offset,
    const char *out_format, int out_w, int out_h, const char *out_colorimetry,
    const char *out_chroma_site, const int *out_stride, const gsize * out_offset,
    const char *config)
{
  return ref_video_converter_new_interlaced (in_format, in_w, in_h, in_colorimetry, in_chroma_site, in_stride, in_offset,
      out_format, out_w, out_h, out_colorimetry, out_chroma_site, out_stride, out_offset, config, 0);
}

/* interlace_mode: GstVideoInterlaceMode of BOTH infos (gst_video_converter_new insists on equal modes, video-converter.c:2435); with
 * 1 = interleaved gst_video_frame_map marks every frame GST_VIDEO_FRAME_FLAG_INTERLACED (video-frame.c), which is what the converter's
 * field-aware paths look at (video_converter_generic :3303, GET_LINE_OFFSETS :3383, setup_scale :7977) */
void *
ref_video_converter_new_interlaced (const char *in_format, int in_w, int in_h, const char *in_colorimetry,
    const char *in_chroma_site, const int *in_stride, const gsize * in_offset,
    const char *out_format, int out_w, int out_h, const char *out_colorimetry,
    const char *out_chroma_site, const int *out_stride, const gsize * out_offset,
    const char *config, int interlace_mode)
{
  RefConverter *rc;
  GstStructure *s = NULL;
  ref_init ();
  rc = g_new0 (RefConverter, 1);
  if (!make_info (&rc->in_info, in_format, in_w, in_h, in_colorimetry, in_chroma_site, in_stride,
          in_offset)
      || !make_info (&rc->out_info, out_format, out_w, out_h, out_colorimetry, out_chroma_site,
          out_stride, out_offset)) {
    g_free (rc);
    return NULL;
  }
  if (config && *config) {
    s = gst_structure_from_string (config, NULL);
    if (!s) {
      g_free (rc);
      return NULL;
    }
  }
  GST_VIDEO_INFO_INTERLACE_MODE (&rc->in_info) = (GstVideoInterlaceMode) interlace_mode;
  GST_VIDEO_INFO_INTERLACE_MODE (&rc->out_info) = (GstVideoInterlaceMode) interlace_mode;
  rc->conv = gst_video_converter_new (&rc->in_info, &rc->out_info, s);
  if (!rc->conv) {
    g_free (rc);
    return NULL;
  }
  return rc;
}

int
ref_video_converter_frame (void *h, const guint8 * in_data, gsize in_size, guint8 * out_data,
    gsize out_size)
{
  RefConverter *rc = h;
  GstBuffer *ib = wrap_frame (&rc->in_info, (guint8 *) in_data, in_size, FALSE);
  GstBuffer *ob = wrap_frame (&rc->out_info, out_data, out_size, TRUE);
  GstVideoFrame inf, outf;
  int ret = -1;
  if (gst_video_frame_map (&inf, &rc->in_info, ib, GST_MAP_READ)) {
    if (gst_video_frame_map (&outf, &rc->out_info, ob, GST_MAP_WRITE)) {
      gst_video_converter_frame (rc->conv, &inf, &outf);
      gst_video_frame_unmap (&outf);
      ret = 0;
    }
    gst_video_frame_unmap (&inf);
  }
  gst_buffer_unref (ib);
  gst_buffer_unref (ob);
  return ret;
}

/* times n_frames back-to-back conversions of the same mapped frame; returns seconds */
double
ref_video_converter_bench (void *h, const guint8 * in_data, gsize in_size, guint8 * out_data,
    gsize out_size, int n_frames)
{
  RefConverter *rc = h;
  GstBuffer *ib = wrap_frame (&rc->in_info, (guint8 *) in_data, in_size, FALSE);
  GstBuffer *ob = wrap_frame (&rc->out_info, out_data, out_size, TRUE);
  GstVideoFrame inf, outf;
  double secs = -1.0;
  int i;
  if (gst_video_frame_map (&inf, &rc->in_info, ib, GST_MAP_READ)) {
    if (gst_video_frame_map (&outf, &rc->out_info, ob, GST_MAP_WRITE)) {
      gint64 t0 = g_get_monotonic_time ();
      for (i = 0; i < n_frames; i++)
        gst_video_converter_frame (rc->conv, &inf, &outf);
      secs = (g_get_monotonic_time () - t0) * 1e-6;
      gst_video_frame_unmap (&outf);
    }
    gst_video_frame_unmap (&inf);
  }
  gst_buffer_unref (ib);
  gst_buffer_unref (ob);
  return secs;
}

void
ref_video_converter_free (void *h)
{
  RefConverter *rc = h;
  if (!rc)
    return;
  gst_video_converter_free (rc->conv);
  g_free (rc);
}

/* ---- compositor ------------------------------------------------------------------------- */

static gboolean
map_simple (GstVideoFrame * f, GstVideoInfo * info, GstBuffer ** buf, const char *format, int w,
    int h, guint8 * data, gsize size, gboolean writable)
{
  if (!make_info (info, format, w, h, NULL, NULL, NULL, NULL))
    return FALSE;
  if (size < info->size)
    return FALSE;
  *buf = wrap_frame (info, data, size, writable);
  return gst_video_frame_map (f, info, *buf, writable ? GST_MAP_READWRITE : GST_MAP_READ);
}

/* func: "blend_bgra" | "blend_argb" | "overlay_bgra" | "overlay_argb" | "blend_i420" | "blend_nv12" ... */
int
ref_compositor_blend (const char *func, const char *format, const guint8 * src, gsize src_size,
    int sw, int sh, int xpos, int ypos, double alpha, guint8 * dst, gsize dst_size, int dw, int dh,
    int y0, int y1, int mode)
{
  BlendFunction fn = NULL;
  GstVideoInfo si, di;
  GstVideoFrame sf, df;
  GstBuffer *sb = NULL, *db = NULL;
  ref_init ();
  if (!strcmp (func, "blend_bgra"))
    fn = gst_compositor_blend_bgra;
  else if (!strcmp (func, "blend_argb"))
    fn = gst_compositor_blend_argb;
  else if (!strcmp (func, "overlay_bgra"))
    fn = gst_compositor_overlay_bgra;
  else if (!strcmp (func, "overlay_argb"))
    fn = gst_compositor_overlay_argb;
  else if (!strcmp (func, "blend_i420"))
    fn = gst_compositor_blend_i420;
  else if (!strcmp (func, "blend_nv12"))
    fn = gst_compositor_blend_nv12;
  else if (!strcmp (func, "blend_y444"))
    fn = gst_compositor_blend_y444;
  else if (!strcmp (func, "blend_yv12"))
    fn = gst_compositor_blend_yv12;
  else if (!strcmp (func, "blend_y42b"))
    fn = gst_compositor_blend_y42b;
  else if (!strcmp (func, "blend_nv21"))
    fn = gst_compositor_blend_nv21;
  else if (!strcmp (func, "blend_bgr"))
    fn = gst_compositor_blend_bgr;
  else if (!strcmp (func, "blend_xrgb"))
    fn = gst_compositor_blend_xrgb;
  else if (!strcmp (func, "blend_rgb"))
    fn = gst_compositor_blend_rgb;
  else if (!strcmp (func, "blend_yuy2"))        /* blend.h: YVYU and UYVY take the same function */
    fn = gst_compositor_blend_yuy2;
  else if (!strcmp (func, "blend_argb64"))
    fn = gst_compositor_blend_argb64;
  else if (!strcmp (func, "overlay_argb64"))
    fn = gst_compositor_overlay_argb64;
  /* the planar canvases of more than 8 bits (blend.c:609-681: PLANAR_YUV_BLEND with compositor_orc_blend_u10 / u12 / u16) */
  else if (!strcmp (func, "blend_i420_10le"))
    fn = gst_compositor_blend_i420_10le;
  else if (!strcmp (func, "blend_i420_12le"))
    fn = gst_compositor_blend_i420_12le;
  else if (!strcmp (func, "blend_i422_10le"))
    fn = gst_compositor_blend_i422_10le;
  else if (!strcmp (func, "blend_i422_12le"))
    fn = gst_compositor_blend_i422_12le;
  else if (!strcmp (func, "blend_y444_10le"))
    fn = gst_compositor_blend_y444_10le;
  else if (!strcmp (func, "blend_y444_12le"))
    fn = gst_compositor_blend_y444_12le;
  else if (!strcmp (func, "blend_y444_16le"))
    fn = gst_compositor_blend_y444_16le;
  if (!fn)
    return -1;
  if (!map_simple (&sf, &si, &sb, format, sw, sh, (guint8 *) src, src_size, FALSE))
    return -2;
  if (!map_simple (&df, &di, &db, format, dw, dh, dst, dst_size, TRUE))
    return -3;
  fn (&sf, xpos, ypos, alpha, &df, y0, y1, (GstCompositorBlendMode) mode);
  gst_video_frame_unmap (&sf);
  gst_video_frame_unmap (&df);
  gst_buffer_unref (sb);
  gst_buffer_unref (db);
  return 0;
}

/* kind: 0 = checker, 1 = fill_color(c1,c2,c3) ; fmt_func e.g. "bgra", "argb", "rgba", "abgr" */
int
ref_compositor_fill (int kind, const char *fmt_func, const char *format, guint8 * dst,
    gsize dst_size, int dw, int dh, int y0, int y1, int c1, int c2, int c3)
{
  GstVideoInfo di;
  GstVideoFrame df;
  GstBuffer *db = NULL;
  ref_init ();
  if (!map_simple (&df, &di, &db, format, dw, dh, dst, dst_size, TRUE))
    return -3;
  if (kind == 0) {
    FillCheckerFunction fn = NULL;
    if (!strcmp (fmt_func, "bgra"))
      fn = gst_compositor_fill_checker_bgra;
    else if (!strcmp (fmt_func, "argb"))
      fn = gst_compositor_fill_checker_argb;
    else if (!strcmp (fmt_func, "ayuv"))
      fn = gst_compositor_fill_checker_ayuv;
    else if (!strcmp (fmt_func, "i420"))
      fn = gst_compositor_fill_checker_i420;
    else if (!strcmp (fmt_func, "yv12"))
      fn = gst_compositor_fill_checker_yv12;
    else if (!strcmp (fmt_func, "y42b"))
      fn = gst_compositor_fill_checker_y42b;
    else if (!strcmp (fmt_func, "y444"))
      fn = gst_compositor_fill_checker_y444;
    else if (!strcmp (fmt_func, "nv12"))
      fn = gst_compositor_fill_checker_nv12;
    else if (!strcmp (fmt_func, "nv21"))
      fn = gst_compositor_fill_checker_nv21;
    else if (!strcmp (fmt_func, "rgb"))
      fn = gst_compositor_fill_checker_rgb;
    else if (!strcmp (fmt_func, "bgr"))
      fn = gst_compositor_fill_checker_bgr;
    else if (!strcmp (fmt_func, "vuya"))
      fn = gst_compositor_fill_checker_vuya;
    else if (!strcmp (fmt_func, "xrgb") || !strcmp (fmt_func, "xbgr"))
      fn = gst_compositor_fill_checker_xrgb;
    else if (!strcmp (fmt_func, "rgbx") || !strcmp (fmt_func, "bgrx"))
      fn = gst_compositor_fill_checker_rgbx;
    else if (!strcmp (fmt_func, "yuy2") || !strcmp (fmt_func, "yvyu"))
      fn = gst_compositor_fill_checker_yuy2;
    else if (!strcmp (fmt_func, "uyvy"))
      fn = gst_compositor_fill_checker_uyvy;
    else if (!strcmp (fmt_func, "argb64"))
      fn = gst_compositor_fill_checker_argb64;
    else if (!strcmp (fmt_func, "ayuv64"))
      fn = gst_compositor_fill_checker_ayuv64;
    else if (!strcmp (fmt_func, "i420_10le") || !strcmp (fmt_func, "i422_10le") || !strcmp (fmt_func, "y444_10le"))
      fn = gst_compositor_fill_checker_i420_10le;          /* blend.h:127-129: one function for the three layouts */
    else if (!strcmp (fmt_func, "i420_12le") || !strcmp (fmt_func, "i422_12le") || !strcmp (fmt_func, "y444_12le"))
      fn = gst_compositor_fill_checker_i420_12le;
    else if (!strcmp (fmt_func, "y444_16le"))
      fn = gst_compositor_fill_checker_y444_16le;
    if (!fn)
      return -1;
    fn (&df, y0, y1);
  } else {
    FillColorFunction fn = NULL;
    if (!strcmp (fmt_func, "bgra"))
      fn = gst_compositor_fill_color_bgra;
    else if (!strcmp (fmt_func, "argb"))
      fn = gst_compositor_fill_color_argb;
    else if (!strcmp (fmt_func, "rgba"))
      fn = gst_compositor_fill_color_rgba;
    else if (!strcmp (fmt_func, "abgr"))
      fn = gst_compositor_fill_color_abgr;
    else if (!strcmp (fmt_func, "ayuv"))
      fn = gst_compositor_fill_color_ayuv;
    else if (!strcmp (fmt_func, "i420"))
      fn = gst_compositor_fill_color_i420;
    else if (!strcmp (fmt_func, "yv12"))
      fn = gst_compositor_fill_color_yv12;
    else if (!strcmp (fmt_func, "y42b"))
      fn = gst_compositor_fill_color_y42b;
    else if (!strcmp (fmt_func, "y444"))
      fn = gst_compositor_fill_color_y444;
    else if (!strcmp (fmt_func, "nv12"))
      fn = gst_compositor_fill_color_nv12;
    else if (!strcmp (fmt_func, "nv21"))
      fn = gst_compositor_fill_color_nv12;      /* blend.h:149: the same component-addressed function */
    else if (!strcmp (fmt_func, "rgb"))
      fn = gst_compositor_fill_color_rgb;
    else if (!strcmp (fmt_func, "bgr"))
      fn = gst_compositor_fill_color_bgr;
    else if (!strcmp (fmt_func, "vuya"))
      fn = gst_compositor_fill_color_vuya;
    else if (!strcmp (fmt_func, "xrgb"))
      fn = gst_compositor_fill_color_xrgb;
    else if (!strcmp (fmt_func, "xbgr"))
      fn = gst_compositor_fill_color_xbgr;
    else if (!strcmp (fmt_func, "rgbx"))
      fn = gst_compositor_fill_color_rgbx;
    else if (!strcmp (fmt_func, "bgrx"))
      fn = gst_compositor_fill_color_bgrx;
    else if (!strcmp (fmt_func, "yuy2"))
      fn = gst_compositor_fill_color_yuy2;
    else if (!strcmp (fmt_func, "yvyu"))
      fn = gst_compositor_fill_color_yvyu;
    else if (!strcmp (fmt_func, "uyvy"))
      fn = gst_compositor_fill_color_uyvy;
    else if (!strcmp (fmt_func, "argb64"))
      fn = gst_compositor_fill_color_argb64;
    else if (!strcmp (fmt_func, "i420_10le") || !strcmp (fmt_func, "i422_10le") || !strcmp (fmt_func, "y444_10le"))
      fn = gst_compositor_fill_color_i420_10le;
    else if (!strcmp (fmt_func, "i420_12le") || !strcmp (fmt_func, "i422_12le") || !strcmp (fmt_func, "y444_12le"))
      fn = gst_compositor_fill_color_i420_12le;
    else if (!strcmp (fmt_func, "y444_16le"))
      fn = gst_compositor_fill_color_y444_16le;
    if (!fn)
      return -1;
    fn (&df, y0, y1, c1, c2, c3);
  }
  gst_video_frame_unmap (&df);
  gst_buffer_unref (db);
  return 0;
}

/* ---- audio resampler -------------------------------------------------------------------- */

/* filter-mode / filter-interpolation for the next ref_audio_resampler_new (-1: leave the option out) */
static int ref_filter_mode = -1, ref_filter_interpolation = -1;
void
ref_audio_resampler_set_filter (int filter_mode, int filter_interpolation)
{
  ref_filter_mode = filter_mode;
  ref_filter_interpolation = filter_interpolation;
}

/* method: GstAudioResamplerMethod (0 nearest,1 linear,2 cubic,3 blackman-nuttall,4 kaiser);
 * quality <0 -> no quality option (library default); options: extra serialized GstStructure or NULL */
void *
ref_audio_resampler_new (int method, int flags, const char *format, int channels, int in_rate,
    int out_rate, int quality, const char *options)
{
  GstStructure *s;
  GstAudioFormat f;
  ref_init ();
  f = gst_audio_format_from_string (format);
  if (f == GST_AUDIO_FORMAT_UNKNOWN)
    return NULL;
  if (options && *options)
    s = gst_structure_from_string (options, NULL);
  else
    s = gst_structure_new_empty ("GstAudioResampler");
  if (!s)
    return NULL;
  if (quality >= 0)
    gst_audio_resampler_options_set_quality ((GstAudioResamplerMethod) method, quality, in_rate,
        out_rate, s);
  /* typed enum options (a serialized structure would need the enum GTypes registered by name first) */
  if (ref_filter_mode >= 0)
    gst_structure_set (s, GST_AUDIO_RESAMPLER_OPT_FILTER_MODE, GST_TYPE_AUDIO_RESAMPLER_FILTER_MODE, ref_filter_mode, NULL);
  if (ref_filter_interpolation >= 0)
    gst_structure_set (s, GST_AUDIO_RESAMPLER_OPT_FILTER_INTERPOLATION, GST_TYPE_AUDIO_RESAMPLER_FILTER_INTERPOLATION,
        ref_filter_interpolation, NULL);
  {
    GstAudioResampler *r = gst_audio_resampler_new ((GstAudioResamplerMethod) method,
        (GstAudioResamplerFlags) flags, f, channels, in_rate, out_rate, s);
    gst_structure_free (s);
    return r;
  }
}

/* gst_audio_resampler_update (audio-resampler.c:1503): with_options 0 passes NULL (keep the previous options), else an
 * options structure built like ref_audio_resampler_new builds it (quality for the rates given, plus the set_filter enums) */
int
ref_audio_resampler_update (void *r, int in_rate, int out_rate, int with_options, int method, int quality,
    int q_in_rate, int q_out_rate, const char *options)
{
  GstStructure *s = NULL;
  gboolean ok;
  if (with_options) {
    if (options && *options)
      s = gst_structure_from_string (options, NULL);
    else
      s = gst_structure_new_empty ("GstAudioResampler");
    if (!s)
      return 0;
    if (quality >= 0)
      gst_audio_resampler_options_set_quality ((GstAudioResamplerMethod) method, quality, q_in_rate, q_out_rate, s);
    if (ref_filter_mode >= 0)
      gst_structure_set (s, GST_AUDIO_RESAMPLER_OPT_FILTER_MODE, GST_TYPE_AUDIO_RESAMPLER_FILTER_MODE, ref_filter_mode, NULL);
    if (ref_filter_interpolation >= 0)
      gst_structure_set (s, GST_AUDIO_RESAMPLER_OPT_FILTER_INTERPOLATION, GST_TYPE_AUDIO_RESAMPLER_FILTER_INTERPOLATION,
          ref_filter_interpolation, NULL);
  }
  ok = gst_audio_resampler_update (r, in_rate, out_rate, s);
  if (s)
    gst_structure_free (s);
  return ok;
}

gsize
ref_audio_resampler_get_out_frames (void *r, gsize in_frames)
{
  return gst_audio_resampler_get_out_frames (r, in_frames);
}

gsize
ref_audio_resampler_get_in_frames (void *r, gsize out_frames)
{
  return gst_audio_resampler_get_in_frames (r, out_frames);
}

gsize
ref_audio_resampler_get_max_latency (void *r)
{
  return gst_audio_resampler_get_max_latency (r);
}

/* interleaved in/out (flags==0): in[0]/out[0] are the single interleaved block; in may be NULL
 * (the reference then feeds silence, audio-resampler.c:1750-1806) */
void
ref_audio_resampler_resample (void *r, const void *in, gsize in_frames, void *out,
    gsize out_frames)
{
  gpointer ina[1] = { (gpointer) in };
  gpointer outa[1] = { out };
  gst_audio_resampler_resample (r, in ? ina : NULL, in_frames, outa, out_frames);
}

/* flags != 0: a non-interleaved side holds `channels` planes one after the other (in_frames / out_frames samples of
 * `bps` bytes apart); the pointer arrays gst_audio_resampler_resample wants are built here */
void
ref_audio_resampler_resample_planar (void *r, const void *in, gsize in_frames, void *out, gsize out_frames,
    int channels, int bps, int in_planar, int out_planar)
{
  gpointer ina[64], outa[64];
  int c;
  if (channels > 64)
    return;
  for (c = 0; c < channels; c++) {
    ina[c] = in_planar ? (gpointer) ((const guint8 *) in + (gsize) c * in_frames * bps) : (gpointer) in;
    outa[c] = out_planar ? (gpointer) ((guint8 *) out + (gsize) c * out_frames * bps) : out;
  }
  gst_audio_resampler_resample (r, in ? ina : NULL, in_frames, outa, out_frames);
}

void
ref_audio_resampler_reset (void *r)
{
  gst_audio_resampler_reset (r);
}

void
ref_audio_resampler_free (void *r)
{
  gst_audio_resampler_free (r);
}

/* ---- audio converter (gst-libs/gst/audio/audio-converter.h) ------------------------------------------------------------------- */
/* in / out: format string ("S16LE" ...), rate, channels, positions (GstAudioChannelPosition values, or NULL for the defaults of
 * gst_audio_info_set_format: mono, stereo, unpositioned beyond).
 * config: a GstStructure string or NULL; mix: NULL or out_ch * in_ch floats ([out][in]) for GstAudioConverter.mix-matrix */
void *
ref_audio_converter_new_positions (int flags, const char *in_fmt, int in_rate, int in_ch, const int *in_pos, const char *out_fmt, int out_rate, int out_ch,
    const int *out_pos, const char *config, const float *mix);

void *
ref_audio_converter_new (int flags, const char *in_fmt, int in_rate, int in_ch, const char *out_fmt, int out_rate, int out_ch,
    const char *config, const float *mix)
{
  return ref_audio_converter_new_positions (flags, in_fmt, in_rate, in_ch, NULL, out_fmt, out_rate, out_ch, NULL, config, mix);
}

void *
ref_audio_converter_new_positions (int flags, const char *in_fmt, int in_rate, int in_ch, const int *in_pos, const char *out_fmt, int out_rate, int out_ch,
    const int *out_pos, const char *config, const float *mix)
{
  GstAudioInfo in, out;
  GstStructure *s = NULL;
  GstAudioChannelPosition ip[64], op[64];
  int k;
  ref_init ();
  for (k = 0; k < in_ch && in_pos; k++)
    ip[k] = (GstAudioChannelPosition) in_pos[k];
  for (k = 0; k < out_ch && out_pos; k++)
    op[k] = (GstAudioChannelPosition) out_pos[k];
  /* positions are written into the info the way the audioconvert element does for layouts gst_audio_info_set_format would not take
     (several mono channels, reordered channels) */
  gst_audio_info_set_format (&in, gst_audio_format_from_string (in_fmt), in_rate, in_ch, NULL);
  gst_audio_info_set_format (&out, gst_audio_format_from_string (out_fmt), out_rate, out_ch, NULL);
  if (in_pos) {
    memcpy (in.position, ip, sizeof (ip[0]) * in_ch);
    if (ip[0] != GST_AUDIO_CHANNEL_POSITION_NONE)
      in.flags &= ~GST_AUDIO_FLAG_UNPOSITIONED;
  }
  if (out_pos) {
    memcpy (out.position, op, sizeof (op[0]) * out_ch);
    if (op[0] != GST_AUDIO_CHANNEL_POSITION_NONE)
      out.flags &= ~GST_AUDIO_FLAG_UNPOSITIONED;
  }
  if (config)
    s = gst_structure_from_string (config, NULL);
  if (mix) {
    GValue m = G_VALUE_INIT;
    int i, j;
    if (!s)
      s = gst_structure_new_empty ("GstAudioConverter");
    g_value_init (&m, GST_TYPE_ARRAY);
    for (j = 0; j < out_ch; j++) {
      GValue row = G_VALUE_INIT;
      g_value_init (&row, GST_TYPE_ARRAY);
      for (i = 0; i < in_ch; i++) {
        GValue v = G_VALUE_INIT;
        g_value_init (&v, G_TYPE_FLOAT);
        g_value_set_float (&v, mix[j * in_ch + i]);
        gst_value_array_append_and_take_value (&row, &v);
      }
      gst_value_array_append_and_take_value (&m, &row);
    }
    gst_structure_take_value (s, GST_AUDIO_CONVERTER_OPT_MIX_MATRIX, &m);
  }
  return gst_audio_converter_new ((GstAudioConverterFlags) flags, &in, &out, s);
}

gsize
ref_audio_converter_get_out_frames (void *c, gsize in_frames)
{
  return gst_audio_converter_get_out_frames (c, in_frames);
}

int
ref_audio_converter_is_passthrough (void *c)
{
  return gst_audio_converter_is_passthrough (c);
}

int
ref_audio_converter_samples (void *c, const void *in, gsize in_frames, void *out, gsize out_frames)
{
  gpointer ina[1] = { (gpointer) in }, outa[1] = { out };
  return gst_audio_converter_samples (c, 0, in ? ina : NULL, in_frames, outa, out_frames);
}

void
ref_audio_converter_reset (void *c)
{
  gst_audio_converter_reset ((GstAudioConverter *) c);
}

void
ref_audio_converter_free (void *c)
{
  gst_audio_converter_free (c);
}
