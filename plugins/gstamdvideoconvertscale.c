/* gstamdvideoconvertscale.c - `videoconvertscale` / `videoconvert` / `videoscale` elements backed by the
 * MI355X kernels (C ABI of include/gstamd_video.h).
 *
 * Mirrors the reference element's contract for this path
 * (subprojects/gst-plugins-base/gst/videoconvertscale/gstvideoconvertscale.c):
 *   - factory names and ranks            gstvideoconvertscaleplugin.c:33-53, gstvideoconvertscale.c:121
 *   - properties method / n-threads / alpha-mode / alpha-value / chroma-mode / matrix-mode / envelope /
 *     sharpness / sharpen / dither-quantization with the reference names and defaults  :130-144, 300-391
 *   - transform_caps: drop format/colorimetry/chroma-site, rangify size                :703-772
 *   - set_caps -> converter config per method                                          :985-1095
 *   - transform: one converter call per buffer                                         :1981
 * It subclasses GstBaseTransform directly (GstVideoFilter would CPU-map every buffer,
 * gstvideofilter.c:285-290).  Frames negotiated as video/x-raw(memory:AMDHIPMemory) never leave HBM between
 * chained elements; plain system-memory caps still work (upload + download inside the element), so a
 * pipeline like BASELINE config 1 negotiates unchanged.  Conversions the GPU path refuses
 * (GSTAMD_ERR_UNSUPPORTED) make set_caps fail -> not-negotiated, like the reference's "no_convert" :1111-1120.
 *
 * fixate_caps follows the reference's format scoring and display-aspect-ratio rules (:1098-1975, restated below).  Not
 * implemented: interlace-mode fields / alternate (such caps are refused; interleaved and mixed content is converted field-aware, see AMD_INTERLACE_MODES)
 * and overlay composition metas.
 */
#include <gst/base/gstbasetransform.h>
#include <gst/gst.h>
#include <gst/video/video.h>
#include <string.h>

#include "../include/gstamd_video.h"
#include "gstamdhipbufferpool.h"
#include "gstamdhipmemory.h"

GST_DEBUG_CATEGORY_STATIC (amd_vcs_debug);
GST_DEBUG_CATEGORY_STATIC (CAT_PERFORMANCE);
#define GST_CAT_DEFAULT amd_vcs_debug

#define AMD_FORMATS "{ NV12, NV21, NV16, NV61, NV24, I420, YV12, Y42B, Y444, YUY2, UYVY, YVYU, VYUY, AYUV, RGB, BGR, RGBx, BGRx, xRGB, xBGR, RGBA, BGRA, ARGB, ABGR }"
/* 10-bit formats run the reference's 16-bit chain (video_deep.h): as sources into 8-bit 4-byte destinations, as destinations from
 * every source format (widen, matrix16, u16 scalers, u16 chroma downsample, dither, pack); the combinations the library has no
 * kernel for make set_caps fail (not-negotiated) */
#define AMD_OUT_FORMATS AMD_IN_FORMATS
/* (RGBP / BGRP 1.20, RBGA / A422 / A444 / GBR_16LE / Y216_LE / Y416_LE later: taken together where the headers are the reference's own) */
#if GST_CHECK_VERSION (1, 29, 0)
#define AMD_128_FORMATS ", BGR10x2_LE, RGB10x2_LE, NV16_10LE40, RGBA_F16LE, RGBA_F16BE"
#elif GST_CHECK_VERSION (1, 28, 0)
#define AMD_128_FORMATS ", BGR10x2_LE, RGB10x2_LE, NV16_10LE40"
#else
#define AMD_128_FORMATS ""
#endif
#if GST_CHECK_VERSION (1, 26, 0)
#define AMD_NEWEST_FORMATS ", RGBP, BGRP, RBGA, A422, A444, GBR_16LE, Y216_LE, Y412_LE, Y416_LE, A420_12LE, A422_12LE, A444_12LE, A420_16LE, A422_16LE, A444_16LE, " \
    "GRAY10_LE16, I420_10BE, I422_10BE, Y444_10BE, I420_12BE, I422_12BE, Y444_12BE, Y444_16BE, P010_10BE, P012_BE, P016_BE, GBR_10BE, GBR_12BE, GBR_16BE, GBRA_10BE, GBRA_12BE, A420_10BE, A422_10BE, A444_10BE, A420_12BE, A422_12BE, A444_12BE, A420_16BE, A422_16BE, A444_16BE, Y212_BE, Y216_BE, Y412_BE, Y416_BE, AV12, NV12_16L32S, NV12_8L128, NV12_10LE40_4L4" AMD_128_FORMATS
#else
#define AMD_NEWEST_FORMATS ""
#endif
#if GST_CHECK_VERSION (1, 20, 0)
#define AMD_NEWER_FORMATS ", NV12_4L4, NV12_32L32, NV12_10LE40, VUYA, Y210, Y410, BGR10A2_LE, P012_LE, P016_LE, Y444_16LE, Y212_LE, RGB10A2_LE, ARGB64_LE, ARGB64_BE, RGBA64_LE, RGBA64_BE, BGRA64_LE, BGRA64_BE, ABGR64_LE, ABGR64_BE"
#elif GST_CHECK_VERSION (1, 18, 0)
#define AMD_NEWER_FORMATS ", NV12_10LE40, VUYA, Y210, Y410, BGR10A2_LE, P012_LE, P016_LE, Y444_16LE, Y212_LE, RGB10A2_LE"
#elif GST_CHECK_VERSION (1, 16, 0)
#define AMD_NEWER_FORMATS ", NV12_10LE40, VUYA, Y210, Y410, BGR10A2_LE"
#else
#define AMD_NEWER_FORMATS ""
#endif
#define AMD_IN_FORMATS "{ NV12, NV21, NV16, NV61, NV24, I420, YV12, Y41B, Y42B, Y444, YUY2, UYVY, YVYU, VYUY, AYUV, RGB, BGR, RGBx, BGRx, xRGB, xBGR, RGBA, BGRA, ARGB, ABGR, P010_10LE, I420_10LE, I422_10LE, Y444_10LE, I420_12LE, I422_12LE, Y444_12LE, ARGB64, AYUV64, v308, IYU2, IYU1, GRAY10_LE32, NV12_10LE32, NV16_10LE32, UYVP, NV12_64Z32, GRAY8, GRAY16_LE, GRAY16_BE, RGB16, BGR16, RGB15, BGR15, A420, A420_10LE, A422_10LE, A444_10LE, GBR, GBRA, GBR_10LE, GBR_12LE, GBRA_10LE, GBRA_12LE, v210, v216, r210" AMD_NEWER_FORMATS AMD_NEWEST_FORMATS " }"

/* progressive, interleaved and mixed content (caps without the field are progressive by definition).  An interleaved frame - every frame of
 * interlace-mode=interleaved, the buffers flagged GST_VIDEO_BUFFER_FLAG_INTERLACED of interlace-mode=mixed, which is what gst_video_frame_map makes
 * of the two modes - goes through a converter made for interlaced infos: field-aware 4:2:0 rows, video_chroma_up_vi2, the interlaced scaler
 * (video-converter.c:3303-3312, 3383, 1651; gstamd_video.h GstAmdVideoInfo::interlace_mode).  fields / alternate are not negotiated. */
/* (the templates carry no interlace-mode field, like the reference's (gstvideoconvertscale.c:100-118): a list there would leave the caps of an upstream
 * capsfilter without the field unfixed.  interlace-mode=alternate needs the caps feature format:Interlaced, which the templates do not offer; fields is
 * refused by set_caps) */
#define AMD_PROGRESSIVE ""
static GstStaticPadTemplate sink_tmpl = GST_STATIC_PAD_TEMPLATE ("sink", GST_PAD_SINK, GST_PAD_ALWAYS,
    GST_STATIC_CAPS (GST_VIDEO_CAPS_MAKE_WITH_FEATURES (GST_CAPS_FEATURE_MEMORY_AMD_HIP, AMD_IN_FORMATS) AMD_PROGRESSIVE ";"
        GST_VIDEO_CAPS_MAKE (AMD_IN_FORMATS) AMD_PROGRESSIVE));
static GstStaticPadTemplate src_tmpl = GST_STATIC_PAD_TEMPLATE ("src", GST_PAD_SRC, GST_PAD_ALWAYS,
    GST_STATIC_CAPS (GST_VIDEO_CAPS_MAKE_WITH_FEATURES (GST_CAPS_FEATURE_MEMORY_AMD_HIP, AMD_OUT_FORMATS) AMD_PROGRESSIVE ";"
        GST_VIDEO_CAPS_MAKE (AMD_OUT_FORMATS) AMD_PROGRESSIVE));

#define AMD_MAX_STREAMS 4
#define AMD_BATCH_MAX 8
#define AMD_BATCH_MAX_AGE_US 2000

/* GstVideoScaleMethod of the reference (gstvideoconvertscale.h) */
typedef enum {
  AMD_SCALE_NEAREST, AMD_SCALE_BILINEAR, AMD_SCALE_4TAP, AMD_SCALE_LANCZOS, AMD_SCALE_BILINEAR2, AMD_SCALE_SINC,
  AMD_SCALE_HERMITE, AMD_SCALE_SPLINE, AMD_SCALE_CATROM, AMD_SCALE_MITCHELL
} AmdScaleMethod;

static GType
amd_scale_method_get_type (void)
{
  static GType t = 0;
  static const GEnumValue v[] = {
    {AMD_SCALE_NEAREST, "Nearest Neighbour", "nearest-neighbour"}, {AMD_SCALE_BILINEAR, "Bilinear (2-tap)", "bilinear"},
    {AMD_SCALE_4TAP, "4-tap Sinc", "4-tap"}, {AMD_SCALE_LANCZOS, "Lanczos", "lanczos"},
    {AMD_SCALE_BILINEAR2, "Bilinear (multi-tap)", "bilinear2"}, {AMD_SCALE_SINC, "Sinc (multi-tap)", "sinc"},
    {AMD_SCALE_HERMITE, "Hermite (multi-tap)", "hermite"}, {AMD_SCALE_SPLINE, "Spline (multi-tap)", "spline"},
    {AMD_SCALE_CATROM, "Catmull-Rom (multi-tap)", "catrom"}, {AMD_SCALE_MITCHELL, "Mitchell (multi-tap)", "mitchell"},
    {0, NULL, NULL},
  };
  if (!t)
    t = g_enum_register_static ("GstAmdVideoScaleMethod", v);
  return t;
}

typedef struct {
  GstBaseTransform parent;
  /* properties (reference names) */
  gint method;
  guint n_threads;             /* accepted for compatibility; the GPU grid replaces the thread slices */
  gint alpha_mode, chroma_mode, matrix_mode, gamma_mode, primaries_mode;
  gdouble alpha_value, envelope, sharpness, sharpen;
  guint dither_quantization;
  gint chroma_resampler;       /* chroma-resampler (:137, :345): GstVideoResamplerMethod for the chroma planes of the plane scaler, default linear */
  gint dither;                 /* dither (:329): GstVideoDitherMethod, default bayer; only matters with dither-quantization > 1 here */
  GstStructure *converter_config;      /* converter-config (:378): when set, the ONLY options the converter gets (:962-967) */
  gboolean add_borders;        /* add-borders (:312): letterbox / pillarbox instead of stretching when the DAR changes */
  gint borders_w, borders_h;
  /* negotiated */
  GstVideoInfo in_info, out_info;
  gboolean in_hip, out_hip;
  GstAmdVideoConverter *convert;
  GstAmdVideoConverter *convert_i;     /* interlace-mode=mixed: the converter of the buffers flagged GST_VIDEO_BUFFER_FLAG_INTERLACED (`convert` takes the others) */
  GstBufferPool *out_pool;     /* HBM output frames are recycled through a GstAmdHipBufferPool */
  /* one ring of HIP streams per element instance (SURVEY 8b Threading): frame k runs on stream k % n_streams, so the launch
   * ramp of one frame overlaps the tail of the previous one and two elements of a process never serialise on the NULL stream;
   * buffers are ordered across streams by the events of gstamdhipmemory.h, never by a host wait */
  gint device_id;              /* device-id property: -1 = the process's current device */
  guint hip_streams;           /* hip-streams property */
  gpointer streams[AMD_MAX_STREAMS];
  guint n_streams, next_stream;
  /* device staging for system-memory pads, one per stream */
  gpointer d_in[AMD_MAX_STREAMS], d_out[AMD_MAX_STREAMS];
  gsize d_in_size[AMD_MAX_STREAMS], d_out_size[AMD_MAX_STREAMS];
  /* batch-buffers: HBM -> HBM frames of consecutive transform calls collected into ONE launch (AmdVcsBatch below) */
  gboolean pinned_pools;       /* GSTAMD_NO_PINNED_POOLS unset: offer / use page-locked host buffers at the system-memory edges */
  GstAmdHipPendingReads *reads;        /* system-memory inputs whose upload is still queued */
  guint batch_buffers;         /* property: 0 = automatic (4 when upstream is not live, 1 = no batching when it is) */
  guint batch_limit;           /* what applies to the negotiated stream */
  struct _AmdVcsBatch *batch;
  GThread *batch_watch;        /* launches a batch that has been sitting for AMD_BATCH_MAX_AGE_US without filling up */
  GstPadChainFunction base_chain;       /* GstBaseTransform's chain function (the sink pad's, before chain_list was installed) */
  /* GSTAMD_ELEMENT_STATS=1: host time spent in transform(), printed at stop (where does a buffer's CPU time go?) */
  gboolean stats;
  guint64 n_list_calls, n_list_launches;        /* buffer lists: converter calls, and the launches of them that each served a whole list */
  gint64 t_wait, t_convert, t_mark, t_total, t_prepare;
  guint64 n_frames;
} GstAmdVideoConvertScale;

typedef struct {
  GstBaseTransformClass parent_class;
  gboolean converts, scales;
} GstAmdVideoConvertScaleClass;

static void amd_vcs_batch_flush (struct _AmdVcsBatch * b, gboolean on_demand);
static void amd_vcs_batch_start (GstAmdVideoConvertScale * s);
static void amd_vcs_batch_stop (GstAmdVideoConvertScale * s);
static void amd_vcs_batch_set_converter (GstAmdVideoConvertScale * s);

enum { PROP_0, PROP_METHOD, PROP_ADD_BORDERS, PROP_N_THREADS, PROP_ALPHA_MODE, PROP_ALPHA_VALUE, PROP_CHROMA_MODE, PROP_MATRIX_MODE,
  PROP_ENVELOPE, PROP_SHARPNESS, PROP_SHARPEN, PROP_DITHER_QUANTIZATION, PROP_DEVICE_ID, PROP_HIP_STREAMS, PROP_BATCH_BUFFERS, PROP_CONVERTER_CONFIG, PROP_DITHER,
  PROP_GAMMA_MODE, PROP_PRIMARIES_MODE, PROP_CHROMA_RESAMPLER };

G_DEFINE_TYPE (GstAmdVideoConvertScale, gst_amd_vcs, GST_TYPE_BASE_TRANSFORM);
#define AMD_VCS(o) ((GstAmdVideoConvertScale *) (o))
#define AMD_VCS_GET_CLASS(o) ((GstAmdVideoConvertScaleClass *) G_OBJECT_GET_CLASS (o))

static void
amd_vcs_set_property (GObject * object, guint id, const GValue * value, GParamSpec * pspec)
{
  GstAmdVideoConvertScale *s = AMD_VCS (object);
  gboolean reconfigure = FALSE;
  GST_OBJECT_LOCK (s);
  switch (id) {
    case PROP_METHOD: s->method = g_value_get_enum (value); break;
    case PROP_ADD_BORDERS: s->add_borders = g_value_get_boolean (value); break;
    case PROP_N_THREADS: s->n_threads = g_value_get_uint (value); break;
    case PROP_ALPHA_MODE: s->alpha_mode = g_value_get_enum (value); break;
    case PROP_ALPHA_VALUE: s->alpha_value = g_value_get_double (value); break;
    case PROP_CHROMA_MODE: s->chroma_mode = g_value_get_enum (value); break;
    case PROP_MATRIX_MODE: s->matrix_mode = g_value_get_enum (value); break;
    case PROP_GAMMA_MODE: s->gamma_mode = g_value_get_enum (value); break;
    case PROP_PRIMARIES_MODE: s->primaries_mode = g_value_get_enum (value); break;
    case PROP_ENVELOPE: s->envelope = g_value_get_double (value); break;
    case PROP_SHARPNESS: s->sharpness = g_value_get_double (value); break;
    case PROP_SHARPEN: s->sharpen = g_value_get_double (value); break;
    case PROP_DITHER_QUANTIZATION: s->dither_quantization = g_value_get_uint (value); break;
    case PROP_DITHER: s->dither = g_value_get_enum (value); break;
    case PROP_CHROMA_RESAMPLER: s->chroma_resampler = g_value_get_enum (value); break;
    case PROP_DEVICE_ID: s->device_id = g_value_get_int (value); break;
    case PROP_HIP_STREAMS: s->hip_streams = g_value_get_uint (value); break;
    case PROP_BATCH_BUFFERS: s->batch_buffers = g_value_get_uint (value); break;
    case PROP_CONVERTER_CONFIG:
      if (s->converter_config)
        gst_structure_free (s->converter_config);
      s->converter_config = g_value_dup_boxed (value);
      reconfigure = TRUE;
      break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID (object, id, pspec); break;
  }
  GST_OBJECT_UNLOCK (s);
  /* the reference applies a changed converter-config to the running converter before the next frame (:1989-2000); here the next
   * buffer renegotiates, which builds the converter anew from the structure */
  if (reconfigure)
    gst_base_transform_reconfigure_src (GST_BASE_TRANSFORM (s));
}

static void
amd_vcs_get_property (GObject * object, guint id, GValue * value, GParamSpec * pspec)
{
  GstAmdVideoConvertScale *s = AMD_VCS (object);
  GST_OBJECT_LOCK (s);
  switch (id) {
    case PROP_METHOD: g_value_set_enum (value, s->method); break;
    case PROP_ADD_BORDERS: g_value_set_boolean (value, s->add_borders); break;
    case PROP_N_THREADS: g_value_set_uint (value, s->n_threads); break;
    case PROP_ALPHA_MODE: g_value_set_enum (value, s->alpha_mode); break;
    case PROP_ALPHA_VALUE: g_value_set_double (value, s->alpha_value); break;
    case PROP_CHROMA_MODE: g_value_set_enum (value, s->chroma_mode); break;
    case PROP_MATRIX_MODE: g_value_set_enum (value, s->matrix_mode); break;
    case PROP_GAMMA_MODE: g_value_set_enum (value, s->gamma_mode); break;
    case PROP_PRIMARIES_MODE: g_value_set_enum (value, s->primaries_mode); break;
    case PROP_ENVELOPE: g_value_set_double (value, s->envelope); break;
    case PROP_SHARPNESS: g_value_set_double (value, s->sharpness); break;
    case PROP_SHARPEN: g_value_set_double (value, s->sharpen); break;
    case PROP_DITHER_QUANTIZATION: g_value_set_uint (value, s->dither_quantization); break;
    case PROP_DITHER: g_value_set_enum (value, s->dither); break;
    case PROP_CHROMA_RESAMPLER: g_value_set_enum (value, s->chroma_resampler); break;
    case PROP_DEVICE_ID: g_value_set_int (value, s->device_id); break;
    case PROP_HIP_STREAMS: g_value_set_uint (value, s->hip_streams); break;
    case PROP_BATCH_BUFFERS: g_value_set_uint (value, s->batch_buffers); break;
    case PROP_CONVERTER_CONFIG: g_value_set_boxed (value, s->converter_config); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID (object, id, pspec); break;
  }
  GST_OBJECT_UNLOCK (s);
}

static void
amd_vcs_finalize (GObject * object)
{
  GstAmdVideoConvertScale *s = AMD_VCS (object);
  if (s->converter_config)
    gst_structure_free (s->converter_config);
  s->converter_config = NULL;
  G_OBJECT_CLASS (gst_amd_vcs_parent_class)->finalize (object);
}

/* One GstVideoConverter / GstVideoResampler option of a converter-config structure as a number: enums, ints, uints, doubles and
 * booleans as the reference's get_opt_* read them (video-converter.c:2163-2210) */
static gboolean
amd_cfg_number (const GstStructure * st, const gchar * key, gdouble * out)
{
  const GValue *v = gst_structure_get_value (st, key);
  if (!v)
    return FALSE;
  if (G_VALUE_HOLDS_ENUM (v))
    *out = g_value_get_enum (v);
  else if (G_VALUE_HOLDS_INT (v))
    *out = g_value_get_int (v);
  else if (G_VALUE_HOLDS_UINT (v))
    *out = g_value_get_uint (v);
  else if (G_VALUE_HOLDS_DOUBLE (v))
    *out = g_value_get_double (v);
  else if (G_VALUE_HOLDS_BOOLEAN (v))
    *out = g_value_get_boolean (v);
  else if (G_VALUE_HOLDS_STRING (v) && g_value_get_string (v)) {
    /* gst-launch hands an untyped field over as a string: a number, or for the enum options the value's nick ("lanczos").  Anything
     * else is a wrongly typed field - the reference ignores such a field and keeps its default, so do we (with a warning) */
    const gchar *txt = g_value_get_string (v);
    gchar *end = NULL;
    const gdouble num = g_ascii_strtod (txt, &end);
    if (end != txt && *end == '\0') {
      *out = num;
      return TRUE;
    }
    {
      static const struct { const gchar *key; GType (*type) (void); } enums[] = {
        {"GstVideoConverter.resampler-method", gst_video_resampler_method_get_type},
        {"GstVideoConverter.chroma-resampler-method", gst_video_resampler_method_get_type},
        {"GstVideoConverter.dither-method", gst_video_dither_method_get_type}, {"GstVideoConverter.alpha-mode", gst_video_alpha_mode_get_type},
        {"GstVideoConverter.chroma-mode", gst_video_chroma_mode_get_type}, {"GstVideoConverter.matrix-mode", gst_video_matrix_mode_get_type},
        {"GstVideoConverter.gamma-mode", gst_video_gamma_mode_get_type}, {"GstVideoConverter.primaries-mode", gst_video_primaries_mode_get_type},
      };
      guint i;
      for (i = 0; i < G_N_ELEMENTS (enums); i++)
        if (g_strcmp0 (key, enums[i].key) == 0) {
          GEnumClass *klass = g_type_class_ref (enums[i].type ());
          const GEnumValue *ev = g_enum_get_value_by_nick (klass, txt);
          if (!ev)
            ev = g_enum_get_value_by_name (klass, txt);
          if (ev)
            *out = ev->value;
          g_type_class_unref (klass);
          if (ev)
            return TRUE;
        }
    }
    GST_WARNING ("converter-config: field %s has the value \"%s\", which is neither a number nor a value of the option's type - ignored", key, txt);
    return FALSE;
  } else
    return FALSE;
  return TRUE;
}

/* A converter-config written on a command line names its enum fields by type - GstVideoConverter.resampler-method=(GstVideoResamplerMethod)lanczos -
 * and gst_structure_from_string only finds types that are registered by then: the elements that take such a structure register them
 * with their class */
void
gst_amd_converter_config_register_types (void)
{
  g_type_class_unref (g_type_class_ref (gst_video_resampler_method_get_type ()));
  g_type_class_unref (g_type_class_ref (gst_video_dither_method_get_type ()));
  g_type_class_unref (g_type_class_ref (gst_video_alpha_mode_get_type ()));
  g_type_class_unref (g_type_class_ref (gst_video_chroma_mode_get_type ()));
  g_type_class_unref (g_type_class_ref (gst_video_matrix_mode_get_type ()));
  g_type_class_unref (g_type_class_ref (gst_video_gamma_mode_get_type ()));
  g_type_class_unref (g_type_class_ref (gst_video_primaries_mode_get_type ()));
}

/* converter-config -> the C ABI's config: library defaults for whatever the structure does not name, exactly what
 * gst_video_converter_new does with it */
void
gst_amd_converter_config_from_structure (const GstStructure * st, GstAmdVideoConverterConfig * cfg)
{
  gdouble d;
#define OPT_I(key, field) if (amd_cfg_number (st, key, &d)) cfg->field = (gint) d
#define OPT_U(key, field) if (amd_cfg_number (st, key, &d)) cfg->field = (guint) d
#define OPT_D(key, field) if (amd_cfg_number (st, key, &d)) cfg->field = d
  OPT_I ("GstVideoConverter.resampler-method", resampler_method);
  OPT_I ("GstVideoConverter.chroma-resampler-method", chroma_resampler_method);
  OPT_U ("GstVideoConverter.resampler-taps", resampler_taps);
  OPT_U ("GstVideoConverter.dither-quantization", dither_quantization);
  OPT_I ("GstVideoConverter.dither-method", dither_method);
  OPT_I ("GstVideoConverter.src-x", src_x);
  OPT_I ("GstVideoConverter.src-y", src_y);
  OPT_I ("GstVideoConverter.src-width", src_width);
  OPT_I ("GstVideoConverter.src-height", src_height);
  OPT_I ("GstVideoConverter.dest-x", dest_x);
  OPT_I ("GstVideoConverter.dest-y", dest_y);
  OPT_I ("GstVideoConverter.dest-width", dest_width);
  OPT_I ("GstVideoConverter.dest-height", dest_height);
  OPT_I ("GstVideoConverter.fill-border", fill_border);
  OPT_U ("GstVideoConverter.border-argb", border_argb);
  OPT_D ("GstVideoConverter.alpha-value", alpha_value);
  OPT_I ("GstVideoConverter.alpha-mode", alpha_mode);
  OPT_I ("GstVideoConverter.chroma-mode", chroma_mode);
  OPT_I ("GstVideoConverter.matrix-mode", matrix_mode);
  OPT_I ("GstVideoConverter.gamma-mode", gamma_mode);
  OPT_I ("GstVideoConverter.primaries-mode", primaries_mode);
  OPT_I ("GstVideoResampler.max-taps", max_taps);
  OPT_D ("GstVideoResampler.cubic-b", cubic_b);
  OPT_D ("GstVideoResampler.cubic-c", cubic_c);
  OPT_D ("GstVideoResampler.envelope", envelope);
  OPT_D ("GstVideoResampler.sharpness", sharpness);
  OPT_D ("GstVideoResampler.sharpen", sharpen);
#undef OPT_I
#undef OPT_U
#undef OPT_D
}

static gboolean
features_convertible (const GstCapsFeatures * f)
{
  if (gst_caps_features_is_any (f))
    return FALSE;
  return gst_caps_features_is_equal (f, GST_CAPS_FEATURES_MEMORY_SYSTEM_MEMORY) ||
      (gst_caps_features_get_size (f) == 1 && gst_caps_features_contains (f, GST_CAPS_FEATURE_MEMORY_AMD_HIP));
}

/* gst_video_convert_caps_remove_format_and_rangify_size_info (:703-748); additionally both memory kinds are
 * offered on the other side, since the element can upload/download itself */
static GstCaps *
amd_vcs_transform_caps (GstBaseTransform * trans, GstPadDirection direction, GstCaps * caps, GstCaps * filter)
{
  GstAmdVideoConvertScaleClass *klass = AMD_VCS_GET_CLASS (trans);
  GstCaps *ret = gst_caps_new_empty ();
  guint i, n = gst_caps_get_size (caps);

  for (i = 0; i < n; i++) {
    GstStructure *st = gst_structure_copy (gst_caps_get_structure (caps, i));
    GstCapsFeatures *f = gst_caps_get_features (caps, i);

    if (features_convertible (f)) {
      if (klass->scales) {
        gst_structure_set (st, "width", GST_TYPE_INT_RANGE, 1, G_MAXINT, "height", GST_TYPE_INT_RANGE, 1, G_MAXINT, NULL);
        if (gst_structure_has_field (st, "pixel-aspect-ratio"))
          gst_structure_set (st, "pixel-aspect-ratio", GST_TYPE_FRACTION_RANGE, 1, G_MAXINT, G_MAXINT, 1, NULL);
      }
      if (klass->converts)
        gst_structure_remove_fields (st, "format", "colorimetry", "chroma-site", NULL);
      gst_caps_append_structure_full (ret, gst_structure_copy (st), gst_caps_features_new (GST_CAPS_FEATURE_MEMORY_AMD_HIP, NULL));
      gst_caps_append_structure_full (ret, st, gst_caps_features_new (GST_CAPS_FEATURE_MEMORY_SYSTEM_MEMORY, NULL));
    } else {
      gst_caps_append_structure_full (ret, st, gst_caps_features_copy (f));
    }
  }
  {
    /* restrict to what the pad template on the other side allows */
    GstPad *other = direction == GST_PAD_SINK ? GST_BASE_TRANSFORM_SRC_PAD (trans) : GST_BASE_TRANSFORM_SINK_PAD (trans);
    GstCaps *tmpl = gst_pad_get_pad_template_caps (other);
    GstCaps *tmp = gst_caps_intersect_full (ret, tmpl, GST_CAPS_INTERSECT_FIRST);
    gst_caps_unref (ret);
    gst_caps_unref (tmpl);
    ret = tmp;
  }
  if (filter) {
    GstCaps *tmp = gst_caps_intersect_full (filter, ret, GST_CAPS_INTERSECT_FIRST);
    gst_caps_unref (ret);
    ret = tmp;
  }
  return ret;
}

/* ---- fixate_caps: the reference's decisions (gstvideoconvertscale.c:1098-1975), restated -------------------------------------
 * format:  the output format that loses least against the input (score table of :1098-1246: a change costs 1, a LOSS of
 *          colourspace 2, depth 4, alpha 8, chroma width 16, chroma height 32, palette 64, colour 128), then colorimetry and
 *          chroma-site carried over from the input where they still mean the same (:1330-1425);
 * size:    keep the display aspect ratio DAR = w/h * PAR through whatever is still free on the other side - both sizes fixed: only
 *          the PAR can follow; one size fixed: the other one follows, through the PAR if that is free; nothing fixed: keep the
 *          input size and let the PAR absorb the change, else scale one dimension (:1488-1930). */
static gint
format_loss (const GstVideoFormatInfo * in, const GstVideoFormatInfo * t)
{
  const guint ignore = GST_VIDEO_FORMAT_FLAG_LE | GST_VIDEO_FORMAT_FLAG_COMPLEX | GST_VIDEO_FORMAT_FLAG_UNPACK;
  const guint cs = GST_VIDEO_FORMAT_FLAG_YUV | GST_VIDEO_FORMAT_FLAG_RGB | GST_VIDEO_FORMAT_FLAG_GRAY;
  const guint fi = GST_VIDEO_FORMAT_INFO_FLAGS (in) & ~ignore, ft = GST_VIDEO_FORMAT_INFO_FLAGS (t) & ~ignore;
  gint loss = 1;                /* any change of format */

  if (in == t)
    return 0;
  if ((ft ^ fi) & GST_VIDEO_FORMAT_FLAG_PALETTE)
    loss += 1 + ((ft & GST_VIDEO_FORMAT_FLAG_PALETTE) ? 64 : 0);
  if ((ft & cs) != (fi & cs))
    loss += 2 + ((ft & GST_VIDEO_FORMAT_FLAG_GRAY) ? 128 : 0);
  if ((ft ^ fi) & GST_VIDEO_FORMAT_FLAG_ALPHA)
    loss += 1 + ((fi & GST_VIDEO_FORMAT_FLAG_ALPHA) ? 8 : 0);
  if (in->h_sub[1] != t->h_sub[1])
    loss += 1 + (in->h_sub[1] < t->h_sub[1] ? 32 : 0);
  if (in->w_sub[1] != t->w_sub[1])
    loss += 1 + (in->w_sub[1] < t->w_sub[1] ? 16 : 0);
  if (in->bits != t->bits)
    loss += 1 + (in->bits > t->bits ? 4 : 0);
  return loss;
}

static void
consider_format (const GstVideoFormatInfo * in, const GValue * v, gint * best_loss, const GstVideoFormatInfo ** best)
{
  const GstVideoFormatInfo *t;
  gint loss;

  if (!G_VALUE_HOLDS_STRING (v) || *best_loss == 0)
    return;
  t = gst_video_format_get_info (gst_video_format_from_string (g_value_get_string (v)));
  if (!t || GST_VIDEO_FORMAT_INFO_FORMAT (t) == GST_VIDEO_FORMAT_UNKNOWN)
    return;
  loss = format_loss (in, t);
  if (loss < *best_loss) {
    *best_loss = loss;
    *best = t;
  }
}

/* every structure of `result` loses its size fields; the first one gets the least lossy format found anywhere in `result` */
static void
pick_format (GstCaps * caps, GstCaps * result)
{
  const gchar *name = gst_structure_get_string (gst_caps_get_structure (caps, 0), "format");
  const GstVideoFormatInfo *in = name ? gst_video_format_get_info (gst_video_format_from_string (name)) : NULL, *best = NULL;
  gint best_loss = G_MAXINT;
  guint i, j;

  if (!in)
    return;
  for (i = 0; i < gst_caps_get_size (result); i++) {
    GstStructure *st = gst_caps_get_structure (result, i);
    const GValue *f = gst_structure_get_value (st, "format");
    gst_structure_remove_fields (st, "height", "width", "pixel-aspect-ratio", "display-aspect-ratio", NULL);
    if (!f)
      continue;
    if (GST_VALUE_HOLDS_LIST (f))
      for (j = 0; j < gst_value_list_get_size (f); j++)
        consider_format (in, gst_value_list_get_value (f, j), &best_loss, &best);
    else
      consider_format (in, f, &best_loss, &best);
  }
  if (best)
    gst_structure_set (gst_caps_get_structure (result, 0), "format", G_TYPE_STRING, GST_VIDEO_FORMAT_INFO_NAME (best), NULL);
}

/* output caps without colorimetry / chroma-site take the input's where that is meaningful: the colorimetry whole inside one colour
 * model, primaries + transfer only across RGB <-> YUV; the chroma siting only between YUV formats of equal subsampling */
static void
inherit_colorimetry (GstCaps * in_caps, GstCaps * out_caps)
{
  GstStructure *os = gst_caps_get_structure (out_caps, 0), *is = gst_caps_get_structure (in_caps, 0), *ts;
  const gboolean have_col = gst_structure_has_field (os, "colorimetry"), have_site = gst_structure_has_field (os, "chroma-site");
  const GValue *in_col = gst_structure_get_value (is, "colorimetry"), *in_site = gst_structure_get_value (is, "chroma-site");
  GstVideoInfo ii, oi;
  GstCaps *probe;
  guint c;

  if ((have_col && have_site) || !gst_video_info_from_caps (&ii, in_caps))
    return;
  /* the output size may still be open here: probe the colour model on a fixated copy that borrows the input size */
  probe = gst_caps_fixate (gst_caps_copy (out_caps));
  ts = gst_caps_get_structure (probe, 0);
  if (!gst_structure_has_field (ts, "width"))
    gst_structure_set_value (ts, "width", gst_structure_get_value (is, "width"));
  if (!gst_structure_has_field (ts, "height"))
    gst_structure_set_value (ts, "height", gst_structure_get_value (is, "height"));
  if (!gst_video_info_from_caps (&oi, probe)) {
    gst_caps_unref (probe);
    return;
  }
  gst_caps_unref (probe);
  if (!have_col && in_col) {
    if ((GST_VIDEO_INFO_IS_YUV (&oi) && GST_VIDEO_INFO_IS_YUV (&ii)) || (GST_VIDEO_INFO_IS_RGB (&oi) && GST_VIDEO_INFO_IS_RGB (&ii)) ||
        (GST_VIDEO_INFO_IS_GRAY (&oi) && GST_VIDEO_INFO_IS_GRAY (&ii))) {
      gst_structure_set_value (os, "colorimetry", in_col);
    } else {
      gchar *str;
      oi.colorimetry.primaries = ii.colorimetry.primaries;
      oi.colorimetry.transfer = ii.colorimetry.transfer;
      str = gst_video_colorimetry_to_string (&oi.colorimetry);
      if (str)
        gst_caps_set_simple (out_caps, "colorimetry", G_TYPE_STRING, str, NULL);
      g_free (str);
    }
  }
  if (!have_site && in_site && GST_VIDEO_INFO_IS_YUV (&oi) && GST_VIDEO_INFO_IS_YUV (&ii) &&
      GST_VIDEO_INFO_N_COMPONENTS (&ii) == GST_VIDEO_INFO_N_COMPONENTS (&oi)) {
    gboolean same = TRUE;
    for (c = 0; c < GST_VIDEO_INFO_N_COMPONENTS (&ii); c++)
      same = same && GST_VIDEO_FORMAT_INFO_W_SUB (ii.finfo, c) == GST_VIDEO_FORMAT_INFO_W_SUB (oi.finfo, c) &&
          GST_VIDEO_FORMAT_INFO_H_SUB (ii.finfo, c) == GST_VIDEO_FORMAT_INFO_H_SUB (oi.finfo, c);
    if (same)
      gst_structure_set_value (os, "chroma-site", in_site);
  }
}

static GstCaps *
fixed_format_caps (GstPadDirection direction, GstCaps * caps, GstCaps * othercaps)
{
  GstCaps *r = gst_caps_intersect (othercaps, caps);

  if (gst_caps_is_empty (r)) {
    gst_caps_unref (r);
    r = gst_caps_copy (othercaps);
  }
  r = gst_caps_make_writable (r);
  pick_format (caps, r);
  r = gst_caps_fixate (r);
  if (direction == GST_PAD_SINK) {
    if (gst_caps_is_subset (caps, r))
      gst_caps_replace (&r, caps);
    else
      inherit_colorimetry (caps, r);
  }
  return r;
}

/* a * b as a reduced fraction; doubles when the integers overflow */
static void
frac_mul (gint an, gint ad, gint bn, gint bd, gint * rn, gint * rd)
{
  if (!gst_util_fraction_multiply (an, ad, bn, bd, rn, rd)) {
    gdouble x, y;
    gst_util_fraction_to_double (an, ad, &x);
    gst_util_fraction_to_double (bn, bd, &y);
    gst_util_double_to_fraction (x * y, rn, rd);
  }
}

/* what field `name` of `st` would become when fixated towards `target` (on a copy) */
static gint
nearest_int (const GstStructure * st, const gchar * name, gint target)
{
  GstStructure *t = gst_structure_copy (st);
  gint v = target;
  gst_structure_fixate_field_nearest_int (t, name, target);
  gst_structure_get_int (t, name, &v);
  gst_structure_free (t);
  return v;
}

/* the PAR the other side would accept closest to n / d (its PAR field, or `to_par` when the field is absent) */
static void
nearest_par (const GstStructure * st, const GValue * to_par, gint n, gint d, gint * set_n, gint * set_d)
{
  GstStructure *t = gst_structure_copy (st);
  if (!gst_structure_has_field (t, "pixel-aspect-ratio"))
    gst_structure_set_value (t, "pixel-aspect-ratio", to_par);
  gst_structure_fixate_field_nearest_fraction (t, "pixel-aspect-ratio", n, d);
  *set_n = n;
  *set_d = d;
  gst_structure_get_fraction (t, "pixel-aspect-ratio", set_n, set_d);
  gst_structure_free (t);
}

static void
put_par (GstStructure * outs, gint n, gint d)
{
  if (gst_structure_has_field (outs, "pixel-aspect-ratio") || n != d)
    gst_structure_set (outs, "pixel-aspect-ratio", GST_TYPE_FRACTION, n, d, NULL);
}

static GstCaps *
fixate_size (GstBaseTransform * base, GstPadDirection direction, GstCaps * caps, GstCaps * othercaps)
{
  GstStructure *ins, *outs;
  const GValue *from_par, *to_par;
  GValue fpar = G_VALUE_INIT, tpar = G_VALUE_INIT;
  gint from_w = 0, from_h = 0, w = 0, h = 0, from_pn = 1, from_pd = 1, dar_n, dar_d;

  othercaps = gst_caps_make_writable (gst_caps_truncate (othercaps));
  ins = gst_caps_get_structure (caps, 0);
  outs = gst_caps_get_structure (othercaps, 0);
  from_par = gst_structure_get_value (ins, "pixel-aspect-ratio");
  to_par = gst_structure_get_value (outs, "pixel-aspect-ratio");
  /* a missing PAR means 1/1 on the side we come from; on the other side it means "anything" when fixating from the sink pad and
   * 1/1 (which is then written into the caps) when fixating from the src pad */
  if (!from_par) {
    g_value_init (&fpar, GST_TYPE_FRACTION);
    gst_value_set_fraction (&fpar, 1, 1);
    from_par = &fpar;
  }
  if (!to_par) {
    if (direction == GST_PAD_SINK) {
      g_value_init (&tpar, GST_TYPE_FRACTION_RANGE);
      gst_value_set_fraction_range_full (&tpar, 1, G_MAXINT, G_MAXINT, 1);
    } else {
      g_value_init (&tpar, GST_TYPE_FRACTION);
      gst_value_set_fraction (&tpar, 1, 1);
      gst_structure_set (outs, "pixel-aspect-ratio", GST_TYPE_FRACTION, 1, 1, NULL);
    }
    to_par = &tpar;
  }
  if (!gst_value_is_fixed (from_par))
    goto done;
  from_pn = gst_value_get_fraction_numerator (from_par);
  from_pd = gst_value_get_fraction_denominator (from_par);
  gst_structure_get_int (ins, "width", &from_w);
  gst_structure_get_int (ins, "height", &from_h);
  gst_structure_get_int (outs, "width", &w);
  gst_structure_get_int (outs, "height", &h);

  if (w && h) {
    /* both sizes given: only the PAR can still follow the display ratio */
    guint n, d;
    if (!gst_value_is_fixed (to_par) && gst_video_calculate_display_ratio (&n, &d, from_w, from_h, from_pn, from_pd, w, h)) {
      if (gst_structure_has_field (outs, "pixel-aspect-ratio"))
        gst_structure_fixate_field_nearest_fraction (outs, "pixel-aspect-ratio", n, d);
      else if (n != d)
        gst_structure_set (outs, "pixel-aspect-ratio", GST_TYPE_FRACTION, n, d, NULL);
    }
    goto done;
  }
  frac_mul (from_w, from_h, from_pn, from_pd, &dar_n, &dar_d);

  if (h || w) {
    /* one dimension given (`fixed`), the other (`free_name`) follows the DAR; written for "height given", mirrored for "width given" */
    const gboolean h_fixed = h != 0;
    const gchar *free_name = h_fixed ? "width" : "height";
    const gint fixed = h_fixed ? h : w, from_free = h_fixed ? from_w : from_h;
    gint num, den, set_free, want_pn, want_pd, set_pn, set_pd;
    guint64 t;

    if (gst_value_is_fixed (to_par)) {
      frac_mul (dar_n, dar_d, gst_value_get_fraction_denominator (to_par), gst_value_get_fraction_numerator (to_par), &num, &den);
      t = h_fixed ? gst_util_uint64_scale_int_round (fixed, num, den) : gst_util_uint64_scale_int_round (fixed, den, num);
      if (t > G_MAXINT) {
        GST_ELEMENT_ERROR (base, CORE, NEGOTIATION, (NULL), ("Error calculating the output scaled size - integer overflow"));
        goto done;
      }
      gst_structure_fixate_field_nearest_int (outs, free_name, (gint) t);
      goto done;
    }
    /* free PAR: keep the input's size in the open dimension and let the PAR carry the DAR ... */
    set_free = nearest_int (outs, free_name, from_free);
    if (h_fixed)
      frac_mul (dar_n, dar_d, fixed, set_free, &want_pn, &want_pd);
    else
      frac_mul (dar_n, dar_d, set_free, fixed, &want_pn, &want_pd);
    nearest_par (outs, to_par, want_pn, want_pd, &set_pn, &set_pd);
    if (set_pn == want_pn && set_pd == want_pd) {
      if (gst_structure_has_field (outs, "pixel-aspect-ratio") || set_pn != set_pd)
        gst_structure_set (outs, free_name, G_TYPE_INT, set_free, "pixel-aspect-ratio", GST_TYPE_FRACTION, set_pn, set_pd, NULL);
      goto done;
    }
    /* ... or, with the PAR the other side does accept, scale the open dimension */
    frac_mul (dar_n, dar_d, set_pd, set_pn, &num, &den);
    t = h_fixed ? gst_util_uint64_scale_int_round (fixed, num, den) : gst_util_uint64_scale_int_round (fixed, den, num);
    if (t > G_MAXINT) {
      GST_ELEMENT_ERROR (base, CORE, NEGOTIATION, (NULL), ("Error calculating the output scaled size - integer overflow"));
      goto done;
    }
    gst_structure_fixate_field_nearest_int (outs, free_name, (gint) t);
    put_par (outs, set_pn, set_pd);
    goto done;
  }

  if (gst_value_is_fixed (to_par)) {
    /* sizes open, PAR given: keep the height (interlacing) and scale the width; failing that keep the width; failing that the
     * pair whose DAR is closest */
    gint num, den, set_h, set_w, alt_w, alt_h;
    guint64 tw, th;

    frac_mul (dar_n, dar_d, gst_value_get_fraction_denominator (to_par), gst_value_get_fraction_numerator (to_par), &num, &den);
    set_h = nearest_int (outs, "height", from_h);
    tw = gst_util_uint64_scale_int_round (set_h, num, den);
    if (tw > G_MAXINT) {
      GST_ELEMENT_ERROR (base, CORE, NEGOTIATION, (NULL), ("Error calculating the output scaled size - integer overflow"));
      goto done;
    }
    set_w = nearest_int (outs, "width", (gint) tw);
    if (set_w == (gint) tw) {
      gst_structure_set (outs, "width", G_TYPE_INT, set_w, "height", G_TYPE_INT, set_h, NULL);
      goto done;
    }
    alt_w = nearest_int (outs, "width", from_w);
    th = gst_util_uint64_scale_int_round (alt_w, den, num);
    if (th > G_MAXINT) {
      GST_ELEMENT_ERROR (base, CORE, NEGOTIATION, (NULL), ("Error calculating the output scaled size - integer overflow"));
      goto done;
    }
    alt_h = nearest_int (outs, "height", (gint) th);
    if (alt_h == (gint) th) {
      gst_structure_set (outs, "width", G_TYPE_INT, alt_w, "height", G_TYPE_INT, alt_h, NULL);
      goto done;
    }
    if ((gint64) alt_w * ABS (alt_h - (gint) th) < (gint64) ABS (set_w - (gint) tw) * set_h) {
      set_w = alt_w;
      set_h = alt_h;
    }
    gst_structure_set (outs, "width", G_TYPE_INT, set_w, "height", G_TYPE_INT, set_h, NULL);
    goto done;
  }

  {
    /* everything open (and passthrough impossible): input size + a PAR that keeps the DAR; else one scaled dimension under the PAR
     * that is accepted; else the first try as it is */
    gint set_h = nearest_int (outs, "height", from_h), set_w = nearest_int (outs, "width", from_w);
    gint want_pn, want_pd, set_pn, set_pd, num, den, got;
    guint64 t;

    frac_mul (dar_n, dar_d, set_h, set_w, &want_pn, &want_pd);
    nearest_par (outs, to_par, want_pn, want_pd, &set_pn, &set_pd);
    if (!(set_pn == want_pn && set_pd == want_pd)) {
      frac_mul (dar_n, dar_d, set_pd, set_pn, &num, &den);
      t = gst_util_uint64_scale_round (set_h, num, den);
      got = nearest_int (outs, "width", (gint) t);
      if (got == (gint) t) {
        set_w = got;
      } else {
        t = gst_util_uint64_scale_round (set_w, den, num);
        got = nearest_int (outs, "height", (gint) t);
        if (got == (gint) t)
          set_h = got;
      }
    }
    gst_structure_set (outs, "width", G_TYPE_INT, set_w, "height", G_TYPE_INT, set_h, NULL);
    put_par (outs, set_pn, set_pd);
  }

done:
  othercaps = gst_caps_fixate (othercaps);
  if (from_par == &fpar)
    g_value_unset (&fpar);
  if (to_par == &tpar)
    g_value_unset (&tpar);
  return othercaps;
}

static GstCaps *
amd_vcs_fixate_caps (GstBaseTransform * trans, GstPadDirection direction, GstCaps * caps, GstCaps * othercaps)
{
  static const gchar *fields[] = { "format", "colorimetry", "chroma-site" };
  GstCaps *format = fixed_format_caps (direction, caps, othercaps);
  guint i;

  if (gst_caps_is_empty (format)) {
    GST_ERROR_OBJECT (trans, "Could not convert formats");
    gst_caps_unref (othercaps);
    return format;
  }
  /* keep the memory kind the format pass settled on: the size pass truncates to the first structure of othercaps */
  othercaps = fixate_size (trans, direction, caps, othercaps);
  if (gst_caps_get_size (othercaps) == 1) {
    GstStructure *fs = gst_caps_get_structure (format, 0), *os;
    othercaps = gst_caps_make_writable (othercaps);
    os = gst_caps_get_structure (othercaps, 0);
    for (i = 0; i < G_N_ELEMENTS (fields); i++) {
      if (gst_structure_has_field (fs, fields[i]))
        gst_structure_set (os, fields[i], G_TYPE_STRING, gst_structure_get_string (fs, fields[i]), NULL);
      else
        gst_structure_remove_field (os, fields[i]);
    }
  }
  gst_caps_unref (format);
  return othercaps;
}

const gchar *
gst_amd_video_formats_string (void)
{
  return AMD_OUT_FORMATS;
}

#define fill_amd_info gst_amd_video_info_fill
gboolean
gst_amd_video_info_fill (const GstVideoInfo * vi, GstAmdVideoInfo * ai)
{
  guint i;
  static const struct { GstVideoFormat f; int a; } map[] = {
    {GST_VIDEO_FORMAT_I420, GSTAMD_VIDEO_FORMAT_I420}, {GST_VIDEO_FORMAT_YV12, GSTAMD_VIDEO_FORMAT_YV12},
    {GST_VIDEO_FORMAT_AYUV, GSTAMD_VIDEO_FORMAT_AYUV}, {GST_VIDEO_FORMAT_RGBx, GSTAMD_VIDEO_FORMAT_RGBx},
    {GST_VIDEO_FORMAT_BGRx, GSTAMD_VIDEO_FORMAT_BGRx}, {GST_VIDEO_FORMAT_xRGB, GSTAMD_VIDEO_FORMAT_xRGB},
    {GST_VIDEO_FORMAT_xBGR, GSTAMD_VIDEO_FORMAT_xBGR}, {GST_VIDEO_FORMAT_RGBA, GSTAMD_VIDEO_FORMAT_RGBA},
    {GST_VIDEO_FORMAT_BGRA, GSTAMD_VIDEO_FORMAT_BGRA}, {GST_VIDEO_FORMAT_ARGB, GSTAMD_VIDEO_FORMAT_ARGB},
    {GST_VIDEO_FORMAT_ABGR, GSTAMD_VIDEO_FORMAT_ABGR}, {GST_VIDEO_FORMAT_Y42B, GSTAMD_VIDEO_FORMAT_Y42B}, {GST_VIDEO_FORMAT_Y41B, GSTAMD_VIDEO_FORMAT_Y41B},
    {GST_VIDEO_FORMAT_Y444, GSTAMD_VIDEO_FORMAT_Y444}, {GST_VIDEO_FORMAT_NV12, GSTAMD_VIDEO_FORMAT_NV12},
    {GST_VIDEO_FORMAT_NV21, GSTAMD_VIDEO_FORMAT_NV21}, {GST_VIDEO_FORMAT_NV16, GSTAMD_VIDEO_FORMAT_NV16},
    {GST_VIDEO_FORMAT_NV61, GSTAMD_VIDEO_FORMAT_NV61}, {GST_VIDEO_FORMAT_NV24, GSTAMD_VIDEO_FORMAT_NV24},
    {GST_VIDEO_FORMAT_YUY2, GSTAMD_VIDEO_FORMAT_YUY2}, {GST_VIDEO_FORMAT_UYVY, GSTAMD_VIDEO_FORMAT_UYVY},
    {GST_VIDEO_FORMAT_YVYU, GSTAMD_VIDEO_FORMAT_YVYU}, {GST_VIDEO_FORMAT_VYUY, GSTAMD_VIDEO_FORMAT_VYUY},
    {GST_VIDEO_FORMAT_RGB, GSTAMD_VIDEO_FORMAT_RGB},
    {GST_VIDEO_FORMAT_BGR, GSTAMD_VIDEO_FORMAT_BGR},
    {GST_VIDEO_FORMAT_P010_10LE, GSTAMD_VIDEO_FORMAT_P010_10LE}, {GST_VIDEO_FORMAT_I420_10LE, GSTAMD_VIDEO_FORMAT_I420_10LE},
    {GST_VIDEO_FORMAT_ARGB64, GSTAMD_VIDEO_FORMAT_ARGB64}, {GST_VIDEO_FORMAT_AYUV64, GSTAMD_VIDEO_FORMAT_AYUV64},
    {GST_VIDEO_FORMAT_v308, GSTAMD_VIDEO_FORMAT_v308}, {GST_VIDEO_FORMAT_IYU2, GSTAMD_VIDEO_FORMAT_IYU2}, {GST_VIDEO_FORMAT_IYU1, GSTAMD_VIDEO_FORMAT_IYU1},
    {GST_VIDEO_FORMAT_GRAY10_LE32, GSTAMD_VIDEO_FORMAT_GRAY10_LE32}, {GST_VIDEO_FORMAT_NV12_10LE32, GSTAMD_VIDEO_FORMAT_NV12_10LE32},
    {GST_VIDEO_FORMAT_NV16_10LE32, GSTAMD_VIDEO_FORMAT_NV16_10LE32}, {GST_VIDEO_FORMAT_UYVP, GSTAMD_VIDEO_FORMAT_UYVP},
    {GST_VIDEO_FORMAT_NV12_64Z32, GSTAMD_VIDEO_FORMAT_NV12_64Z32},
    {GST_VIDEO_FORMAT_GRAY8, GSTAMD_VIDEO_FORMAT_GRAY8}, {GST_VIDEO_FORMAT_GBR, GSTAMD_VIDEO_FORMAT_GBR}, {GST_VIDEO_FORMAT_v210, GSTAMD_VIDEO_FORMAT_v210},
    {GST_VIDEO_FORMAT_I422_10LE, GSTAMD_VIDEO_FORMAT_I422_10LE}, {GST_VIDEO_FORMAT_Y444_10LE, GSTAMD_VIDEO_FORMAT_Y444_10LE},
    {GST_VIDEO_FORMAT_I420_12LE, GSTAMD_VIDEO_FORMAT_I420_12LE}, {GST_VIDEO_FORMAT_I422_12LE, GSTAMD_VIDEO_FORMAT_I422_12LE},
    {GST_VIDEO_FORMAT_Y444_12LE, GSTAMD_VIDEO_FORMAT_Y444_12LE},
    {GST_VIDEO_FORMAT_GRAY16_LE, GSTAMD_VIDEO_FORMAT_GRAY16_LE}, {GST_VIDEO_FORMAT_GRAY16_BE, GSTAMD_VIDEO_FORMAT_GRAY16_BE},
    {GST_VIDEO_FORMAT_A420, GSTAMD_VIDEO_FORMAT_A420}, {GST_VIDEO_FORMAT_GBRA, GSTAMD_VIDEO_FORMAT_GBRA},
    {GST_VIDEO_FORMAT_GBR_10LE, GSTAMD_VIDEO_FORMAT_GBR_10LE}, {GST_VIDEO_FORMAT_GBR_12LE, GSTAMD_VIDEO_FORMAT_GBR_12LE},
    {GST_VIDEO_FORMAT_GBRA_10LE, GSTAMD_VIDEO_FORMAT_GBRA_10LE}, {GST_VIDEO_FORMAT_GBRA_12LE, GSTAMD_VIDEO_FORMAT_GBRA_12LE},
    {GST_VIDEO_FORMAT_v216, GSTAMD_VIDEO_FORMAT_v216}, {GST_VIDEO_FORMAT_r210, GSTAMD_VIDEO_FORMAT_r210},
    {GST_VIDEO_FORMAT_A420_10LE, GSTAMD_VIDEO_FORMAT_A420_10LE}, {GST_VIDEO_FORMAT_A422_10LE, GSTAMD_VIDEO_FORMAT_A422_10LE},
    {GST_VIDEO_FORMAT_A444_10LE, GSTAMD_VIDEO_FORMAT_A444_10LE},
#if GST_CHECK_VERSION (1, 26, 0)
    {GST_VIDEO_FORMAT_RGBP, GSTAMD_VIDEO_FORMAT_RGBP}, {GST_VIDEO_FORMAT_BGRP, GSTAMD_VIDEO_FORMAT_BGRP}, {GST_VIDEO_FORMAT_RBGA, GSTAMD_VIDEO_FORMAT_RBGA},
    {GST_VIDEO_FORMAT_A422, GSTAMD_VIDEO_FORMAT_A422}, {GST_VIDEO_FORMAT_A444, GSTAMD_VIDEO_FORMAT_A444}, {GST_VIDEO_FORMAT_GBR_16LE, GSTAMD_VIDEO_FORMAT_GBR_16LE},
    {GST_VIDEO_FORMAT_Y216_LE, GSTAMD_VIDEO_FORMAT_Y216_LE}, {GST_VIDEO_FORMAT_Y412_LE, GSTAMD_VIDEO_FORMAT_Y412_LE}, {GST_VIDEO_FORMAT_Y416_LE, GSTAMD_VIDEO_FORMAT_Y416_LE},
    {GST_VIDEO_FORMAT_A420_12LE, GSTAMD_VIDEO_FORMAT_A420_12LE}, {GST_VIDEO_FORMAT_A422_12LE, GSTAMD_VIDEO_FORMAT_A422_12LE}, {GST_VIDEO_FORMAT_A444_12LE, GSTAMD_VIDEO_FORMAT_A444_12LE},
    {GST_VIDEO_FORMAT_A420_16LE, GSTAMD_VIDEO_FORMAT_A420_16LE}, {GST_VIDEO_FORMAT_A422_16LE, GSTAMD_VIDEO_FORMAT_A422_16LE}, {GST_VIDEO_FORMAT_A444_16LE, GSTAMD_VIDEO_FORMAT_A444_16LE},
    {GST_VIDEO_FORMAT_GRAY10_LE16, GSTAMD_VIDEO_FORMAT_GRAY10_LE16},
    {GST_VIDEO_FORMAT_I420_10BE, GSTAMD_VIDEO_FORMAT_I420_10BE},
    {GST_VIDEO_FORMAT_I422_10BE, GSTAMD_VIDEO_FORMAT_I422_10BE},
    {GST_VIDEO_FORMAT_Y444_10BE, GSTAMD_VIDEO_FORMAT_Y444_10BE},
    {GST_VIDEO_FORMAT_I420_12BE, GSTAMD_VIDEO_FORMAT_I420_12BE},
    {GST_VIDEO_FORMAT_I422_12BE, GSTAMD_VIDEO_FORMAT_I422_12BE},
    {GST_VIDEO_FORMAT_Y444_12BE, GSTAMD_VIDEO_FORMAT_Y444_12BE},
    {GST_VIDEO_FORMAT_Y444_16BE, GSTAMD_VIDEO_FORMAT_Y444_16BE},
    {GST_VIDEO_FORMAT_P010_10BE, GSTAMD_VIDEO_FORMAT_P010_10BE},
    {GST_VIDEO_FORMAT_P012_BE, GSTAMD_VIDEO_FORMAT_P012_BE},
    {GST_VIDEO_FORMAT_P016_BE, GSTAMD_VIDEO_FORMAT_P016_BE},
    {GST_VIDEO_FORMAT_GBR_10BE, GSTAMD_VIDEO_FORMAT_GBR_10BE},
    {GST_VIDEO_FORMAT_GBR_12BE, GSTAMD_VIDEO_FORMAT_GBR_12BE},
    {GST_VIDEO_FORMAT_GBR_16BE, GSTAMD_VIDEO_FORMAT_GBR_16BE},
    {GST_VIDEO_FORMAT_GBRA_10BE, GSTAMD_VIDEO_FORMAT_GBRA_10BE},
    {GST_VIDEO_FORMAT_GBRA_12BE, GSTAMD_VIDEO_FORMAT_GBRA_12BE},
    {GST_VIDEO_FORMAT_A420_10BE, GSTAMD_VIDEO_FORMAT_A420_10BE},
    {GST_VIDEO_FORMAT_A422_10BE, GSTAMD_VIDEO_FORMAT_A422_10BE},
    {GST_VIDEO_FORMAT_A444_10BE, GSTAMD_VIDEO_FORMAT_A444_10BE},
    {GST_VIDEO_FORMAT_A420_12BE, GSTAMD_VIDEO_FORMAT_A420_12BE},
    {GST_VIDEO_FORMAT_A422_12BE, GSTAMD_VIDEO_FORMAT_A422_12BE},
    {GST_VIDEO_FORMAT_A444_12BE, GSTAMD_VIDEO_FORMAT_A444_12BE},
    {GST_VIDEO_FORMAT_A420_16BE, GSTAMD_VIDEO_FORMAT_A420_16BE}, {GST_VIDEO_FORMAT_AV12, GSTAMD_VIDEO_FORMAT_AV12},
    {GST_VIDEO_FORMAT_A422_16BE, GSTAMD_VIDEO_FORMAT_A422_16BE},
    {GST_VIDEO_FORMAT_A444_16BE, GSTAMD_VIDEO_FORMAT_A444_16BE},
    {GST_VIDEO_FORMAT_Y212_BE, GSTAMD_VIDEO_FORMAT_Y212_BE},
    {GST_VIDEO_FORMAT_Y216_BE, GSTAMD_VIDEO_FORMAT_Y216_BE},
    {GST_VIDEO_FORMAT_Y412_BE, GSTAMD_VIDEO_FORMAT_Y412_BE},
    {GST_VIDEO_FORMAT_Y416_BE, GSTAMD_VIDEO_FORMAT_Y416_BE},
#endif
    {GST_VIDEO_FORMAT_RGB16, GSTAMD_VIDEO_FORMAT_RGB16}, {GST_VIDEO_FORMAT_BGR16, GSTAMD_VIDEO_FORMAT_BGR16},
    {GST_VIDEO_FORMAT_RGB15, GSTAMD_VIDEO_FORMAT_RGB15}, {GST_VIDEO_FORMAT_BGR15, GSTAMD_VIDEO_FORMAT_BGR15},
    /* VUYA, Y210, Y410 (1.16), P012_LE, P016_LE, Y444_16LE, Y212_LE (1.18) joined the format enum after 1.14: there when the headers this is compiled
     * against have them (plugins/build.py also type-checks every element against the reference's own 1.29 headers) */
#if GST_CHECK_VERSION (1, 16, 0)
    {GST_VIDEO_FORMAT_VUYA, GSTAMD_VIDEO_FORMAT_VUYA}, {GST_VIDEO_FORMAT_Y210, GSTAMD_VIDEO_FORMAT_Y210}, {GST_VIDEO_FORMAT_Y410, GSTAMD_VIDEO_FORMAT_Y410},
    {GST_VIDEO_FORMAT_BGR10A2_LE, GSTAMD_VIDEO_FORMAT_BGR10A2_LE}, {GST_VIDEO_FORMAT_NV12_10LE40, GSTAMD_VIDEO_FORMAT_NV12_10LE40},
#endif
#if GST_CHECK_VERSION (1, 18, 0)
    {GST_VIDEO_FORMAT_P012_LE, GSTAMD_VIDEO_FORMAT_P012_LE}, {GST_VIDEO_FORMAT_P016_LE, GSTAMD_VIDEO_FORMAT_P016_LE},
    {GST_VIDEO_FORMAT_Y444_16LE, GSTAMD_VIDEO_FORMAT_Y444_16LE}, {GST_VIDEO_FORMAT_Y212_LE, GSTAMD_VIDEO_FORMAT_Y212_LE},
    {GST_VIDEO_FORMAT_RGB10A2_LE, GSTAMD_VIDEO_FORMAT_RGB10A2_LE},
#endif
#if GST_CHECK_VERSION (1, 26, 0)
    {GST_VIDEO_FORMAT_NV12_16L32S, GSTAMD_VIDEO_FORMAT_NV12_16L32S}, {GST_VIDEO_FORMAT_NV12_8L128, GSTAMD_VIDEO_FORMAT_NV12_8L128},
    {GST_VIDEO_FORMAT_NV12_10LE40_4L4, GSTAMD_VIDEO_FORMAT_NV12_10LE40_4L4},
#endif
#if GST_CHECK_VERSION (1, 28, 0)
    {GST_VIDEO_FORMAT_BGR10x2_LE, GSTAMD_VIDEO_FORMAT_BGR10x2_LE}, {GST_VIDEO_FORMAT_RGB10x2_LE, GSTAMD_VIDEO_FORMAT_RGB10x2_LE},
    {GST_VIDEO_FORMAT_NV16_10LE40, GSTAMD_VIDEO_FORMAT_NV16_10LE40},
#endif
#if GST_CHECK_VERSION (1, 29, 0)
    {GST_VIDEO_FORMAT_RGBA_F16LE, GSTAMD_VIDEO_FORMAT_RGBA_F16LE}, {GST_VIDEO_FORMAT_RGBA_F16BE, GSTAMD_VIDEO_FORMAT_RGBA_F16BE},
#endif
#if GST_CHECK_VERSION (1, 20, 0)
    {GST_VIDEO_FORMAT_NV12_4L4, GSTAMD_VIDEO_FORMAT_NV12_4L4}, {GST_VIDEO_FORMAT_NV12_32L32, GSTAMD_VIDEO_FORMAT_NV12_32L32},
    {GST_VIDEO_FORMAT_ARGB64_LE, GSTAMD_VIDEO_FORMAT_ARGB64_LE}, {GST_VIDEO_FORMAT_ARGB64_BE, GSTAMD_VIDEO_FORMAT_ARGB64_BE},
    {GST_VIDEO_FORMAT_RGBA64_LE, GSTAMD_VIDEO_FORMAT_RGBA64_LE}, {GST_VIDEO_FORMAT_RGBA64_BE, GSTAMD_VIDEO_FORMAT_RGBA64_BE},
    {GST_VIDEO_FORMAT_BGRA64_LE, GSTAMD_VIDEO_FORMAT_BGRA64_LE}, {GST_VIDEO_FORMAT_BGRA64_BE, GSTAMD_VIDEO_FORMAT_BGRA64_BE},
    {GST_VIDEO_FORMAT_ABGR64_LE, GSTAMD_VIDEO_FORMAT_ABGR64_LE}, {GST_VIDEO_FORMAT_ABGR64_BE, GSTAMD_VIDEO_FORMAT_ABGR64_BE},
#endif
  };
  int fmt = 0;
  for (i = 0; i < G_N_ELEMENTS (map); i++)
    if (map[i].f == GST_VIDEO_INFO_FORMAT (vi))
      fmt = map[i].a;
  if (!fmt || gstamd_video_info_set_format (ai, fmt, GST_VIDEO_INFO_WIDTH (vi), GST_VIDEO_INFO_HEIGHT (vi)) != GSTAMD_OK)
    return FALSE;
  for (i = 0; i < GST_VIDEO_INFO_N_PLANES (vi); i++) {
    ai->stride[i] = GST_VIDEO_INFO_PLANE_STRIDE (vi, i);
    ai->offset[i] = GST_VIDEO_INFO_PLANE_OFFSET (vi, i);
  }
  ai->size = GST_VIDEO_INFO_SIZE (vi);
  /* GstVideoColorRange / GstVideoColorMatrix / GstVideoChromaSite share their numeric values with the ABI */
  ai->color_range = vi->colorimetry.range;
  ai->color_matrix = vi->colorimetry.matrix;
  ai->chroma_site = vi->chroma_site;
  /* GstVideoTransferFunction / GstVideoColorPrimaries as they are (the enums only ever grew at the end) */
  ai->color_transfer = vi->colorimetry.transfer;
  ai->color_primaries = vi->colorimetry.primaries;
  return TRUE;
}

static gboolean
caps_are_hip (GstCaps * caps)
{
  GstCapsFeatures *f = gst_caps_get_features (caps, 0);
  return f && gst_caps_features_contains (f, GST_CAPS_FEATURE_MEMORY_AMD_HIP);
}

static gboolean
amd_vcs_set_caps (GstBaseTransform * trans, GstCaps * incaps, GstCaps * outcaps)
{
  GstAmdVideoConvertScale *s = AMD_VCS (trans);
  GstAmdVideoInfo ai, ao;
  GstAmdVideoConverterConfig cfg;
  int status = 0;

  if (!gst_video_info_from_caps (&s->in_info, incaps) || !gst_video_info_from_caps (&s->out_info, outcaps))
    return FALSE;
  if (GST_VIDEO_INFO_INTERLACE_MODE (&s->in_info) != GST_VIDEO_INFO_INTERLACE_MODE (&s->out_info) ||
      (GST_VIDEO_INFO_IS_INTERLACED (&s->in_info) && GST_VIDEO_INFO_INTERLACE_MODE (&s->in_info) != GST_VIDEO_INTERLACE_MODE_INTERLEAVED &&
          GST_VIDEO_INFO_INTERLACE_MODE (&s->in_info) != GST_VIDEO_INTERLACE_MODE_MIXED)) {
    GST_ERROR_OBJECT (s, "interlace-mode %s is not supported by the HIP converter (the modes of both sides must be equal, as gst_video_converter_new wants them)",
        gst_video_interlace_mode_to_string (GST_VIDEO_INFO_INTERLACE_MODE (&s->in_info)));
    return FALSE;
  }
  gst_amd_hip_select_device (s->device_id);
  s->in_hip = caps_are_hip (incaps);
  s->out_hip = caps_are_hip (outcaps);
  if (s->out_pool) {
    gst_buffer_pool_set_active (s->out_pool, FALSE);
    gst_object_unref (s->out_pool);
    s->out_pool = NULL;
  }
  if (s->out_hip && !(s->out_pool = gst_amd_hip_buffer_pool_new_for_caps (outcaps, 8))) {
    GST_ERROR_OBJECT (s, "could not set up the HBM output pool");
    return FALSE;
  }
  if (!fill_amd_info (&s->in_info, &ai) || !fill_amd_info (&s->out_info, &ao)) {
    GST_ERROR_OBJECT (s, "format not supported by the HIP converter");
    return FALSE;
  }
  gstamd_video_converter_config_init (&cfg);
  /* method -> converter options, as gstvideoconvertscale.c:993-1062 */
  switch (s->method) {
    case AMD_SCALE_NEAREST: cfg.resampler_method = GSTAMD_RESAMPLER_METHOD_NEAREST; break;
    case AMD_SCALE_BILINEAR: cfg.resampler_method = GSTAMD_RESAMPLER_METHOD_LINEAR; cfg.max_taps = 2; break;
    case AMD_SCALE_4TAP: cfg.resampler_method = GSTAMD_RESAMPLER_METHOD_SINC; cfg.max_taps = 4; break;
    case AMD_SCALE_LANCZOS: cfg.resampler_method = GSTAMD_RESAMPLER_METHOD_LANCZOS; break;
    case AMD_SCALE_BILINEAR2: cfg.resampler_method = GSTAMD_RESAMPLER_METHOD_LINEAR; break;
    case AMD_SCALE_SINC: cfg.resampler_method = GSTAMD_RESAMPLER_METHOD_SINC; break;
    case AMD_SCALE_HERMITE: cfg.resampler_method = GSTAMD_RESAMPLER_METHOD_CUBIC; cfg.cubic_b = 0.0; cfg.cubic_c = 0.0; break;
    case AMD_SCALE_SPLINE: cfg.resampler_method = GSTAMD_RESAMPLER_METHOD_CUBIC; cfg.cubic_b = 1.0; cfg.cubic_c = 0.0; break;
    case AMD_SCALE_CATROM: cfg.resampler_method = GSTAMD_RESAMPLER_METHOD_CUBIC; cfg.cubic_b = 0.0; cfg.cubic_c = 0.5; break;
    case AMD_SCALE_MITCHELL: cfg.resampler_method = GSTAMD_RESAMPLER_METHOD_CUBIC; cfg.cubic_b = 1.0 / 3.0; cfg.cubic_c = 1.0 / 3.0; break;
  }
  /* borders that keep the display aspect ratio (gstvideoconvertscale.c:920-957, 1068-1072) */
  s->borders_w = s->borders_h = 0;
  {
    gint from_n, from_d, to_n, to_d;
    if (!gst_util_fraction_multiply (s->in_info.width, s->in_info.height, s->in_info.par_n, s->in_info.par_d, &from_n, &from_d))
      from_n = from_d = -1;
    if (!gst_util_fraction_multiply (s->out_info.width, s->out_info.height, s->out_info.par_n, s->out_info.par_d, &to_n, &to_d))
      to_n = to_d = -1;
    if ((to_n != from_n || to_d != from_d) && s->add_borders) {
      gint n, d;
      if (from_n != -1 && from_d != -1 && gst_util_fraction_multiply (from_n, from_d, s->out_info.par_d, s->out_info.par_n, &n, &d)) {
        const gint to_h = gst_util_uint64_scale_int (s->out_info.width, d, n);
        if (to_h <= s->out_info.height)
          s->borders_h = s->out_info.height - to_h;
        else
          s->borders_w = s->out_info.width - (gint) gst_util_uint64_scale_int (s->out_info.height, n, d);
      } else {
        GST_WARNING_OBJECT (s, "Can't calculate borders");
      }
    }
  }
  cfg.dest_x = s->borders_w / 2;
  cfg.dest_y = s->borders_h / 2;
  cfg.dest_width = s->out_info.width - s->borders_w;
  cfg.dest_height = s->out_info.height - s->borders_h;
  cfg.envelope = s->envelope;
  cfg.sharpness = s->sharpness;
  cfg.sharpen = s->sharpen;
  cfg.alpha_mode = s->alpha_mode;
  cfg.alpha_value = s->alpha_value;
  cfg.chroma_mode = s->chroma_mode;
  cfg.matrix_mode = s->matrix_mode;
  cfg.gamma_mode = s->gamma_mode;
  cfg.primaries_mode = s->primaries_mode;
  cfg.dither_quantization = s->dither_quantization;
  cfg.dither_method = s->dither;
  cfg.chroma_resampler_method = s->chroma_resampler;       /* GST_VIDEO_CONVERTER_OPT_CHROMA_RESAMPLER_METHOD (:1076) */

  if (s->converter_config) {
    /* a user-provided converter-config replaces the element's own options altogether (gstvideoconvertscale.c:962-967): method,
     * borders, alpha / chroma / matrix modes then come from the structure or are the library's defaults */
    gstamd_video_converter_config_init (&cfg);
    gst_amd_converter_config_from_structure (s->converter_config, &cfg);
    GST_DEBUG_OBJECT (s, "using the user-provided converter-config %" GST_PTR_FORMAT, s->converter_config);
  }
  s->batch_limit = 1;
  amd_vcs_batch_set_converter (s);      /* launches what is pending and detaches the old converter */
  if (s->convert)
    gstamd_video_converter_free (s->convert);
  if (s->convert_i)
    gstamd_video_converter_free (s->convert_i);
  s->convert_i = NULL;
  /* interleaved: every frame is an interlaced one; mixed: the flagged buffers are (a second converter), the others are progressive frames */
  ai.interlace_mode = ao.interlace_mode = GST_VIDEO_INFO_INTERLACE_MODE (&s->in_info) == GST_VIDEO_INTERLACE_MODE_INTERLEAVED ?
      GSTAMD_INTERLACE_MODE_INTERLEAVED : GSTAMD_INTERLACE_MODE_PROGRESSIVE;
  s->convert = gstamd_video_converter_new (&ai, &ao, &cfg, &status);
  if (!s->convert) {
    GST_ERROR_OBJECT (s, "no HIP conversion for these caps: %s", gstamd_last_error ());
    return FALSE;
  }
  if (GST_VIDEO_INFO_INTERLACE_MODE (&s->in_info) == GST_VIDEO_INTERLACE_MODE_MIXED) {
    ai.interlace_mode = ao.interlace_mode = GSTAMD_INTERLACE_MODE_INTERLEAVED;
    s->convert_i = gstamd_video_converter_new (&ai, &ao, &cfg, &status);
    if (!s->convert_i) {
      GST_ERROR_OBJECT (s, "no HIP conversion for the interlaced frames of these caps: %s", gstamd_last_error ());
      gstamd_video_converter_free (s->convert);
      s->convert = NULL;
      return FALSE;
    }
    GST_CAT_DEBUG_OBJECT (CAT_PERFORMANCE, s, "HIP plan of the flagged frames: %s", gstamd_video_converter_describe (s->convert_i));
  }
  GST_CAT_DEBUG_OBJECT (CAT_PERFORMANCE, s, "HIP plan: %s", gstamd_video_converter_describe (s->convert));
  {
    /* the stream ring: as many streams as the plan allows in flight at once (plans with a scratch image: one) */
    guint want = gstamd_video_converter_is_reentrant (s->convert) ? CLAMP (s->hip_streams, 1, AMD_MAX_STREAMS) : 1, i;
    for (i = 0; i < AMD_MAX_STREAMS; i++)
      if (i < want && !s->streams[i])
        s->streams[i] = gstamd_stream_new ();
    for (i = 0; i < want && s->streams[i]; i++);
    if (i == 0) {
      GST_ERROR_OBJECT (s, "could not create a HIP stream: %s", gstamd_last_error ());
      return FALSE;
    }
    s->n_streams = i;
  }
  /* deferred launches (batch-buffers): HBM on both sides and a plan without per-converter scratch; automatic = only when upstream is
   * not live - a live pipeline would have to report the frames it holds back as latency */
  if (s->in_hip && s->out_hip && gstamd_video_converter_is_reentrant (s->convert)) {
    guint want = s->batch_buffers;
    if (want == 0) {
      GstQuery *q = gst_query_new_latency ();
      gboolean live = TRUE;
      if (gst_pad_peer_query (GST_BASE_TRANSFORM_SINK_PAD (trans), q))
        gst_query_parse_latency (q, &live, NULL, NULL);
      gst_query_unref (q);
      want = live ? 1 : 4;
    }
    s->batch_limit = MIN (want, AMD_BATCH_MAX);
    if (s->batch_limit > 1) {
      amd_vcs_batch_start (s);
      amd_vcs_batch_set_converter (s);
    }
  }
  return TRUE;
}

/* upstream asks how to allocate the frames it will send us (gst_video_filter_propose_allocation, gstvideofilter.c:56-103, adds
 * GstVideoMeta support; here HBM caps additionally get the HIP allocator and a GstAmdHipBufferPool, so a source or decoder that
 * honours the query writes straight into device memory) */
static gboolean
amd_vcs_propose_allocation (GstBaseTransform * trans, GstQuery * decide_query, GstQuery * query)
{
  GstAmdVideoConvertScale *s = AMD_VCS (trans);
  GstCaps *caps = NULL;
  gboolean need_pool = FALSE;

  if (!GST_BASE_TRANSFORM_CLASS (gst_amd_vcs_parent_class)->propose_allocation (trans, decide_query, query))
    return FALSE;
  if (decide_query == NULL)           /* passthrough: the downstream answer was forwarded */
    return TRUE;
  gst_query_parse_allocation (query, &caps, &need_pool);
  if (caps && caps_are_hip (caps)) {
    GstVideoInfo info;
    if (!gst_video_info_from_caps (&info, caps))
      return FALSE;
    gst_amd_hip_select_device (s->device_id);
    gst_query_add_allocation_param (query, gst_amd_hip_allocator_get (), NULL);
    if (need_pool) {
      GstBufferPool *pool = gst_amd_hip_buffer_pool_new ();
      GstStructure *config = gst_buffer_pool_get_config (pool);
      gst_buffer_pool_config_set_params (config, caps, GST_VIDEO_INFO_SIZE (&info), 2, 0);
      gst_buffer_pool_config_add_option (config, GST_BUFFER_POOL_OPTION_VIDEO_META);
      if (!gst_buffer_pool_set_config (pool, config)) {
        gst_object_unref (pool);
        return FALSE;
      }
      gst_query_add_allocation_pool (query, pool, GST_VIDEO_INFO_SIZE (&info), 2, 0);
      gst_object_unref (pool);
    } else {
      gst_query_add_allocation_pool (query, NULL, GST_VIDEO_INFO_SIZE (&info), 2, 0);
    }
  }
  else if (caps && need_pool && s->pinned_pools) {
    /* system-memory frames that are about to be uploaded: offer page-locked host buffers, the upload is then a DMA transfer at the
       link's rate (pageable memory is staged by the runtime at a fraction of it) */
    GstVideoInfo info;
    if (gst_video_info_from_caps (&info, caps)) {
      GstBufferPool *pool = gst_amd_hip_buffer_pool_new_pinned_host ();
      GstStructure *config = gst_buffer_pool_get_config (pool);
      gst_amd_hip_select_device (s->device_id);
      gst_buffer_pool_config_set_params (config, caps, GST_VIDEO_INFO_SIZE (&info), 2, 0);
      if (gst_buffer_pool_set_config (pool, config))
        gst_query_add_allocation_pool (query, pool, GST_VIDEO_INFO_SIZE (&info), 2, 0);
      gst_object_unref (pool);
    }
  }
  gst_query_add_allocation_meta (query, GST_VIDEO_META_API_TYPE, NULL);
  return TRUE;
}

/* system-memory output that downstream has no pool for: page-locked host buffers of our own (the download is a DMA transfer) */
static gboolean
amd_vcs_decide_allocation (GstBaseTransform * trans, GstQuery * query)
{
  GstAmdVideoConvertScale *s = AMD_VCS (trans);
  GstCaps *caps = NULL;

  gst_query_parse_allocation (query, &caps, NULL);
  if (caps && !caps_are_hip (caps) && s->pinned_pools) {
    GstBufferPool *theirs = NULL;
    GstVideoInfo info;
    if (gst_query_get_n_allocation_pools (query) > 0)
      gst_query_parse_nth_allocation_pool (query, 0, &theirs, NULL, NULL, NULL);
    if (!theirs && gst_video_info_from_caps (&info, caps)) {
      GstBufferPool *pool = gst_amd_hip_buffer_pool_new_pinned_host ();
      gst_amd_hip_select_device (s->device_id);
      if (gst_query_get_n_allocation_pools (query) > 0)
        gst_query_set_nth_allocation_pool (query, 0, pool, GST_VIDEO_INFO_SIZE (&info), 2, 0);
      else
        gst_query_add_allocation_pool (query, pool, GST_VIDEO_INFO_SIZE (&info), 2, 0);
      gst_object_unref (pool);
    }
    if (theirs)
      gst_object_unref (theirs);
  }
  return GST_BASE_TRANSFORM_CLASS (gst_amd_vcs_parent_class)->decide_allocation (trans, query);
}

static gboolean
amd_vcs_get_unit_size (GstBaseTransform * trans, GstCaps * caps, gsize * size)
{
  GstVideoInfo info;
  if (!gst_video_info_from_caps (&info, caps))
    return FALSE;
  *size = GST_VIDEO_INFO_SIZE (&info);
  return TRUE;
}

/* HIP output: frames come from our own GstAmdHipBufferPool; system output: default */
static GstFlowReturn
amd_vcs_prepare_output_buffer (GstBaseTransform * trans, GstBuffer * inbuf, GstBuffer ** outbuf)
{
  GstAmdVideoConvertScale *s = AMD_VCS (trans);

  if (gst_base_transform_is_passthrough (trans)) {
    *outbuf = inbuf;
    return GST_FLOW_OK;
  }
  if (!s->out_hip)
    return GST_BASE_TRANSFORM_CLASS (gst_amd_vcs_parent_class)->prepare_output_buffer (trans, inbuf, outbuf);
  gst_amd_hip_select_device (s->device_id);
  {
    const gint64 t0 = s->stats ? g_get_monotonic_time () : 0;
    const GstFlowReturn fr = s->out_pool ? gst_buffer_pool_acquire_buffer (s->out_pool, outbuf, NULL) : GST_FLOW_ERROR;
    if (fr != GST_FLOW_OK)
      return fr;                /* FLUSHING during a seek / shutdown is the pool's answer, not an error of this element */
    gst_buffer_copy_into (*outbuf, inbuf, GST_BUFFER_COPY_FLAGS | GST_BUFFER_COPY_TIMESTAMPS, 0, -1);
    if (s->stats)
      s->t_prepare += g_get_monotonic_time () - t0;
    return GST_FLOW_OK;
  }
  {
    const GstFlowReturn fr = s->out_pool ? gst_buffer_pool_acquire_buffer (s->out_pool, outbuf, NULL) : GST_FLOW_ERROR;
    if (fr != GST_FLOW_OK)
      return fr;
  }
  gst_buffer_copy_into (*outbuf, inbuf, GST_BUFFER_COPY_FLAGS | GST_BUFFER_COPY_TIMESTAMPS, 0, -1);
  return GST_FLOW_OK;
}

static gboolean
ensure_staging (gpointer * p, gsize * have, gsize need)
{
  if (*have >= need)
    return TRUE;
  gstamd_device_free (*p);
  *p = gstamd_device_alloc (need);
  *have = *p ? need : 0;
  return *p != NULL;
}

/* plane pointers / pitches of a frame whose first byte is `base`: the buffer's GstVideoMeta when it has one (a pool or decoder may pad
 * rows or planes), else the negotiated GstVideoInfo */
static void
frame_planes (GstBuffer * buf, const GstVideoInfo * info, guint8 * base, gpointer planes[GSTAMD_VIDEO_MAX_PLANES],
    gint32 strides[GSTAMD_VIDEO_MAX_PLANES])
{
  GstVideoMeta *vm = gst_buffer_get_video_meta (buf);
  guint i, n = GST_VIDEO_INFO_N_PLANES (info);

  for (i = 0; i < GSTAMD_VIDEO_MAX_PLANES; i++) {
    planes[i] = NULL;
    strides[i] = 0;
  }
  for (i = 0; i < n && i < GSTAMD_VIDEO_MAX_PLANES; i++) {
    if (vm && vm->n_planes == n) {
      planes[i] = base + vm->offset[i];
      strides[i] = vm->stride[i];
    } else {
      planes[i] = base + GST_VIDEO_INFO_PLANE_OFFSET (info, i);
      strides[i] = GST_VIDEO_INFO_PLANE_STRIDE (info, i);
    }
  }
}

/* ---- deferred launches ------------------------------------------------------------------------------------------------------
 * Two dependent kernels on one HIP stream are ~4 us apart on this stack however little the host does (scripts/launch_cost.cpp: 11.3 us
 * per 4K frame back to back against 7.1 in a list of 32; 6.0 against 2.1 at 1080p), and an event record per buffer adds ~3 more.  With
 * batch-buffers > 1 transform () does not launch: it files the (input, output) pair in the open batch and hangs a DEFERRED ticket
 * (gstamdhipmemory.h) on both memories.  The batch becomes one gstamd_video_converter_frames call - one launch, one event - when it is
 * full, when anybody needs one of its frames (a stream about to read it, a CPU map: the ticket's launch hook), at EOS / flush / caps
 * change / stop, or when it has been sitting for AMD_BATCH_MAX_AGE_US.  The buffers are referenced until then, so neither side's pool
 * can recycle them under the pending launch.  The object outlives the element for as long as a ticket points at it. */
typedef struct _AmdVcsBatch {
  gint refcount;
  GMutex lock;
  GCond cond;                   /* wakes the watcher */
  GstAmdVideoConverter *convert;        /* the element's; cleared (after a flush) before the element frees it */
  gint device_id;
  gboolean lazy;                /* the element has one stream: its tickets need no event until another stream or the host asks */
  gpointer stream;              /* of the open batch */
  guint n;
  GstBuffer *in[AMD_BATCH_MAX], *out[AMD_BATCH_MAX];
  const void *src[AMD_BATCH_MAX];
  void *dst[AMD_BATCH_MAX];
  GstAmdHipTicket *ticket;      /* deferred, shared by the frames of the open batch */
  gint64 opened_at;             /* monotonic us */
  gboolean quit;
  guint64 n_launches, n_frames, n_on_demand, n_aged;
} AmdVcsBatch;

static AmdVcsBatch *
amd_vcs_batch_ref (AmdVcsBatch * b)
{
  g_atomic_int_inc (&b->refcount);
  return b;
}

static void
amd_vcs_batch_unref (gpointer p)
{
  AmdVcsBatch *b = p;
  if (!g_atomic_int_dec_and_test (&b->refcount))
    return;
  g_mutex_clear (&b->lock);
  g_cond_clear (&b->cond);
  g_free (b);
}

/* lock held.  The buffers to release are handed back: dropping the last reference of a memory waits for its tickets on the host,
 * which is nothing to do under the lock. */
static guint
amd_vcs_batch_launch_locked (AmdVcsBatch * b, GstBuffer ** drop)
{
  guint i, n_drop = 0;
  int r = GSTAMD_ERR_INVALID;

  if (!b->n)
    return 0;
  gst_amd_hip_select_device (b->device_id);
  if (b->convert)
    r = gstamd_video_converter_frames (b->convert, (int) b->n, b->src, b->dst, b->stream);
  if (r != GSTAMD_OK)
    GST_ERROR ("deferred HIP conversion of %u frames failed: %s", b->n, gstamd_last_error ());
  gst_amd_hip_ticket_resolve (b->ticket, r == GSTAMD_OK ? b->stream : NULL, b->lazy);
  gst_amd_hip_ticket_unref (b->ticket);
  b->ticket = NULL;
  for (i = 0; i < b->n; i++) {
    drop[n_drop++] = b->in[i];
    drop[n_drop++] = b->out[i];
  }
  b->n_launches++;
  b->n_frames += b->n;
  b->n = 0;
  return n_drop;
}

static void
amd_vcs_batch_flush (AmdVcsBatch * b, gboolean on_demand)
{
  GstBuffer *drop[2 * AMD_BATCH_MAX];
  guint n, i;

  if (!b)
    return;
  g_mutex_lock (&b->lock);
  if (b->n && on_demand)
    b->n_on_demand++;
  n = amd_vcs_batch_launch_locked (b, drop);
  g_mutex_unlock (&b->lock);
  for (i = 0; i < n; i++)
    gst_buffer_unref (drop[i]);
}

/* the deferred ticket's hook: somebody needs a frame of the open batch */
static void
amd_vcs_batch_launch_hook (gpointer owner)
{
  amd_vcs_batch_flush (owner, TRUE);
}

static gpointer
amd_vcs_batch_watch (gpointer data)
{
  AmdVcsBatch *b = data;

  g_mutex_lock (&b->lock);
  while (!b->quit) {
    if (!b->n) {
      g_cond_wait (&b->cond, &b->lock);
      continue;
    }
    {
      const gint64 due = b->opened_at + AMD_BATCH_MAX_AGE_US;
      if (g_get_monotonic_time () < due) {
        g_cond_wait_until (&b->cond, &b->lock, due);
        continue;
      }
    }
    {
      GstBuffer *drop[2 * AMD_BATCH_MAX];
      guint n, i;
      b->n_aged++;
      n = amd_vcs_batch_launch_locked (b, drop);
      g_mutex_unlock (&b->lock);
      for (i = 0; i < n; i++)
        gst_buffer_unref (drop[i]);
      g_mutex_lock (&b->lock);
    }
  }
  g_mutex_unlock (&b->lock);
  amd_vcs_batch_unref (b);
  return NULL;
}

static gboolean buffer_is_plain_hip_frame (GstBuffer * buf, const GstVideoInfo * info);

/* transform () for one HBM -> HBM pair while batching is on: TRUE when the pair has been filed (or launched with the batch it filled) */
static gboolean
amd_vcs_batch_add (GstAmdVideoConvertScale * s, GstBuffer * inbuf, GstBuffer * outbuf)
{
  AmdVcsBatch *b = s->batch;
  GstMemory *imem = gst_buffer_peek_memory (inbuf, 0), *omem = gst_buffer_peek_memory (outbuf, 0);
  GstMapInfo imap, omap;
  GstBuffer *drop[2 * AMD_BATCH_MAX + 2];
  GstAmdHipTicket *t;
  gpointer stream;
  guint n_drop = 0, i;

  gint64 ta = 0, tb = 0, tc = 0;

  if (s->stats)
    ta = g_get_monotonic_time ();
  g_mutex_lock (&b->lock);
  if (!b->n)
    b->stream = s->streams[s->next_stream++ % s->n_streams];
  stream = b->stream;
  g_mutex_unlock (&b->lock);
  /* device pointers (a device map also uploads what a CPU write left in the staging copy) and the stream's waits, outside the lock:
   * they may run the launch hooks of other elements' tickets */
  if (!gst_memory_map (imem, &imap, GST_MAP_READ | GST_MAP_AMDHIP))
    return FALSE;
  if (!gst_memory_map (omem, &omap, GST_MAP_WRITE | GST_MAP_AMDHIP)) {
    gst_memory_unmap (imem, &imap);
    return FALSE;
  }
  gst_amd_hip_memory_wait_written (imem, stream);
  gst_amd_hip_memory_wait_idle (omem, stream);
  if (s->stats)
    tb = g_get_monotonic_time ();
  g_mutex_lock (&b->lock);
  if (b->n && b->stream != stream) {
    /* cannot happen with one streaming thread; keep the waits and the launch on one stream whatever happened */
    n_drop = amd_vcs_batch_launch_locked (b, drop);
  }
  if (!b->n) {
    b->stream = stream;         /* a hook may have launched the batch meanwhile: a new one opens on the stream the waits went to */
    b->opened_at = g_get_monotonic_time ();
  }
  if (!b->ticket)
    b->ticket = gst_amd_hip_ticket_new_deferred (amd_vcs_batch_launch_hook, amd_vcs_batch_ref (b), amd_vcs_batch_unref);
  t = gst_amd_hip_ticket_ref (b->ticket);
  b->in[b->n] = gst_buffer_ref (inbuf);
  b->out[b->n] = gst_buffer_ref (outbuf);
  b->src[b->n] = imap.data;
  b->dst[b->n] = omap.data;
  b->n++;
  if (b->n >= s->batch_limit)
    n_drop += amd_vcs_batch_launch_locked (b, drop + n_drop);
  else if (b->n == 1)
    g_cond_signal (&b->cond);
  g_mutex_unlock (&b->lock);
  if (s->stats)
    tc = g_get_monotonic_time ();
  /* deferred or resolved by now, the ticket is the frame's either way */
  gst_amd_hip_memory_set_read (imem, t);
  gst_amd_hip_memory_set_written (omem, t);
  gst_amd_hip_ticket_unref (t);
  gst_memory_unmap (omem, &omap);
  gst_memory_unmap (imem, &imap);
  for (i = 0; i < n_drop; i++)
    gst_buffer_unref (drop[i]);
  if (s->stats) {
    s->t_wait += tb - ta;
    s->t_convert += tc - tb;
    s->t_mark += g_get_monotonic_time () - tc;
  }
  return TRUE;
}

static void
amd_vcs_batch_start (GstAmdVideoConvertScale * s)
{
  AmdVcsBatch *b;

  if (s->batch)
    return;
  b = g_new0 (AmdVcsBatch, 1);
  b->refcount = 1;
  g_mutex_init (&b->lock);
  g_cond_init (&b->cond);
  b->device_id = s->device_id;
  s->batch = b;
  s->batch_watch = g_thread_new ("amdvcs-batch", amd_vcs_batch_watch, amd_vcs_batch_ref (b));
}

/* flush, detach from the element's converter, stop the watcher; tickets may keep the (empty) object alive */
static void
amd_vcs_batch_stop (GstAmdVideoConvertScale * s)
{
  AmdVcsBatch *b = s->batch;

  if (!b)
    return;
  amd_vcs_batch_flush (b, FALSE);
  g_mutex_lock (&b->lock);
  b->convert = NULL;
  b->quit = TRUE;
  g_cond_signal (&b->cond);
  g_mutex_unlock (&b->lock);
  if (s->batch_watch)
    g_thread_join (s->batch_watch);
  s->batch_watch = NULL;
  if (s->stats && b->n_launches)
    g_printerr ("videoconvertscale deferred launches: %" G_GUINT64_FORMAT " frames in %" G_GUINT64_FORMAT " launches (%" G_GUINT64_FORMAT
        " on demand, %" G_GUINT64_FORMAT " aged)\n", b->n_frames, b->n_launches, b->n_on_demand, b->n_aged);
  s->batch = NULL;
  amd_vcs_batch_unref (b);
}

/* after the element's converter changed (or is about to be freed): pending frames go out with the one they were filed for */
static void
amd_vcs_batch_set_converter (GstAmdVideoConvertScale * s)
{
  AmdVcsBatch *b = s->batch;

  if (!b)
    return;
  amd_vcs_batch_flush (b, FALSE);
  g_mutex_lock (&b->lock);
  b->convert = s->batch_limit > 1 ? s->convert : NULL;
  b->device_id = s->device_id;
  b->lazy = s->n_streams == 1;
  g_mutex_unlock (&b->lock);
}

/* serialized events must not overtake the frames filed before them */
static gboolean
amd_vcs_sink_event (GstBaseTransform * trans, GstEvent * event)
{
  GstAmdVideoConvertScale *s = AMD_VCS (trans);

  if (s->batch && (GST_EVENT_IS_SERIALIZED (event) || GST_EVENT_TYPE (event) == GST_EVENT_FLUSH_START))
    amd_vcs_batch_flush (s->batch, FALSE);
  /* input buffers a queued upload still reads are not kept across a flush or the end of the stream */
  if (GST_EVENT_TYPE (event) == GST_EVENT_FLUSH_STOP || GST_EVENT_TYPE (event) == GST_EVENT_EOS || GST_EVENT_TYPE (event) == GST_EVENT_GAP) {
    gst_amd_hip_select_device (s->device_id);
    gst_amd_hip_pending_reads_drain (s->reads);
  }
  return GST_BASE_TRANSFORM_CLASS (gst_amd_vcs_parent_class)->sink_event (trans, event);
}


static GstFlowReturn
amd_vcs_transform (GstBaseTransform * trans, GstBuffer * inbuf, GstBuffer * outbuf)
{
  GstAmdVideoConvertScale *s = AMD_VCS (trans);
  GstMapInfo imap, omap;
  GstMemory *imem = gst_buffer_peek_memory (inbuf, 0), *omem = gst_buffer_peek_memory (outbuf, 0);
  const gboolean in_dev = gst_buffer_n_memory (inbuf) == 1 && gst_is_amd_hip_memory (imem);
  const gboolean out_dev = gst_buffer_n_memory (outbuf) == 1 && gst_is_amd_hip_memory (omem);
  gpointer sp[GSTAMD_VIDEO_MAX_PLANES], dp[GSTAMD_VIDEO_MAX_PLANES], stream;
  gint32 ss[GSTAMD_VIDEO_MAX_PLANES], ds[GSTAMD_VIDEO_MAX_PLANES];
  guint8 *src, *dst;
  guint k;
  int r;
  GstAmdVideoConverter *conv;

  gint64 ta = 0, tb = 0, tc = 0, td = 0;
  if (!s->convert || !s->n_streams)
    return GST_FLOW_NOT_NEGOTIATED;
  if (s->stats)
    ta = g_get_monotonic_time ();
  gst_amd_hip_select_device (s->device_id);       /* the streaming thread's current device */
  gst_amd_hip_pending_reads_retire (s->reads);
  /* interlace-mode=mixed: gst_video_frame_map marks the frames of buffers with GST_VIDEO_BUFFER_FLAG_INTERLACED (video-frame.c) */
  conv = s->convert_i && GST_BUFFER_FLAG_IS_SET (inbuf, GST_VIDEO_BUFFER_FLAG_INTERLACED) ? s->convert_i : s->convert;
  if (s->batch && s->batch_limit > 1) {
    if (conv == s->convert && in_dev && out_dev && buffer_is_plain_hip_frame (inbuf, &s->in_info) && buffer_is_plain_hip_frame (outbuf, &s->out_info)) {
      if (amd_vcs_batch_add (s, inbuf, outbuf)) {
        if (s->stats) {
          s->t_total += g_get_monotonic_time () - ta;
          s->n_frames++;
        }
        return GST_FLOW_OK;
      }
      return GST_FLOW_ERROR;
    }
    amd_vcs_batch_flush (s->batch, FALSE);        /* a frame that takes the direct path: keep the order of the launches */
  }
  k = s->next_stream++ % s->n_streams;
  stream = s->streams[k];
  /* source */
  if (in_dev) {
    if (!gst_memory_map (imem, &imap, GST_MAP_READ | GST_MAP_AMDHIP))
      return GST_FLOW_ERROR;
    gst_amd_hip_memory_wait_written (imem, stream);
    src = imap.data;
  } else {
    if (!gst_buffer_map (inbuf, &imap, GST_MAP_READ))
      return GST_FLOW_ERROR;
    if (!ensure_staging (&s->d_in[k], &s->d_in_size[k], imap.size) ||
        gstamd_device_upload_async (s->d_in[k], imap.data, imap.size, stream) != GSTAMD_OK) {
      gst_buffer_unmap (inbuf, &imap);
      return GST_FLOW_ERROR;
    }
    /* from pageable memory the copy call returns once the source has been staged; from page-locked memory - the pool this element offers
     * upstream (propose_allocation) - it is only QUEUED: the input stays referenced, out of its pool, until the transfer is over */
    if (!s->reads)
      s->reads = gst_amd_hip_pending_reads_new ();
    gst_amd_hip_pending_reads_hold (s->reads, inbuf, imap.data, stream);
    src = s->d_in[k];
  }
  /* destination */
  if (out_dev) {
    if (!gst_memory_map (omem, &omap, GST_MAP_WRITE | GST_MAP_AMDHIP))
      goto map_fail;
    gst_amd_hip_memory_wait_idle (omem, stream);    /* a recycled pool buffer may still be read downstream */
    dst = omap.data;
  } else {
    if (!gst_buffer_map (outbuf, &omap, GST_MAP_WRITE))
      goto map_fail;
    if (!ensure_staging (&s->d_out[k], &s->d_out_size[k], omap.size)) {
      gst_buffer_unmap (outbuf, &omap);
      goto map_fail;
    }
    dst = s->d_out[k];
  }

  GST_CAT_DEBUG_OBJECT (CAT_PERFORMANCE, s, "HIP convert %s on stream %u", gstamd_video_converter_describe (conv), k);
  frame_planes (inbuf, &s->in_info, src, sp, ss);
  frame_planes (outbuf, &s->out_info, dst, dp, ds);
  if (s->stats)
    tb = g_get_monotonic_time ();
  r = gstamd_video_converter_frame_planes (conv, (const void *const *) sp, ss, dp, ds, stream);
  if (s->stats)
    tc = g_get_monotonic_time ();
  if (r == GSTAMD_OK && (out_dev || in_dev)) {
    GstAmdHipTicket *t = s->n_streams == 1 ? gst_amd_hip_ticket_new_lazy (stream) : gst_amd_hip_ticket_new (stream);       /* one event for the launch, shared by both buffers */
    if (out_dev)
      gst_amd_hip_memory_set_written (omem, t);
    if (in_dev)
      gst_amd_hip_memory_set_read (imem, t);
    gst_amd_hip_ticket_unref (t);
  }
  if (s->stats)
    td = g_get_monotonic_time ();
  if (r == GSTAMD_OK && !out_dev) {
    r = gstamd_device_download_async (omap.data, s->d_out[k], omap.size, stream);
    if (r == GSTAMD_OK)
      r = gstamd_stream_synchronize (stream);         /* the CPU is about to look at outbuf */
    gst_amd_hip_pending_reads_retire (s->reads);      /* ... and this stream's upload is over: its input goes back to its pool now */
  }

  if (out_dev)
    gst_memory_unmap (omem, &omap);
  else
    gst_buffer_unmap (outbuf, &omap);
  if (in_dev)
    gst_memory_unmap (imem, &imap);
  else
    gst_buffer_unmap (inbuf, &imap);
  if (r != GSTAMD_OK) {
    GST_ELEMENT_ERROR (s, LIBRARY, FAILED, ("HIP conversion failed"), ("%s", gstamd_last_error ()));
    return GST_FLOW_ERROR;
  }
  if (s->stats) {
    s->t_wait += tb - ta;
    s->t_convert += tc - tb;
    s->t_mark += td - tc;
    s->t_total += g_get_monotonic_time () - ta;
    s->n_frames++;
  }
  return GST_FLOW_OK;

map_fail:
  if (in_dev)
    gst_memory_unmap (imem, &imap);
  else
    gst_buffer_unmap (inbuf, &imap);
  return GST_FLOW_ERROR;
}

/* A GstBufferList arriving on the sink pad (gst_pad_push_list; the core would otherwise feed it to chain() buffer by buffer,
 * gstpad.c gst_pad_chain_list_default): the list becomes ONE gstamd_video_converter_frames call (one kernel launch where the plan
 * allows it), one ticket, and one GstBufferList pushed downstream (while a (re)negotiation is pending the first buffer takes the
 * regular GstBaseTransform path and settles it).  The per-launch host cost (~7 us of HIP runtime per kernel on
 * this stack) is then paid once per list instead of once per frame. */
static gboolean
buffer_is_plain_hip_frame (GstBuffer * buf, const GstVideoInfo * info)
{
  GstVideoMeta *vm;
  guint i;

  if (gst_buffer_n_memory (buf) != 1 || !gst_is_amd_hip_memory (gst_buffer_peek_memory (buf, 0)))
    return FALSE;
  vm = gst_buffer_get_video_meta (buf);
  if (!vm)
    return TRUE;
  if (vm->n_planes != GST_VIDEO_INFO_N_PLANES (info))
    return FALSE;
  for (i = 0; i < vm->n_planes; i++)
    if (vm->offset[i] != GST_VIDEO_INFO_PLANE_OFFSET (info, i) || vm->stride[i] != GST_VIDEO_INFO_PLANE_STRIDE (info, i))
      return FALSE;
  return TRUE;
}

#define AMD_LIST_CHUNK 32

static GstFlowReturn
amd_vcs_chain_list (GstPad * pad, GstObject * parent, GstBufferList * list)
{
  GstAmdVideoConvertScale *s = AMD_VCS (parent);
  GstBaseTransform *trans = GST_BASE_TRANSFORM (parent);
  const guint n = gst_buffer_list_length (list);
  GstFlowReturn ret = GST_FLOW_OK;
  guint i = 0;

  if (n == 0) {
    gst_buffer_list_unref (list);
    return GST_FLOW_OK;
  }
  if (s->batch)
    amd_vcs_batch_flush (s->batch, FALSE);
  /* a negotiated, stable element (caps events are handled before the buffers that follow them; a pending downstream
   * reconfigure shows on the src pad) converts the whole list at once; otherwise the first buffer takes the regular path and
   * settles the negotiation */
  if (!(s->convert && s->n_streams && !gst_pad_needs_reconfigure (GST_BASE_TRANSFORM_SRC_PAD (trans)))) {
    ret = s->base_chain (pad, parent, gst_buffer_ref (gst_buffer_list_get (list, 0)));
    i = 1;
  }
  while (ret == GST_FLOW_OK && i < n) {
    gboolean batch = s->convert && s->n_streams && s->in_hip && s->out_hip && s->out_pool && !gst_base_transform_is_passthrough (trans) &&
        !gst_pad_needs_reconfigure (GST_BASE_TRANSFORM_SRC_PAD (trans));
    guint cnt = MIN (n - i, AMD_LIST_CHUNK), k;
    for (k = 0; batch && k < cnt; k++)
      batch = buffer_is_plain_hip_frame (gst_buffer_list_get (list, i + k), &s->in_info) &&
          !(s->convert_i && GST_BUFFER_FLAG_IS_SET (gst_buffer_list_get (list, i + k), GST_VIDEO_BUFFER_FLAG_INTERLACED));
    if (!batch) {
      ret = s->base_chain (pad, parent, gst_buffer_ref (gst_buffer_list_get (list, i)));
      i++;
      continue;
    }
    {
      GstBuffer *outs[AMD_LIST_CHUNK];
      GstMapInfo imaps[AMD_LIST_CHUNK], omaps[AMD_LIST_CHUNK];
      const void *srcs[AMD_LIST_CHUNK];
      void *dsts[AMD_LIST_CHUNK];
      gpointer stream;
      guint got = 0, mapped = 0;
      int r = GSTAMD_OK;
      GstFlowReturn pool_flow = GST_FLOW_OK;

      gst_amd_hip_select_device (s->device_id);
      stream = s->streams[s->next_stream++ % s->n_streams];
      for (k = 0; k < cnt; k++) {
        if ((pool_flow = gst_buffer_pool_acquire_buffer (s->out_pool, &outs[k], NULL)) != GST_FLOW_OK)
          break;
        got++;
        gst_buffer_copy_into (outs[k], gst_buffer_list_get (list, i + k), GST_BUFFER_COPY_FLAGS | GST_BUFFER_COPY_TIMESTAMPS, 0, -1);
      }
      for (k = 0; k < got; k++) {
        GstMemory *im = gst_buffer_peek_memory (gst_buffer_list_get (list, i + k), 0), *om = gst_buffer_peek_memory (outs[k], 0);
        if (!gst_memory_map (im, &imaps[k], GST_MAP_READ | GST_MAP_AMDHIP))
          break;
        if (!gst_memory_map (om, &omaps[k], GST_MAP_WRITE | GST_MAP_AMDHIP)) {
          gst_memory_unmap (im, &imaps[k]);
          break;
        }
        gst_amd_hip_memory_wait_written (im, stream);
        gst_amd_hip_memory_wait_idle (om, stream);
        srcs[k] = imaps[k].data;
        dsts[k] = omaps[k].data;
        mapped++;
      }
      if (mapped == cnt) {
        r = gstamd_video_converter_frames (s->convert, (int) cnt, srcs, dsts, stream);
        s->n_list_calls++;
        s->n_list_launches += (guint64) gstamd_video_converter_list_launches (s->convert);
      } else
        r = GSTAMD_ERR_INVALID;
      if (r == GSTAMD_OK) {
        GstAmdHipTicket *t = s->n_streams == 1 ? gst_amd_hip_ticket_new_lazy (stream) : gst_amd_hip_ticket_new (stream);
        for (k = 0; k < cnt; k++) {
          gst_amd_hip_memory_set_read (gst_buffer_peek_memory (gst_buffer_list_get (list, i + k), 0), t);
          gst_amd_hip_memory_set_written (gst_buffer_peek_memory (outs[k], 0), t);
        }
        gst_amd_hip_ticket_unref (t);
      }
      for (k = 0; k < mapped; k++) {
        gst_memory_unmap (gst_buffer_peek_memory (gst_buffer_list_get (list, i + k), 0), &imaps[k]);
        gst_memory_unmap (gst_buffer_peek_memory (outs[k], 0), &omaps[k]);
      }
      if (r != GSTAMD_OK) {
        for (k = 0; k < got; k++)
          gst_buffer_unref (outs[k]);
        if (pool_flow != GST_FLOW_OK && pool_flow != GST_FLOW_ERROR) {
          ret = pool_flow;      /* the pool is flushing (seek, state change): pass that on, nothing failed */
          break;
        }
        GST_ELEMENT_ERROR (s, LIBRARY, FAILED, ("HIP conversion of a buffer list failed"), ("%s", gstamd_last_error ()));
        ret = GST_FLOW_ERROR;
        break;
      }
      {
        GstBufferList *out_list = gst_buffer_list_new_sized (cnt);
        for (k = 0; k < cnt; k++)
          gst_buffer_list_add (out_list, outs[k]);
        ret = gst_pad_push_list (GST_BASE_TRANSFORM_SRC_PAD (trans), out_list);
      }
      i += cnt;
    }
  }
  gst_buffer_list_unref (list);
  return ret;
}

/* gst_video_convert_scale_src_event (gstvideoconvertscale.c:2008-2037): a navigation event travelling upstream names a point of the OUTPUT picture;
 * upstream of a scaler it is the point of the input picture that lands there.  (The coordinates are the event structure's pointer_x / pointer_y
 * fields on every GStreamer version; gst_navigation_event_set_coordinates, 1.22, writes the same two fields.) */
static gboolean
amd_vcs_src_event (GstBaseTransform * trans, GstEvent * event)
{
  GstAmdVideoConvertScale *s = AMD_VCS (trans);

  if (GST_EVENT_TYPE (event) == GST_EVENT_NAVIGATION && s->convert &&
      (s->in_info.width != s->out_info.width || s->in_info.height != s->out_info.height) && s->out_info.width > 0 && s->out_info.height > 0) {
    GstStructure *st;
    gdouble x, y;
    event = gst_event_make_writable (event);
    st = gst_event_writable_structure (event);
    if (st && gst_structure_get_double (st, "pointer_x", &x) && gst_structure_get_double (st, "pointer_y", &y))
      gst_structure_set (st, "pointer_x", G_TYPE_DOUBLE, x * s->in_info.width / s->out_info.width,
          "pointer_y", G_TYPE_DOUBLE, y * s->in_info.height / s->out_info.height, NULL);
  }
  return GST_BASE_TRANSFORM_CLASS (gst_amd_vcs_parent_class)->src_event (trans, event);
}

/* gst_video_convert_scale_transform_meta (gstvideoconvertscale.c:773-829): metas whose tags are all of {video, orientation, size} survive the
 * conversion - copied as they are, or, when they carry the size tag, through their own transform function (GstVideoMetaTransformMatrix on 1.28+ with
 * the picture's place inside the borders, else the older "gst-video-scale" transform with the two infos: video crop, region-of-interest ...);
 * colorspace-tagged and any other metas go to the base class's rule */
static gboolean
amd_vcs_transform_meta (GstBaseTransform * trans, GstBuffer * outbuf, GstMeta * meta, GstBuffer * inbuf)
{
  GstAmdVideoConvertScale *s = AMD_VCS (trans);
  const GstMetaInfo *info = meta->info;
  const gchar *const *tags = gst_meta_api_type_get_tags (info->api);
  static const gchar *const valid[] = { GST_META_TAG_VIDEO_STR, GST_META_TAG_VIDEO_ORIENTATION_STR, GST_META_TAG_VIDEO_SIZE_STR, NULL };
  guint i, k;

  /* (gst_meta_api_type_tags_contain_only, 1.24: every tag of the API is one of the list; an API without tags qualifies) */
  for (i = 0; tags && tags[i]; i++) {
    for (k = 0; valid[k] && strcmp (tags[i], valid[k]) != 0; k++);
    if (!valid[k])
      return GST_BASE_TRANSFORM_CLASS (gst_amd_vcs_parent_class)->transform_meta (trans, outbuf, meta, inbuf);
  }
  if (gst_meta_api_type_has_tag (info->api, g_quark_from_static_string (GST_META_TAG_VIDEO_SIZE_STR))) {
    if (info->transform_func) {
      GstVideoMetaTransform sc = { &s->in_info, &s->out_info };
#if GST_CHECK_VERSION (1, 27, 0)
      GstVideoMetaTransformMatrix mx;
      const GstVideoRectangle in_rect = { 0, 0, GST_VIDEO_INFO_WIDTH (&s->in_info), GST_VIDEO_INFO_HEIGHT (&s->in_info) };
      const GstVideoRectangle out_rect = { s->borders_w / 2, s->borders_h / 2, GST_VIDEO_INFO_WIDTH (&s->out_info) - s->borders_w,
        GST_VIDEO_INFO_HEIGHT (&s->out_info) - s->borders_h
      };
      gst_video_meta_transform_matrix_init (&mx, &s->in_info, &in_rect, &s->out_info, &out_rect);
      if (info->transform_func (outbuf, meta, inbuf, gst_video_meta_transform_matrix_get_quark (), &mx))
        return FALSE;
#endif
      info->transform_func (outbuf, meta, inbuf, gst_video_meta_transform_scale_get_quark (), &sc);
    }
    return FALSE;          /* transformed (or not transformable): never copied as it is */
  }
  return TRUE;
}

static gboolean
amd_vcs_stop (GstBaseTransform * trans)
{
  GstAmdVideoConvertScale *s = AMD_VCS (trans);
  gst_amd_hip_select_device (s->device_id);
  amd_vcs_batch_stop (s);
  if (s->convert)
    gstamd_video_converter_free (s->convert);
  s->convert = NULL;
  if (s->convert_i)
    gstamd_video_converter_free (s->convert_i);
  s->convert_i = NULL;
  if (s->out_pool) {
    gst_buffer_pool_set_active (s->out_pool, FALSE);
    gst_object_unref (s->out_pool);
    s->out_pool = NULL;
  }
  gst_amd_hip_select_device (s->device_id);
  if (s->stats && s->n_frames)
    g_printerr ("videoconvertscale host time per buffer over %" G_GUINT64_FORMAT " buffers: prepare_output %.2f us, transform %.2f us "
        "(map + stream waits %.2f, converter call %.2f, event records %.2f)\n", s->n_frames, (double) s->t_prepare / s->n_frames,
        (double) s->t_total / s->n_frames, (double) s->t_wait / s->n_frames, (double) s->t_convert / s->n_frames,
        (double) s->t_mark / s->n_frames);
  if (s->stats && s->n_list_calls)
    g_printerr ("videoconvertscale buffer lists: %" G_GUINT64_FORMAT " converter calls, %" G_GUINT64_FORMAT " list launches\n", s->n_list_calls,
        s->n_list_launches);
  s->n_list_calls = s->n_list_launches = 0;
  if (s->reads) {
    gst_amd_hip_pending_reads_free (s->reads);          /* waits for the queued uploads, gives their input buffers back */
    s->reads = NULL;
  }
  {
    guint i;
    for (i = 0; i < AMD_MAX_STREAMS; i++) {
      if (s->streams[i]) {
        gst_amd_hip_stream_retire (s->streams[i]);      /* waits for it; lazy tickets of the stream count as done from here on */
        gstamd_stream_free (s->streams[i]);
      }
      s->streams[i] = NULL;
      gstamd_device_free (s->d_in[i]);
      gstamd_device_free (s->d_out[i]);
      s->d_in[i] = s->d_out[i] = NULL;
      s->d_in_size[i] = s->d_out_size[i] = 0;
    }
    s->n_streams = 0;
  }
  return TRUE;
}

static GType
amd_enum_type (const gchar * name, const GEnumValue * values)
{
  GType t = g_type_from_name (name);
  return t ? t : g_enum_register_static (name, values);
}

static void
gst_amd_vcs_class_init (GstAmdVideoConvertScaleClass * klass)
{
  GObjectClass *oc = G_OBJECT_CLASS (klass);
  GstElementClass *ec = GST_ELEMENT_CLASS (klass);
  GstBaseTransformClass *tc = GST_BASE_TRANSFORM_CLASS (klass);
  static const GEnumValue alpha_v[] = {{0, "copy", "copy"}, {1, "set", "set"}, {2, "mult", "mult"}, {0, NULL, NULL}};
  static const GEnumValue chroma_v[] = {{0, "full", "full"}, {1, "upsample-only", "upsample-only"},
    {2, "downsample-only", "downsample-only"}, {3, "none", "none"}, {0, NULL, NULL}};
  static const GEnumValue matrix_v[] = {{0, "full", "full"}, {1, "input-only", "input-only"},
    {2, "output-only", "output-only"}, {3, "none", "none"}, {0, NULL, NULL}};
  /* GstVideoGammaMode / GstVideoPrimariesMode (video-converter.h:176-206) */
  static const GEnumValue gamma_v[] = {{0, "disable gamma handling", "none"}, {1, "convert between input and output gamma", "remap"}, {0, NULL, NULL}};
  static const GEnumValue primaries_v[] = {{0, "disable conversion between primaries", "none"},
    {1, "do conversion between primaries only when it can be merged with color matrix conversion", "merge-only"},
    {2, "fast conversion between primaries", "fast"}, {0, NULL, NULL}};

  gst_amd_converter_config_register_types ();

  GST_DEBUG_CATEGORY_INIT (amd_vcs_debug, "amdvideoconvertscale", 0, "MI355X videoconvertscale");
  GST_DEBUG_CATEGORY_GET (CAT_PERFORMANCE, "GST_PERFORMANCE");
  oc->set_property = amd_vcs_set_property;
  oc->get_property = amd_vcs_get_property;
  oc->finalize = amd_vcs_finalize;
  g_object_class_install_property (oc, PROP_METHOD, g_param_spec_enum ("method", "method", "method",
          amd_scale_method_get_type (), AMD_SCALE_BILINEAR, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_ADD_BORDERS, g_param_spec_boolean ("add-borders", "Add Borders",
          "Add black borders if necessary to keep the display aspect ratio", TRUE, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_N_THREADS, g_param_spec_uint ("n-threads", "Threads",
          "Accepted for compatibility (the GPU grid replaces CPU thread slices)", 0, G_MAXUINT, 1,
          G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_ALPHA_MODE, g_param_spec_enum ("alpha-mode", "Alpha Mode",
          "Alpha Mode to use", amd_enum_type ("GstAmdVideoAlphaMode", alpha_v), 0, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_ALPHA_VALUE, g_param_spec_double ("alpha-value", "Alpha Value",
          "Alpha Value to use", 0.0, 1.0, 1.0, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_CHROMA_MODE, g_param_spec_enum ("chroma-mode", "Chroma Mode",
          "Chroma Resampling Mode", amd_enum_type ("GstAmdVideoChromaMode", chroma_v), 0, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_MATRIX_MODE, g_param_spec_enum ("matrix-mode", "Matrix Mode",
          "Matrix Conversion Mode", amd_enum_type ("GstAmdVideoMatrixMode", matrix_v), 0, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_GAMMA_MODE, g_param_spec_enum ("gamma-mode", "Gamma Mode",
          "Gamma Conversion Mode", amd_enum_type ("GstAmdVideoGammaMode", gamma_v), 0, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_PRIMARIES_MODE, g_param_spec_enum ("primaries-mode", "Primaries Mode",
          "Primaries Conversion Mode", amd_enum_type ("GstAmdVideoPrimariesMode", primaries_v), 0, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_ENVELOPE, g_param_spec_double ("envelope", "Envelope",
          "Size of filter envelope", 1.0, 5.0, 2.0, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_SHARPNESS, g_param_spec_double ("sharpness", "Sharpness",
          "Sharpness of filter", 0.5, 1.5, 1.0, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_SHARPEN, g_param_spec_double ("sharpen", "Sharpen",
          "Sharpening", 0.0, 1.0, 0.0, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_DITHER_QUANTIZATION, g_param_spec_uint ("dither-quantization",
          "Dither Quantize", "Quantizer to use", 0, G_MAXUINT, 1, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));

  /* beyond the reference's properties: which GPU (one stream per GPU, device-id per element instance) and how many HIP
   * streams the instance rotates its frames over */
  g_object_class_install_property (oc, PROP_DEVICE_ID, g_param_spec_int ("device-id", "Device ID",
          "HIP device this instance runs on (-1 = the process's current device)", -1, G_MAXINT, -1,
          G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_DITHER, g_param_spec_enum ("dither", "Dither", "Apply dithering while converting",
          GST_TYPE_VIDEO_DITHER_METHOD, GST_VIDEO_DITHER_BAYER, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  {
    static GType rm_type = 0;
    static const GEnumValue rm_values[] = {
      {GSTAMD_RESAMPLER_METHOD_NEAREST, "Duplicates the samples when upsampling and drops when downsampling", "nearest"},
      {GSTAMD_RESAMPLER_METHOD_LINEAR, "Uses linear interpolation to reconstruct missing samples and averaging to downsample", "linear"},
      {GSTAMD_RESAMPLER_METHOD_CUBIC, "Uses cubic interpolation", "cubic"},
      {GSTAMD_RESAMPLER_METHOD_SINC, "Uses sinc interpolation", "sinc"},
      {GSTAMD_RESAMPLER_METHOD_LANCZOS, "Uses lanczos interpolation", "lanczos"}, {0, NULL, NULL}};
    /* the library's own GstVideoResamplerMethod type where the runtime has it registered (same nicks and values) */
    if (!rm_type && !(rm_type = g_type_from_name ("GstVideoResamplerMethod")))
      rm_type = g_enum_register_static ("GstAmdVideoResamplerMethod", rm_values);
    g_object_class_install_property (oc, PROP_CHROMA_RESAMPLER, g_param_spec_enum ("chroma-resampler", "Chroma resampler", "Chroma resampler method",
            rm_type, GSTAMD_RESAMPLER_METHOD_LINEAR, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  }
  g_object_class_install_property (oc, PROP_CONVERTER_CONFIG, g_param_spec_boxed ("converter-config", "Converter configuration",
          "A GstStructure describing the configuration that should be used. This configuration, if set, takes precedence over the "
          "other similar conversion properties.", GST_TYPE_STRUCTURE, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_HIP_STREAMS, g_param_spec_uint ("hip-streams", "HIP streams",
          "HIP streams the instance rotates its frames over (the launch ramp of one frame overlaps the tail of the previous one)",
          1, AMD_MAX_STREAMS, 3, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));

  g_object_class_install_property (oc, PROP_BATCH_BUFFERS, g_param_spec_uint ("batch-buffers", "Buffers per launch",
          "HBM frames of consecutive buffers converted by one kernel launch (the launch is deferred until that many have arrived, "
          "somebody needs one of them, or 2 ms have passed); 0 = 4 when upstream is not live, 1 otherwise; 1 = one launch per buffer",
          0, AMD_BATCH_MAX, 0, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));

  gst_element_class_set_static_metadata (ec, "Video colorspace converter and scaler (MI355X/HIP)",
      "Filter/Converter/Video/Scaler/Colorspace",
      "Converts and resizes raw video on an AMD Instinct GPU with kernels bit-exact to GstVideoConverter", "gstreamer_amd");
  gst_element_class_add_static_pad_template (ec, &sink_tmpl);
  gst_element_class_add_static_pad_template (ec, &src_tmpl);

  tc->passthrough_on_same_caps = TRUE;
  tc->transform_caps = GST_DEBUG_FUNCPTR (amd_vcs_transform_caps);
  tc->fixate_caps = GST_DEBUG_FUNCPTR (amd_vcs_fixate_caps);
  tc->set_caps = GST_DEBUG_FUNCPTR (amd_vcs_set_caps);
  tc->get_unit_size = GST_DEBUG_FUNCPTR (amd_vcs_get_unit_size);
  tc->prepare_output_buffer = GST_DEBUG_FUNCPTR (amd_vcs_prepare_output_buffer);
  tc->propose_allocation = GST_DEBUG_FUNCPTR (amd_vcs_propose_allocation);
  tc->decide_allocation = GST_DEBUG_FUNCPTR (amd_vcs_decide_allocation);
  tc->transform = GST_DEBUG_FUNCPTR (amd_vcs_transform);
  tc->stop = GST_DEBUG_FUNCPTR (amd_vcs_stop);
  tc->src_event = GST_DEBUG_FUNCPTR (amd_vcs_src_event);
  tc->transform_meta = GST_DEBUG_FUNCPTR (amd_vcs_transform_meta);
  tc->sink_event = GST_DEBUG_FUNCPTR (amd_vcs_sink_event);
  klass->converts = TRUE;
  klass->scales = TRUE;
}

static void
gst_amd_vcs_init (GstAmdVideoConvertScale * s)
{
  s->method = AMD_SCALE_BILINEAR;          /* DEFAULT_PROP_METHOD (:130) */
  s->chroma_resampler = GSTAMD_RESAMPLER_METHOD_LINEAR;        /* DEFAULT_PROP_CHROMA_RESAMPLER (:137) */
  s->add_borders = TRUE;                   /* DEFAULT_PROP_ADD_BORDERS (:131) */
  s->n_threads = 1;
  s->alpha_mode = GSTAMD_ALPHA_MODE_COPY;
  s->alpha_value = 1.0;
  s->chroma_mode = GSTAMD_CHROMA_MODE_FULL;
  s->matrix_mode = GSTAMD_MATRIX_MODE_FULL;
  s->envelope = 2.0;
  s->sharpness = 1.0;
  s->sharpen = 0.0;
  s->dither_quantization = 1;
  s->dither = GST_VIDEO_DITHER_BAYER;
  s->device_id = -1;
  s->hip_streams = 3;
  s->stats = g_getenv ("GSTAMD_ELEMENT_STATS") != NULL;
  s->pinned_pools = g_getenv ("GSTAMD_NO_PINNED_POOLS") == NULL;
  s->base_chain = GST_PAD_CHAINFUNC (GST_BASE_TRANSFORM_SINK_PAD (s));
  gst_pad_set_chain_list_function (GST_BASE_TRANSFORM_SINK_PAD (s), GST_DEBUG_FUNCPTR (amd_vcs_chain_list));
  gst_amd_hip_allocator_get ();
}

GType
gst_amd_video_convert_scale_get_type (void)
{
  return gst_amd_vcs_get_type ();
}

/* `videoconvert` and `videoscale`: subclasses that only convert / only scale, exactly as the reference derives them from
 * GstVideoConvertScale (gstvideoconvert.c:50-58 sets scales = FALSE, gstvideoscale.c:110-118 sets converts = FALSE) */
typedef GstAmdVideoConvertScale GstAmdVideoConvert;
typedef GstAmdVideoConvertScaleClass GstAmdVideoConvertClass;
typedef GstAmdVideoConvertScale GstAmdVideoScale;
typedef GstAmdVideoConvertScaleClass GstAmdVideoScaleClass;
G_DEFINE_TYPE (GstAmdVideoConvert, gst_amd_video_convert, gst_amd_vcs_get_type ());
G_DEFINE_TYPE (GstAmdVideoScale, gst_amd_video_scale, gst_amd_vcs_get_type ());

static void
gst_amd_video_convert_class_init (GstAmdVideoConvertClass * klass)
{
  klass->converts = TRUE;
  klass->scales = FALSE;
  gst_element_class_set_static_metadata (GST_ELEMENT_CLASS (klass), "Video colorspace converter (MI355X/HIP)", "Filter/Converter/Video/Colorspace",
      "Converts video from one colorspace to another on an AMD Instinct GPU, bit-exact to GstVideoConverter", "gstreamer_amd");
}

static void
gst_amd_video_convert_init (GstAmdVideoConvert * s)
{
}

static void
gst_amd_video_scale_class_init (GstAmdVideoScaleClass * klass)
{
  klass->converts = FALSE;
  klass->scales = TRUE;
  gst_element_class_set_static_metadata (GST_ELEMENT_CLASS (klass), "Video scaler (MI355X/HIP)", "Filter/Converter/Video/Scaler",
      "Resizes video on an AMD Instinct GPU, bit-exact to GstVideoConverter", "gstreamer_amd");
}

static void
gst_amd_video_scale_init (GstAmdVideoScale * s)
{
}

GType gst_amd_video_convert_element_get_type (void) { return gst_amd_video_convert_get_type (); }
GType gst_amd_video_scale_element_get_type (void) { return gst_amd_video_scale_get_type (); }
