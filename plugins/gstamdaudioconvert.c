/* gstamdaudioconvert.c - `audioconvert` element backed by the device GstAudioConverter (include/gstamd_audio.h gstamd_audio_converter_*).
 *
 * Mirrors the reference element's contract for this path (subprojects/gst-plugins-base/gst/audioconvert/gstaudioconvert.c):
 * factory name / rank PRIMARY (gstaudioconvert.c:190-195), properties dithering (default tpdf), noise-shaping, mix-matrix,
 * dithering-threshold (:350-391), the caps transformation that frees format, layout and - for positioned layouts or with a
 * mix-matrix - channels (:1094-1196), passthrough-first fixation (:1498-1546), a converter made per caps with the element's
 * properties as its config (:1548-1636), silence for GAP buffers (:1733-1747), the GstRequestAudioMixMatrix upstream event
 * (:309-337).  Samples are staged to HBM per buffer like the audioresample element's.
 * Not implemented: input-channels-reorder / -mode, non-interleaved layouts, the depth / sign scoring of fixate_format (:1237-1340;
 * this element prefers the input's format, then the widest one the peer offers of the same kind).
 */
#include <gst/audio/audio.h>
#include <gst/base/gstbasetransform.h>
#include <gst/gst.h>
#include <string.h>

#include "../include/gstamd_audio.h"
#include "../include/gstamd_video.h"
#include "gstamdhipmemory.h"

GST_DEBUG_CATEGORY_STATIC (amd_ac_debug);
#define GST_CAT_DEFAULT amd_ac_debug

#define AMD_AC_CAPS "audio/x-raw, format = (string) { F64LE, F32LE, S32LE, S24_32LE, S24LE, S16LE, S8, U8 }, rate = (int) [ 1, MAX ], " \
    "channels = (int) [ 1, 8 ], layout = (string) interleaved"

static GstStaticPadTemplate ac_sink = GST_STATIC_PAD_TEMPLATE ("sink", GST_PAD_SINK, GST_PAD_ALWAYS, GST_STATIC_CAPS (AMD_AC_CAPS));
static GstStaticPadTemplate ac_src = GST_STATIC_PAD_TEMPLATE ("src", GST_PAD_SRC, GST_PAD_ALWAYS, GST_STATIC_CAPS (AMD_AC_CAPS));

typedef struct {
  GstBaseTransform parent;
  gint dither, ns;
  guint dither_threshold;
  GValue mix_matrix;            /* GST_TYPE_ARRAY of rows ([out][in]) */
  gboolean mix_matrix_is_set;
  gboolean conv_dirty;          /* options changed: the streaming thread re-makes the converter (object lock) */
  gboolean have_caps;
  GstAudioInfo in, out;
  GstAmdAudioConverter *conv;
  gpointer d_in, d_out;
  gsize d_in_size, d_out_size;
  gpointer stream;
  gint device_id;
} GstAmdAudioConvert;
typedef struct { GstBaseTransformClass parent_class; } GstAmdAudioConvertClass;

enum { PROP_0, PROP_DITHERING, PROP_NOISE_SHAPING, PROP_MIX_MATRIX, PROP_DITHERING_THRESHOLD, PROP_DEVICE_ID };
G_DEFINE_TYPE (GstAmdAudioConvert, gst_amd_ac, GST_TYPE_BASE_TRANSFORM);
#define AMD_AC(o) ((GstAmdAudioConvert *) (o))

static void
amd_ac_drop_converter (GstAmdAudioConvert * s)
{
  if (s->conv)
    gstamd_audio_converter_free (s->conv);
  s->conv = NULL;
}

/* gst_audio_convert_set_mix_matrix (gstaudioconvert.c:1858-1907): an empty array keeps "set" (the converter then makes a truncated
 * identity); rows must have one length */
static void
amd_ac_set_mix_matrix (GstAmdAudioConvert * s, const GValue * value)
{
  gboolean ok = TRUE;
  guint i;

  GST_OBJECT_LOCK (s);
  if (G_IS_VALUE (&s->mix_matrix))
    g_value_unset (&s->mix_matrix);
  g_value_init (&s->mix_matrix, GST_TYPE_ARRAY);
  if (gst_value_array_get_size (value)) {
    const guint cols = gst_value_array_get_size (gst_value_array_get_value (value, 0));
    for (i = 1; i < gst_value_array_get_size (value); i++)
      ok = ok && gst_value_array_get_size (gst_value_array_get_value (value, i)) == cols;
    if (ok)
      g_value_copy (value, &s->mix_matrix);
    else
      g_warning ("Invalid mix-matrix: rows of different lengths");
  }
  s->mix_matrix_is_set = ok;
  /* the converter is NOT freed here: this runs on the application's (or, for GstRequestAudioMixMatrix, an upstream) thread while the
   * streaming thread may be inside gstamd_audio_converter_samples.  It is flagged and re-made by the streaming thread at the top of the
   * next transform - the reference re-creates it lazily there too (gstaudioconvert.c:1700 gst_audio_convert_ensure_converter) - and the
   * element leaves passthrough so that transform is called at all (gstaudioconvert.c:1883-1886) */
  s->conv_dirty = TRUE;
  GST_OBJECT_UNLOCK (s);
  gst_base_transform_set_passthrough (GST_BASE_TRANSFORM (s), FALSE);
  if (ok)
    gst_base_transform_reconfigure_sink (GST_BASE_TRANSFORM (s));
}

static void
amd_ac_set_property (GObject * o, guint id, const GValue * v, GParamSpec * p)
{
  GstAmdAudioConvert *s = AMD_AC (o);
  switch (id) {
    case PROP_DITHERING: s->dither = g_value_get_enum (v); break;
    case PROP_NOISE_SHAPING: s->ns = g_value_get_enum (v); break;
    case PROP_DITHERING_THRESHOLD: s->dither_threshold = g_value_get_uint (v); break;
    case PROP_MIX_MATRIX: amd_ac_set_mix_matrix (s, v); break;
    case PROP_DEVICE_ID: s->device_id = g_value_get_int (v); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID (o, id, p); break;
  }
}

static void
amd_ac_get_property (GObject * o, guint id, GValue * v, GParamSpec * p)
{
  GstAmdAudioConvert *s = AMD_AC (o);
  switch (id) {
    case PROP_DITHERING: g_value_set_enum (v, s->dither); break;
    case PROP_NOISE_SHAPING: g_value_set_enum (v, s->ns); break;
    case PROP_DITHERING_THRESHOLD: g_value_set_uint (v, s->dither_threshold); break;
    case PROP_MIX_MATRIX:
      GST_OBJECT_LOCK (s);
      if (s->mix_matrix_is_set && G_IS_VALUE (&s->mix_matrix))
        g_value_copy (&s->mix_matrix, v);
      GST_OBJECT_UNLOCK (s);
      break;
    case PROP_DEVICE_ID: g_value_set_int (v, s->device_id); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID (o, id, p); break;
  }
}

/* gst_audio_convert_transform_caps (gstaudioconvert.c:1148-1196) */
static GstCaps *
amd_ac_transform_caps (GstBaseTransform * t, GstPadDirection direction, GstCaps * caps, GstCaps * filter)
{
  GstAmdAudioConvert *s = AMD_AC (t);
  GstCaps *tmp = gst_caps_copy (caps);
  guint i;
  gint other_channels = 0;

  GST_OBJECT_LOCK (s);
  if (s->mix_matrix_is_set && gst_value_array_get_size (&s->mix_matrix))
    other_channels = direction == GST_PAD_SRC ? (gint) gst_value_array_get_size (gst_value_array_get_value (&s->mix_matrix, 0)) :
        (gint) gst_value_array_get_size (&s->mix_matrix);
  for (i = 0; i < gst_caps_get_size (tmp); i++) {
    GstStructure *st = gst_caps_get_structure (tmp, i);
    guint64 mask;
    gint channels;
    gst_structure_remove_field (st, "format");
    gst_structure_remove_field (st, "layout");
    /* channels stay only for an unpositioned layout of more than one channel without a mix-matrix (remove_channels_from_structure) */
    if (s->mix_matrix_is_set || !gst_structure_get (st, "channel-mask", GST_TYPE_BITMASK, &mask, NULL) ||
        (mask != 0 || (gst_structure_get_int (st, "channels", &channels) && channels == 1)))
      gst_structure_remove_fields (st, "channel-mask", "channels", NULL);
    if (other_channels)
      gst_structure_set (st, "channels", G_TYPE_INT, other_channels, NULL);
  }
  GST_OBJECT_UNLOCK (s);
  if (filter) {
    GstCaps *tmp2 = gst_caps_intersect_full (filter, tmp, GST_CAPS_INTERSECT_FIRST);
    gst_caps_unref (tmp);
    tmp = tmp2;
  }
  return tmp;
}

static gint
amd_ac_format_score (const GstAudioFormatInfo * in, const GstAudioFormatInfo * f)
{
  /* the input's own format first; then the same kind (integer / float) without losing depth, as narrow as possible; then wider kinds */
  gint score = 0;
  if (f->format == in->format)
    return 1 << 20;
  if (GST_AUDIO_FORMAT_INFO_IS_INTEGER (f) == GST_AUDIO_FORMAT_INFO_IS_INTEGER (in))
    score += 1 << 10;
  if (GST_AUDIO_FORMAT_INFO_DEPTH (f) >= GST_AUDIO_FORMAT_INFO_DEPTH (in))
    score += (1 << 9) - GST_AUDIO_FORMAT_INFO_DEPTH (f);
  else
    score += GST_AUDIO_FORMAT_INFO_DEPTH (f);
  return score;
}

static void
amd_ac_fixate_format (GstStructure * ins, GstStructure * outs)
{
  const gchar *in_name = gst_structure_get_string (ins, "format");
  const GValue *fv = gst_structure_get_value (outs, "format");
  const GstAudioFormatInfo *in_info, *best = NULL;
  gint best_score = -1;
  guint i;

  if (!in_name || !fv || !GST_VALUE_HOLDS_LIST (fv))
    return;
  in_info = gst_audio_format_get_info (gst_audio_format_from_string (in_name));
  for (i = 0; i < gst_value_list_get_size (fv); i++) {
    const GValue *v = gst_value_list_get_value (fv, i);
    const GstAudioFormatInfo *f;
    gint score;
    if (!G_VALUE_HOLDS_STRING (v))
      continue;
    f = gst_audio_format_get_info (gst_audio_format_from_string (g_value_get_string (v)));
    if (!f || f->format == GST_AUDIO_FORMAT_UNKNOWN)
      continue;
    score = amd_ac_format_score (in_info, f);
    if (score > best_score) {
      best_score = score;
      best = f;
    }
  }
  if (best)
    gst_structure_set (outs, "format", G_TYPE_STRING, GST_AUDIO_FORMAT_INFO_NAME (best), NULL);
}

/* the channel count nearest to the input's; a positioned layout for it (the input's own mask when the count is the same, the
 * fallback mask otherwise) when the peer left the mask open */
static void
amd_ac_fixate_channels (GstStructure * ins, GstStructure * outs)
{
  gint in_ch = 0, out_ch = 0;
  guint64 in_mask = 0;
  const gboolean in_has_mask = gst_structure_get (ins, "channel-mask", GST_TYPE_BITMASK, &in_mask, NULL);

  if (!gst_structure_get_int (ins, "channels", &in_ch))
    return;
  if (!gst_structure_has_field (outs, "channels"))
    gst_structure_set (outs, "channels", G_TYPE_INT, in_ch, NULL);
  else if (!gst_structure_get_int (outs, "channels", &out_ch))
    gst_structure_fixate_field_nearest_int (outs, "channels", in_ch);
  if (!gst_structure_get_int (outs, "channels", &out_ch))
    return;
  if (out_ch <= 2) {
    /* mono and stereo need no mask (gst_audio_info_from_caps gives them their default positions) */
    if (gst_structure_has_field (outs, "channel-mask") && !gst_structure_get (outs, "channel-mask", GST_TYPE_BITMASK, &in_mask, NULL))
      gst_structure_remove_field (outs, "channel-mask");
    return;
  }
  if (!gst_structure_has_field (outs, "channel-mask") || !gst_structure_get (outs, "channel-mask", GST_TYPE_BITMASK, &in_mask, NULL)) {
    guint64 mask = 0;
    if (out_ch == in_ch && in_has_mask)
      gst_structure_get (ins, "channel-mask", GST_TYPE_BITMASK, &mask, NULL);
    else
      mask = gst_audio_channel_get_fallback_mask (out_ch);
    gst_structure_set (outs, "channel-mask", GST_TYPE_BITMASK, mask, NULL);
  }
}

/* gst_audio_convert_fixate_caps (gstaudioconvert.c:1498-1546): what lets the buffers pass untouched first */
static GstCaps *
amd_ac_fixate_caps (GstBaseTransform * t, GstPadDirection direction, GstCaps * caps, GstCaps * othercaps)
{
  GstCaps *result = gst_caps_intersect (othercaps, caps);
  GstStructure *ins, *outs;

  if (gst_caps_is_empty (result)) {
    gst_caps_unref (result);
    result = othercaps;
  } else {
    gst_caps_unref (othercaps);
  }
  result = gst_caps_truncate (gst_caps_make_writable (result));
  ins = gst_caps_get_structure (caps, 0);
  outs = gst_caps_get_structure (result, 0);
  amd_ac_fixate_channels (ins, outs);
  amd_ac_fixate_format (ins, outs);
  return gst_caps_fixate (result);
}

static gboolean
amd_ac_info (const GstAudioInfo * i, GstAmdAudioInfo * a)
{
  gint c;
  memset (a, 0, sizeof (*a));
  switch (GST_AUDIO_INFO_FORMAT (i)) {
    case GST_AUDIO_FORMAT_S8: a->format = GSTAMD_AFMT_S8; break;
    case GST_AUDIO_FORMAT_U8: a->format = GSTAMD_AFMT_U8; break;
    case GST_AUDIO_FORMAT_S16LE: a->format = GSTAMD_AFMT_S16LE; break;
    case GST_AUDIO_FORMAT_S24_32LE: a->format = GSTAMD_AFMT_S24_32LE; break;
    case GST_AUDIO_FORMAT_S32LE: a->format = GSTAMD_AFMT_S32LE; break;
    case GST_AUDIO_FORMAT_S24LE: a->format = GSTAMD_AFMT_S24LE; break;
    case GST_AUDIO_FORMAT_F32LE: a->format = GSTAMD_AFMT_F32LE; break;
    case GST_AUDIO_FORMAT_F64LE: a->format = GSTAMD_AFMT_F64LE; break;
    default: return FALSE;
  }
  if (GST_AUDIO_INFO_CHANNELS (i) > GSTAMD_AUDIO_MAX_CHANNELS || GST_AUDIO_INFO_LAYOUT (i) != GST_AUDIO_LAYOUT_INTERLEAVED)
    return FALSE;
  a->rate = GST_AUDIO_INFO_RATE (i);
  a->channels = GST_AUDIO_INFO_CHANNELS (i);
  a->layout = 0;
  a->unpositioned = GST_AUDIO_INFO_IS_UNPOSITIONED (i) ? 1 : 0;
  for (c = 0; c < a->channels; c++)
    a->position[c] = (int32_t) i->position[c];          /* GstAudioChannelPosition values as they are */
  return TRUE;
}

/* gst_audio_convert_ensure_converter (gstaudioconvert.c:1590-1684): (re)make the converter for s->in / s->out with the element's current
 * options; streaming thread (set_caps, transform) */
static gboolean
amd_ac_ensure_converter (GstAmdAudioConvert * s)
{
  GstBaseTransform *t = GST_BASE_TRANSFORM (s);
  GstAmdAudioInfo ai, ao;
  GstAmdAudioConverterConfig cfg;
  int status = 0;

  GST_OBJECT_LOCK (s);
  if (s->conv && !s->conv_dirty) {
    GST_OBJECT_UNLOCK (s);
    return TRUE;
  }
  s->conv_dirty = FALSE;
  GST_OBJECT_UNLOCK (s);
  if (!s->have_caps)
    return FALSE;
  gst_amd_hip_select_device (s->device_id);
  amd_ac_drop_converter (s);
  if (!amd_ac_info (&s->in, &ai) || !amd_ac_info (&s->out, &ao)) {
    GST_ERROR_OBJECT (s, "caps outside the device converter's formats / channel counts");
    return FALSE;
  }
  gstamd_audio_converter_config_init (&cfg);
  cfg.dither_method = s->dither;
  cfg.noise_shaping = s->ns;
  cfg.dither_threshold = s->dither_threshold;
  GST_OBJECT_LOCK (s);
  if (s->mix_matrix_is_set) {
    const guint rows = gst_value_array_get_size (&s->mix_matrix);
    guint r, c;
    cfg.has_mix_matrix = 1;
    if (rows == 0) {
      /* an empty matrix: gst_audio_channel_mixer_new_with_matrix makes a (truncated) identity (audio-channel-mixer.c:1160-1172) */
      for (r = 0; r < GSTAMD_AUDIO_MAX_CHANNELS; r++)
        cfg.mix_matrix[r][r] = 1.0f;
    } else {
      const guint cols = gst_value_array_get_size (gst_value_array_get_value (&s->mix_matrix, 0));
      if (rows != (guint) ao.channels || cols != (guint) ai.channels) {
        GST_OBJECT_UNLOCK (s);
        GST_ERROR_OBJECT (s, "mix-matrix is %u x %u, the caps have %d input and %d output channels", rows, cols, ai.channels, ao.channels);
        return FALSE;
      }
      for (r = 0; r < rows; r++)
        for (c = 0; c < cols; c++) {
          const GValue *v = gst_value_array_get_value (gst_value_array_get_value (&s->mix_matrix, r), c);
          cfg.mix_matrix[r][c] = G_VALUE_HOLDS_FLOAT (v) ? g_value_get_float (v) : G_VALUE_HOLDS_DOUBLE (v) ? (float) g_value_get_double (v) :
              G_VALUE_HOLDS_INT (v) ? (float) g_value_get_int (v) : 0.0f;
        }
    }
  }
  GST_OBJECT_UNLOCK (s);
  s->conv = gstamd_audio_converter_new (0, &ai, &ao, &cfg, &status);
  if (!s->conv) {
    GST_ERROR_OBJECT (s, "Failed to make converter (status %d): %s", status, gstamd_last_error ());
    return FALSE;
  }
  gst_base_transform_set_passthrough (t, gstamd_audio_converter_is_passthrough (s->conv));
  return TRUE;
}

/* gst_audio_convert_set_caps (gstaudioconvert.c:1548-1588) */
static gboolean
amd_ac_set_caps (GstBaseTransform * t, GstCaps * incaps, GstCaps * outcaps)
{
  GstAmdAudioConvert *s = AMD_AC (t);
  GstAudioInfo in, out;

  if (!gst_audio_info_from_caps (&in, incaps) || !gst_audio_info_from_caps (&out, outcaps))
    return FALSE;
  s->in = in;
  s->out = out;
  s->have_caps = TRUE;
  GST_OBJECT_LOCK (s);
  s->conv_dirty = TRUE;
  GST_OBJECT_UNLOCK (s);
  return amd_ac_ensure_converter (s);
}

static gboolean
amd_ac_get_unit_size (GstBaseTransform * t, GstCaps * caps, gsize * size)
{
  GstAudioInfo i;
  if (!gst_audio_info_from_caps (&i, caps))
    return FALSE;
  *size = GST_AUDIO_INFO_BPF (&i);
  return TRUE;
}

static gboolean
ac_staging (gpointer * p, gsize * have, gsize need)
{
  if (*have >= need && *p)
    return TRUE;
  gstamd_device_free (*p);
  *p = gstamd_device_alloc (need + 64);
  *have = *p ? need + 64 : 0;
  return *p != NULL;
}

static GstFlowReturn
amd_ac_transform (GstBaseTransform * t, GstBuffer * inbuf, GstBuffer * outbuf)
{
  GstAmdAudioConvert *s = AMD_AC (t);
  GstMapInfo im, om;
  gsize frames;
  gboolean ok = TRUE;

  if (!amd_ac_ensure_converter (s))
    return GST_FLOW_NOT_NEGOTIATED;
  if (gstamd_audio_converter_is_passthrough (s->conv) && inbuf != outbuf && gst_buffer_get_size (inbuf) <= gst_buffer_get_size (outbuf)) {
    /* the new options made the conversion an identity while the base class was not in passthrough for this buffer yet */
    if (!gst_buffer_map (inbuf, &im, GST_MAP_READ))
      return GST_FLOW_ERROR;
    gst_buffer_fill (outbuf, 0, im.data, im.size);
    gst_buffer_set_size (outbuf, im.size);
    gst_buffer_unmap (inbuf, &im);
    return GST_FLOW_OK;
  }
  if (!gst_buffer_map (inbuf, &im, GST_MAP_READ))
    return GST_FLOW_ERROR;
  if (!gst_buffer_map (outbuf, &om, GST_MAP_WRITE)) {
    gst_buffer_unmap (inbuf, &im);
    return GST_FLOW_ERROR;
  }
  frames = im.size / GST_AUDIO_INFO_BPF (&s->in);
  if (frames > om.size / GST_AUDIO_INFO_BPF (&s->out))
    frames = om.size / GST_AUDIO_INFO_BPF (&s->out);
  if (GST_BUFFER_FLAG_IS_SET (inbuf, GST_BUFFER_FLAG_GAP)) {
    /* gstaudioconvert.c:1733-1747: a gap stays silence, the converter does not see it */
    gst_audio_format_fill_silence (s->out.finfo, om.data, frames * GST_AUDIO_INFO_BPF (&s->out));
  } else if (frames) {
    const gsize in_bytes = frames * GST_AUDIO_INFO_BPF (&s->in), out_bytes = frames * GST_AUDIO_INFO_BPF (&s->out);
    gst_amd_hip_select_device (s->device_id);
    ok = (s->stream || (s->stream = gstamd_stream_new ()) != NULL) && ac_staging (&s->d_in, &s->d_in_size, in_bytes) &&
        ac_staging (&s->d_out, &s->d_out_size, out_bytes) && gstamd_device_upload_async (s->d_in, im.data, in_bytes, s->stream) == GSTAMD_OK &&
        gstamd_audio_converter_samples (s->conv, 0, s->d_in, frames, s->d_out, frames, s->stream) == GSTAMD_OK &&
        gstamd_device_download_async (om.data, s->d_out, out_bytes, s->stream) == GSTAMD_OK && gstamd_stream_synchronize (s->stream) == GSTAMD_OK;
  }
  gst_buffer_unmap (outbuf, &om);
  gst_buffer_unmap (inbuf, &im);
  if (!ok) {
    GST_ELEMENT_ERROR (s, LIBRARY, FAILED, ("HIP audio conversion failed"), ("%s", gstamd_last_error ()));
    return GST_FLOW_ERROR;
  }
  gst_buffer_set_size (outbuf, frames * GST_AUDIO_INFO_BPF (&s->out));
  return GST_FLOW_OK;
}

/* GstRequestAudioMixMatrix (gstaudioconvert.c:309-337) */
static gboolean
amd_ac_src_event (GstBaseTransform * t, GstEvent * event)
{
  if (GST_EVENT_TYPE (event) == GST_EVENT_CUSTOM_UPSTREAM) {
    const GstStructure *st = gst_event_get_structure (event);
    if (st && gst_structure_has_name (st, "GstRequestAudioMixMatrix")) {
      const GValue *m = gst_structure_get_value (st, "matrix");
      if (m) {
        amd_ac_set_mix_matrix (AMD_AC (t), m);
        g_object_notify (G_OBJECT (t), "mix-matrix");
      }
      gst_event_unref (event);
      return TRUE;
    }
  }
  return GST_BASE_TRANSFORM_CLASS (gst_amd_ac_parent_class)->src_event (t, event);
}

static gboolean
amd_ac_sink_event (GstBaseTransform * t, GstEvent * event)
{
  GstAmdAudioConvert *s = AMD_AC (t);
  if (GST_EVENT_TYPE (event) == GST_EVENT_FLUSH_STOP && s->conv)
    gstamd_audio_converter_reset (s->conv);
  return GST_BASE_TRANSFORM_CLASS (gst_amd_ac_parent_class)->sink_event (t, event);
}

static gboolean
amd_ac_stop (GstBaseTransform * t)
{
  GstAmdAudioConvert *s = AMD_AC (t);
  gst_amd_hip_select_device (s->device_id);
  amd_ac_drop_converter (s);
  s->have_caps = FALSE;
  if (s->stream) {
    gstamd_stream_synchronize (s->stream);
    gstamd_stream_free (s->stream);
    s->stream = NULL;
  }
  gstamd_device_free (s->d_in);
  gstamd_device_free (s->d_out);
  s->d_in = s->d_out = NULL;
  s->d_in_size = s->d_out_size = 0;
  return TRUE;
}

static void
amd_ac_finalize (GObject * o)
{
  GstAmdAudioConvert *s = AMD_AC (o);
  if (G_IS_VALUE (&s->mix_matrix))
    g_value_unset (&s->mix_matrix);
  G_OBJECT_CLASS (gst_amd_ac_parent_class)->finalize (o);
}

static void
gst_amd_ac_class_init (GstAmdAudioConvertClass * klass)
{
  GObjectClass *oc = G_OBJECT_CLASS (klass);
  GstElementClass *ec = GST_ELEMENT_CLASS (klass);
  GstBaseTransformClass *tc = GST_BASE_TRANSFORM_CLASS (klass);
  /* nicks of GstAudioDitherMethod / GstAudioNoiseShapingMethod (audio-quantize.h:45-72) */
  static const GEnumValue dv[] = {{0, "No dithering", "none"}, {1, "Rectangular dithering", "rpdf"}, {2, "Triangular dithering (default)", "tpdf"},
    {3, "High frequency triangular dithering", "tpdf-hf"}, {0, NULL, NULL}};
  static const GEnumValue nv[] = {{0, "No noise shaping (default)", "none"}, {1, "Error feedback", "error-feedback"},
    {2, "Simple 2-pole noise shaping", "simple"}, {3, "Medium 5-pole noise shaping", "medium"}, {4, "High 8-pole noise shaping", "high"}, {0, NULL, NULL}};
  GType dt = g_type_from_name ("GstAmdAudioDitherMethod"), nt = g_type_from_name ("GstAmdAudioNoiseShapingMethod");

  if (!dt)
    dt = g_enum_register_static ("GstAmdAudioDitherMethod", dv);
  if (!nt)
    nt = g_enum_register_static ("GstAmdAudioNoiseShapingMethod", nv);
  GST_DEBUG_CATEGORY_INIT (amd_ac_debug, "amdaudioconvert", 0, "MI355X audioconvert");
  oc->set_property = amd_ac_set_property;
  oc->get_property = amd_ac_get_property;
  oc->finalize = amd_ac_finalize;
  g_object_class_install_property (oc, PROP_DITHERING, g_param_spec_enum ("dithering", "Dithering", "Selects between different dithering methods.",
          dt, 2, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_NOISE_SHAPING, g_param_spec_enum ("noise-shaping", "Noise shaping",
          "Selects between different noise shaping methods.", nt, 0, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_MIX_MATRIX, gst_param_spec_array ("mix-matrix", "Input/output channel matrix",
          "Transformation matrix for input/output channels.", gst_param_spec_array ("matrix-rows", "rows", "rows",
              g_param_spec_float ("matrix-cols", "cols", "cols", -1, 1, 0, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS),
              G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS), G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_DITHERING_THRESHOLD, g_param_spec_uint ("dithering-threshold", "Dithering Threshold",
          "Threshold for the output bit depth at/below which to apply dithering.", 0, 32, 20, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_DEVICE_ID, g_param_spec_int ("device-id", "Device ID",
          "HIP device this instance runs on (-1 = the process's current device)", -1, G_MAXINT, -1, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  gst_element_class_set_static_metadata (ec, "Audio converter (MI355X/HIP)", "Filter/Converter/Audio",
      "Convert audio to different formats and channel layouts on an AMD Instinct GPU, bit-exact to GstAudioConverter", "gstreamer_amd");
  gst_element_class_add_static_pad_template (ec, &ac_sink);
  gst_element_class_add_static_pad_template (ec, &ac_src);
  tc->passthrough_on_same_caps = FALSE;
  tc->transform_caps = GST_DEBUG_FUNCPTR (amd_ac_transform_caps);
  tc->fixate_caps = GST_DEBUG_FUNCPTR (amd_ac_fixate_caps);
  tc->set_caps = GST_DEBUG_FUNCPTR (amd_ac_set_caps);
  tc->get_unit_size = GST_DEBUG_FUNCPTR (amd_ac_get_unit_size);
  tc->transform = GST_DEBUG_FUNCPTR (amd_ac_transform);
  tc->src_event = GST_DEBUG_FUNCPTR (amd_ac_src_event);
  tc->sink_event = GST_DEBUG_FUNCPTR (amd_ac_sink_event);
  tc->stop = GST_DEBUG_FUNCPTR (amd_ac_stop);
}

static void
gst_amd_ac_init (GstAmdAudioConvert * s)
{
  s->dither = GSTAMD_AUDIO_DITHER_TPDF;        /* the element's default, not the library's (gstaudioconvert.c:477-483) */
  s->ns = 0;
  s->dither_threshold = 20;
  s->device_id = -1;
  g_value_init (&s->mix_matrix, GST_TYPE_ARRAY);
}

GType
gst_amd_audio_convert_get_type (void)
{
  return gst_amd_ac_get_type ();
}
