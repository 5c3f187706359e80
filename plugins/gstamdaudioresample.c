/* gstamdaudioresample.c - `audioresample` element backed by the MI355X polyphase FIR (include/gstamd_audio.h).
 *
 * Mirrors the reference element's contract for this path
 * (subprojects/gst-plugins-base/gst/audioresample/gstaudioresample.c): factory name / rank PRIMARY (:139-142),
 * properties quality (0-10, default 4), resample-method, sinc-filter-mode / -auto-threshold / -interpolation (:150-185), caps audio/x-raw {F32,F64,S16,S32}
 * interleaved with the rate made a range by transform_caps, output sizes from
 * gst_audio_resampler_get_out_frames, one resample() per buffer (:745-860), drain of the filter history at EOS.
 * Samples are staged to HBM per buffer (audio buffers are tiny; a HIP-memory audio path would not change
 * throughput).  DISCONT buffers and timestamp jumps drain + reset the filter (gstaudioresample.c:705-739, 898-940).  Not implemented: GAP-buffer
 * accounting, latency query.
 */
#include <gst/audio/audio.h>
#include <gst/base/gstbasetransform.h>
#include <gst/gst.h>
#include <string.h>

#include "../include/gstamd_audio.h"
#include "../include/gstamd_video.h"
#include "gstamdhipmemory.h"

GST_DEBUG_CATEGORY_STATIC (amd_ar_debug);
#define GST_CAT_DEFAULT amd_ar_debug

#define AMD_AUDIO_CAPS "audio/x-raw, format = (string) { F32LE, F64LE, S16LE, S32LE }, rate = (int) [ 1, MAX ], " \
    "channels = (int) [ 1, MAX ], layout = (string) { interleaved, non-interleaved }"

static GstStaticPadTemplate ar_sink = GST_STATIC_PAD_TEMPLATE ("sink", GST_PAD_SINK, GST_PAD_ALWAYS, GST_STATIC_CAPS (AMD_AUDIO_CAPS));
static GstStaticPadTemplate ar_src = GST_STATIC_PAD_TEMPLATE ("src", GST_PAD_SRC, GST_PAD_ALWAYS, GST_STATIC_CAPS (AMD_AUDIO_CAPS));

typedef struct {
  GstBaseTransform parent;
  gint quality, method;
  gint sinc_filter_mode, sinc_filter_interpolation;      /* sinc-filter-mode (auto), sinc-filter-interpolation (cubic) */
  guint sinc_filter_auto_threshold;                      /* sinc-filter-auto-threshold (1 MiB) */
  GstAudioInfo in, out;
  GstAmdAudioResampler *r;
  gint r_method;                                         /* the method s->r was made with */
  gpointer d_in, d_out;
  gsize d_in_size, d_out_size;
  gpointer h_in, h_out;                                  /* page-locked staging of the list path: one upload and one download per list */
  gsize h_in_size, h_out_size;
  gpointer stream;             /* this instance's HIP stream */
  gint device_id;              /* device-id property: -1 = the process's current device */
  guint64 samples_in, samples_out;
  guint64 in_offset0, out_offset0;
  gboolean need_discont;
  GstClockTime t0;
  GstPadChainFunction base_chain;      /* GstBaseTransform's chain function (the sink pad's, before chain_list was installed) */
  guint64 n_list_calls, n_list_buffers; /* buffer lists resampled in one call / the buffers they held (GSTAMD_ELEMENT_STATS) */
} GstAmdAudioResample;
typedef struct { GstBaseTransformClass parent_class; } GstAmdAudioResampleClass;

enum { PROP_0, PROP_QUALITY, PROP_METHOD, PROP_SINC_FILTER_MODE, PROP_SINC_FILTER_AUTO_THRESHOLD, PROP_SINC_FILTER_INTERPOLATION, PROP_DEVICE_ID };
G_DEFINE_TYPE (GstAmdAudioResample, gst_amd_ar, GST_TYPE_BASE_TRANSFORM);
#define AMD_AR(o) ((GstAmdAudioResample *) (o))

static void
amd_ar_set_property (GObject * o, guint id, const GValue * v, GParamSpec * p)
{
  GstAmdAudioResample *s = AMD_AR (o);
  if (id == PROP_QUALITY)
    s->quality = g_value_get_int (v);
  else if (id == PROP_METHOD)
    s->method = g_value_get_enum (v);
  else if (id == PROP_SINC_FILTER_MODE)
    s->sinc_filter_mode = g_value_get_enum (v);
  else if (id == PROP_SINC_FILTER_AUTO_THRESHOLD)
    s->sinc_filter_auto_threshold = g_value_get_uint (v);
  else if (id == PROP_SINC_FILTER_INTERPOLATION)
    s->sinc_filter_interpolation = g_value_get_enum (v);
  else if (id == PROP_DEVICE_ID)
    s->device_id = g_value_get_int (v);
  else
    G_OBJECT_WARN_INVALID_PROPERTY_ID (o, id, p);
}

static void
amd_ar_get_property (GObject * o, guint id, GValue * v, GParamSpec * p)
{
  GstAmdAudioResample *s = AMD_AR (o);
  if (id == PROP_QUALITY)
    g_value_set_int (v, s->quality);
  else if (id == PROP_METHOD)
    g_value_set_enum (v, s->method);
  else if (id == PROP_SINC_FILTER_MODE)
    g_value_set_enum (v, s->sinc_filter_mode);
  else if (id == PROP_SINC_FILTER_AUTO_THRESHOLD)
    g_value_set_uint (v, s->sinc_filter_auto_threshold);
  else if (id == PROP_SINC_FILTER_INTERPOLATION)
    g_value_set_enum (v, s->sinc_filter_interpolation);
  else if (id == PROP_DEVICE_ID)
    g_value_set_int (v, s->device_id);
  else
    G_OBJECT_WARN_INVALID_PROPERTY_ID (o, id, p);
}

static GstCaps *
amd_ar_transform_caps (GstBaseTransform * t, GstPadDirection d, GstCaps * caps, GstCaps * filter)
{
  GstCaps *res = gst_caps_new_empty ();
  guint i;
  for (i = 0; i < gst_caps_get_size (caps); i++) {
    GstStructure *st = gst_structure_copy (gst_caps_get_structure (caps, i));
    gst_structure_set (st, "rate", GST_TYPE_INT_RANGE, 1, G_MAXINT, NULL);
    gst_caps_append_structure (res, st);
  }
  if (filter) {
    GstCaps *tmp = gst_caps_intersect_full (filter, res, GST_CAPS_INTERSECT_FIRST);
    gst_caps_unref (res);
    res = tmp;
  }
  return res;
}

static GstCaps *
amd_ar_fixate_caps (GstBaseTransform * t, GstPadDirection d, GstCaps * caps, GstCaps * othercaps)
{
  gint rate = 0;
  othercaps = gst_caps_truncate (gst_caps_make_writable (othercaps));
  if (gst_structure_get_int (gst_caps_get_structure (caps, 0), "rate", &rate))
    gst_structure_fixate_field_nearest_int (gst_caps_get_structure (othercaps, 0), "rate", rate);
  return gst_caps_fixate (othercaps);
}

static int
amd_format (const GstAudioInfo * i)
{
  /* the ABI takes the GstAudioFormat itself (gst_audio_resampler_new, audio-resampler.h:218); these four are what the resampler accepts */
  switch (GST_AUDIO_INFO_FORMAT (i)) {
    case GST_AUDIO_FORMAT_S16LE:
    case GST_AUDIO_FORMAT_S32LE:
    case GST_AUDIO_FORMAT_F32LE:
    case GST_AUDIO_FORMAT_F64LE:
      return (int) GST_AUDIO_INFO_FORMAT (i);
    default: return -1;
  }
}

static void amd_ar_drain (GstAmdAudioResample * s);

static gboolean
amd_ar_set_caps (GstBaseTransform * t, GstCaps * incaps, GstCaps * outcaps)
{
  GstAmdAudioResample *s = AMD_AR (t);
  GstAmdAudioResamplerOptions o;
  int status = 0;

  GstAudioInfo in, out;
  gboolean same_stream;

  if (!gst_audio_info_from_caps (&in, incaps) || !gst_audio_info_from_caps (&out, outcaps))
    return FALSE;
  if (amd_format (&in) < 0 || GST_AUDIO_INFO_FORMAT (&in) != GST_AUDIO_INFO_FORMAT (&out) ||
      GST_AUDIO_INFO_CHANNELS (&in) != GST_AUDIO_INFO_CHANNELS (&out) || GST_AUDIO_INFO_LAYOUT (&in) != GST_AUDIO_INFO_LAYOUT (&out))
    return FALSE;
  /* gst_audio_resample_update_state (gstaudioresample.c:398-446): a change of format, channels or layout destroys the
   * resampler; a change of rates only goes through gst_audio_resampler_update with freshly made options, so the
   * stream keeps its history and phase */
  same_stream = s->r != NULL && GST_AUDIO_INFO_FORMAT (&in) == GST_AUDIO_INFO_FORMAT (&s->in) &&
      GST_AUDIO_INFO_CHANNELS (&in) == GST_AUDIO_INFO_CHANNELS (&s->in) && GST_AUDIO_INFO_LAYOUT (&in) == GST_AUDIO_INFO_LAYOUT (&s->in);
  /* gst_audio_resample_set_caps (gstaudioresample.c:519-538): a change of either side drains the old stream with its own caps,
   * resets the history and restarts the timestamp tracking */
  if (s->r && (!gst_audio_info_is_equal (&in, &s->in) || !gst_audio_info_is_equal (&out, &s->out))) {
    amd_ar_drain (s);
    gstamd_audio_resampler_reset (s->r);
    s->samples_in = s->samples_out = 0;
    s->t0 = GST_CLOCK_TIME_NONE;
    s->need_discont = TRUE;
  }
  s->in = in;
  s->out = out;
  gstamd_audio_resampler_options_init (&o);
  gstamd_audio_resampler_options_set_quality (s->method, (unsigned) s->quality, GST_AUDIO_INFO_RATE (&s->in),
      GST_AUDIO_INFO_RATE (&s->out), &o);
  /* make_options (gstaudioresample.c:374-395): the three sinc-filter-* properties go into the options */
  o.filter_mode = s->sinc_filter_mode;
  o.filter_mode_threshold = (int32_t) s->sinc_filter_auto_threshold;
  o.filter_interpolation = s->sinc_filter_interpolation;
  if (same_stream && s->r_method == s->method) {
    status = gstamd_audio_resampler_update (s->r, GST_AUDIO_INFO_RATE (&s->in), GST_AUDIO_INFO_RATE (&s->out), &o);
    if (status != 0) {
      GST_ERROR_OBJECT (s, "failed to update resampler (status %d)", status);
      return FALSE;
    }
    return TRUE;
  }
  if (s->r)
    gstamd_audio_resampler_free (s->r);
  /* non-interleaved buffers hold their planes back to back, one buffer's frames apart (gstaudioresample.c:~960 builds the plane
   * pointers the same way); GST_AUDIO_RESAMPLER_FLAG_NON_INTERLEAVED_IN | _OUT, and _VARIABLE_RATE as the element's converter
   * is made with GST_AUDIO_CONVERTER_FLAG_VARIABLE_RATE (gstaudioresample.c:422, audio-converter.c:929) */
  s->r = gstamd_audio_resampler_new (s->method, (GST_AUDIO_INFO_LAYOUT (&s->in) == GST_AUDIO_LAYOUT_NON_INTERLEAVED ? 3 : 0) | 4, amd_format (&s->in), GST_AUDIO_INFO_CHANNELS (&s->in),
      GST_AUDIO_INFO_RATE (&s->in), GST_AUDIO_INFO_RATE (&s->out), &o, &status);
  if (!s->r) {
    GST_ERROR_OBJECT (s, "HIP resampler refused this configuration (status %d)", status);
    return FALSE;
  }
  s->r_method = s->method;
  s->samples_in = s->samples_out = 0;
  s->t0 = GST_CLOCK_TIME_NONE;
  s->need_discont = TRUE;
  return TRUE;
}

static gboolean
amd_ar_get_unit_size (GstBaseTransform * t, GstCaps * caps, gsize * size)
{
  GstAudioInfo i;
  if (!gst_audio_info_from_caps (&i, caps))
    return FALSE;
  *size = GST_AUDIO_INFO_BPF (&i);
  return TRUE;
}

static gboolean
amd_ar_transform_size (GstBaseTransform * t, GstPadDirection d, GstCaps * caps, gsize size, GstCaps * othercaps, gsize * othersize)
{
  GstAmdAudioResample *s = AMD_AR (t);
  GstAudioInfo i, o;
  if (!s->r || !gst_audio_info_from_caps (&i, caps) || !gst_audio_info_from_caps (&o, othercaps))
    return FALSE;
  if (d == GST_PAD_SINK)
    *othersize = gstamd_audio_resampler_get_out_frames (s->r, size / GST_AUDIO_INFO_BPF (&i)) * GST_AUDIO_INFO_BPF (&o);
  else
    *othersize = gstamd_audio_resampler_get_in_frames (s->r, size / GST_AUDIO_INFO_BPF (&i)) * GST_AUDIO_INFO_BPF (&o);
  return TRUE;
}

static gboolean
ar_staging (gpointer * p, gsize * have, gsize need)
{
  if (*have >= need && *p)
    return TRUE;
  gstamd_device_free (*p);
  *p = gstamd_device_alloc (need + 64);
  *have = *p ? need + 64 : 0;
  return *p != NULL;
}

static gboolean
ar_host_staging (gpointer * p, gsize * have, gsize need)
{
  if (*have >= need && *p)
    return TRUE;
  gstamd_host_free (*p);
  *p = gstamd_host_alloc (need + 4096);
  *have = *p ? need + 4096 : 0;
  return *p != NULL;
}

/* resample in_frames from host memory (NULL = silence) into a host buffer; returns frames produced or -1.
 * in_planes: NULL, or one pointer per channel (a non-interleaved buffer whose GstAudioMeta places the planes; they are gathered
 * back to back in the device staging buffer, which is the layout the resampler was created for) */
static gssize
amd_ar_process (GstAmdAudioResample * s, const guint8 * in, gpointer * in_planes, gsize in_frames, guint8 * out, gsize out_cap_frames)
{
  const gsize bpf = GST_AUDIO_INFO_BPF (&s->in);
  gsize out_frames = gstamd_audio_resampler_get_out_frames (s->r, in_frames);

  gst_amd_hip_select_device (s->device_id);
  if (!s->stream && !(s->stream = gstamd_stream_new ()))
    return -1;
  if (out_frames > out_cap_frames)
    out_frames = out_cap_frames;
  if ((in || in_planes) && !ar_staging (&s->d_in, &s->d_in_size, in_frames * bpf))
    return -1;
  if (in_planes) {
    const gsize plane = in_frames * (bpf / GST_AUDIO_INFO_CHANNELS (&s->in));
    gint c;
    for (c = 0; c < GST_AUDIO_INFO_CHANNELS (&s->in); c++)
      if (gstamd_device_upload_async ((guint8 *) s->d_in + c * plane, in_planes[c], plane, s->stream) != GSTAMD_OK)
        return -1;
  } else if (in && gstamd_device_upload_async (s->d_in, in, in_frames * bpf, s->stream) != GSTAMD_OK) {
    return -1;
  }
  if (!ar_staging (&s->d_out, &s->d_out_size, out_frames * bpf))
    return -1;
  if (gstamd_audio_resampler_resample (s->r, (in || in_planes) ? s->d_in : NULL, in_frames, s->d_out, out_frames, s->stream) != GSTAMD_OK)
    return -1;
  if (out_frames && gstamd_device_download_async (out, s->d_out, out_frames * bpf, s->stream) != GSTAMD_OK)
    return -1;
  if (gstamd_stream_synchronize (s->stream) != GSTAMD_OK)       /* the CPU reads `out` next; the staging buffers are free again */
    return -1;
  return (gssize) out_frames;
}

/* a non-interleaved output buffer of `frames` frames: planes back to back + (GStreamer >= 1.16) the GstAudioMeta that
 * gst_audio_buffer_map requires for this layout */
static void
amd_ar_finish_layout (GstAmdAudioResample * s, GstBuffer * buf, gsize frames)
{
#if GST_CHECK_VERSION (1, 16, 0)
  if (GST_AUDIO_INFO_LAYOUT (&s->out) == GST_AUDIO_LAYOUT_NON_INTERLEAVED)
    gst_buffer_add_audio_meta (buf, &s->out, frames, NULL);
#endif
}

static void
amd_ar_stamp (GstAmdAudioResample * s, GstBuffer * buf, gsize frames)
{
  const gint rate = GST_AUDIO_INFO_RATE (&s->out);
  if (GST_CLOCK_TIME_IS_VALID (s->t0)) {
    GST_BUFFER_PTS (buf) = s->t0 + gst_util_uint64_scale_int_round (s->samples_out, GST_SECOND, rate);
    GST_BUFFER_DURATION (buf) = s->t0 + gst_util_uint64_scale_int_round (s->samples_out + frames, GST_SECOND, rate) - GST_BUFFER_PTS (buf);
  }
  if (s->out_offset0 != GST_BUFFER_OFFSET_NONE) {
    GST_BUFFER_OFFSET (buf) = s->out_offset0 + s->samples_out;
    GST_BUFFER_OFFSET_END (buf) = s->out_offset0 + s->samples_out + frames;
  } else {
    GST_BUFFER_OFFSET (buf) = GST_BUFFER_OFFSET_NONE;
    GST_BUFFER_OFFSET_END (buf) = GST_BUFFER_OFFSET_NONE;
  }
  s->samples_out += frames;
}

/* gst_audio_resample_check_discont (gstaudioresample.c:705-739): the DISCONT flag, or a timestamp that is more than rate / 32 samples away from
 * where the stream should be */
static gboolean
amd_ar_check_discont (GstAmdAudioResample * s, GstBuffer * buf)
{
  guint64 offset, delta;
  if (GST_BUFFER_IS_DISCONT (buf))
    return TRUE;
  if (!(GST_BUFFER_PTS_IS_VALID (buf) && GST_CLOCK_TIME_IS_VALID (s->t0)) || GST_BUFFER_PTS (buf) < s->t0)
    return FALSE;
  offset = gst_util_uint64_scale_int_round (GST_BUFFER_PTS (buf) - s->t0, GST_AUDIO_INFO_RATE (&s->in), GST_SECOND);
  delta = offset > s->samples_in ? offset - s->samples_in : s->samples_in - offset;
  if (delta <= (guint64) (GST_AUDIO_INFO_RATE (&s->in) >> 5))
    return FALSE;
  GST_WARNING_OBJECT (s, "encountered timestamp discontinuity of %" G_GUINT64_FORMAT " samples", delta);
  return TRUE;
}

static GstFlowReturn
amd_ar_transform (GstBaseTransform * t, GstBuffer * inbuf, GstBuffer * outbuf)
{
  GstAmdAudioResample *s = AMD_AR (t);
  GstMapInfo im, om;
  gssize n;

  if (!s->r)
    return GST_FLOW_NOT_NEGOTIATED;
  /* transform (:898-940): drain what the filter still holds with the old timing, reset it, restart the counters at this buffer */
  if (s->samples_in > 0 && amd_ar_check_discont (s, inbuf)) {
    amd_ar_drain (s);
    gstamd_audio_resampler_reset (s->r);
    s->need_discont = TRUE;
  }
  if (s->need_discont) {
    s->samples_in = s->samples_out = 0;
    s->t0 = GST_BUFFER_PTS_IS_VALID (inbuf) ? GST_BUFFER_PTS (inbuf) : GST_CLOCK_TIME_NONE;
    if (GST_BUFFER_OFFSET_IS_VALID (inbuf)) {
      s->in_offset0 = GST_BUFFER_OFFSET (inbuf);
      s->out_offset0 = gst_util_uint64_scale_int_round (s->in_offset0, GST_AUDIO_INFO_RATE (&s->out), GST_AUDIO_INFO_RATE (&s->in));
    } else {
      s->in_offset0 = s->out_offset0 = GST_BUFFER_OFFSET_NONE;
    }
    GST_BUFFER_FLAG_SET (outbuf, GST_BUFFER_FLAG_DISCONT);
    s->need_discont = FALSE;
  }
  if (!GST_CLOCK_TIME_IS_VALID (s->t0))
    s->t0 = GST_BUFFER_PTS_IS_VALID (inbuf) ? GST_BUFFER_PTS (inbuf) : 0;
  if (!gst_buffer_map (inbuf, &im, GST_MAP_READ))
    return GST_FLOW_ERROR;
  if (!gst_buffer_map (outbuf, &om, GST_MAP_WRITE)) {
    gst_buffer_unmap (inbuf, &im);
    return GST_FLOW_ERROR;
  }
  {
    gpointer *planes = NULL;
    gsize in_frames = im.size / GST_AUDIO_INFO_BPF (&s->in);
#if GST_CHECK_VERSION (1, 16, 0)
    /* non-interleaved input: the planes are where the buffer's GstAudioMeta says, not necessarily back to back */
    GstAudioMeta *ameta = gst_buffer_get_audio_meta (inbuf);
    gpointer plane_ptrs[64];
    if (ameta && GST_AUDIO_INFO_LAYOUT (&s->in) == GST_AUDIO_LAYOUT_NON_INTERLEAVED && GST_AUDIO_INFO_CHANNELS (&s->in) <= 64) {
      gint c;
      for (c = 0; c < GST_AUDIO_INFO_CHANNELS (&s->in); c++)
        plane_ptrs[c] = im.data + ameta->offsets[c];
      planes = plane_ptrs;
      in_frames = ameta->samples;
    }
#endif
    n = amd_ar_process (s, im.data, planes, in_frames, om.data, om.size / GST_AUDIO_INFO_BPF (&s->out));
    s->samples_in += in_frames;
  }
  gst_buffer_unmap (outbuf, &om);
  gst_buffer_unmap (inbuf, &im);
  if (n < 0) {
    GST_ELEMENT_ERROR (s, LIBRARY, FAILED, ("HIP resampling failed"), ("%s", gstamd_last_error ()));
    return GST_FLOW_ERROR;
  }
  gst_buffer_set_size (outbuf, (gsize) n * GST_AUDIO_INFO_BPF (&s->out));
  amd_ar_finish_layout (s, outbuf, (gsize) n);
  amd_ar_stamp (s, outbuf, (gsize) n);
  return n > 0 ? GST_FLOW_OK : GST_BASE_TRANSFORM_FLOW_DROPPED;
}

/* amd_ar_check_discont for a buffer that would arrive after `samples_in` frames of the stream */
static gboolean
amd_ar_discont_at (GstAmdAudioResample * s, GstBuffer * buf, guint64 samples_in)
{
  guint64 offset, delta;
  if (GST_BUFFER_IS_DISCONT (buf))
    return TRUE;
  if (!(GST_BUFFER_PTS_IS_VALID (buf) && GST_CLOCK_TIME_IS_VALID (s->t0)) || GST_BUFFER_PTS (buf) < s->t0)
    return FALSE;
  offset = gst_util_uint64_scale_int_round (GST_BUFFER_PTS (buf) - s->t0, GST_AUDIO_INFO_RATE (&s->in), GST_SECOND);
  delta = offset > samples_in ? offset - samples_in : samples_in - offset;
  return delta > (guint64) (GST_AUDIO_INFO_RATE (&s->in) >> 5);
}

/* A GstBufferList on the sink pad (the core would feed it to chain () buffer by buffer, gstpad.c gst_pad_chain_list_default; the reference's element
 * has no list path, gstaudioresample.c).  Consecutive buffers of one stream are ONE stretch of samples: the run is uploaded back to back, resampled by
 * one gstamd_audio_resampler_resample call (one launch instead of one per buffer - at 1024 frames a buffer the launch is the cost), downloaded, and cut
 * where buffer by buffer calls would have cut it: the number of frames n input frames produce from a given state does not depend on how they are
 * chunked (gst_audio_resampler_get_out_frames, audio-resampler.c:1640-1668: a function of the running phase), so output k holds
 * out (in_0 + .. + in_k) - out (in_0 + .. + in_k-1) frames - sample for sample and timestamp for timestamp what the per-buffer path gives
 * (plugins/tests/live_props.c audio-list).  Interleaved streams; a DISCONT or a timestamp jump ends the run and takes the regular path. */
#define AMD_AR_LIST_CHUNK 64
static GstFlowReturn
amd_ar_chain_list (GstPad * pad, GstObject * parent, GstBufferList * list)
{
  GstAmdAudioResample *s = AMD_AR (parent);
  GstBaseTransform *trans = GST_BASE_TRANSFORM (parent);
  const guint n = gst_buffer_list_length (list);
  GstFlowReturn ret = GST_FLOW_OK;
  guint i = 0;

  while (ret == GST_FLOW_OK && i < n) {
    const gsize bpf = s->r ? GST_AUDIO_INFO_BPF (&s->in) : 0;
    gboolean batch = s->r && bpf && !s->need_discont && s->samples_in > 0 && !gst_base_transform_is_passthrough (trans) &&
        !gst_pad_needs_reconfigure (GST_BASE_TRANSFORM_SRC_PAD (trans)) && GST_AUDIO_INFO_LAYOUT (&s->in) == GST_AUDIO_LAYOUT_INTERLEAVED;
    gsize in_frames[AMD_AR_LIST_CHUNK], out_cum[AMD_AR_LIST_CHUNK + 1];
    guint cnt = 0, k;
    guint64 sin = s->samples_in;
    gsize total_in = 0;

    for (k = i; batch && k < n && cnt < AMD_AR_LIST_CHUNK; k++) {
      GstBuffer *b = gst_buffer_list_get (list, k);
      const gsize sz = gst_buffer_get_size (b);
      if (sz == 0 || sz % bpf || amd_ar_discont_at (s, b, sin))
        break;
      in_frames[cnt++] = sz / bpf;
      total_in += sz / bpf;
      sin += sz / bpf;
    }
    if (cnt < 2) {               /* nothing to gain (or the stream starts / restarts here): the regular path, which also settles negotiation */
      ret = s->base_chain (pad, parent, gst_buffer_ref (gst_buffer_list_get (list, i)));
      i++;
      continue;
    }
    {
      GstBufferList *out_list = gst_buffer_list_new_sized (cnt);
      GstBuffer *outs[AMD_AR_LIST_CHUNK];
      GstMapInfo imaps[AMD_AR_LIST_CHUNK], omaps[AMD_AR_LIST_CHUNK];
      const gsize obpf = GST_AUDIO_INFO_BPF (&s->out);
      gsize acc = 0, off = 0, total_out;
      guint mapped_in = 0, mapped_out = 0;
      gboolean ok = TRUE;

      out_cum[0] = 0;
      for (k = 0; k < cnt; k++) {
        acc += in_frames[k];
        out_cum[k + 1] = gstamd_audio_resampler_get_out_frames (s->r, acc);
      }
      total_out = out_cum[cnt];
      gst_amd_hip_select_device (s->device_id);
      if (!s->stream && !(s->stream = gstamd_stream_new ()))
        ok = FALSE;
      ok = ok && ar_staging (&s->d_in, &s->d_in_size, total_in * bpf) && ar_staging (&s->d_out, &s->d_out_size, total_out * obpf + 64) &&
          ar_host_staging (&s->h_in, &s->h_in_size, total_in * bpf) && ar_host_staging (&s->h_out, &s->h_out_size, total_out * obpf + 64);
      /* the run gathered in page-locked memory (a few KB a buffer: a memcpy), ONE upload, one launch, ONE download, then cut into the output buffers -
         three asynchronous calls and one synchronisation per list, whatever its length (a copy call per buffer costs more than the kernel) */
      for (k = 0; ok && k < cnt; k++) {
        if (!gst_buffer_map (gst_buffer_list_get (list, i + k), &imaps[k], GST_MAP_READ)) {
          ok = FALSE;
          break;
        }
        mapped_in++;
        memcpy ((guint8 *) s->h_in + off, imaps[k].data, in_frames[k] * bpf);
        off += in_frames[k] * bpf;
      }
      for (k = 0; k < mapped_in; k++)
        gst_buffer_unmap (gst_buffer_list_get (list, i + k), &imaps[k]);
      mapped_in = 0;
      ok = ok && gstamd_device_upload_async (s->d_in, s->h_in, total_in * bpf, s->stream) == GSTAMD_OK;
      ok = ok && gstamd_audio_resampler_resample (s->r, s->d_in, total_in, s->d_out, total_out, s->stream) == GSTAMD_OK;
      if (ok && total_out)
        ok = gstamd_device_download_async (s->h_out, s->d_out, total_out * obpf, s->stream) == GSTAMD_OK;
      if (s->stream && gstamd_stream_synchronize (s->stream) != GSTAMD_OK)
        ok = FALSE;
      for (k = 0; k < cnt; k++)
        outs[k] = NULL;
      for (k = 0; ok && k < cnt; k++) {
        const gsize of = out_cum[k + 1] - out_cum[k];
        if (of == 0)
          continue;             /* (an input buffer too short to complete an output frame: GST_BASE_TRANSFORM_FLOW_DROPPED on the regular path) */
        outs[k] = gst_buffer_new_and_alloc (of * obpf);
        if (!gst_buffer_map (outs[k], &omaps[k], GST_MAP_WRITE)) {
          ok = FALSE;
          break;
        }
        memcpy (omaps[k].data, (guint8 *) s->h_out + out_cum[k] * obpf, of * obpf);
        gst_buffer_unmap (outs[k], &omaps[k]);
      }
      (void) mapped_out;
      if (!ok) {
        for (k = 0; k < cnt; k++)
          if (outs[k])
            gst_buffer_unref (outs[k]);
        gst_buffer_list_unref (out_list);
        GST_ELEMENT_ERROR (s, LIBRARY, FAILED, ("HIP resampling of a buffer list failed"), ("%s", gstamd_last_error ()));
        ret = GST_FLOW_ERROR;
        break;
      }
      for (k = 0; k < cnt; k++) {
        s->samples_in += in_frames[k];
        if (!outs[k])
          continue;
        amd_ar_finish_layout (s, outs[k], out_cum[k + 1] - out_cum[k]);
        amd_ar_stamp (s, outs[k], out_cum[k + 1] - out_cum[k]);
        gst_buffer_list_add (out_list, outs[k]);
      }
      s->n_list_calls++;
      s->n_list_buffers += cnt;
      if (g_getenv ("GSTAMD_ELEMENT_STATS"))
        g_printerr ("amdaudioresample: %u buffers of a list (%" G_GSIZE_FORMAT " frames) in one resample call\n", cnt, total_in);
      if (gst_buffer_list_length (out_list) > 0)
        ret = gst_pad_push_list (GST_BASE_TRANSFORM_SRC_PAD (trans), out_list);
      else
        gst_buffer_list_unref (out_list);
      i += cnt;
    }
  }
  gst_buffer_list_unref (list);
  return ret;
}

/* drain: feed max-latency frames of silence, push what comes out */
static void
amd_ar_drain (GstAmdAudioResample * s)
{
  gsize lat, out_frames;
  GstBuffer *buf;
  GstMapInfo om;
  gssize n;

  if (!s->r || s->samples_out == 0)
    return;
  lat = gstamd_audio_resampler_get_max_latency (s->r);
  out_frames = gstamd_audio_resampler_get_out_frames (s->r, lat);
  if (out_frames == 0)
    return;
  buf = gst_buffer_new_and_alloc (out_frames * GST_AUDIO_INFO_BPF (&s->out));
  gst_buffer_map (buf, &om, GST_MAP_WRITE);
  n = amd_ar_process (s, NULL, NULL, lat, om.data, out_frames);
  gst_buffer_unmap (buf, &om);
  if (n <= 0) {
    gst_buffer_unref (buf);
    return;
  }
  gst_buffer_set_size (buf, (gsize) n * GST_AUDIO_INFO_BPF (&s->out));
  amd_ar_finish_layout (s, buf, (gsize) n);
  amd_ar_stamp (s, buf, (gsize) n);
  gst_pad_push (GST_BASE_TRANSFORM_SRC_PAD (s), buf);
}

static gboolean
amd_ar_sink_event (GstBaseTransform * t, GstEvent * event)
{
  GstAmdAudioResample *s = AMD_AR (t);
  if (GST_EVENT_TYPE (event) == GST_EVENT_EOS)
    amd_ar_drain (s);
  else if (GST_EVENT_TYPE (event) == GST_EVENT_FLUSH_STOP && s->r) {
    gstamd_audio_resampler_reset (s->r);
    s->samples_in = s->samples_out = 0;
    s->t0 = GST_CLOCK_TIME_NONE;
    s->need_discont = TRUE;
  }
  return GST_BASE_TRANSFORM_CLASS (gst_amd_ar_parent_class)->sink_event (t, event);
}

static gboolean
amd_ar_stop (GstBaseTransform * t)
{
  GstAmdAudioResample *s = AMD_AR (t);
  if (g_getenv ("GSTAMD_ELEMENT_STATS") && s->n_list_calls)
    g_printerr ("amdaudioresample: %" G_GUINT64_FORMAT " buffers of lists in %" G_GUINT64_FORMAT " resample calls\n", s->n_list_buffers, s->n_list_calls);
  if (s->r)
    gstamd_audio_resampler_free (s->r);
  s->r = NULL;
  gst_amd_hip_select_device (s->device_id);
  if (s->stream) {
    gstamd_stream_synchronize (s->stream);
    gstamd_stream_free (s->stream);
    s->stream = NULL;
  }
  gstamd_device_free (s->d_in);
  gstamd_device_free (s->d_out);
  s->d_in = s->d_out = NULL;
  s->d_in_size = s->d_out_size = 0;
  gstamd_host_free (s->h_in);
  gstamd_host_free (s->h_out);
  s->h_in = s->h_out = NULL;
  s->h_in_size = s->h_out_size = 0;
  return TRUE;
}

static void
gst_amd_ar_class_init (GstAmdAudioResampleClass * klass)
{
  GObjectClass *oc = G_OBJECT_CLASS (klass);
  GstElementClass *ec = GST_ELEMENT_CLASS (klass);
  GstBaseTransformClass *tc = GST_BASE_TRANSFORM_CLASS (klass);
  static const GEnumValue mv[] = {{0, "nearest", "nearest"}, {1, "linear", "linear"}, {2, "cubic", "cubic"},
    {3, "blackman-nuttall", "blackman-nuttall"}, {4, "kaiser", "kaiser"}, {0, NULL, NULL}};
  GType mt = g_type_from_name ("GstAmdAudioResamplerMethod");

  if (!mt)
    mt = g_enum_register_static ("GstAmdAudioResamplerMethod", mv);
  GST_DEBUG_CATEGORY_INIT (amd_ar_debug, "amdaudioresample", 0, "MI355X audioresample");
  oc->set_property = amd_ar_set_property;
  oc->get_property = amd_ar_get_property;
  g_object_class_install_property (oc, PROP_QUALITY, g_param_spec_int ("quality", "Quality",
          "Resample quality with 0 being the lowest and 10 being the best", 0, 10, 4,
          G_PARAM_READWRITE | G_PARAM_CONSTRUCT | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_METHOD, g_param_spec_enum ("resample-method", "Resample method to use",
          "What resample method to use", mt, 4, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  {
    static const GEnumValue fm[] = {{0, "Use interpolated filter tables", "interpolated"}, {1, "Use full filter table", "full"},
      {2, "Automatically choose based on filter table size", "auto"}, {0, NULL, NULL}};
    static const GEnumValue fi[] = {{0, "No interpolation", "none"}, {1, "Linear interpolation of the filter coefficients", "linear"},
      {2, "Cubic interpolation of the filter coefficients", "cubic"}, {0, NULL, NULL}};
    GType fmt = g_type_from_name ("GstAmdAudioResamplerFilterMode"), fit = g_type_from_name ("GstAmdAudioResamplerFilterInterpolation");
    if (!fmt)
      fmt = g_enum_register_static ("GstAmdAudioResamplerFilterMode", fm);
    if (!fit)
      fit = g_enum_register_static ("GstAmdAudioResamplerFilterInterpolation", fi);
    /* names, ranges and defaults of gstaudioresample.c:165-185 */
    g_object_class_install_property (oc, PROP_SINC_FILTER_MODE, g_param_spec_enum ("sinc-filter-mode", "Sinc filter table mode",
            "What sinc filter table mode to use", fmt, 2, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
    g_object_class_install_property (oc, PROP_SINC_FILTER_AUTO_THRESHOLD, g_param_spec_uint ("sinc-filter-auto-threshold",
            "Sinc filter auto mode threshold", "Memory usage threshold to use if sinc filter mode is AUTO, given in bytes", 0, G_MAXUINT,
            1048576, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
    g_object_class_install_property (oc, PROP_SINC_FILTER_INTERPOLATION, g_param_spec_enum ("sinc-filter-interpolation",
            "Sinc filter interpolation", "How to interpolate the sinc filter table", fit, 2, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  }
  g_object_class_install_property (oc, PROP_DEVICE_ID, g_param_spec_int ("device-id", "Device ID",
          "HIP device this instance runs on (-1 = the process's current device)", -1, G_MAXINT, -1, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  gst_element_class_set_static_metadata (ec, "Audio resampler (MI355X/HIP)", "Filter/Converter/Audio",
      "Resamples audio with a polyphase FIR on an AMD Instinct GPU, bit-exact to GstAudioResampler", "gstreamer_amd");
  gst_element_class_add_static_pad_template (ec, &ar_sink);
  gst_element_class_add_static_pad_template (ec, &ar_src);
  tc->passthrough_on_same_caps = TRUE;
  tc->transform_caps = GST_DEBUG_FUNCPTR (amd_ar_transform_caps);
  tc->fixate_caps = GST_DEBUG_FUNCPTR (amd_ar_fixate_caps);
  tc->set_caps = GST_DEBUG_FUNCPTR (amd_ar_set_caps);
  tc->get_unit_size = GST_DEBUG_FUNCPTR (amd_ar_get_unit_size);
  tc->transform_size = GST_DEBUG_FUNCPTR (amd_ar_transform_size);
  tc->transform = GST_DEBUG_FUNCPTR (amd_ar_transform);
  tc->sink_event = GST_DEBUG_FUNCPTR (amd_ar_sink_event);
  tc->stop = GST_DEBUG_FUNCPTR (amd_ar_stop);
}

static void
gst_amd_ar_init (GstAmdAudioResample * s)
{
  s->quality = 4;
  s->method = GSTAMD_AUDIO_RESAMPLER_METHOD_KAISER;
  s->sinc_filter_mode = GSTAMD_AUDIO_FILTER_MODE_AUTO;
  s->sinc_filter_interpolation = GSTAMD_AUDIO_FILTER_INTERPOLATION_CUBIC;
  s->sinc_filter_auto_threshold = 1048576;
  s->t0 = GST_CLOCK_TIME_NONE;
  s->device_id = -1;
  s->need_discont = TRUE;
  s->in_offset0 = s->out_offset0 = GST_BUFFER_OFFSET_NONE;
  s->base_chain = GST_PAD_CHAINFUNC (GST_BASE_TRANSFORM_SINK_PAD (s));
  gst_pad_set_chain_list_function (GST_BASE_TRANSFORM_SINK_PAD (s), GST_DEBUG_FUNCPTR (amd_ar_chain_list));
}

GType
gst_amd_audio_resample_get_type (void)
{
  return gst_amd_ar_get_type ();
}
