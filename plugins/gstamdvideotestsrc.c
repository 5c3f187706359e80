/* gstamdvideotestsrc.c - `amdhipvideotestsrc`: GstVideoTestSrc's frames born in HBM (round 6).
 *
 * The reference's videotestsrc (gst/videotestsrc/gstvideotestsrc.c) paints every frame on the CPU; in front of the HIP elements that is a host frame and
 * an upload per buffer.  This source paints in device memory (gstamd_video_test_pattern_*, include/gstamd_video.h: one kernel for the painted lines,
 * the library's chroma downsampler + packer for the caps' format) and pushes video/x-raw(memory:AMDHIPMemory) buffers of its own HBM pool.
 * Byte for byte the reference element's frames (tests/test_plugin_gpu.py) for the patterns the library paints; the others are refused at set_caps.
 *
 * Properties mirror the reference's where they exist here: pattern (same enum values and nicks, gstvideotestsrc.c:134-175), foreground-color,
 * background-color, is-live, timestamp-offset; GstBaseSrc's num-buffers.  Timestamps, durations and offsets as gst_video_test_src_fill
 * (gstvideotestsrc.c:1269-1330): running frame count over the caps' framerate. */
#include <gst/gst.h>
#include <gst/base/gstpushsrc.h>
#include <gst/video/video.h>
#include <string.h>

#include "../include/gstamd_video.h"
#include "gstamdhipmemory.h"
#include "gstamdhipbufferpool.h"

GST_DEBUG_CATEGORY_STATIC (amd_vts_debug);
#define GST_CAT_DEFAULT amd_vts_debug

typedef struct _GstAmdVideoTestSrc {
  GstPushSrc parent;
  gint pattern;
  guint foreground_color, background_color;
  gint64 timestamp_offset;
  gint device_id;
  GstVideoInfo info;
  GstAmdVideoTestPattern *painter;
  gpointer stream;
  GstBufferPool *pool;
  guint64 n_frames;
  GstClockTime running_time;
  gint64 accum_frames;
  GstClockTime accum_rtime;
} GstAmdVideoTestSrc;

typedef struct _GstAmdVideoTestSrcClass {
  GstPushSrcClass parent_class;
} GstAmdVideoTestSrcClass;

GType gst_amd_video_test_src_get_type (void);
G_DEFINE_TYPE (GstAmdVideoTestSrc, gst_amd_video_test_src, GST_TYPE_PUSH_SRC);
#define AMD_VTS(o) ((GstAmdVideoTestSrc *) (o))

enum { PROP_0, PROP_PATTERN, PROP_FOREGROUND, PROP_BACKGROUND, PROP_IS_LIVE, PROP_TIMESTAMP_OFFSET, PROP_DEVICE_ID };

/* GstVideoTestSrcPattern (gstvideotestsrc.h:84-112, gstvideotestsrc.c:134-175): values and nicks */
static GType
amd_vts_pattern_get_type (void)
{
  static GType t = 0;
  static const GEnumValue v[] = {
    {0, "SMPTE 100% color bars", "smpte"}, {1, "Random (television snow)", "snow"}, {2, "100% Black", "black"}, {3, "100% White", "white"},
    {4, "Red", "red"}, {5, "Green", "green"}, {6, "Blue", "blue"}, {7, "Checkers 1px", "checkers-1"}, {8, "Checkers 2px", "checkers-2"},
    {9, "Checkers 4px", "checkers-4"}, {10, "Checkers 8px", "checkers-8"}, {11, "Circular", "circular"}, {12, "Blink", "blink"},
    {13, "SMPTE 75% color bars", "smpte75"}, {14, "Zone plate", "zone-plate"}, {15, "Gamut checkers", "gamut"}, {16, "Chroma zone plate", "chroma-zone-plate"},
    {17, "Solid color", "solid-color"}, {18, "Moving ball", "ball"}, {19, "SMPTE 100% color bars", "smpte100"}, {20, "Bar", "bar"}, {21, "Pinwheel", "pinwheel"},
    {22, "Spokes", "spokes"}, {23, "Gradient", "gradient"}, {24, "Colors", "colors"}, {25, "SMPTE test pattern, RP 219 conformant", "smpte-rp-219"},
    {0, NULL, NULL}
  };
  if (g_once_init_enter (&t)) {
    GType n = g_enum_register_static ("GstAmdVideoTestSrcPattern", v);
    g_once_init_leave (&t, n);
  }
  return t;
}

static void
amd_vts_set_property (GObject * o, guint id, const GValue * value, GParamSpec * ps)
{
  GstAmdVideoTestSrc *s = AMD_VTS (o);
  switch (id) {
    case PROP_PATTERN: s->pattern = g_value_get_enum (value); break;
    case PROP_FOREGROUND: s->foreground_color = g_value_get_uint (value); break;
    case PROP_BACKGROUND: s->background_color = g_value_get_uint (value); break;
    case PROP_IS_LIVE: gst_base_src_set_live (GST_BASE_SRC (s), g_value_get_boolean (value)); break;
    case PROP_TIMESTAMP_OFFSET: s->timestamp_offset = g_value_get_int64 (value); break;
    case PROP_DEVICE_ID: s->device_id = g_value_get_int (value); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID (o, id, ps); break;
  }
}

static void
amd_vts_get_property (GObject * o, guint id, GValue * value, GParamSpec * ps)
{
  GstAmdVideoTestSrc *s = AMD_VTS (o);
  switch (id) {
    case PROP_PATTERN: g_value_set_enum (value, s->pattern); break;
    case PROP_FOREGROUND: g_value_set_uint (value, s->foreground_color); break;
    case PROP_BACKGROUND: g_value_set_uint (value, s->background_color); break;
    case PROP_IS_LIVE: g_value_set_boolean (value, gst_base_src_is_live (GST_BASE_SRC (s))); break;
    case PROP_TIMESTAMP_OFFSET: g_value_set_int64 (value, s->timestamp_offset); break;
    case PROP_DEVICE_ID: g_value_set_int (value, s->device_id); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID (o, id, ps); break;
  }
}

/* gst_video_test_src_src_fixate (gstvideotestsrc.c:780-830): 320 x 240 at 30/1 where the peer leaves them open, the first format */
static GstCaps *
amd_vts_fixate (GstBaseSrc * bsrc, GstCaps * caps)
{
  GstStructure *st;

  caps = gst_caps_make_writable (caps);
  caps = gst_caps_truncate (caps);
  st = gst_caps_get_structure (caps, 0);
  gst_structure_fixate_field_nearest_int (st, "width", 320);
  gst_structure_fixate_field_nearest_int (st, "height", 240);
  if (gst_structure_has_field (st, "framerate"))
    gst_structure_fixate_field_nearest_fraction (st, "framerate", 30, 1);
  else
    gst_structure_set (st, "framerate", GST_TYPE_FRACTION, 30, 1, NULL);
  if (gst_structure_has_field (st, "pixel-aspect-ratio"))
    gst_structure_fixate_field_nearest_fraction (st, "pixel-aspect-ratio", 1, 1);
  if (gst_structure_has_field (st, "interlace-mode"))
    gst_structure_fixate_field_string (st, "interlace-mode", "progressive");
  return GST_BASE_SRC_CLASS (gst_amd_video_test_src_parent_class)->fixate (bsrc, caps);
}

static void
amd_vts_release (GstAmdVideoTestSrc * s)
{
  if (s->painter) {
    gst_amd_hip_select_device (s->device_id);
    if (s->stream)
      gstamd_stream_synchronize (s->stream);
    gstamd_video_test_pattern_free (s->painter);
    s->painter = NULL;
  }
  if (s->pool) {
    gst_buffer_pool_set_active (s->pool, FALSE);
    gst_object_unref (s->pool);
    s->pool = NULL;
  }
}

static gboolean
amd_vts_set_caps (GstBaseSrc * bsrc, GstCaps * caps)
{
  GstAmdVideoTestSrc *s = AMD_VTS (bsrc);
  GstAmdVideoInfo ai;
  int status = 0;

  if (!gst_video_info_from_caps (&s->info, caps))
    return FALSE;
  gst_amd_hip_select_device (s->device_id);
  amd_vts_release (s);
  if (!gst_amd_video_info_fill (&s->info, &ai)) {
    GST_ERROR_OBJECT (s, "format not supported by the HIP test source");
    return FALSE;
  }
  if (GST_VIDEO_INFO_IS_INTERLACED (&s->info)) {
    GST_ERROR_OBJECT (s, "progressive frames only");
    return FALSE;
  }
  s->painter = gstamd_video_test_pattern_new (&ai, s->pattern, s->foreground_color, s->background_color, &status);
  if (!s->painter) {
    GST_ERROR_OBJECT (s, "pattern %d on these caps: %s", s->pattern, gstamd_last_error ());
    return FALSE;
  }
  GST_DEBUG_OBJECT (s, "HIP painter: %s", gstamd_video_test_pattern_describe (s->painter));
  if (!s->stream)
    s->stream = gstamd_stream_new ();
  s->pool = gst_amd_hip_buffer_pool_new_for_caps (caps, 4);
  if (!s->pool || !gst_buffer_pool_set_active (s->pool, TRUE)) {
    GST_ERROR_OBJECT (s, "could not set up the HBM buffer pool");
    return FALSE;
  }
  s->accum_rtime += s->running_time;
  s->accum_frames += (gint64) s->n_frames;
  s->running_time = 0;
  s->n_frames = 0;
  return TRUE;
}

static gboolean
amd_vts_start (GstBaseSrc * bsrc)
{
  GstAmdVideoTestSrc *s = AMD_VTS (bsrc);
  s->running_time = 0;
  s->n_frames = 0;
  s->accum_frames = 0;
  s->accum_rtime = 0;
  gst_video_info_init (&s->info);
  return TRUE;
}

static gboolean
amd_vts_stop (GstBaseSrc * bsrc)
{
  GstAmdVideoTestSrc *s = AMD_VTS (bsrc);
  amd_vts_release (s);
  if (s->stream) {
    gst_amd_hip_select_device (s->device_id);
    gst_amd_hip_stream_retire (s->stream);
    gstamd_stream_free (s->stream);
    s->stream = NULL;
  }
  return TRUE;
}

static gboolean
amd_vts_is_seekable (GstBaseSrc * bsrc)
{
  return FALSE;                 /* (the reference's source seeks by frame number; this one runs forward only) */
}

static gboolean
amd_vts_decide_allocation (GstBaseSrc * bsrc, GstQuery * query)
{
  return TRUE;                  /* frames come from the element's own HBM pool (create) */
}

static void
amd_vts_get_times (GstBaseSrc * bsrc, GstBuffer * buffer, GstClockTime * start, GstClockTime * end)
{
  /* gst_video_test_src_get_times (gstvideotestsrc.c:1105-1125): sync to the clock only when live */
  if (gst_base_src_is_live (bsrc)) {
    GstClockTime ts = GST_BUFFER_PTS (buffer);
    if (GST_CLOCK_TIME_IS_VALID (ts)) {
      GstClockTime d = GST_BUFFER_DURATION (buffer);
      if (GST_CLOCK_TIME_IS_VALID (d))
        *end = ts + d;
      *start = ts;
    }
  } else {
    *start = *end = GST_CLOCK_TIME_NONE;
  }
}

static GstFlowReturn
amd_vts_create (GstPushSrc * psrc, GstBuffer ** out)
{
  GstAmdVideoTestSrc *s = AMD_VTS (psrc);
  GstBuffer *buf = NULL;
  GstMemory *mem;
  GstMapInfo map;
  GstFlowReturn fr;
  GstClockTime next;
  int r;

  if (!s->painter || !s->pool)
    return GST_FLOW_NOT_NEGOTIATED;
  /* 0 framerate and we are at the second frame: eos (gst_video_test_src_fill :1284) */
  if (s->info.fps_n == 0 && s->n_frames == 1)
    return GST_FLOW_EOS;
  gst_amd_hip_select_device (s->device_id);
  if ((fr = gst_buffer_pool_acquire_buffer (s->pool, &buf, NULL)) != GST_FLOW_OK)
    return fr;
  mem = gst_buffer_peek_memory (buf, 0);
  if (!gst_memory_map (mem, &map, GST_MAP_WRITE | GST_MAP_AMDHIP)) {
    gst_buffer_unref (buf);
    return GST_FLOW_ERROR;
  }
  gst_amd_hip_memory_wait_idle (mem, s->stream);          /* a recycled pool frame may still be read downstream */
  r = gstamd_video_test_pattern_frame (s->painter, s->n_frames, map.data, s->stream);
  gst_memory_unmap (mem, &map);
  if (r != GSTAMD_OK) {
    GST_ELEMENT_ERROR (s, LIBRARY, FAILED, ("HIP test pattern failed"), ("%s", gstamd_last_error ()));
    gst_buffer_unref (buf);
    return GST_FLOW_ERROR;
  }
  gst_amd_hip_memory_mark_written (mem, s->stream);
  /* timestamps as gst_video_test_src_fill (:1300-1330) */
  GST_BUFFER_PTS (buf) = s->accum_rtime + s->timestamp_offset + s->running_time;
  GST_BUFFER_DTS (buf) = GST_CLOCK_TIME_NONE;
  GST_BUFFER_OFFSET (buf) = s->accum_frames + s->n_frames;
  s->n_frames++;
  GST_BUFFER_OFFSET_END (buf) = GST_BUFFER_OFFSET (buf) + 1;
  if (s->info.fps_n) {
    next = gst_util_uint64_scale (s->n_frames, s->info.fps_d * GST_SECOND, s->info.fps_n);
    GST_BUFFER_DURATION (buf) = next - s->running_time;
  } else {
    next = s->timestamp_offset;
    GST_BUFFER_DURATION (buf) = GST_CLOCK_TIME_NONE;
  }
  s->running_time = next;
  *out = buf;
  return GST_FLOW_OK;
}

static void
amd_vts_finalize (GObject * o)
{
  G_OBJECT_CLASS (gst_amd_video_test_src_parent_class)->finalize (o);
}

static void
gst_amd_video_test_src_class_init (GstAmdVideoTestSrcClass * klass)
{
  GObjectClass *oc = G_OBJECT_CLASS (klass);
  GstElementClass *ec = GST_ELEMENT_CLASS (klass);
  GstBaseSrcClass *bc = GST_BASE_SRC_CLASS (klass);
  GstPushSrcClass *pc = GST_PUSH_SRC_CLASS (klass);
  GstCaps *caps;
  gchar *str;

  GST_DEBUG_CATEGORY_INIT (amd_vts_debug, "amdhipvideotestsrc", 0, "HIP video test source");
  oc->set_property = amd_vts_set_property;
  oc->get_property = amd_vts_get_property;
  oc->finalize = amd_vts_finalize;
  g_object_class_install_property (oc, PROP_PATTERN, g_param_spec_enum ("pattern", "Pattern", "Type of test pattern to generate", amd_vts_pattern_get_type (), 0,
          G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_FOREGROUND, g_param_spec_uint ("foreground-color", "Foreground Color",
          "Foreground color to use (big-endian ARGB)", 0, G_MAXUINT32, 0xffffffff, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_BACKGROUND, g_param_spec_uint ("background-color", "Background Color",
          "Background color to use (big-endian ARGB)", 0, G_MAXUINT32, 0xff000000, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_IS_LIVE, g_param_spec_boolean ("is-live", "Is Live", "Whether to act as a live source", FALSE,
          G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_TIMESTAMP_OFFSET, g_param_spec_int64 ("timestamp-offset", "Timestamp offset",
          "An offset added to timestamps set on buffers (in ns)", 0, (G_MAXLONG == G_MAXINT64) ? G_MAXINT64 : (G_MAXLONG * GST_SECOND - 1), 0,
          G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_DEVICE_ID, g_param_spec_int ("device-id", "Device", "HIP device the frames are painted on (-1: the process's current device)",
          -1, 255, -1, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  gst_element_class_set_static_metadata (ec, "Video test source (MI355X/HIP)", "Source/Video",
      "Paints GstVideoTestSrc's test patterns into HBM frames, bit-exact to videotestsrc", "gstreamer_amd");
  str = g_strdup_printf ("video/x-raw(" GST_CAPS_FEATURE_MEMORY_AMD_HIP "), format=(string)%s, width=(int)[1, 32767], height=(int)[1, 32767], "
      "framerate=(fraction)[0/1, 2147483647/1]", gst_amd_video_formats_string ());
  caps = gst_caps_from_string (str);
  g_free (str);
  gst_element_class_add_pad_template (ec, gst_pad_template_new ("src", GST_PAD_SRC, GST_PAD_ALWAYS, caps));
  gst_caps_unref (caps);
  bc->set_caps = amd_vts_set_caps;
  bc->fixate = amd_vts_fixate;
  bc->is_seekable = amd_vts_is_seekable;
  bc->get_times = amd_vts_get_times;
  bc->start = amd_vts_start;
  bc->stop = amd_vts_stop;
  bc->decide_allocation = amd_vts_decide_allocation;
  pc->create = amd_vts_create;
}

static void
gst_amd_video_test_src_init (GstAmdVideoTestSrc * s)
{
  s->pattern = 0;
  s->foreground_color = 0xffffffff;
  s->background_color = 0xff000000;
  s->timestamp_offset = 0;
  s->device_id = -1;
  gst_base_src_set_format (GST_BASE_SRC (s), GST_FORMAT_TIME);
  gst_base_src_set_live (GST_BASE_SRC (s), FALSE);
}
