/* gstamdhiptransfer.c - `amdhipupload` / `amdhipdownload`: system memory <-> memory:AMDHIPMemory at a pipeline's edges, so that the
 * elements in between negotiate HBM caps and never copy (the GL / CUDA / HIP plugins of the reference ship the same pair, e.g.
 * subprojects/gst-plugins-bad/sys/hip/gsthipmemorycopy.c: contract only, no code from there).
 *
 *   upload:   video/x-raw -> video/x-raw(memory:AMDHIPMemory); output frames from a GstAmdHipBufferPool, the copy enqueued on the
 *             instance's stream and published with a ticket (gstamdhipmemory.h) - no host wait
 *   download: the reverse; waits for the producer's ticket on the stream, copies, and synchronises (the CPU owns the result)
 * Buffers already in the target memory pass through. */
#include <gst/base/gstbasetransform.h>
#include <gst/gst.h>
#include <gst/video/video.h>
#include <string.h>

#include "../include/gstamd_video.h"
#include "gstamdhipbufferpool.h"
#include "gstamdhipmemory.h"

GST_DEBUG_CATEGORY_STATIC (amd_transfer_debug);
#define GST_CAT_DEFAULT amd_transfer_debug

typedef struct {
  GstBaseTransform parent;
  gint device_id;
  gpointer stream;
  GstBufferPool *pool;          /* upload: HBM output frames */
  GstVideoInfo info;
  GstAmdHipPendingReads *reads; /* upload: input buffers whose transfer is still queued */
} GstAmdHipTransfer;

typedef struct {
  GstBaseTransformClass parent_class;
  gboolean upload;
} GstAmdHipTransferClass;

G_DEFINE_TYPE (GstAmdHipTransfer, gst_amd_hip_transfer, GST_TYPE_BASE_TRANSFORM);
#define AMD_TR(o) ((GstAmdHipTransfer *) (o))
#define AMD_TR_CLASS(o) ((GstAmdHipTransferClass *) G_OBJECT_GET_CLASS (o))

static GstStaticPadTemplate tr_sink = GST_STATIC_PAD_TEMPLATE ("sink", GST_PAD_SINK, GST_PAD_ALWAYS,
    GST_STATIC_CAPS ("video/x-raw(" GST_CAPS_FEATURE_MEMORY_AMD_HIP "); video/x-raw"));
static GstStaticPadTemplate tr_src = GST_STATIC_PAD_TEMPLATE ("src", GST_PAD_SRC, GST_PAD_ALWAYS,
    GST_STATIC_CAPS ("video/x-raw(" GST_CAPS_FEATURE_MEMORY_AMD_HIP "); video/x-raw"));

enum { PROP_0, PROP_DEVICE_ID };

static void
tr_set_property (GObject * o, guint id, const GValue * v, GParamSpec * p)
{
  if (id == PROP_DEVICE_ID)
    AMD_TR (o)->device_id = g_value_get_int (v);
  else
    G_OBJECT_WARN_INVALID_PROPERTY_ID (o, id, p);
}

static void
tr_get_property (GObject * o, guint id, GValue * v, GParamSpec * p)
{
  if (id == PROP_DEVICE_ID)
    g_value_set_int (v, AMD_TR (o)->device_id);
  else
    G_OBJECT_WARN_INVALID_PROPERTY_ID (o, id, p);
}

/* the other side carries the same video description in the target memory (first choice) or unchanged (passthrough) */
static GstCaps *
tr_transform_caps (GstBaseTransform * trans, GstPadDirection direction, GstCaps * caps, GstCaps * filter)
{
  const gboolean to_hip = (direction == GST_PAD_SINK) == AMD_TR_CLASS (trans)->upload;
  GstCaps *ret = gst_caps_new_empty ();
  guint i, n = gst_caps_get_size (caps);

  for (i = 0; i < n; i++) {
    GstStructure *st = gst_caps_get_structure (caps, i);
    gst_caps_append_structure_full (ret, gst_structure_copy (st),
        to_hip ? gst_caps_features_new (GST_CAPS_FEATURE_MEMORY_AMD_HIP, NULL) : gst_caps_features_new (GST_CAPS_FEATURE_MEMORY_SYSTEM_MEMORY, NULL));
  }
  ret = gst_caps_merge (ret, gst_caps_copy (caps));
  if (filter) {
    GstCaps *tmp = gst_caps_intersect_full (filter, ret, GST_CAPS_INTERSECT_FIRST);
    gst_caps_unref (ret);
    ret = tmp;
  }
  return ret;
}

static gboolean
caps_hip (GstCaps * caps)
{
  GstCapsFeatures *f = gst_caps_get_features (caps, 0);
  return f && gst_caps_features_contains (f, GST_CAPS_FEATURE_MEMORY_AMD_HIP);
}

static gboolean
tr_set_caps (GstBaseTransform * trans, GstCaps * incaps, GstCaps * outcaps)
{
  GstAmdHipTransfer *s = AMD_TR (trans);

  if (!gst_video_info_from_caps (&s->info, incaps))
    return FALSE;
  gst_amd_hip_select_device (s->device_id);
  if (s->pool) {
    gst_buffer_pool_set_active (s->pool, FALSE);
    gst_object_unref (s->pool);
    s->pool = NULL;
  }
  gst_base_transform_set_passthrough (trans, caps_hip (incaps) == caps_hip (outcaps));
  if (caps_hip (outcaps) && !caps_hip (incaps) && !(s->pool = gst_amd_hip_buffer_pool_new_for_caps (outcaps, 4)))
    return FALSE;
  if (!s->stream && !(s->stream = gstamd_stream_new ()))
    return FALSE;
  return TRUE;
}

static gboolean
tr_get_unit_size (GstBaseTransform * trans, GstCaps * caps, gsize * size)
{
  GstVideoInfo info;
  if (!gst_video_info_from_caps (&info, caps))
    return FALSE;
  *size = GST_VIDEO_INFO_SIZE (&info);
  return TRUE;
}

static GstFlowReturn
tr_prepare_output_buffer (GstBaseTransform * trans, GstBuffer * inbuf, GstBuffer ** outbuf)
{
  GstAmdHipTransfer *s = AMD_TR (trans);

  if (gst_base_transform_is_passthrough (trans)) {
    *outbuf = inbuf;
    return GST_FLOW_OK;
  }
  if (!s->pool)
    return GST_BASE_TRANSFORM_CLASS (gst_amd_hip_transfer_parent_class)->prepare_output_buffer (trans, inbuf, outbuf);
  gst_amd_hip_select_device (s->device_id);
  {
    const GstFlowReturn fr = gst_buffer_pool_acquire_buffer (s->pool, outbuf, NULL);
    if (fr != GST_FLOW_OK)
      return fr;                /* FLUSHING is the pool's answer during a seek / shutdown, not an error */
  }
  gst_buffer_copy_into (*outbuf, inbuf, GST_BUFFER_COPY_FLAGS | GST_BUFFER_COPY_TIMESTAMPS, 0, -1);
  return GST_FLOW_OK;
}

/* the metas of the input travel with the frame (GstBaseTransform's default copy_metadata runs for pool buffers too); the GstVideoMeta
 * does not: it describes the INPUT's plane layout, the output frame has the pool's */
static gboolean
tr_transform_meta (GstBaseTransform * trans, GstBuffer * outbuf, GstMeta * meta, GstBuffer * inbuf)
{
  if (meta->info->api == GST_VIDEO_META_API_TYPE)
    return FALSE;
  return GST_BASE_TRANSFORM_CLASS (gst_amd_hip_transfer_parent_class)->transform_meta (trans, outbuf, meta, inbuf);
}

/* rows x bytes of plane k of a frame (gst_video_format_info component geometry; the packed formats have one plane of whole pixels) */
static void
tr_plane_geometry (const GstVideoInfo * info, guint k, gsize * row_bytes, gsize * rows)
{
  const GstVideoFormatInfo *f = info->finfo;
  gint comp[GST_VIDEO_MAX_COMPONENTS];
  guint c, n = 0;
  gsize bytes = 0;
  /* the components stored in plane k (gst_video_format_info_component exists from 1.18 on only) */
  for (c = 0; c < GST_VIDEO_FORMAT_INFO_N_COMPONENTS (f); c++)
    if (GST_VIDEO_FORMAT_INFO_PLANE (f, c) == k)
      comp[n++] = (gint) c;
  /* whole pixels: width x pixel stride of the widest component of the plane (the last sample's own bytes are too few where a pixel is
   * wider than its last component: RGB16 / RGB15, Y410, r210).  Formats whose samples do not sit at a fixed stride (pixel stride 0 /
   * GST_VIDEO_FORMAT_FLAG_COMPLEX: v210, UYVP ...) copy the rows of the format's default layout. */
  {
    gboolean fixed = !(GST_VIDEO_FORMAT_INFO_FLAGS (f) & GST_VIDEO_FORMAT_FLAG_COMPLEX);
    for (c = 0; c < n && fixed; c++)
      fixed = GST_VIDEO_FORMAT_INFO_PSTRIDE (f, comp[c]) > 0;
    if (fixed) {
      for (c = 0; c < n; c++) {
        const gint cw = GST_VIDEO_FORMAT_INFO_SCALE_WIDTH (f, comp[c], GST_VIDEO_INFO_WIDTH (info));
        bytes = MAX (bytes, (gsize) cw * (gsize) GST_VIDEO_FORMAT_INFO_PSTRIDE (f, comp[c]));
      }
    } else if (n) {
      GstVideoInfo def;
      gst_video_info_init (&def);
      if (gst_video_info_set_format (&def, GST_VIDEO_INFO_FORMAT (info), GST_VIDEO_INFO_WIDTH (info), GST_VIDEO_INFO_HEIGHT (info)))
        bytes = (gsize) GST_VIDEO_INFO_PLANE_STRIDE (&def, k);
    }
  }
  *row_bytes = bytes;
  *rows = n ? GST_VIDEO_FORMAT_INFO_SCALE_HEIGHT (f, comp[0], GST_VIDEO_INFO_HEIGHT (info)) : 0;
}

/* TRUE: the buffer's planes lie where `info` (the layout of the pool's frames: default strides) puts them - one flat copy serves */
static gboolean
tr_layout_is_default (GstBuffer * buf, const GstVideoInfo * info)
{
  const GstVideoMeta *m = gst_buffer_get_video_meta (buf);
  guint k;
  if (!m)
    return TRUE;
  if (m->n_planes != GST_VIDEO_INFO_N_PLANES (info))
    return FALSE;
  for (k = 0; k < m->n_planes; k++)
    if (m->offset[k] != GST_VIDEO_INFO_PLANE_OFFSET (info, k) || m->stride[k] != GST_VIDEO_INFO_PLANE_STRIDE (info, k))
      return FALSE;
  return TRUE;
}

/* plane-by-plane pitched copy between a system-memory frame described by its GstVideoMeta and a frame in the default layout */
static int
tr_copy_planes (GstAmdHipTransfer * s, gboolean upload, guint8 * dev, guint8 * host, gsize host_size, const GstVideoMeta * m)
{
  guint k;
  int r = GSTAMD_OK;
  for (k = 0; k < GST_VIDEO_INFO_N_PLANES (&s->info) && r == GSTAMD_OK; k++) {
    gsize row_bytes, rows;
    tr_plane_geometry (&s->info, k, &row_bytes, &rows);
    /* a complex format's row is its default layout's row, padding included: a tightly packed producer's own pitch may be smaller */
    if (m->stride[k] > 0 && row_bytes > (gsize) m->stride[k])
      row_bytes = (gsize) m->stride[k];
    if (rows && m->offset[k] + (rows - 1) * (gsize) m->stride[k] + row_bytes > host_size)
      return GSTAMD_ERR_INVALID;
    if (upload)
      r = gstamd_device_upload_2d_async (dev + GST_VIDEO_INFO_PLANE_OFFSET (&s->info, k), GST_VIDEO_INFO_PLANE_STRIDE (&s->info, k),
          host + m->offset[k], m->stride[k], row_bytes, rows, s->stream);
    else
      r = gstamd_device_download_2d_async (host + m->offset[k], m->stride[k], dev + GST_VIDEO_INFO_PLANE_OFFSET (&s->info, k),
          GST_VIDEO_INFO_PLANE_STRIDE (&s->info, k), row_bytes, rows, s->stream);
  }
  return r;
}

static GstFlowReturn
tr_transform (GstBaseTransform * trans, GstBuffer * inbuf, GstBuffer * outbuf)
{
  GstAmdHipTransfer *s = AMD_TR (trans);
  GstMemory *imem = gst_buffer_peek_memory (inbuf, 0), *omem = gst_buffer_peek_memory (outbuf, 0);
  const gboolean in_dev = gst_buffer_n_memory (inbuf) == 1 && gst_is_amd_hip_memory (imem);
  const gboolean out_dev = gst_buffer_n_memory (outbuf) == 1 && gst_is_amd_hip_memory (omem);
  GstMapInfo im, om;
  int r = GSTAMD_ERR_INVALID;

  gst_amd_hip_select_device (s->device_id);
  if (!in_dev && out_dev) {               /* upload */
    if (!gst_buffer_map (inbuf, &im, GST_MAP_READ))
      return GST_FLOW_ERROR;
    if (gst_memory_map (omem, &om, GST_MAP_WRITE | GST_MAP_AMDHIP)) {
      gst_amd_hip_memory_wait_idle (omem, s->stream);
      /* A source with its
       * own strides / plane offsets (GstVideoMeta of a decoder or an aligned pool) is copied plane by plane into the pool frame's
       * default layout - a flat copy would shear every row after the first */
      if (tr_layout_is_default (inbuf, &s->info))
        r = gstamd_device_upload_async (om.data, im.data, MIN (im.size, om.size), s->stream);
      else
        r = tr_copy_planes (s, TRUE, om.data, im.data, im.size, gst_buffer_get_video_meta (inbuf));
      if (r == GSTAMD_OK) {
        gst_amd_hip_memory_mark_written (omem, s->stream);
        /* from page-locked memory (an upstream element's pinned pool) the copy is only QUEUED when the call returns: the input stays
         * referenced - out of its pool - until the transfer is over */
        if (!s->reads)
          s->reads = gst_amd_hip_pending_reads_new ();
        gst_amd_hip_pending_reads_hold (s->reads, inbuf, im.data, s->stream);
      }
      gst_memory_unmap (omem, &om);
    }
    gst_buffer_unmap (inbuf, &im);
  } else if (in_dev && !out_dev) {        /* download */
    if (!gst_memory_map (imem, &im, GST_MAP_READ | GST_MAP_AMDHIP))
      return GST_FLOW_ERROR;
    if (gst_buffer_map (outbuf, &om, GST_MAP_WRITE)) {
      gst_amd_hip_memory_wait_written (imem, s->stream);
      if (tr_layout_is_default (outbuf, &s->info))
        r = gstamd_device_download_async (om.data, im.data, MIN (im.size, om.size), s->stream);
      else
        r = tr_copy_planes (s, FALSE, im.data, om.data, om.size, gst_buffer_get_video_meta (outbuf));
      if (r == GSTAMD_OK)
        r = gstamd_stream_synchronize (s->stream);
      gst_buffer_unmap (outbuf, &om);
    }
    gst_memory_unmap (imem, &im);
  } else {                                /* same kind on both sides although not passthrough: plain copy through the maps */
    if (gst_buffer_map (inbuf, &im, GST_MAP_READ)) {
      if (gst_buffer_map (outbuf, &om, GST_MAP_WRITE)) {
        memcpy (om.data, im.data, MIN (im.size, om.size));
        gst_buffer_unmap (outbuf, &om);
        r = GSTAMD_OK;
      }
      gst_buffer_unmap (inbuf, &im);
    }
  }
  if (r != GSTAMD_OK) {
    GST_ELEMENT_ERROR (s, RESOURCE, FAILED, ("HIP transfer failed"), ("%s", gstamd_last_error ()));
    return GST_FLOW_ERROR;
  }
  return GST_FLOW_OK;
}

/* input buffers a queued upload still reads are not kept across a flush or the end of the stream */
static gboolean
tr_sink_event (GstBaseTransform * trans, GstEvent * event)
{
  GstAmdHipTransfer *s = AMD_TR (trans);

  if (GST_EVENT_TYPE (event) == GST_EVENT_FLUSH_STOP || GST_EVENT_TYPE (event) == GST_EVENT_EOS || GST_EVENT_TYPE (event) == GST_EVENT_GAP) {
    gst_amd_hip_select_device (s->device_id);
    gst_amd_hip_pending_reads_drain (s->reads);
  }
  return GST_BASE_TRANSFORM_CLASS (gst_amd_hip_transfer_parent_class)->sink_event (trans, event);
}

static gboolean
tr_stop (GstBaseTransform * trans)
{
  GstAmdHipTransfer *s = AMD_TR (trans);

  gst_amd_hip_select_device (s->device_id);
  if (s->pool) {
    gst_buffer_pool_set_active (s->pool, FALSE);
    gst_object_unref (s->pool);
    s->pool = NULL;
  }
  if (s->reads) {
    gst_amd_hip_pending_reads_free (s->reads);
    s->reads = NULL;
  }
  if (s->stream) {
    gstamd_stream_synchronize (s->stream);
    gstamd_stream_free (s->stream);
    s->stream = NULL;
  }
  return TRUE;
}

static void
gst_amd_hip_transfer_class_init (GstAmdHipTransferClass * klass)
{
  GObjectClass *oc = G_OBJECT_CLASS (klass);
  GstElementClass *ec = GST_ELEMENT_CLASS (klass);
  GstBaseTransformClass *tc = GST_BASE_TRANSFORM_CLASS (klass);

  GST_DEBUG_CATEGORY_INIT (amd_transfer_debug, "amdhiptransfer", 0, "MI355X upload / download");
  oc->set_property = tr_set_property;
  oc->get_property = tr_get_property;
  g_object_class_install_property (oc, PROP_DEVICE_ID, g_param_spec_int ("device-id", "Device ID",
          "HIP device this instance runs on (-1 = the process's current device)", -1, G_MAXINT, -1, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  gst_element_class_add_static_pad_template (ec, &tr_sink);
  gst_element_class_add_static_pad_template (ec, &tr_src);
  tc->passthrough_on_same_caps = TRUE;
  tc->transform_caps = GST_DEBUG_FUNCPTR (tr_transform_caps);
  tc->set_caps = GST_DEBUG_FUNCPTR (tr_set_caps);
  tc->get_unit_size = GST_DEBUG_FUNCPTR (tr_get_unit_size);
  tc->prepare_output_buffer = GST_DEBUG_FUNCPTR (tr_prepare_output_buffer);
  tc->transform = GST_DEBUG_FUNCPTR (tr_transform);
  tc->transform_meta = GST_DEBUG_FUNCPTR (tr_transform_meta);
  tc->stop = GST_DEBUG_FUNCPTR (tr_stop);
  tc->sink_event = GST_DEBUG_FUNCPTR (tr_sink_event);
  klass->upload = TRUE;
}

static void
gst_amd_hip_transfer_init (GstAmdHipTransfer * s)
{
  s->device_id = -1;
  gst_amd_hip_allocator_get ();
}

/* the two factories: one implementation, direction in the class */
typedef GstAmdHipTransfer GstAmdHipUpload;
typedef GstAmdHipTransferClass GstAmdHipUploadClass;
typedef GstAmdHipTransfer GstAmdHipDownload;
typedef GstAmdHipTransferClass GstAmdHipDownloadClass;
G_DEFINE_TYPE (GstAmdHipUpload, gst_amd_hip_upload, gst_amd_hip_transfer_get_type ());
G_DEFINE_TYPE (GstAmdHipDownload, gst_amd_hip_download, gst_amd_hip_transfer_get_type ());

static void
gst_amd_hip_upload_class_init (GstAmdHipUploadClass * klass)
{
  klass->upload = TRUE;
  gst_element_class_set_static_metadata (GST_ELEMENT_CLASS (klass), "HIP uploader (MI355X)", "Filter/Video",
      "Copies system-memory video frames into MI355X HBM (memory:AMDHIPMemory)", "gstreamer_amd");
}

static void
gst_amd_hip_upload_init (GstAmdHipUpload * s)
{
}

static void
gst_amd_hip_download_class_init (GstAmdHipDownloadClass * klass)
{
  klass->upload = FALSE;
  gst_element_class_set_static_metadata (GST_ELEMENT_CLASS (klass), "HIP downloader (MI355X)", "Filter/Video",
      "Copies video frames from MI355X HBM (memory:AMDHIPMemory) into system memory", "gstreamer_amd");
}

static void
gst_amd_hip_download_init (GstAmdHipDownload * s)
{
}

GType gst_amd_hip_upload_element_get_type (void) { return gst_amd_hip_upload_get_type (); }
GType gst_amd_hip_download_element_get_type (void) { return gst_amd_hip_download_get_type (); }
