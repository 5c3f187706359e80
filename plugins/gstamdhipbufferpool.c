/* gstamdhipbufferpool.c - see gstamdhipbufferpool.h */
#include "gstamdhipbufferpool.h"

#include "gstamdhipmemory.h"
#include "../include/gstamd_video.h"

GST_DEBUG_CATEGORY_STATIC (amd_pool_debug);
#define GST_CAT_DEFAULT amd_pool_debug

typedef struct {
  GstBufferPool parent;
  GstVideoInfo info;
  gboolean is_video;
  gboolean add_videometa;
  gboolean pinned_host;         /* buffers are page-locked host memory wrapped as system memory, not HBM */
  gsize size;
} GstAmdHipBufferPool;

typedef struct { GstBufferPoolClass parent_class; } GstAmdHipBufferPoolClass;

G_DEFINE_TYPE (GstAmdHipBufferPool, gst_amd_hip_buffer_pool, GST_TYPE_BUFFER_POOL);

static const gchar **
amd_pool_get_options (GstBufferPool * pool)
{
  static const gchar *options[] = { GST_BUFFER_POOL_OPTION_VIDEO_META, NULL };
  return options;
}

static gboolean
amd_pool_set_config (GstBufferPool * bpool, GstStructure * config)
{
  GstAmdHipBufferPool *pool = (GstAmdHipBufferPool *) bpool;
  GstCaps *caps = NULL;
  guint size = 0, min = 0, max = 0;

  if (!gst_buffer_pool_config_get_params (config, &caps, &size, &min, &max) || !caps) {
    GST_WARNING_OBJECT (pool, "invalid pool config");
    return FALSE;
  }
  pool->is_video = gst_video_info_from_caps (&pool->info, caps);
  if (pool->is_video && size < GST_VIDEO_INFO_SIZE (&pool->info))
    size = GST_VIDEO_INFO_SIZE (&pool->info);
  pool->size = size;
  /* device frames always carry their layout: downstream maps the device pointer, not a GstVideoFrame; host frames carry it when the
   * pool's user asked for it */
  pool->add_videometa = pool->is_video && (!pool->pinned_host || gst_buffer_pool_config_has_option (config, GST_BUFFER_POOL_OPTION_VIDEO_META));
  gst_buffer_pool_config_set_params (config, caps, size, min, max);
  return GST_BUFFER_POOL_CLASS (gst_amd_hip_buffer_pool_parent_class)->set_config (bpool, config);
}

static GstFlowReturn
amd_pool_alloc_buffer (GstBufferPool * bpool, GstBuffer ** buffer, GstBufferPoolAcquireParams * params)
{
  GstAmdHipBufferPool *pool = (GstAmdHipBufferPool *) bpool;
  GstBuffer *buf;

  if (pool->pinned_host) {
    gpointer host = gstamd_host_alloc (pool->size);
    if (!host) {
      GST_ERROR_OBJECT (pool, "page-locked host allocation of %" G_GSIZE_FORMAT " bytes failed", pool->size);
      return GST_FLOW_ERROR;
    }
    buf = gst_buffer_new ();
    gst_buffer_append_memory (buf, gst_memory_new_wrapped (0, host, pool->size, 0, pool->size, host, (GDestroyNotify) gstamd_host_free));
    if (pool->is_video && pool->add_videometa)
      gst_buffer_add_video_meta_full (buf, GST_VIDEO_FRAME_FLAG_NONE, GST_VIDEO_INFO_FORMAT (&pool->info), GST_VIDEO_INFO_WIDTH (&pool->info),
          GST_VIDEO_INFO_HEIGHT (&pool->info), GST_VIDEO_INFO_N_PLANES (&pool->info), pool->info.offset, pool->info.stride);
    *buffer = buf;
    return GST_FLOW_OK;
  }
  buf = pool->is_video ? gst_amd_hip_buffer_new_video (&pool->info) : gst_amd_hip_buffer_new (pool->size);

  if (!buf) {
    GST_ERROR_OBJECT (pool, "HIP allocation of %" G_GSIZE_FORMAT " bytes failed", pool->size);
    return GST_FLOW_ERROR;
  }
  *buffer = buf;
  return GST_FLOW_OK;
}

static void
gst_amd_hip_buffer_pool_class_init (GstAmdHipBufferPoolClass * klass)
{
  GstBufferPoolClass *pc = GST_BUFFER_POOL_CLASS (klass);

  GST_DEBUG_CATEGORY_INIT (amd_pool_debug, "amdhipbufferpool", 0, "MI355X HBM buffer pool");
  pc->get_options = amd_pool_get_options;
  pc->set_config = amd_pool_set_config;
  pc->alloc_buffer = amd_pool_alloc_buffer;
}

static void
gst_amd_hip_buffer_pool_init (GstAmdHipBufferPool * pool)
{
  pool->is_video = FALSE;
  pool->size = 0;
}

GstBufferPool *
gst_amd_hip_buffer_pool_new (void)
{
  GstBufferPool *pool = g_object_new (GST_TYPE_AMD_HIP_BUFFER_POOL, NULL);
  gst_object_ref_sink (pool);
  return pool;
}

GstBufferPool *
gst_amd_hip_buffer_pool_new_pinned_host (void)
{
  GstBufferPool *pool = gst_amd_hip_buffer_pool_new ();
  ((GstAmdHipBufferPool *) pool)->pinned_host = TRUE;
  return pool;
}

GstBufferPool *
gst_amd_hip_buffer_pool_new_for_caps (GstCaps * caps, guint min_buffers)
{
  GstBufferPool *pool = gst_amd_hip_buffer_pool_new ();
  GstStructure *config = gst_buffer_pool_get_config (pool);
  GstVideoInfo info;
  guint size = gst_video_info_from_caps (&info, caps) ? (guint) GST_VIDEO_INFO_SIZE (&info) : 0;

  gst_buffer_pool_config_set_params (config, caps, size, min_buffers, 0);
  gst_buffer_pool_config_add_option (config, GST_BUFFER_POOL_OPTION_VIDEO_META);
  if (!size || !gst_buffer_pool_set_config (pool, config) || !gst_buffer_pool_set_active (pool, TRUE)) {
    gst_object_unref (pool);
    return NULL;
  }
  return pool;
}
