/* gstamdhipmemory.c - see gstamdhipmemory.h.  Device memory comes from the C ABI's helpers
 * (gstamd_device_alloc & co., include/gstamd_video.h), so this file has no HIP dependency of its own. */
#include "gstamdhipmemory.h"

#include <string.h>

#include "../include/gstamd_video.h"

typedef struct { GstAllocator parent; } GstAmdHipAllocator;
typedef struct { GstAllocatorClass parent_class; } GstAmdHipAllocatorClass;

G_DEFINE_TYPE (GstAmdHipAllocator, gst_amd_hip_allocator, GST_TYPE_ALLOCATOR);

static GstMemory *
amd_hip_alloc (GstAllocator * allocator, gsize size, GstAllocationParams * params)
{
  GstAmdHipMemory *m = g_new0 (GstAmdHipMemory, 1);
  gsize maxsize = size + (params ? params->prefix + params->padding : 0);

  m->device_ptr = gstamd_device_alloc (maxsize);
  if (!m->device_ptr) {
    GST_ERROR ("HIP allocation of %" G_GSIZE_FORMAT " bytes failed: %s", maxsize, gstamd_last_error ());
    g_free (m);
    return NULL;
  }
  g_mutex_init (&m->lock);
  gst_memory_init (GST_MEMORY_CAST (m), 0, allocator, NULL, maxsize, 255, params ? params->prefix : 0, size);
  return GST_MEMORY_CAST (m);
}

static void
amd_hip_free (GstAllocator * allocator, GstMemory * mem)
{
  GstAmdHipMemory *m = (GstAmdHipMemory *) mem;

  gstamd_device_free (m->device_ptr);
  g_free (m->host_staging);
  g_mutex_clear (&m->lock);
  g_free (m);
}

static gpointer
amd_hip_map_full (GstMemory * mem, GstMapInfo * info, gsize maxsize)
{
  GstAmdHipMemory *m = (GstAmdHipMemory *) mem;
  gpointer ret;

  g_mutex_lock (&m->lock);
  if (info->flags & GST_MAP_AMDHIP) {
    /* device map: pending host writes were uploaded at their unmap; the host mirror goes stale on write */
    if (info->flags & GST_MAP_WRITE)
      m->host_valid = FALSE;
    ret = m->device_ptr;
  } else {
    if (!m->host_staging)
      m->host_staging = g_malloc (mem->maxsize);
    if ((info->flags & GST_MAP_READ) && !m->host_valid) {
      /* kernels run asynchronously on the default stream: download synchronises before the CPU looks */
      if (gstamd_device_download (m->host_staging, m->device_ptr, mem->maxsize, NULL) != GSTAMD_OK) {
        g_mutex_unlock (&m->lock);
        return NULL;
      }
      m->host_valid = TRUE;
    }
    if (info->flags & GST_MAP_WRITE)
      m->device_dirty_from_host = TRUE;
    ret = m->host_staging;
  }
  g_mutex_unlock (&m->lock);
  return ret;
}

static void
amd_hip_unmap_full (GstMemory * mem, GstMapInfo * info)
{
  GstAmdHipMemory *m = (GstAmdHipMemory *) mem;

  g_mutex_lock (&m->lock);
  if (!(info->flags & GST_MAP_AMDHIP) && (info->flags & GST_MAP_WRITE) && m->device_dirty_from_host) {
    gstamd_device_upload (m->device_ptr, m->host_staging, mem->maxsize, NULL);
    gstamd_stream_synchronize (NULL);
    m->device_dirty_from_host = FALSE;
    m->host_valid = TRUE;
  }
  g_mutex_unlock (&m->lock);
}

static void
gst_amd_hip_allocator_class_init (GstAmdHipAllocatorClass * klass)
{
  GstAllocatorClass *ac = GST_ALLOCATOR_CLASS (klass);

  ac->alloc = amd_hip_alloc;
  ac->free = amd_hip_free;
}

static void
gst_amd_hip_allocator_init (GstAmdHipAllocator * self)
{
  GstAllocator *a = GST_ALLOCATOR_CAST (self);

  a->mem_type = GST_AMD_HIP_MEMORY_TYPE;
  a->mem_map_full = amd_hip_map_full;
  a->mem_unmap_full = amd_hip_unmap_full;
  /* mem_copy / mem_share / mem_is_span: GstAllocator's defaults (copy goes through map) */
  GST_OBJECT_FLAG_SET (self, GST_ALLOCATOR_FLAG_CUSTOM_ALLOC);
}

GstAllocator *
gst_amd_hip_allocator_get (void)
{
  static gsize once = 0;
  static GstAllocator *allocator = NULL;

  if (g_once_init_enter (&once)) {
    allocator = g_object_new (gst_amd_hip_allocator_get_type (), NULL);
    gst_object_ref_sink (allocator);
    gst_allocator_register (GST_AMD_HIP_MEMORY_TYPE, gst_object_ref (allocator));
    g_once_init_leave (&once, 1);
  }
  return allocator;
}

gboolean
gst_is_amd_hip_memory (GstMemory * mem)
{
  return mem != NULL && mem->allocator != NULL && g_strcmp0 (mem->allocator->mem_type, GST_AMD_HIP_MEMORY_TYPE) == 0;
}

GstBuffer *
gst_amd_hip_buffer_new (gsize size)
{
  GstMemory *mem = amd_hip_alloc (gst_amd_hip_allocator_get (), size, NULL);
  GstBuffer *buf;

  if (!mem)
    return NULL;
  buf = gst_buffer_new ();
  gst_buffer_append_memory (buf, mem);
  return buf;
}

GstBuffer *
gst_amd_hip_buffer_new_video (const GstVideoInfo * info)
{
  GstBuffer *buf = gst_amd_hip_buffer_new (GST_VIDEO_INFO_SIZE (info));

  if (buf)
    gst_buffer_add_video_meta_full (buf, GST_VIDEO_FRAME_FLAG_NONE, GST_VIDEO_INFO_FORMAT (info),
        GST_VIDEO_INFO_WIDTH (info), GST_VIDEO_INFO_HEIGHT (info), GST_VIDEO_INFO_N_PLANES (info),
        (gsize *) info->offset, (gint *) info->stride);
  return buf;
}
