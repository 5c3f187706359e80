/* gstamdhipmemory.h - GstAllocator for frames that stay in MI355X HBM across chained elements.
 *
 * Boundary contract ("Memory / ownership"): a GstAllocator subclass with its own mem_type and
 * caps feature; a CPU gst_buffer_map() stages through host memory (download on READ, upload on unmap after
 * WRITE), a device map (GST_MAP_AMDHIP) hands out the HBM pointer with no copy.  Precedent for the flag
 * convention in the reference: GST_MAP_HIP = GST_MAP_FLAG_LAST << 1
 * (subprojects/gst-plugins-bad/gst-libs/gst/hip/gsthipmemory.h:48) - names here are our own, and none of that
 * plugin's code is used. */
#ifndef GST_AMD_HIP_MEMORY_H
#define GST_AMD_HIP_MEMORY_H

#include <gst/gst.h>
#include <gst/video/video.h>

G_BEGIN_DECLS

#define GST_AMD_HIP_MEMORY_TYPE "AMDHIPMemory"
#define GST_CAPS_FEATURE_MEMORY_AMD_HIP "memory:AMDHIPMemory"
#define GST_MAP_AMDHIP (GST_MAP_FLAG_LAST << 1)

typedef struct _GstAmdHipMemory {
  GstMemory mem;
  gpointer device_ptr;      /* HBM */
  gpointer host_staging;    /* lazily allocated mirror for CPU maps */
  gboolean host_valid;      /* staging holds the current contents */
  gboolean device_dirty_from_host; /* a CPU WRITE map is outstanding / needs upload at unmap */
  GMutex lock;
} GstAmdHipMemory;

GType gst_amd_hip_allocator_get_type (void);
GstAllocator *gst_amd_hip_allocator_get (void);          /* singleton, transfer none */
gboolean gst_is_amd_hip_memory (GstMemory * mem);
/* buffer with one AMDHIPMemory of info->size bytes + a GstVideoMeta carrying pitches/offsets */
GstBuffer *gst_amd_hip_buffer_new_video (const GstVideoInfo * info);
GstBuffer *gst_amd_hip_buffer_new (gsize size);

G_END_DECLS
#endif
