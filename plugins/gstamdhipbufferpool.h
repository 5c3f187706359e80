/* gstamdhipbufferpool.h - GstBufferPool whose buffers hold one AMDHIPMemory (HBM) each, so a streaming element recycles
 * its output frames instead of calling hipMalloc / hipFree per buffer.
 *
 * Boundary contract ("Memory / ownership"): a GstBufferPool subclass with the vfuncs of
 * subprojects/gstreamer/gst/gstbufferpool.h:139-269 (get_options, set_config, alloc_buffer; start / stop / acquire /
 * release / reset come from the base class), negotiated through the ALLOCATION query like any other pool.  Buffers
 * of video caps carry a GstVideoMeta with the pitches / offsets of the negotiated GstVideoInfo. */
#ifndef GST_AMD_HIP_BUFFER_POOL_H
#define GST_AMD_HIP_BUFFER_POOL_H

#include <gst/gst.h>
#include <gst/video/video.h>

G_BEGIN_DECLS

GType gst_amd_hip_buffer_pool_get_type (void);
#define GST_TYPE_AMD_HIP_BUFFER_POOL (gst_amd_hip_buffer_pool_get_type ())

GstBufferPool *gst_amd_hip_buffer_pool_new (void);
/* the same pool class handing out PAGE-LOCKED HOST memory (hipHostMalloc) as ordinary system memory: what an element at the edge of the
 * HBM part of a pipeline offers its system-memory neighbours, so that the copies to and from the device are DMA transfers at the
 * link's rate instead of staged copies of pageable memory */
GstBufferPool *gst_amd_hip_buffer_pool_new_pinned_host (void);
/* configured and activated pool for frames of `caps` (video/x-raw; size from the caps), or NULL */
GstBufferPool *gst_amd_hip_buffer_pool_new_for_caps (GstCaps * caps, guint min_buffers);

G_END_DECLS
#endif
