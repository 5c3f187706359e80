/* gstamdhipmemory.h - GstAllocator for frames that stay in MI355X HBM across chained elements.
 *
 * Boundary contract ("Memory / ownership"): a GstAllocator subclass with its own mem_type and
 * caps feature; a CPU gst_buffer_map() stages through (page-locked) host memory (download on READ, upload on unmap
 * after WRITE), a device map (GST_MAP_AMDHIP) hands out the HBM pointer with no copy.  Precedent for the flag
 * convention in the reference: GST_MAP_HIP = GST_MAP_FLAG_LAST << 1
 * (subprojects/gst-plugins-bad/gst-libs/gst/hip/gsthipmemory.h:48) - names here are our own, and none of that
 * plugin's code is used.
 *
 * Ordering without host synchronisation (SURVEY 8b Threading: one HIP stream per element instance): an element records ONE event
 * per launch (a "ticket", reference counted, events recycled through a free list) on its stream and hangs it on every memory
 * the launch touched - as `written` on the frames it produced, as `read` on the frames it consumed.  An element that READS a
 * memory on another stream first makes its stream wait on the `written` ticket (gst_amd_hip_memory_wait_written); the next
 * writer of a recycled pool buffer waits on both tickets (gst_amd_hip_memory_wait_idle).  A CPU map waits on the host. */
#ifndef GST_AMD_HIP_MEMORY_H
#define GST_AMD_HIP_MEMORY_H

#include <gst/gst.h>
#include <gst/video/video.h>

G_BEGIN_DECLS

#define GST_AMD_HIP_MEMORY_TYPE "AMDHIPMemory"
#define GST_CAPS_FEATURE_MEMORY_AMD_HIP "memory:AMDHIPMemory"
#define GST_MAP_AMDHIP (GST_MAP_FLAG_LAST << 1)
#define GST_AMD_HIP_MAX_READERS 4

typedef struct _GstAmdHipMemory {
  GstMemory mem;
  gpointer device_ptr;      /* HBM */
  gint device_id;           /* device the allocation lives on */
  gpointer host_staging;    /* lazily allocated page-locked mirror for CPU maps */
  gboolean host_valid;      /* staging holds the current contents */
  gboolean device_dirty_from_host; /* a CPU WRITE map is outstanding / needs upload at unmap */
  struct _GstAmdHipTicket *written;         /* the launch that last wrote this memory, NULL: nothing pending */
  struct _GstAmdHipTicket *read[GST_AMD_HIP_MAX_READERS];    /* launches still reading it (several consumers of one buffer) */
  GMutex lock;
} GstAmdHipMemory;

/* one recorded event shared by everything a launch touched */
typedef struct _GstAmdHipTicket {
  gpointer event;           /* gstamd event */
  gint refcount;
  gint stream_slot;         /* index of the recording stream in the module's stream table */
  guint64 seq;              /* position among the tickets recorded on that stream: work on a stream completes in order, so a
                             * ticket is known to be done as soon as any later one of its stream has been seen done - most
                             * completion checks then cost no HIP call (hipEventQuery is ~1.5 us on this stack) */
  gpointer waited_stream;   /* last stream that was made to wait on this ticket (a second wait would be redundant) */
  /* a DEFERRED ticket stands for work its owner has collected but not launched yet (an element batching the buffers of several
   * transform calls into one kernel launch): whoever needs the work - a stream about to read the memory, a CPU map, the free of
   * the allocation - first calls launch (owner), which enqueues the work and resolves the ticket; nothing else changes for them */
  /* a LAZY ticket knows its stream and its place in that stream's order but has no event yet: work enqueued later on the same stream
   * needs none, and an element with a single stream then never records one; the first stream or host that does have to wait records
   * it at that point (it then covers whatever the stream was given in between, which only makes the wait longer) */
  gpointer stream;
  volatile gint lazy;
  volatile gint deferred;
  void (*launch) (gpointer owner);
  gpointer owner;           /* the ticket holds a reference (owner_unref) */
  GDestroyNotify owner_unref;
} GstAmdHipTicket;
gboolean gst_amd_hip_ticket_is_done (GstAmdHipTicket * t);
GstAmdHipTicket *gst_amd_hip_ticket_new (gpointer stream);    /* records on `stream`; NULL when no event could be had (the stream was synchronised instead) */
/* a ticket for work that `launch (owner)` will enqueue on demand; the owner calls _resolve once it has (stream NULL: the work failed,
 * the ticket then counts as done) */
GstAmdHipTicket *gst_amd_hip_ticket_new_lazy (gpointer stream);
/* before a stream that tickets were made for is destroyed: waits for it and marks everything it was given as done */
void gst_amd_hip_stream_retire (gpointer stream);
GstAmdHipTicket *gst_amd_hip_ticket_new_deferred (void (*launch) (gpointer owner), gpointer owner, GDestroyNotify owner_unref);
void gst_amd_hip_ticket_resolve (GstAmdHipTicket * t, gpointer stream, gboolean lazy);
GstAmdHipTicket *gst_amd_hip_ticket_ref (GstAmdHipTicket * t);
void gst_amd_hip_ticket_unref (GstAmdHipTicket * t);

GType gst_amd_hip_allocator_get_type (void);
GstAllocator *gst_amd_hip_allocator_get (void);          /* singleton, transfer none */
gboolean gst_is_amd_hip_memory (GstMemory * mem);
/* buffer with one AMDHIPMemory of info->size bytes + a GstVideoMeta carrying pitches/offsets */
GstBuffer *gst_amd_hip_buffer_new_video (const GstVideoInfo * info);
GstBuffer *gst_amd_hip_buffer_new (gsize size);

/* stream ordering, see the header comment; `stream` is the gstamd stream the caller enqueues its work on */
void gst_amd_hip_memory_wait_written (GstMemory * mem, gpointer stream);    /* before reading on `stream` */
void gst_amd_hip_memory_wait_idle (GstMemory * mem, gpointer stream);       /* before overwriting on `stream` */
void gst_amd_hip_memory_set_written (GstMemory * mem, GstAmdHipTicket * t); /* after the writer's ticket was recorded (t may be NULL) */
void gst_amd_hip_memory_set_read (GstMemory * mem, GstAmdHipTicket * t);    /* after the reader's ticket was recorded */
/* convenience: a ticket of its own for one memory */
void gst_amd_hip_memory_mark_written (GstMemory * mem, gpointer stream);
void gst_amd_hip_memory_mark_read (GstMemory * mem, gpointer stream);

/* the host waits until the device work pending on the memory is done (no copy; a CPU map does this and then downloads) */
void gst_amd_hip_memory_host_wait (GstMemory * mem);

/* device every thread of this process should select before touching HIP for an element: -1 = leave the current one */
void gst_amd_hip_select_device (gint device_id);

/* Host buffers a queued host -> HBM transfer still reads.  From page-locked memory (our own pinned pools, or anybody's) the copy call
 * returns before a byte has moved, so the buffer must not go back to its pool - and be overwritten upstream - when transform () returns:
 * it stays referenced until an event recorded behind the copy has fired. */
typedef struct _GstAmdHipPendingReads GstAmdHipPendingReads;
GstAmdHipPendingReads *gst_amd_hip_pending_reads_new (void);
/* right after queueing the copy from `host` (the mapped bytes of `buf`).  Pageable memory is not held (the copy call has staged it).  A
 * buffer of a bounded pool (max-buffers <= 3: v4l2-style fixed pools) is waited for here instead of held: upstream would block in
 * acquire_buffer while this element waits for the next buffer before letting go of the last. */
void gst_amd_hip_pending_reads_hold (GstAmdHipPendingReads * p, GstBuffer * buf, gconstpointer host, gpointer stream);
void gst_amd_hip_pending_reads_retire (GstAmdHipPendingReads * p);      /* drop the references of the transfers that are over; never waits */
void gst_amd_hip_pending_reads_drain (GstAmdHipPendingReads * p);        /* wait for every transfer, drop the references (stop, flush) */
void gst_amd_hip_pending_reads_free (GstAmdHipPendingReads * p);

G_END_DECLS
/* converter-config (a GstStructure of GstVideoConverter.* / GstVideoResampler.* options) -> the C ABI's config: library defaults for whatever
 * the structure does not name (gstamdvideoconvertscale.c); shared with the compositor's per-pad converters */
struct GstAmdVideoConverterConfig;
void gst_amd_converter_config_from_structure (const GstStructure * st, struct GstAmdVideoConverterConfig * cfg);
void gst_amd_converter_config_register_types (void);
/* GstVideoInfo -> the C ABI's info (format, size, pitches, colorimetry, chroma site; FALSE: a format the library does not know), and the format list of the
 * elements' pad templates ("{ NV12, ... }") - gstamdvideoconvertscale.c, shared with the test source */
struct GstAmdVideoInfo;
gboolean gst_amd_video_info_fill (const GstVideoInfo * vi, struct GstAmdVideoInfo * ai);
const gchar *gst_amd_video_formats_string (void);

#endif
