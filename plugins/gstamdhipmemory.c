/* gstamdhipmemory.c - see gstamdhipmemory.h.  Device memory, streams and events come from the C ABI's helpers
 * (include/gstamd_video.h), so this file has no HIP dependency of its own. */
#include "gstamdhipmemory.h"

#include <string.h>

#include "../include/gstamd_video.h"

typedef struct { GstAllocator parent; } GstAmdHipAllocator;
typedef struct { GstAllocatorClass parent_class; } GstAmdHipAllocatorClass;

G_DEFINE_TYPE (GstAmdHipAllocator, gst_amd_hip_allocator, GST_TYPE_ALLOCATOR);

void
gst_amd_hip_select_device (gint device_id)
{
  if (device_id >= 0 && gstamd_get_device () != device_id)
    gstamd_set_device (device_id);
}

/* the allocator's own transfer stream, one per device: CPU maps of HBM buffers copy on it and wait for IT alone - the NULL stream
 * would order the copy against (and its synchronisation wait for) every blocking stream of the process */
#define AMD_XFER_DEVICES 16
static GMutex xfer_lock;
static gpointer xfer_streams[AMD_XFER_DEVICES];

static gpointer
amd_hip_xfer_stream (gint device_id)
{
  gint d = device_id >= 0 ? device_id : gstamd_get_device ();
  gpointer st;
  if (d < 0 || d >= AMD_XFER_DEVICES)
    return NULL;                /* the NULL stream still works, only less politely */
  g_mutex_lock (&xfer_lock);
  if (!xfer_streams[d])
    xfer_streams[d] = gstamd_stream_new ();
  st = xfer_streams[d];
  g_mutex_unlock (&xfer_lock);
  return st;
}

/* ---- tickets ----------------------------------------------------------------------------------------------------------- */
static GMutex ticket_lock;
static GSList *free_events = NULL;      /* recycled gstamd events (creating one costs more than recording it) */

#define AMD_STREAM_SLOTS 64
static struct { gpointer stream; guint64 next_seq, done_seq; } stream_table[AMD_STREAM_SLOTS];   /* under ticket_lock */

static gint
stream_slot_locked (gpointer stream)
{
  gint i, free_slot = -1;
  for (i = 0; i < AMD_STREAM_SLOTS; i++) {
    if (stream_table[i].next_seq && stream_table[i].stream == stream)
      return i;
    if (!stream_table[i].next_seq && free_slot < 0)
      free_slot = i;
  }
  if (free_slot >= 0) {
    stream_table[free_slot].stream = stream;
    stream_table[free_slot].next_seq = 1;
    stream_table[free_slot].done_seq = 0;
  }
  return free_slot;             /* -1: table full, the ticket falls back to plain event queries */
}

/* record an event for `t` on `stream` (the free list first: creating an event costs more than recording it) */
static gboolean
ticket_record (GstAmdHipTicket * t, gpointer stream)
{
  gpointer ev = NULL;

  g_mutex_lock (&ticket_lock);
  if (free_events) {
    ev = free_events->data;
    free_events = g_slist_delete_link (free_events, free_events);
  }
  g_mutex_unlock (&ticket_lock);
  if (!ev)
    ev = gstamd_event_new ();
  if (!ev || gstamd_event_record (ev, stream) != GSTAMD_OK) {
    gstamd_event_free (ev);
    gstamd_stream_synchronize (stream);       /* no event: fall back to a host wait, nothing is pending afterwards */
    return FALSE;
  }
  t->event = ev;
  t->stream = stream;
  g_mutex_lock (&ticket_lock);
  t->stream_slot = stream_slot_locked (stream);
  if (t->stream_slot >= 0)
    t->seq = stream_table[t->stream_slot].next_seq++;
  g_mutex_unlock (&ticket_lock);
  return TRUE;
}

/* a place in the stream's order, no event; FALSE when the stream table is full (the caller records an event instead) */
static gboolean
ticket_make_lazy (GstAmdHipTicket * t, gpointer stream)
{
  g_mutex_lock (&ticket_lock);
  t->stream_slot = stream_slot_locked (stream);
  if (t->stream_slot >= 0)
    t->seq = stream_table[t->stream_slot].next_seq++;
  g_mutex_unlock (&ticket_lock);
  if (t->stream_slot < 0)
    return FALSE;
  t->stream = stream;
  g_atomic_int_set (&t->lazy, 1);
  return TRUE;
}

/* the event of a lazy ticket, recorded now unless its stream has been retired or seen past it; the table lock keeps the stream alive */
static void
ticket_unlazy (GstAmdHipTicket * t)
{
  if (!t || !g_atomic_int_get (&t->lazy))
    return;
  g_mutex_lock (&ticket_lock);
  if (g_atomic_int_get (&t->lazy)) {
    if (t->seq > stream_table[t->stream_slot].done_seq) {
      gpointer ev = NULL;
      if (free_events) {
        ev = free_events->data;
        free_events = g_slist_delete_link (free_events, free_events);
      }
      if (!ev)
        ev = gstamd_event_new ();
      if (ev && gstamd_event_record (ev, t->stream) == GSTAMD_OK) {
        t->event = ev;
      } else {
        gstamd_event_free (ev);
        gstamd_stream_synchronize (t->stream);
      }
    }
    g_atomic_int_set (&t->lazy, 0);
  }
  g_mutex_unlock (&ticket_lock);
}

void
gst_amd_hip_stream_retire (gpointer stream)
{
  gint i;

  if (!stream)
    return;
  gstamd_stream_synchronize (stream);
  g_mutex_lock (&ticket_lock);
  for (i = 0; i < AMD_STREAM_SLOTS; i++)
    if (stream_table[i].next_seq && stream_table[i].stream == stream)
      stream_table[i].done_seq = stream_table[i].next_seq - 1;
  g_mutex_unlock (&ticket_lock);
}

GstAmdHipTicket *
gst_amd_hip_ticket_new_lazy (gpointer stream)
{
  GstAmdHipTicket *t = g_new0 (GstAmdHipTicket, 1);

  t->refcount = 1;
  t->stream_slot = -1;
  if (!ticket_make_lazy (t, stream) && !ticket_record (t, stream)) {
    g_free (t);
    return NULL;
  }
  return t;
}

GstAmdHipTicket *
gst_amd_hip_ticket_new (gpointer stream)
{
  GstAmdHipTicket *t = g_new0 (GstAmdHipTicket, 1);

  t->refcount = 1;
  t->stream_slot = -1;
  if (!ticket_record (t, stream)) {
    g_free (t);
    return NULL;
  }
  return t;
}

GstAmdHipTicket *
gst_amd_hip_ticket_new_deferred (void (*launch) (gpointer owner), gpointer owner, GDestroyNotify owner_unref)
{
  GstAmdHipTicket *t = g_new0 (GstAmdHipTicket, 1);

  t->refcount = 1;
  t->stream_slot = -1;
  t->launch = launch;
  t->owner = owner;
  t->owner_unref = owner_unref;
  t->deferred = 1;
  return t;
}

void
gst_amd_hip_ticket_resolve (GstAmdHipTicket * t, gpointer stream, gboolean lazy)
{
  if (!t || !g_atomic_int_get (&t->deferred))
    return;
  if (stream && !(lazy && ticket_make_lazy (t, stream)))
    ticket_record (t, stream);          /* no event (failure or none to be had): the ticket counts as done */
  g_atomic_int_set (&t->deferred, 0);
}

/* before anybody looks at the ticket's event: have deferred work launched (the owner serialises concurrent callers and resolves the
 * ticket before launch returns) */
static void
ticket_settle (GstAmdHipTicket * t)
{
  if (t && g_atomic_int_get (&t->deferred))
    t->launch (t->owner);
}

static void
ticket_host_wait (GstAmdHipTicket * t)
{
  if (!t)
    return;
  ticket_settle (t);
  ticket_unlazy (t);
  if (t->event)
    gstamd_event_synchronize (t->event);
}

gboolean
gst_amd_hip_ticket_is_done (GstAmdHipTicket * t)
{
  gboolean done = FALSE;

  if (!t)
    return TRUE;
  if (g_atomic_int_get (&t->deferred))
    return FALSE;               /* not even launched; a mere completion check does not force the launch */
  if (g_atomic_int_get (&t->lazy)) {
    g_mutex_lock (&ticket_lock);
    done = t->seq <= stream_table[t->stream_slot].done_seq;
    g_mutex_unlock (&ticket_lock);
    return done;                /* no event to ask; a later ticket of the stream (or its retirement) will tell */
  }
  if (!t->event)
    return TRUE;
  g_mutex_lock (&ticket_lock);
  if (t->stream_slot >= 0 && t->seq <= stream_table[t->stream_slot].done_seq)
    done = TRUE;
  g_mutex_unlock (&ticket_lock);
  if (done)
    return TRUE;
  if (gstamd_event_query (t->event) != 1)
    return FALSE;
  g_mutex_lock (&ticket_lock);
  if (t->stream_slot >= 0 && stream_table[t->stream_slot].done_seq < t->seq)
    stream_table[t->stream_slot].done_seq = t->seq;
  g_mutex_unlock (&ticket_lock);
  return TRUE;
}

GstAmdHipTicket *
gst_amd_hip_ticket_ref (GstAmdHipTicket * t)
{
  if (t)
    g_atomic_int_inc (&t->refcount);
  return t;
}

void
gst_amd_hip_ticket_unref (GstAmdHipTicket * t)
{
  if (!t || !g_atomic_int_dec_and_test (&t->refcount))
    return;
  if (t->event) {
    g_mutex_lock (&ticket_lock);
    free_events = g_slist_prepend (free_events, t->event);
    g_mutex_unlock (&ticket_lock);
  }
  if (t->owner && t->owner_unref)
    t->owner_unref (t->owner);
  g_free (t);
}

/* the allocation a memory stands for: a sub-memory made by mem_share has no state of its own - its device pointer, host mirror and tickets are its
 * parent's (whatever touches part of an allocation is ordered against everything that touches the allocation) */
static inline GstAmdHipMemory *
amd_hip_root (GstMemory * mem)
{
  while (mem->parent)
    mem = mem->parent;
  return (GstAmdHipMemory *) mem;
}

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
  m->device_id = gstamd_get_device ();
  g_mutex_init (&m->lock);
  /* shareable (amd_hip_mem_share): gst_buffer_copy_region & co. then make sub-memories of the same HBM allocation instead of copying */
  gst_memory_init (GST_MEMORY_CAST (m), 0, allocator, NULL, maxsize, 255, params ? params->prefix : 0, size);
  return GST_MEMORY_CAST (m);
}

static void
amd_hip_free (GstAllocator * allocator, GstMemory * mem)
{
  GstAmdHipMemory *m = (GstAmdHipMemory *) mem;

  if (mem->parent) {            /* a sub-memory: nothing of its own but the struct (the core drops its reference on the parent) */
    g_mutex_clear (&m->lock);
    g_free (m);
    return;
  }

  /* work that still reads or writes the allocation must finish before it goes back to the driver */
  ticket_host_wait (m->written);
  gst_amd_hip_ticket_unref (m->written);
  {
    guint i;
    for (i = 0; i < GST_AMD_HIP_MAX_READERS; i++) {
      ticket_host_wait (m->read[i]);
      gst_amd_hip_ticket_unref (m->read[i]);
    }
  }
  gstamd_device_free (m->device_ptr);
  gstamd_host_free (m->host_staging);
  g_mutex_clear (&m->lock);
  g_free (m);
}

static void
host_wait (GstAmdHipMemory * m, gboolean also_reads)
{
  if (m->written) {
    ticket_host_wait (m->written);
    gst_amd_hip_ticket_unref (m->written);
    m->written = NULL;
  }
  if (also_reads) {
    guint i;
    for (i = 0; i < GST_AMD_HIP_MAX_READERS; i++)
      if (m->read[i]) {
        ticket_host_wait (m->read[i]);
        gst_amd_hip_ticket_unref (m->read[i]);
        m->read[i] = NULL;
      }
  }
}

void
gst_amd_hip_memory_host_wait (GstMemory * mem)
{
  GstAmdHipMemory *m = amd_hip_root (mem);

  g_mutex_lock (&m->lock);
  host_wait (m, TRUE);
  g_mutex_unlock (&m->lock);
}

static gpointer
amd_hip_map_full (GstMemory * mem, GstMapInfo * info, gsize maxsize)
{
  GstAmdHipMemory *m = amd_hip_root (mem);
  gpointer ret;

  g_mutex_lock (&m->lock);
  if (info->flags & GST_MAP_AMDHIP) {
    /* device map: pending host writes were uploaded at their unmap; the host mirror goes stale on write.  Ordering against
     * other streams is the caller's job (gst_amd_hip_memory_wait_written / _wait_idle) */
    if (info->flags & GST_MAP_WRITE)
      m->host_valid = FALSE;
    ret = m->device_ptr;
  } else {
    gst_amd_hip_select_device (m->device_id);
    if (!m->host_staging)
      m->host_staging = gstamd_host_alloc (mem->maxsize);
    if (!m->host_staging) {
      g_mutex_unlock (&m->lock);
      return NULL;
    }
    /* a READ needs the device's bytes; so does a WRITE-only map of a stale mirror, or the unmap's upload of the whole mirror
     * would clobber the bytes the caller did not write */
    if (!m->host_valid) {
      host_wait (m, FALSE);
      if (gstamd_device_download (m->host_staging, m->device_ptr, mem->maxsize, amd_hip_xfer_stream (m->device_id)) != GSTAMD_OK) {
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
  GstAmdHipMemory *m = amd_hip_root (mem);

  g_mutex_lock (&m->lock);
  if (!(info->flags & GST_MAP_AMDHIP) && (info->flags & GST_MAP_WRITE) && m->device_dirty_from_host) {
    gst_amd_hip_select_device (m->device_id);
    host_wait (m, TRUE);                /* kernels still reading the old contents */
    {
      gpointer xs = amd_hip_xfer_stream (m->device_id);
      gstamd_device_upload (m->device_ptr, m->host_staging, mem->maxsize, xs);
      gstamd_stream_synchronize (xs);     /* this stream only: other pipelines' streams keep running */
    }
    m->device_dirty_from_host = FALSE;
    m->host_valid = TRUE;
  }
  g_mutex_unlock (&m->lock);
}

/* make `stream` wait for ticket *t unless its event has been reached already (then the ticket is dropped) */
static void
stream_wait_ticket (GstAmdHipTicket ** t, gpointer stream)
{
  if (!*t)
    return;
  ticket_settle (*t);
  if ((*t)->stream == stream && !g_atomic_int_get (&(*t)->deferred))
    return;                     /* the same stream: what it is given now runs after the ticket's work anyway - no wait, not even a query */
  if (gst_amd_hip_ticket_is_done (*t)) {
    gst_amd_hip_ticket_unref (*t);
    *t = NULL;
    return;
  }
  ticket_unlazy (*t);
  if (!(*t)->event) {           /* became done (or synchronised) while the event was being made */
    gst_amd_hip_ticket_unref (*t);
    *t = NULL;
  } else if ((*t)->waited_stream != stream) {
    gstamd_stream_wait_event (stream, (*t)->event);
    (*t)->waited_stream = stream;     /* the same stream need not wait on it again (buffers of one list share their ticket) */
  }
}

void
gst_amd_hip_memory_wait_written (GstMemory * mem, gpointer stream)
{
  GstAmdHipMemory *m = amd_hip_root (mem);

  g_mutex_lock (&m->lock);
  stream_wait_ticket (&m->written, stream);
  g_mutex_unlock (&m->lock);
}

void
gst_amd_hip_memory_wait_idle (GstMemory * mem, gpointer stream)
{
  GstAmdHipMemory *m = amd_hip_root (mem);

  g_mutex_lock (&m->lock);
  stream_wait_ticket (&m->written, stream);
  {
    guint i;
    for (i = 0; i < GST_AMD_HIP_MAX_READERS; i++)
      stream_wait_ticket (&m->read[i], stream);
  }
  g_mutex_unlock (&m->lock);
}

void
gst_amd_hip_memory_set_written (GstMemory * mem, GstAmdHipTicket * t)
{
  GstAmdHipMemory *m = amd_hip_root (mem);

  g_mutex_lock (&m->lock);
  gst_amd_hip_ticket_unref (m->written);        /* the writer waited for it (wait_idle) before it wrote */
  m->written = gst_amd_hip_ticket_ref (t);
  m->host_valid = FALSE;
  g_mutex_unlock (&m->lock);
}

void
gst_amd_hip_memory_set_read (GstMemory * mem, GstAmdHipTicket * t)
{
  GstAmdHipMemory *m = amd_hip_root (mem);

  g_mutex_lock (&m->lock);
  /* readers do not wait for each other: the memory remembers up to GST_AMD_HIP_MAX_READERS launches that may still be reading it
   * (finished ones leave their slot); only when every slot holds a running reader does the host wait for the oldest */
  {
    guint i, slot = GST_AMD_HIP_MAX_READERS;
    for (i = 0; t && i < GST_AMD_HIP_MAX_READERS; i++) {
      /* the launch is in already, or an earlier one of the same stream is: the later ticket stands for both */
      if (m->read[i] == t) {
        g_mutex_unlock (&m->lock);
        return;
      }
      if (m->read[i] && t->stream && m->read[i]->stream == t->stream && !g_atomic_int_get (&m->read[i]->deferred) &&
          !g_atomic_int_get (&t->deferred) && m->read[i]->stream_slot >= 0 && m->read[i]->stream_slot == t->stream_slot && m->read[i]->seq < t->seq) {
        gst_amd_hip_ticket_unref (m->read[i]);
        m->read[i] = gst_amd_hip_ticket_ref (t);
        g_mutex_unlock (&m->lock);
        return;
      }
    }
    for (i = 0; i < GST_AMD_HIP_MAX_READERS; i++) {
      if (m->read[i] && gst_amd_hip_ticket_is_done (m->read[i])) {
        gst_amd_hip_ticket_unref (m->read[i]);
        m->read[i] = NULL;
      }
      if (!m->read[i] && slot == GST_AMD_HIP_MAX_READERS)
        slot = i;
    }
    if (slot == GST_AMD_HIP_MAX_READERS) {
      ticket_host_wait (m->read[0]);
      gst_amd_hip_ticket_unref (m->read[0]);
      slot = 0;
    }
    m->read[slot] = gst_amd_hip_ticket_ref (t);
  }
  g_mutex_unlock (&m->lock);
}

void
gst_amd_hip_memory_mark_written (GstMemory * mem, gpointer stream)
{
  GstAmdHipTicket *t = gst_amd_hip_ticket_new (stream);
  gst_amd_hip_memory_set_written (mem, t);
  gst_amd_hip_ticket_unref (t);
}

void
gst_amd_hip_memory_mark_read (GstMemory * mem, gpointer stream)
{
  GstAmdHipTicket *t = gst_amd_hip_ticket_new (stream);
  gst_amd_hip_memory_set_read (mem, t);
  gst_amd_hip_ticket_unref (t);
}

/* GstAllocator::mem_copy in HBM (the default copies through a CPU map: a download and an upload over PCIe for every gst_buffer_copy_deep / tee with a
 * writer behind it).  Precedent: gst-plugins-bad gst-libs/gst/hip/gsthipmemory.cpp hip_mem_copy.  The copy runs on the transfer stream behind the launch
 * that wrote the source (its ticket; deferred work is launched), and the two memories carry the copy's ticket like any reader / writer. */
static GstMemory *
amd_hip_mem_copy (GstMemory * mem, gssize offset, gssize size)
{
  GstAmdHipMemory *m = amd_hip_root (mem), *c;
  GstMemory *copy;
  GstAmdHipTicket *t;
  gpointer xs;

  if (size == -1)
    size = (gssize) mem->size > offset ? (gssize) mem->size - offset : 0;
  gst_amd_hip_select_device (m->device_id);
  copy = amd_hip_alloc (mem->allocator, (gsize) size, NULL);
  if (!copy)
    return NULL;
  c = (GstAmdHipMemory *) copy;
  xs = amd_hip_xfer_stream (m->device_id);
  gst_amd_hip_memory_wait_written (mem, xs);
  if (size > 0 && gstamd_device_copy (c->device_ptr, (const guint8 *) m->device_ptr + mem->offset + offset, (gsize) size, xs) != GSTAMD_OK) {
    GST_ERROR ("HBM copy of %" G_GSSIZE_FORMAT " bytes failed: %s", size, gstamd_last_error ());
    gst_memory_unref (copy);
    return NULL;
  }
  t = gst_amd_hip_ticket_new (xs);
  gst_amd_hip_memory_set_read (mem, t);
  gst_amd_hip_memory_set_written (copy, t);
  gst_amd_hip_ticket_unref (t);
  return copy;
}

/* GstAllocator::mem_share: a window into the same HBM allocation (gst_memory_share: gst_buffer_copy_region, gst_buffer_resize on shared memory ...).
 * The sub-memory keeps a reference on its parent and borrows everything from it (amd_hip_root). */
static GstMemory *
amd_hip_mem_share (GstMemory * mem, gssize offset, gssize size)
{
  GstMemory *parent = mem->parent ? mem->parent : mem;
  GstAmdHipMemory *root = amd_hip_root (mem), *sub;

  if (size == -1)
    size = (gssize) mem->size - offset;
  sub = g_new0 (GstAmdHipMemory, 1);
  sub->device_ptr = root->device_ptr;
  sub->device_id = root->device_id;
  g_mutex_init (&sub->lock);
  gst_memory_init (GST_MEMORY_CAST (sub), GST_MINI_OBJECT_FLAGS (parent) | GST_MINI_OBJECT_FLAG_LOCK_READONLY, mem->allocator, parent, mem->maxsize, mem->align,
      mem->offset + offset, (gsize) size);
  return GST_MEMORY_CAST (sub);
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
  a->mem_copy = amd_hip_mem_copy;
  a->mem_share = amd_hip_mem_share;
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

/* ---- host buffers under a queued transfer ------------------------------------------------------------------------------------------- */
typedef struct
{
  gpointer event;
  GstBuffer *buf;
} AmdPendingRead;

struct _GstAmdHipPendingReads
{
  GMutex lock;
  GQueue items;
};

GstAmdHipPendingReads *
gst_amd_hip_pending_reads_new (void)
{
  GstAmdHipPendingReads *p = g_new0 (GstAmdHipPendingReads, 1);
  g_mutex_init (&p->lock);
  g_queue_init (&p->items);
  return p;
}

/* drop the references of the transfers that are over (all of them when `wait`); events fire in stream order but the element may use
 * several streams, so the whole queue is looked at */
static void
amd_pending_reads_retire (GstAmdHipPendingReads * p, gboolean wait)
{
  GList *l = p->items.head;
  while (l) {
    GList *next = l->next;
    AmdPendingRead *r = l->data;
    if (wait)
      gstamd_event_synchronize (r->event);
    if (wait || gstamd_event_query (r->event) != 0) {
      gstamd_event_free (r->event);
      gst_buffer_unref (r->buf);
      g_free (r);
      g_queue_delete_link (&p->items, l);
    }
    l = next;
  }
}

/* at most this many transfers stay un-waited-for: a source that runs ahead does not pile its buffers up here */
#define AMD_PENDING_READS_MAX 4

static gboolean
amd_buffer_pool_is_tight (GstBuffer * buf)
{
  GstBufferPool *pool = buf->pool;
  GstStructure *config;
  guint size = 0, min = 0, max = 0;
  gboolean tight = FALSE;

  if (!pool)
    return FALSE;
  config = gst_buffer_pool_get_config (pool);
  if (config) {
    if (gst_buffer_pool_config_get_params (config, NULL, &size, &min, &max))
      tight = max != 0 && max <= 3;
    gst_structure_free (config);
  }
  return tight;
}

void
gst_amd_hip_pending_reads_hold (GstAmdHipPendingReads * p, GstBuffer * buf, gconstpointer host, gpointer stream)
{
  AmdPendingRead *r;
  gpointer ev;

  gst_amd_hip_pending_reads_retire (p);
  if (!gstamd_host_is_pinned (host))
    return;                     /* pageable: hipMemcpyAsync returned with the source staged */
  ev = gstamd_event_new ();
  if (!ev || gstamd_event_record (ev, stream) != GSTAMD_OK || amd_buffer_pool_is_tight (buf)) {
    /* no event, or a pool that cannot spare the buffer: the safe answer is to wait for the transfer here */
    if (ev) {
      gstamd_event_synchronize (ev);
      gstamd_event_free (ev);
    } else
      gstamd_stream_synchronize (stream);
    return;
  }
  r = g_new0 (AmdPendingRead, 1);
  r->event = ev;
  r->buf = gst_buffer_ref (buf);
  g_mutex_lock (&p->lock);
  g_queue_push_tail (&p->items, r);
  while (g_queue_get_length (&p->items) > AMD_PENDING_READS_MAX) {
    AmdPendingRead *o = g_queue_pop_head (&p->items);
    gstamd_event_synchronize (o->event);
    gstamd_event_free (o->event);
    gst_buffer_unref (o->buf);
    g_free (o);
  }
  g_mutex_unlock (&p->lock);
}

void
gst_amd_hip_pending_reads_retire (GstAmdHipPendingReads * p)
{
  if (!p)
    return;
  g_mutex_lock (&p->lock);
  amd_pending_reads_retire (p, FALSE);
  g_mutex_unlock (&p->lock);
}

void
gst_amd_hip_pending_reads_drain (GstAmdHipPendingReads * p)
{
  if (!p)
    return;
  g_mutex_lock (&p->lock);
  amd_pending_reads_retire (p, TRUE);
  g_mutex_unlock (&p->lock);
}

void
gst_amd_hip_pending_reads_free (GstAmdHipPendingReads * p)
{
  if (!p)
    return;
  gst_amd_hip_pending_reads_drain (p);
  g_mutex_clear (&p->lock);
  g_free (p);
}
