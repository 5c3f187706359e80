/* live_props.c - properties changed on a RUNNING element (test tool, not part of the plugin).
 *
 *   live_props mix-matrix      audioconvert, F32 stereo -> F32 stereo: a buffer with the default (identity) conversion - the element is
 *                              in passthrough -, then `mix-matrix` is set to a channel swap while the caps stay the same, then two more
 *                              buffers.  The reference re-makes its converter lazily at the top of transform (gstaudioconvert.c:1700
 *                              gst_audio_convert_ensure_converter) and leaves passthrough in set_mix_matrix (:1883-1886); round 2's
 *                              element freed its converter in the setter and answered the next buffer with NOT_NEGOTIATED.
 *   live_props upload-meta     amdhipupload fed an NV12 frame whose GstVideoMeta has padded strides and a gap between the planes (what a
 *                              decoder or an aligned pool hands over): the HBM frame must hold the picture in the default layout.  Round
 *                              2's element copied the bytes flat and sheared every row after the first.
 *   live_props nav-meta        videoconvertscale while scaling: navigation events travelling upstream get input coordinates, size-tagged metas
 *                              (GstVideoCropMeta) are scaled onto the output buffer.
 *   live_props hip-memory      the HIP allocator's mem_copy (device to device) and mem_share (windows into one allocation).
 *   live_props audio-list      audioresample fed a GstBufferList: the run of buffers is resampled in one call and cut where buffer by buffer calls
 *                              cut it - same buffers, bytes, timestamps and offsets as the same stream pushed one buffer at a time.
 * prints "ok" and exits 0 when every output buffer is what the options in force at that buffer say. */
#include <gst/check/gstharness.h>
#include <gst/gst.h>
#include <gst/video/video.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../gstamdhipmemory.h"

static GstBuffer *
make_buffer (int frames, float base)
{
  GstBuffer *b = gst_buffer_new_allocate (NULL, frames * 2 * sizeof (float), NULL);
  GstMapInfo m;
  gst_buffer_map (b, &m, GST_MAP_WRITE);
  float *f = (float *) m.data;
  for (int i = 0; i < frames; i++) {
    f[2 * i] = base + i * 0.001f;
    f[2 * i + 1] = -base - i * 0.002f;
  }
  gst_buffer_unmap (b, &m);
  return b;
}

static int
check (GstBuffer * out, int frames, float base, int swapped)
{
  GstMapInfo m;
  int bad = 0;
  if (!out)
    return 1;
  gst_buffer_map (out, &m, GST_MAP_READ);
  if (m.size != frames * 2 * sizeof (float))
    bad = 1;
  else {
    const float *f = (const float *) m.data;
    for (int i = 0; i < frames && !bad; i++) {
      const float l = base + i * 0.001f, r = -base - i * 0.002f;
      bad = swapped ? (f[2 * i] != r || f[2 * i + 1] != l) : (f[2 * i] != l || f[2 * i + 1] != r);
    }
  }
  gst_buffer_unmap (out, &m);
  gst_buffer_unref (out);
  return bad;
}

static int
upload_meta (void)
{
  const int w = 64, h = 32, ys = 80, cs = 96;
  const gsize coff = (gsize) ys * h + 64, size = coff + (gsize) cs * (h / 2);
  GstHarness *hn = gst_harness_new ("amdhipupload");
  if (!hn) {
    fprintf (stderr, "no amdhipupload element (GST_PLUGIN_PATH?)\n");
    return 1;
  }
  gst_harness_set_caps_str (hn, "video/x-raw,format=NV12,width=64,height=32,framerate=30/1",
      "video/x-raw(memory:AMDHIPMemory),format=NV12,width=64,height=32,framerate=30/1");
  GstBuffer *b = gst_buffer_new_allocate (NULL, size, NULL);
  GstMapInfo m;
  gst_buffer_map (b, &m, GST_MAP_WRITE);
  memset (m.data, 0xee, size);                  /* padding bytes: must not show up in the picture */
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++)
      m.data[y * ys + x] = (guint8) (y * 7 + x);
  for (int y = 0; y < h / 2; y++)
    for (int x = 0; x < w; x++)
      m.data[coff + y * cs + x] = (guint8) (200 - y * 3 + x);
  gst_buffer_unmap (b, &m);
  gsize offset[GST_VIDEO_MAX_PLANES] = { 0, coff, 0, 0 };
  gint stride[GST_VIDEO_MAX_PLANES] = { ys, cs, 0, 0 };
  gst_buffer_add_video_meta_full (b, GST_VIDEO_FRAME_FLAG_NONE, GST_VIDEO_FORMAT_NV12, w, h, 2, offset, stride);
  if (gst_harness_push (hn, b) != GST_FLOW_OK) {
    fprintf (stderr, "push failed\n");
    return 1;
  }
  GstBuffer *out = gst_harness_try_pull (hn);
  int bad = out == NULL;
  if (out) {
    GstVideoInfo oi;
    gst_video_info_set_format (&oi, GST_VIDEO_FORMAT_NV12, w, h);
    if (!gst_buffer_map (out, &m, GST_MAP_READ))        /* a CPU map of the HBM frame: waits for the upload, downloads */
      bad = 1;
    else {
      if (m.size < GST_VIDEO_INFO_SIZE (&oi))
        bad = 1;
      for (int y = 0; y < h && !bad; y++)
        for (int x = 0; x < w && !bad; x++)
          bad = m.data[GST_VIDEO_INFO_PLANE_OFFSET (&oi, 0) + y * GST_VIDEO_INFO_PLANE_STRIDE (&oi, 0) + x] != (guint8) (y * 7 + x);
      for (int y = 0; y < h / 2 && !bad; y++)
        for (int x = 0; x < w && !bad; x++)
          bad = m.data[GST_VIDEO_INFO_PLANE_OFFSET (&oi, 1) + y * GST_VIDEO_INFO_PLANE_STRIDE (&oi, 1) + x] != (guint8) (200 - y * 3 + x);
      gst_buffer_unmap (out, &m);
    }
    gst_buffer_unref (out);
  }
  gst_harness_teardown (hn);
  if (bad) {
    fprintf (stderr, "upload of a frame with padded strides: wrong picture in HBM\n");
    return 1;
  }
  printf ("ok\n");
  return 0;
}

/* videoconvertscale 320x240 -> 160x60: a navigation event sent upstream from the src pad arrives with its coordinates in INPUT pixels
 * (gst_video_convert_scale_src_event, gstvideoconvertscale.c:2008), and a GstVideoCropMeta on the input buffer comes out scaled to the output size
 * (gst_video_convert_scale_transform_meta :773: the meta's own "gst-video-scale" transform) */
static int
nav_meta (void)
{
  GstHarness *hn = gst_harness_new ("videoconvertscale");
  if (!hn) {
    fprintf (stderr, "no videoconvertscale element (GST_PLUGIN_PATH?)\n");
    return 1;
  }
  gst_harness_set_caps_str (hn, "video/x-raw,format=BGRA,width=320,height=240,framerate=30/1", "video/x-raw,format=BGRA,width=160,height=60,framerate=30/1");
  GstBuffer *b = gst_buffer_new_allocate (NULL, 320 * 240 * 4, NULL);
  gst_buffer_memset (b, 0, 0x40, 320 * 240 * 4);
  GstVideoCropMeta *cm = gst_buffer_add_video_crop_meta (b);
  cm->x = 40, cm->y = 80, cm->width = 100, cm->height = 120;
  if (gst_harness_push (hn, b) != GST_FLOW_OK) {
    fprintf (stderr, "push failed\n");
    return 1;
  }
  GstBuffer *out = gst_harness_try_pull (hn);
  int bad = out == NULL;
  if (out) {
    GstVideoCropMeta *oc = gst_buffer_get_video_crop_meta (out);
    if (!oc || oc->x != 20 || oc->y != 20 || oc->width != 50 || oc->height != 30) {
      fprintf (stderr, "crop meta on the output: %s (%u, %u, %u x %u), expected (20, 20, 50 x 30)\n", oc ? "wrong" : "missing", oc ? oc->x : 0, oc ? oc->y : 0,
          oc ? oc->width : 0, oc ? oc->height : 0);
      bad = 1;
    }
    gst_buffer_unref (out);
  }
  /* the navigation event: structure fields pointer_x / pointer_y (what gst_navigation_event_new_mouse_move makes) */
  GstStructure *st = gst_structure_new ("application/x-gst-navigation", "event", G_TYPE_STRING, "mouse-move", "pointer_x", G_TYPE_DOUBLE, 80.0,
      "pointer_y", G_TYPE_DOUBLE, 30.0, NULL);
  while (gst_harness_try_pull_upstream_event (hn));          /* (whatever the negotiation sent upstream) */
  if (!gst_harness_push_upstream_event (hn, gst_event_new_navigation (st))) {
    fprintf (stderr, "navigation event refused\n");
    bad = 1;
  }
  GstEvent *ev;
  int seen = 0;
  while ((ev = gst_harness_try_pull_upstream_event (hn))) {
    if (GST_EVENT_TYPE (ev) == GST_EVENT_NAVIGATION) {
      gdouble x = 0, y = 0;
      const GstStructure *es = gst_event_get_structure (ev);
      seen = 1;
      if (!gst_structure_get_double (es, "pointer_x", &x) || !gst_structure_get_double (es, "pointer_y", &y) || x != 160.0 || y != 120.0) {
        fprintf (stderr, "navigation event upstream at (%g, %g), expected (160, 120)\n", x, y);
        bad = 1;
      }
    }
    gst_event_unref (ev);
  }
  if (!seen) {
    fprintf (stderr, "no navigation event arrived upstream\n");
    bad = 1;
  }
  gst_harness_teardown (hn);
  if (!bad)
    printf ("ok\n");
  return bad;
}

/* GstAllocator::mem_copy / mem_share of the HIP allocator: a deep copy of an HBM buffer is another HBM allocation with the same bytes (made on the
 * device: the source's host mirror is not touched), a shared window shows the parent's bytes at its offset and keeps the parent alive */
static int
hip_memory (void)
{
  const gsize size = 1 << 20;
  int bad = 0;
  GstBuffer *b = gst_amd_hip_buffer_new (size);
  GstMapInfo m;
  if (!b || !gst_buffer_map (b, &m, GST_MAP_WRITE)) {
    fprintf (stderr, "no HBM buffer\n");
    return 1;
  }
  for (gsize i = 0; i < size; i++)
    m.data[i] = (guint8) (i * 7 + (i >> 9));
  gst_buffer_unmap (b, &m);
  GstMemory *src = gst_buffer_peek_memory (b, 0);
  GstBuffer *c = gst_buffer_copy_deep (b);
  GstMemory *cm = c ? gst_buffer_peek_memory (c, 0) : NULL;
  GstMapInfo sd, cd;
  if (!cm || !gst_is_amd_hip_memory (cm)) {
    fprintf (stderr, "the deep copy is not HIP memory\n");
    bad = 1;
  } else {
    if (gst_memory_map (src, &sd, GST_MAP_READ | GST_MAP_AMDHIP) && gst_memory_map (cm, &cd, GST_MAP_READ | GST_MAP_AMDHIP)) {
      if (sd.data == cd.data) {
        fprintf (stderr, "the deep copy shares the source's allocation\n");
        bad = 1;
      }
      gst_memory_unmap (cm, &cd);
      gst_memory_unmap (src, &sd);
    }
    if (!gst_buffer_map (c, &m, GST_MAP_READ))
      bad = 1;
    else {
      for (gsize i = 0; i < size && !bad; i++)
        bad = m.data[i] != (guint8) (i * 7 + (i >> 9));
      if (bad)
        fprintf (stderr, "the deep copy holds other bytes\n");
      gst_buffer_unmap (c, &m);
    }
  }
  /* a shared window, then a copy of part of it */
  GstMemory *win = gst_memory_share (src, 4096, 8192);
  if (!win || !gst_is_amd_hip_memory (win) || win->parent != src) {
    fprintf (stderr, "gst_memory_share did not make a sub-memory of the allocation\n");
    bad = 1;
  } else {
    if (gst_memory_map (win, &m, GST_MAP_READ)) {
      for (gsize i = 0; i < 8192 && !bad; i++)
        bad = m.data[i] != (guint8) ((i + 4096) * 7 + ((i + 4096) >> 9));
      if (bad)
        fprintf (stderr, "the shared window shows other bytes\n");
      gst_memory_unmap (win, &m);
    } else
      bad = 1;
    GstMemory *part = gst_memory_copy (win, 100, 1000);
    if (!part || !gst_memory_map (part, &m, GST_MAP_READ))
      bad = 1;
    else {
      for (gsize i = 0; i < 1000 && !bad; i++)
        bad = m.data[i] != (guint8) ((i + 4196) * 7 + ((i + 4196) >> 9));
      if (bad || m.size != 1000)
        fprintf (stderr, "the copy of a region of the window is wrong\n"), bad = 1;
      gst_memory_unmap (part, &m);
    }
    if (part)
      gst_memory_unref (part);
    gst_buffer_unref (b);          /* the window keeps the allocation */
    b = NULL;
    if (gst_memory_map (win, &m, GST_MAP_READ)) {
      bad |= m.data[5] != (guint8) ((5 + 4096) * 7 + ((5 + 4096) >> 9));
      gst_memory_unmap (win, &m);
    }
    gst_memory_unref (win);
  }
  if (b)
    gst_buffer_unref (b);
  if (c)
    gst_buffer_unref (c);
  if (!bad)
    printf ("ok\n");
  return bad;
}

/* buffer k of the test stream: `frames` frames of F32 stereo at 48 kHz, stamped like a live source would stamp them */
static GstBuffer *
audio_buffer (guint64 first_frame, int frames)
{
  GstBuffer *b = gst_buffer_new_allocate (NULL, frames * 2 * sizeof (float), NULL);
  GstMapInfo m;
  gst_buffer_map (b, &m, GST_MAP_WRITE);
  float *f = (float *) m.data;
  for (int i = 0; i < frames; i++) {
    const guint64 t = first_frame + i;
    f[2 * i] = (float) ((t * 2654435761u) % 20011u) / 10005.5f - 1.0f;
    f[2 * i + 1] = (float) ((t * 40503u + 977u) % 30011u) / 15005.5f - 1.0f;
  }
  gst_buffer_unmap (b, &m);
  GST_BUFFER_PTS (b) = gst_util_uint64_scale_int_round (first_frame, GST_SECOND, 48000);
  GST_BUFFER_DURATION (b) = gst_util_uint64_scale_int_round (first_frame + frames, GST_SECOND, 48000) - GST_BUFFER_PTS (b);
  GST_BUFFER_OFFSET (b) = first_frame;
  GST_BUFFER_OFFSET_END (b) = first_frame + frames;
  return b;
}

static int
audio_list (void)
{
  static const int sizes[] = {1024, 1024, 1024, 1, 1024, 333, 1024, 1024, 2, 1024, 1024, 4096, 1024};
  const int n = (int) G_N_ELEMENTS (sizes);
  const char *in_caps = "audio/x-raw,format=F32LE,rate=48000,channels=2,layout=interleaved,channel-mask=(bitmask)0x3";
  const char *out_caps = "audio/x-raw,format=F32LE,rate=44100,channels=2,layout=interleaved,channel-mask=(bitmask)0x3";
  GstHarness *ha = gst_harness_new ("amdaudioresample"), *hb = gst_harness_new ("amdaudioresample"), *hc = gst_harness_new ("amdaudioresample");
  int bad = 0;
  if (!ha || !hb || !hc) {
    fprintf (stderr, "no audioresample element (GST_PLUGIN_PATH?)\n");
    return 1;
  }
  gst_harness_set_caps_str (ha, in_caps, out_caps);
  gst_harness_set_caps_str (hb, in_caps, out_caps);
  gst_harness_set_caps_str (hc, in_caps, out_caps);
  /* C: one buffer at a time as well (a second element: what differs between A and C is not the list path's) */
  {
    guint64 p2 = 0;
    for (int k = 0; k < n; k++) {
      if (gst_harness_push (hc, audio_buffer (p2, sizes[k])) != GST_FLOW_OK)
        bad |= 32;
      p2 += sizes[k];
    }
  }
  /* A: one buffer at a time.  B: the first buffer alone (it starts the stream), the others as two lists */
  guint64 pos = 0;
  for (int k = 0; k < n; k++) {
    if (gst_harness_push (ha, audio_buffer (pos, sizes[k])) != GST_FLOW_OK)
      bad |= 1;
    pos += sizes[k];
  }
  pos = 0;
  if (gst_harness_push (hb, audio_buffer (pos, sizes[0])) != GST_FLOW_OK)
    bad |= 2;
  pos += sizes[0];
  for (int part = 0; part < 2; part++) {
    const int k0 = part == 0 ? 1 : 7, k1 = part == 0 ? 7 : n;
    GstBufferList *list = gst_buffer_list_new ();
    for (int k = k0; k < k1; k++) {
      gst_buffer_list_add (list, audio_buffer (pos, sizes[k]));
      pos += sizes[k];
    }
    if (gst_pad_push_list (hb->srcpad, list) != GST_FLOW_OK)
      bad |= 4;
  }
  const guint na = gst_harness_buffers_in_queue (ha), nb = gst_harness_buffers_in_queue (hb);
  if (na != nb || na == 0) {
    fprintf (stderr, "audio-list: %u buffers one by one, %u through lists\n", na, nb);
    bad |= 8;
  }
  for (guint k = 0; k < na && k < nb; k++) {
    GstBuffer *a = gst_harness_pull (ha), *b = gst_harness_pull (hb), *c = gst_harness_try_pull (hc);
    GstMapInfo ma, mb;
    gst_buffer_map (a, &ma, GST_MAP_READ);
    gst_buffer_map (b, &mb, GST_MAP_READ);
    if (c) {
      GstMapInfo mc;
      gst_buffer_map (c, &mc, GST_MAP_READ);
      if (ma.size != mc.size || memcmp (ma.data, mc.data, ma.size) != 0) {
        fprintf (stderr, "audio-list: output %u of two per-buffer runs differs\n", k);
        bad |= 64;
      }
      gst_buffer_unmap (c, &mc);
      gst_buffer_unref (c);
    }
    if (ma.size != mb.size || memcmp (ma.data, mb.data, ma.size) != 0 || GST_BUFFER_PTS (a) != GST_BUFFER_PTS (b) || GST_BUFFER_DURATION (a) != GST_BUFFER_DURATION (b) ||
        GST_BUFFER_OFFSET (a) != GST_BUFFER_OFFSET (b) || GST_BUFFER_OFFSET_END (a) != GST_BUFFER_OFFSET_END (b)) {
      fprintf (stderr, "audio-list: output %u differs (sizes %zu / %zu, pts %" GST_TIME_FORMAT " / %" GST_TIME_FORMAT ")\n", k, ma.size, mb.size,
          GST_TIME_ARGS (GST_BUFFER_PTS (a)), GST_TIME_ARGS (GST_BUFFER_PTS (b)));
      bad |= 16;
    }
    gst_buffer_unmap (a, &ma);
    gst_buffer_unmap (b, &mb);
    gst_buffer_unref (a);
    gst_buffer_unref (b);
  }
  gst_harness_teardown (ha);
  gst_harness_teardown (hb);
  gst_harness_teardown (hc);
  if (!bad)
    printf ("ok (%u buffers)\n", na);
  return bad;
}

/* live_props audio-list-bench [buffers per list]: wall time per 1024-frame buffer through amdaudioresample (system memory in and out: upload, kernel,
 * download, synchronise), one buffer at a time against lists */
static int
audio_list_bench (int per_list)
{
  const char *in_caps = "audio/x-raw,format=F32LE,rate=48000,channels=2,layout=interleaved,channel-mask=(bitmask)0x3";
  const char *out_caps = "audio/x-raw,format=F32LE,rate=44100,channels=2,layout=interleaved,channel-mask=(bitmask)0x3";
  const int total = 2048;
  for (int mode = 0; mode < 2; mode++) {
    GstHarness *h = gst_harness_new ("amdaudioresample");
    if (!h)
      return 1;
    gst_harness_set_caps_str (h, in_caps, out_caps);
    guint64 pos = 0;
    for (int k = 0; k < 64; k++, pos += 1024)             /* warm up: stream start, staging buffers, clocks */
      gst_harness_push (h, audio_buffer (pos, 1024));
    while (gst_harness_buffers_in_queue (h))
      gst_buffer_unref (gst_harness_pull (h));
    /* the input made ahead of the clock */
    GPtrArray *items = g_ptr_array_new ();
    for (int k = 0; k < total; k += mode ? per_list : 1) {
      if (!mode) {
        g_ptr_array_add (items, audio_buffer (pos, 1024));
        pos += 1024;
      } else {
        GstBufferList *l = gst_buffer_list_new ();
        for (int j = 0; j < per_list; j++, pos += 1024)
          gst_buffer_list_add (l, audio_buffer (pos, 1024));
        g_ptr_array_add (items, l);
      }
    }
    const gint64 t0 = g_get_monotonic_time ();
    for (guint k = 0; k < items->len; k++) {
      if (!mode)
        gst_harness_push (h, (GstBuffer *) g_ptr_array_index (items, k));
      else
        gst_pad_push_list (h->srcpad, (GstBufferList *) g_ptr_array_index (items, k));
    }
    const gint64 t1 = g_get_monotonic_time ();
    const guint got = gst_harness_buffers_in_queue (h);
    printf ("{\"case\": \"%s\", \"buffers\": %d, \"frames_per_buffer\": 1024, \"outputs\": %u, \"us_per_buffer\": %.2f}\n",
        mode ? "buffer lists" : "one buffer at a time", total, got, (double) (t1 - t0) / total);
    if (mode)
      printf ("{\"buffers_per_list\": %d}\n", per_list);
    while (gst_harness_buffers_in_queue (h))
      gst_buffer_unref (gst_harness_pull (h));
    g_ptr_array_free (items, TRUE);
    gst_harness_teardown (h);
  }
  return 0;
}

int
main (int argc, char **argv)
{
  gst_init (&argc, &argv);
  if (argc >= 2 && strcmp (argv[1], "audio-list-bench") == 0)
    return audio_list_bench (argc >= 3 ? atoi (argv[2]) : 8);
  if (argc >= 2 && strcmp (argv[1], "audio-list") == 0)
    return audio_list ();
  if (argc >= 2 && strcmp (argv[1], "hip-memory") == 0)
    return hip_memory ();
  if (argc >= 2 && strcmp (argv[1], "upload-meta") == 0)
    return upload_meta ();
  if (argc >= 2 && strcmp (argv[1], "nav-meta") == 0)
    return nav_meta ();
  if (argc < 2 || strcmp (argv[1], "mix-matrix") != 0) {
    fprintf (stderr, "usage: %s mix-matrix | upload-meta\n", argv[0]);
    return 2;
  }
  GstHarness *hn = gst_harness_new ("audioconvert");
  if (!hn) {
    fprintf (stderr, "no audioconvert element (GST_PLUGIN_PATH?)\n");
    return 1;
  }
  const char *caps = "audio/x-raw,format=F32LE,rate=48000,channels=2,layout=interleaved,channel-mask=(bitmask)0x3";
  gst_harness_set_caps_str (hn, caps, caps);
  const int frames = 480;
  int bad = 0;
  if (gst_harness_push (hn, make_buffer (frames, 0.1f)) != GST_FLOW_OK)
    bad |= 1;
  bad |= check (gst_harness_try_pull (hn), frames, 0.1f, 0) << 1;
  /* the swap, as the property's GstValueArray of rows */
  GValue m = G_VALUE_INIT, row = G_VALUE_INIT, v = G_VALUE_INIT;
  g_value_init (&m, GST_TYPE_ARRAY);
  for (int r = 0; r < 2; r++) {
    g_value_init (&row, GST_TYPE_ARRAY);
    for (int c = 0; c < 2; c++) {
      g_value_init (&v, G_TYPE_FLOAT);
      g_value_set_float (&v, r != c ? 1.0f : 0.0f);
      gst_value_array_append_value (&row, &v);
      g_value_unset (&v);
    }
    gst_value_array_append_value (&m, &row);
    g_value_unset (&row);
  }
  g_object_set_property (G_OBJECT (hn->element), "mix-matrix", &m);
  g_value_unset (&m);
  for (int k = 0; k < 2; k++) {
    const GstFlowReturn fr = gst_harness_push (hn, make_buffer (frames, 0.3f + k));
    if (fr != GST_FLOW_OK) {
      fprintf (stderr, "buffer %d after the property change: %s\n", k, gst_flow_get_name (fr));
      bad |= 4;
    }
    bad |= check (gst_harness_try_pull (hn), frames, 0.3f + k, 1) << (3 + k);
  }
  gst_harness_teardown (hn);
  if (bad) {
    fprintf (stderr, "live mix-matrix change: wrong output (mask %d)\n", bad);
    return 1;
  }
  printf ("ok\n");
  return 0;
}
