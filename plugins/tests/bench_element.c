/* bench_element.c - frames/s THROUGH THE ELEMENT, frames resident in HBM (test / measurement tool, not part of the plugin).
 *
 * A GstHarness drives one `videoconvertscale` instance the way gst_base_transform_chain does in a pipeline
 * (gstbasetransform.c:2351 -> transform): video/x-raw(memory:AMDHIPMemory) buffers from a pool of distinct frames are pushed
 * into the sink pad, the converted HBM buffers are pulled from the src pad and released back to the element's pool.  No host
 * synchronisation happens inside the timed loop; the clock stops after a CPU map of the last output (which waits for it).
 *
 *   bench_element <in_fmt> <w> <h> <out_fmt> <ow> <oh> <frames> <hip-streams> [method] [list] [batch-buffers]
 * list > 1: the frames are pushed as GstBufferLists of that many buffers (gst_pad_push_list), the element's chain_list path
 * prints one JSON line. */
#include <gst/check/gstharness.h>
#include <gst/gst.h>
#include <gst/video/video.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../gstamdhipmemory.h"

static void
wait_for (GstBuffer * buf)
{
  /* wait for the producer's ticket only: a CPU map would add a pinned allocation and a 33 MB download to the timed region */
  gst_amd_hip_memory_host_wait (gst_buffer_peek_memory (buf, 0));
}

int
main (int argc, char **argv)
{
  setenv ("HIP_FORCE_DEV_KERNARG", "1", 0);     /* kernel arguments in HBM; the launcher's decision, single-threaded here (tuning.cpp) */
  if (argc < 9) {
    fprintf (stderr, "usage: %s in_fmt w h out_fmt ow oh frames hip-streams [method]\n", argv[0]);
    return 2;
  }
  const char *ifmt = argv[1], *ofmt = argv[4], *method = argc > 9 ? argv[9] : "bilinear";
  const int list_n = argc > 10 ? atoi (argv[10]) : 1;
  const int batch_n = argc > 11 ? atoi (argv[11]) : 1;          /* the harness answers latency queries as a live source: say it */
  const int w = atoi (argv[2]), h = atoi (argv[3]), ow = atoi (argv[5]), oh = atoi (argv[6]), streams = atoi (argv[8]);
  int frames = atoi (argv[7]);
  frames = (frames + list_n - 1) / list_n * list_n;
  gst_init (&argc, &argv);
  GstHarness *hn = gst_harness_new ("videoconvertscale");
  if (!hn) {
    fprintf (stderr, "no videoconvertscale element (GST_PLUGIN_PATH?)\n");
    return 1;
  }
  gst_util_set_object_arg (G_OBJECT (hn->element), "method", method);
  g_object_set (hn->element, "hip-streams", (guint) streams, "batch-buffers", (guint) batch_n, NULL);
  gchar *in_caps = g_strdup_printf ("video/x-raw(memory:AMDHIPMemory),format=%s,width=%d,height=%d,framerate=30/1", ifmt, w, h);
  gchar *out_caps = g_strdup_printf ("video/x-raw(memory:AMDHIPMemory),format=%s,width=%d,height=%d,framerate=30/1", ofmt, ow, oh);
  gst_harness_set_caps_str (hn, in_caps, out_caps);

  GstVideoInfo ii;
  GstCaps *c = gst_caps_from_string (in_caps);
  gst_video_info_from_caps (&ii, c);
  gst_caps_unref (c);
  /* distinct input frames worth > 256 MiB (the Infinity Cache), xorshift bytes */
  int n_in = (int) (400e6 / GST_VIDEO_INFO_SIZE (&ii)) + 1;
  if (n_in < 4)
    n_in = 4;
  GstBuffer **in = g_new0 (GstBuffer *, n_in);
  guint64 x = 0x9E3779B97F4A7C15ull;
  for (int i = 0; i < n_in; i++) {
    GstMapInfo map;
    in[i] = gst_amd_hip_buffer_new_video (&ii);
    if (!in[i] || !gst_buffer_map (in[i], &map, GST_MAP_WRITE)) {
      fprintf (stderr, "HBM allocation failed\n");
      return 1;
    }
    for (gsize k = 0; k + 8 <= map.size; k += 8) {
      x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
      guint64 v = x * 0x2545F4914F6CDD1Dull;
      memcpy (map.data + k, &v, 8);
    }
    gst_buffer_unmap (in[i], &map);
  }
  const int warm = ((frames / 4 + 8 + list_n - 1) / list_n) * list_n;      /* whole lists */
  GstBuffer *last = NULL;
  gint64 t0 = 0;
  for (int i = 0; i < warm + frames; i++) {
    if (i == warm) {
      if (last)
        wait_for (last);
      t0 = g_get_monotonic_time ();
    }
    if (list_n > 1) {
      GstBufferList *bl = gst_buffer_list_new_sized (list_n);
      for (int k = 0; k < list_n; k++)
        gst_buffer_list_add (bl, gst_buffer_ref (in[(i + k) % n_in]));
      if (gst_pad_push_list (hn->srcpad, bl) != GST_FLOW_OK) {
        fprintf (stderr, "push_list failed at frame %d\n", i);
        return 1;
      }
      for (int k = 0; k < list_n; k++) {
        GstBuffer *out = gst_harness_pull (hn);
        if (last)
          gst_buffer_unref (last);
        last = out;
      }
      i += list_n - 1;
      continue;
    }
    if (gst_harness_push (hn, gst_buffer_ref (in[i % n_in])) != GST_FLOW_OK) {
      fprintf (stderr, "push failed at frame %d\n", i);
      return 1;
    }
    GstBuffer *out = gst_harness_pull (hn);
    if (last)
      gst_buffer_unref (last);        /* back to the element's pool: reused a few frames later, ordered by events */
    last = out;
  }
  wait_for (last);
  const double secs = (g_get_monotonic_time () - t0) * 1e-6;
  guint64 sum = 0;
  if (g_getenv ("GSTAMD_BENCH_SUM")) {        /* checksum of the last output frame (same input frame for every list size) */
    GstMapInfo map;
    if (gst_buffer_map (last, &map, GST_MAP_READ)) {
      for (gsize k = 0; k < map.size; k++)
        sum = sum * 1099511628211ull + map.data[k];
      gst_buffer_unmap (last, &map);
    }
  }
  const double in_bytes = (double) GST_VIDEO_INFO_SIZE (&ii), out_bytes = (double) gst_buffer_get_size (last);
  printf ("{\"element\": \"videoconvertscale\", \"in\": \"%s %dx%d\", \"out\": \"%s %dx%d\", \"method\": \"%s\", \"hip_streams\": %d, "
      "\"buffers_per_list\": %d, \"batch_buffers\": %d, \"frames\": %d, \"frames_per_s\": %.1f, \"us_per_frame\": %.3f, \"algorithmic_gb_per_s\": %.1f, \"input_pool_frames\": %d, \"last_frame_sum\": %" G_GUINT64_FORMAT "}\n",
      ifmt, w, h, ofmt, ow, oh, method, streams, list_n, batch_n, frames, frames / secs, secs * 1e6 / frames, (in_bytes + out_bytes) * frames / secs / 1e9,
      n_in, sum);
  gst_buffer_unref (last);
  for (int i = 0; i < n_in; i++)
    gst_buffer_unref (in[i]);
  gst_harness_teardown (hn);
  return 0;
}
