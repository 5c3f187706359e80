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
 * prints "ok" and exits 0 when every output buffer is what the options in force at that buffer say. */
#include <gst/check/gstharness.h>
#include <gst/gst.h>
#include <gst/video/video.h>
#include <stdio.h>
#include <string.h>

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

int
main (int argc, char **argv)
{
  gst_init (&argc, &argv);
  if (argc >= 2 && strcmp (argv[1], "upload-meta") == 0)
    return upload_meta ();
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
