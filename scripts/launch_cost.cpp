// launch_cost.cpp - host time per call of the library's per-frame entry point next to the bare HIP calls it is made of
// (measurement tool).   g++ -O2 scripts/launch_cost.cpp -Iinclude -Lgstreamer_amd/lib -lgstamddsp -o scripts/launch_cost
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gstamd_video.h"

static double now_us ()
{
  return std::chrono::duration<double, std::micro> (std::chrono::steady_clock::now ().time_since_epoch ()).count ();
}

int main (int argc, char **argv)
{
  const int w = argc > 1 ? atoi (argv[1]) : 3840, h = argc > 2 ? atoi (argv[2]) : 2160, n = argc > 3 ? atoi (argv[3]) : 2000;
  GstAmdVideoInfo ii, oi;
  gstamd_video_info_set_format (&ii, GSTAMD_VIDEO_FORMAT_NV12, w, h);
  gstamd_video_info_set_format (&oi, GSTAMD_VIDEO_FORMAT_BGRA, w, h);
  GstAmdVideoConverterConfig cfg;
  gstamd_video_converter_config_init (&cfg);
  cfg.resampler_method = 1;
  int status = 0;
  GstAmdVideoConverter *c = gstamd_video_converter_new (&ii, &oi, &cfg, &status);
  if (!c) {
    fprintf (stderr, "no converter: %s\n", gstamd_last_error ());
    return 1;
  }
  const int pool = 8;
  std::vector<void *> in (pool), out (pool);
  for (int i = 0; i < pool; i++) {
    in[i] = gstamd_device_alloc (ii.size);
    out[i] = gstamd_device_alloc (oi.size);
  }
  void *stream = gstamd_stream_new ();
  void *ev = gstamd_event_new ();
  for (int i = 0; i < 50; i++)
    gstamd_video_converter_frame (c, in[i % pool], out[i % pool], stream);
  gstamd_stream_synchronize (stream);
  // (a) converter calls back to back, queue never drained: includes back-pressure when the GPU is the slower side
  double t0 = now_us ();
  for (int i = 0; i < n; i++)
    gstamd_video_converter_frame (c, in[i % pool], out[i % pool], stream);
  double t1 = now_us ();
  gstamd_stream_synchronize (stream);
  double t2 = now_us ();
  printf ("{\"what\": \"converter_frame back to back\", \"w\": %d, \"h\": %d, \"host_us_per_call\": %.3f, \"wall_us_per_frame\": %.3f}\n", w, h, (t1 - t0) / n, (t2 - t0) / n);
  // (b) one call at a time on an idle queue: the pure host cost of the call
  double acc = 0;
  for (int i = 0; i < 300; i++) {
    gstamd_stream_synchronize (stream);
    double a = now_us ();
    gstamd_video_converter_frame (c, in[i % pool], out[i % pool], stream);
    acc += now_us () - a;
  }
  printf ("{\"what\": \"converter_frame on an idle queue\", \"host_us_per_call\": %.3f}\n", acc / 300);
  // (c) the same with an event record after every call
  gstamd_stream_synchronize (stream);
  t0 = now_us ();
  for (int i = 0; i < n; i++) {
    gstamd_video_converter_frame (c, in[i % pool], out[i % pool], stream);
    gstamd_event_record (ev, stream);
  }
  t1 = now_us ();
  gstamd_stream_synchronize (stream);
  t2 = now_us ();
  printf ("{\"what\": \"converter_frame + event record\", \"host_us_per_call\": %.3f, \"wall_us_per_frame\": %.3f}\n", (t1 - t0) / n, (t2 - t0) / n);
  // (d) event record alone on an idle stream
  acc = 0;
  for (int i = 0; i < 300; i++) {
    double a = now_us ();
    gstamd_event_record (ev, stream);
    acc += now_us () - a;
  }
  printf ("{\"what\": \"event record alone\", \"host_us_per_call\": %.3f}\n", acc / 300);
  // (f) a ring of streams, frames round robin, with and without an event record (from a ring of events) per frame
  for (int ring : {2, 3, 4}) {
    std::vector<void *> st (ring), evs (64);
    for (auto &x : st)
      x = gstamd_stream_new ();
    for (auto &x : evs)
      x = gstamd_event_new ();
    for (int with_ev = 0; with_ev < 2; with_ev++) {
      for (auto &x : st)
        gstamd_stream_synchronize (x);
      t0 = now_us ();
      for (int i = 0; i < n; i++) {
        gstamd_video_converter_frame (c, in[i % pool], out[i % pool], st[i % ring]);
        if (with_ev)
          gstamd_event_record (evs[i % 64], st[i % ring]);
      }
      t1 = now_us ();
      for (auto &x : st)
        gstamd_stream_synchronize (x);
      t2 = now_us ();
      printf ("{\"what\": \"ring of %d streams%s\", \"host_us_per_call\": %.3f, \"wall_us_per_frame\": %.3f}\n", ring, with_ev ? " + event record" : "", (t1 - t0) / n, (t2 - t0) / n);
    }
  }
  // (e) lists of 4 and 32
  for (int list_n : {4, 32}) {
    std::vector<const void *> src (list_n);
    std::vector<void *> dst (list_n);
    for (int i = 0; i < list_n; i++) {
      src[i] = in[i % pool];
      dst[i] = out[i % pool];
    }
    gstamd_stream_synchronize (stream);
    t0 = now_us ();
    for (int i = 0; i < n / list_n; i++)
      gstamd_video_converter_frames (c, list_n, src.data (), dst.data (), stream);
    t1 = now_us ();
    gstamd_stream_synchronize (stream);
    t2 = now_us ();
    printf ("{\"what\": \"converter_frames list of %d\", \"host_us_per_frame\": %.3f, \"wall_us_per_frame\": %.3f}\n", list_n, (t1 - t0) / (n / list_n * list_n), (t2 - t0) / (n / list_n * list_n));
  }
  return 0;
}
