// audio_kernels.hip - gfx950 kernels + C ABI (include/gstamd_audio.h) of the polyphase FIR resampler.
//
// Two kernels per resample() call, both on the caller's stream:
//   k_fir      one lane per output sample-channel; the phase's taps row (<= a few hundred bytes) and the
//              input window come through L1/L2 (the whole 147 x 72 f32 table of 48k->44.1k is 42 KB);
//   k_history  writes the frames the next call still needs into the other history buffer.
// Input samples are read straight from the caller's buffer (no deinterleaved staging copy as in the
// reference, audio-resampler.c:879-897): algorithmic traffic is in + out once.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gstamd_audio.h"
#include "audio_device.h"
#include "audio_taps.h"
#include "tuning.h"

using namespace gstamd;

template <typename T>
__global__ __launch_bounds__ (256) void k_fir (FirParams p, const T *__restrict__ hist, const T *__restrict__ in,
    const T *__restrict__ table, T *__restrict__ out, long long n_out)
{
  const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;   // i = j * channels + c
  if (i >= n_out * p.channels)
    return;
  const long long j = i / p.channels;
  const int c = (int) (i % p.channels);
  out[fir_out_index (p, j, c)] = fir_output<T> (p, hist, in, table, j, c);
}

template <typename T>
__global__ __launch_bounds__ (256) void k_history (FirParams p, const T *__restrict__ hist, const T *__restrict__ in,
    T *__restrict__ new_hist, long long src_start, long long moved, long long keep)
{
  const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= keep * p.channels)
    return;
  new_hist[i] = history_sample<T> (p, hist, in, src_start, moved, i / p.channels, (int) (i % p.channels));
}

// FULL-table mode through LDS (audio_device.h, second half): FIR_LDS_FRAMES output frames per 256-lane workgroup, four lanes per
// frame.  The workgroups past the FIR ones write the history the next call needs (k_history's job, one launch instead of two).
template <typename T>
__device__ __forceinline__ void fir_lds_block (const FirParams &p, const FirLdsGeom &g, const T *__restrict__ hist, const T *__restrict__ in,
    const T *__restrict__ table, T *__restrict__ out, long long n_out, int fir_blocks, T *__restrict__ new_hist, long long src_start,
    long long moved, long long keep, int bx)
{
  extern __shared__ __attribute__ ((aligned (16))) unsigned char fir_lds_raw[];
  if (bx >= fir_blocks) {
    const long long i = (long long) (bx - fir_blocks) * blockDim.x + threadIdx.x;
    if (i < keep * p.channels)
      new_hist[i] = history_sample<T> (p, hist, in, src_start, moved, i / p.channels, (int) (i % p.channels));
    return;
  }
  typedef typename Acc<T>::type A;
  T *rows = (T *) fir_lds_raw, *win = rows + FIR_LDS_FRAMES * g.row_stride;
  int *pos = (int *) (win + (size_t) p.channels * g.win_frames);
  const long long jb = (long long) bx * FIR_LDS_FRAMES;
  const int nj = n_out - jb < FIR_LDS_FRAMES ? (int) (n_out - jb) : FIR_LDS_FRAMES;
  fir_lds_positions (p, jb, nj, pos, (int) threadIdx.x, 256);
  __syncthreads ();
  fir_lds_stage<T> (p, g, hist, in, table, jb, nj, pos, rows, win, (int) threadIdx.x, 256);
  __syncthreads ();
  const int fr = (int) threadIdx.x >> 2, q = (int) threadIdx.x & 3;
  const int frc = fr < nj ? fr : nj - 1;        /* idle quads repeat the last frame (no divergence before the shuffles) */
  for (int c = 0; c < p.channels; c++) {
    const A r = fir_lds_partial<T> (p, g, pos, rows, win, frc, q, c);
    const int lane = (int) (threadIdx.x & 63), base = lane & ~3;
    const A r0 = __shfl (r, base, 64), r1 = __shfl (r, base + 1, 64), r2 = __shfl (r, base + 2, 64), r3 = __shfl (r, base + 3, 64);
    if (fr < nj && q == (c & 3))
      out[fir_out_index (p, jb + fr, c)] = fir_lds_combine<T> (r0, r1, r2, r3);
  }
}

template <typename T>
__global__ __launch_bounds__ (256) void k_fir_lds (FirParams p, FirLdsGeom g, const T *__restrict__ hist, const T *__restrict__ in,
    const T *__restrict__ table, T *__restrict__ out, long long n_out, int fir_blocks, T *__restrict__ new_hist, long long src_start,
    long long moved, long long keep)
{
  fir_lds_block<T> (p, g, hist, in, table, out, n_out, fir_blocks, new_hist, src_start, moved, keep, (int) blockIdx.x);
}

// ---- many independent resamplers of one filter in ONE launch (round 5: gstamd_audio_resampler_resample_many) ---------------------------------
// A buffer of 1024 frames is 5.7 us of launch for 0.2 us of work; "independent streams" is this path's parallelism, so the streams of a
// mixer / a multi-channel capture / a transcoding farm go into one grid: blockIdx.y = stream.  What differs between the streams - the
// buffers and where each stands in its own history - fits 56 bytes, 64 streams fit the kernel arguments; the filter (rates, taps table,
// channel count, sample type) is the same for all.  Each stream's blocks do exactly what its own k_fir_lds launch would have done.
#define GSTAMD_AUDIO_MANY_MAX 64
struct FirManyStream {
  const void *hist, *in;
  void *out, *new_hist;
  int samp_index0, samp_phase0, hist_frames, in_frames, n_out;
  int pad;
};
struct FirMany {
  FirManyStream s[GSTAMD_AUDIO_MANY_MAX];
};

template <typename T>
__global__ __launch_bounds__ (256) void k_fir_lds_many (FirParams shared, FirLdsGeom g, const T *__restrict__ table, FirMany many)
{
  const FirManyStream &m = many.s[blockIdx.y];
  FirParams p = shared;
  p.samp_index0 = m.samp_index0;
  p.samp_phase0 = m.samp_phase0;
  p.hist_frames = m.hist_frames;
  p.total_frames = (long long) m.hist_frames + m.in_frames;
  p.in_is_null = m.in == nullptr;
  if (p.in_plane_stride)
    p.in_plane_stride = m.in_frames;            /* non-interleaved sides: the planes follow each other, as in _resample */
  if (p.out_plane_stride)
    p.out_plane_stride = m.n_out;
  // the history hand-over of audio_step (audio_taps.cpp), from the same numbers
  long long src_start = 0, moved = p.total_frames, keep = p.total_frames;
  if (m.n_out > 0) {
    const long long tot = (long long) m.samp_phase0 + (long long) m.n_out * p.samp_frac;
    const long long end_index = (long long) m.samp_index0 + (long long) m.n_out * p.samp_inc + tot / p.out_rate;
    const long long consumed = end_index - m.samp_index0;
    if (p.total_frames > end_index) {
      src_start = end_index;
      moved = p.total_frames - end_index;
    } else {
      src_start = 0;
      moved = 0;
    }
    keep = consumed > 0 ? (p.total_frames - consumed > 0 ? p.total_frames - consumed : 0) : p.total_frames;
  }
  const int fir_blocks = (int) ((m.n_out + FIR_LDS_FRAMES - 1) / FIR_LDS_FRAMES);
  const int hist_blocks = keep > 0 ? (int) ((keep * p.channels + 255) / 256) : 0;
  if ((int) blockIdx.x >= fir_blocks + hist_blocks)
    return;
  fir_lds_block<T> (p, g, (const T *) m.hist, (const T *) m.in, table, (T *) m.out, m.n_out, fir_blocks, (T *) m.new_hist, src_start, moved, keep,
      (int) blockIdx.x);
}

struct GstAmdAudioResampler {
  AudioPlan plan;
  std::mutex lock;
  AudioState st;
  // device state
  bool device_ready = false;
  void *table_dev = nullptr;
  void *hist[2] = {nullptr, nullptr};
  size_t hist_cap[2] = {0, 0};      // frames
  int cur = 0;
  unsigned long long table_hash = 0;      // of plan.table: resamplers with the same hash, rates, taps and sample type share one launch (_resample_many)
  std::string divergence;                 // gstamd_audio_resampler_divergence: set by an update that enlarges the filter past the history (audio_update)
};

extern "C" const char *gstamd_last_error (void);
extern "C" void gstamd_internal_set_error (const char *msg);

static int audio_hip_fail (const char *where)
{
  const hipError_t e = hipGetLastError ();
  gstamd_internal_set_error ((std::string ("audio resampler, ") + where + ": " + (e != hipSuccess ? hipGetErrorString (e) : "HIP call failed")).c_str ());
  return GSTAMD_ERR_HIP;
}

static int ensure_hist (GstAmdAudioResampler *r, int which, size_t frames)
{
  if (r->hist_cap[which] >= frames)
    return GSTAMD_OK;
  size_t cap = frames + frames / 2 + 256;
  void *n = nullptr;
  const size_t fbytes = (size_t) r->plan.bps * r->plan.channels;
  /* A (re)allocation is rare and synchronous, on BOTH sides.  hipMemset and a device-to-device hipMemcpy run on the null stream and return before they
     have executed; the caller's stream is usually a non-blocking one, which the null stream does not order against: without the two synchronisations the
     memset of a fresh history buffer could land AFTER the first kernel had written the next call's history into it (round 6, found by
     plugins/tests/live_props audio-list: the SECOND buffer of a stream came out as if its history were silence, about one run in ten), and the copy out
     of the old buffer could read it before the kernels still writing it were done. */
  if (hipDeviceSynchronize () != hipSuccess)
    return audio_hip_fail (__func__);
  if (hipMalloc (&n, cap * fbytes) != hipSuccess)
    return audio_hip_fail (__func__);
  if (hipMemset (n, 0, cap * fbytes) != hipSuccess)
    return audio_hip_fail (__func__);
  if (r->hist[which]) {
    /* keep the valid frames (only matters for the current buffer) */
    if (hipMemcpy (n, r->hist[which], r->hist_cap[which] * fbytes, hipMemcpyDeviceToDevice) != hipSuccess)
      return audio_hip_fail (__func__);
  }
  if (hipDeviceSynchronize () != hipSuccess)
    return audio_hip_fail (__func__);
  if (r->hist[which])
    (void) hipFree (r->hist[which]);
  r->hist[which] = n;
  r->hist_cap[which] = cap;
  return GSTAMD_OK;
}

static int ensure_device (GstAmdAudioResampler *r)
{
  if (r->device_ready)
    return GSTAMD_OK;
  if (!r->plan.table.empty ()) {
    if (hipMalloc (&r->table_dev, r->plan.table.size ()) != hipSuccess)
      return audio_hip_fail (__func__);
    if (hipMemcpy (r->table_dev, r->plan.table.data (), r->plan.table.size (), hipMemcpyHostToDevice) != hipSuccess)
      return audio_hip_fail (__func__);
    unsigned long long h = 1469598103934665603ull;
    for (unsigned char b : r->plan.table)
      h = (h ^ b) * 1099511628211ull;
    r->table_hash = h ^ r->plan.table.size ();
  }
  int e = ensure_hist (r, 0, (size_t) r->plan.n_taps + 64);
  if (e == GSTAMD_OK)
    e = ensure_hist (r, 1, (size_t) r->plan.n_taps + 64);
  if (e != GSTAMD_OK)
    return e;
  r->device_ready = true;
  return GSTAMD_OK;
}

static FirParams make_fir_params (const AudioPlan &pl, const AudioStep &s, bool in_null, long long in_stride, long long out_stride)
{
  FirParams p;
  memset (&p, 0, sizeof (p));
  p.channels = pl.channels;
  p.n_taps_padded = pl.taps_stride;
  p.interp = pl.filter_mode == GSTAMD_AUDIO_FILTER_MODE_INTERPOLATED && pl.method != GSTAMD_AUDIO_RESAMPLER_METHOD_NEAREST ?
      (pl.filter_interpolation == GSTAMD_AUDIO_FILTER_INTERPOLATION_CUBIC ? 2 : 1) : 0;
  p.oversample = pl.oversample;
  p.nearest = (pl.method == GSTAMD_AUDIO_RESAMPLER_METHOD_NEAREST || pl.in_rate == pl.out_rate) ? 1 : 0;
  p.samp_inc = pl.samp_inc;
  p.samp_frac = pl.samp_frac;
  p.out_rate = pl.out_rate;
  p.samp_index0 = s.samp_index0;
  p.samp_phase0 = s.samp_phase0;
  p.hist_frames = s.hist_frames;
  p.total_frames = s.total_frames;
  p.in_is_null = in_null;
  p.in_plane_stride = pl.in_planar ? in_stride : 0;
  p.out_plane_stride = pl.out_planar ? out_stride : 0;
  return p;
}

template <typename T>
static int run_resample (GstAmdAudioResampler *r, const void *in, size_t in_frames, void *out, size_t out_frames,
    long long in_stride, long long out_stride, hipStream_t stream)
{
  const AudioPlan &pl = r->plan;
  const AudioStep s = audio_step (pl, &r->st, in_frames, out_frames);
  if (s.skipped_all)
    return GSTAMD_OK;
  const FirParams p = make_fir_params (pl, s, in == nullptr, in_stride, out_stride);
  const int cur = r->cur, nxt = cur ^ 1;
  /* FULL mode: one launch of the LDS-staged kernel does the FIR and the history hand-over */
  if (s.run_fir && !p.nearest && !p.interp && !tuning_on ("GSTAMD_NO_FIR_LDS")) {
    FirLdsGeom g;
    g.row_stride = p.n_taps_padded + 4;
    const int span_max = FIR_LDS_FRAMES * (p.samp_inc + 1) + p.n_taps_padded + 2;
    g.win_frames = ((span_max + 31) & ~31) + 16;
    const size_t lds = ((size_t) FIR_LDS_FRAMES * g.row_stride + (size_t) pl.channels * g.win_frames) * sizeof (T) + 2 * FIR_LDS_FRAMES * sizeof (int);
    if (lds <= 64 * 1024) {
      if (s.keep > 0) {
        int e = ensure_hist (r, nxt, (size_t) s.keep + 64);
        if (e != GSTAMD_OK)
          return e;
      }
      const int fir_blocks = (int) ((s.n_out + FIR_LDS_FRAMES - 1) / FIR_LDS_FRAMES);
      const int hist_blocks = s.keep > 0 ? (int) ((s.keep * pl.channels + 255) / 256) : 0;
      hipLaunchKernelGGL (k_fir_lds<T>, dim3 ((unsigned) (fir_blocks + hist_blocks)), dim3 (256), lds, stream, p, g, (const T *) r->hist[cur],
          (const T *) in, (const T *) r->table_dev, (T *) out, s.n_out, fir_blocks, (T *) r->hist[nxt], s.src_start, s.moved, s.keep);
      if (hipGetLastError () != hipSuccess)
        return audio_hip_fail (__func__);
      r->cur = nxt;
      return GSTAMD_OK;
    }
  }
  if (s.run_fir) {
    const long long total = s.n_out * pl.channels;
    hipLaunchKernelGGL (k_fir<T>, dim3 ((unsigned) ((total + 255) / 256)), dim3 (256), 0, stream, p, (const T *) r->hist[cur],
        (const T *) in, (const T *) r->table_dev, (T *) out, s.n_out);
    if (hipGetLastError () != hipSuccess)
      return audio_hip_fail (__func__);
  }
  if (s.keep > 0) {
    int e = ensure_hist (r, nxt, (size_t) s.keep + 64);
    if (e != GSTAMD_OK)
      return e;
    const long long total = s.keep * pl.channels;
    hipLaunchKernelGGL (k_history<T>, dim3 ((unsigned) ((total + 255) / 256)), dim3 (256), 0, stream, p, (const T *) r->hist[cur],
        (const T *) in, (T *) r->hist[nxt], s.src_start, s.moved, s.keep);
    if (hipGetLastError () != hipSuccess)
      return audio_hip_fail (__func__);
  }
  r->cur = nxt;
  return GSTAMD_OK;
}

template <typename T>
static int run_many (int n, GstAmdAudioResampler *const *rs, const void *const *in, const size_t *in_frames, void *const *out, const size_t *out_frames,
    hipStream_t stream)
{
  const AudioPlan &pl = rs[0]->plan;
  FirMany many;
  memset ((void *) &many, 0, sizeof (many));
  AudioStep first;
  memset (&first, 0, sizeof (first));
  int max_blocks = 0, live = 0;
  /* every allocation first, on a COPY of each stream's state: a failure for stream i must not leave streams 0 .. i - 1 advanced (their input consumed,
     r->cur pointing at a history buffer no kernel has written) with nothing launched */
  for (int i = 0; i < n; i++) {
    GstAmdAudioResampler *r = rs[i];
    auto probe = r->st;
    const AudioStep s = audio_step (r->plan, &probe, in_frames[i], out_frames[i]);
    if (!s.skipped_all && s.keep > 0) {
      const int e = ensure_hist (r, r->cur ^ 1, (size_t) s.keep + 64);
      if (e != GSTAMD_OK)
        return e;
    }
  }
  for (int i = 0; i < n; i++) {
    GstAmdAudioResampler *r = rs[i];
    const AudioStep s = audio_step (r->plan, &r->st, in_frames[i], out_frames[i]);
    if (s.skipped_all)
      continue;
    FirManyStream &m = many.s[live++];
    m.hist = r->hist[r->cur];
    m.new_hist = r->hist[r->cur ^ 1];
    m.in = in[i];
    m.out = out[i];
    m.samp_index0 = (int) s.samp_index0;
    m.samp_phase0 = s.samp_phase0;
    m.hist_frames = (int) s.hist_frames;
    m.in_frames = (int) (s.total_frames - s.hist_frames);
    m.n_out = s.run_fir ? (int) s.n_out : 0;
    const int blocks = (int) ((m.n_out + FIR_LDS_FRAMES - 1) / FIR_LDS_FRAMES) + (s.keep > 0 ? (int) ((s.keep * pl.channels + 255) / 256) : 0);
    max_blocks = blocks > max_blocks ? blocks : max_blocks;
    r->cur ^= 1;
    first = s;
  }
  if (!live || !max_blocks)
    return GSTAMD_OK;
  const FirParams p = make_fir_params (pl, first, false, 1, 1);          /* the per-stream fields are filled in by the kernel */
  FirLdsGeom g;
  g.row_stride = p.n_taps_padded + 4;
  const int span_max = FIR_LDS_FRAMES * (p.samp_inc + 1) + p.n_taps_padded + 2;
  g.win_frames = ((span_max + 31) & ~31) + 16;
  const size_t lds = ((size_t) FIR_LDS_FRAMES * g.row_stride + (size_t) pl.channels * g.win_frames) * sizeof (T) + 2 * FIR_LDS_FRAMES * sizeof (int);
  hipLaunchKernelGGL (k_fir_lds_many<T>, dim3 ((unsigned) max_blocks, (unsigned) live), dim3 (256), lds, stream, p, g, (const T *) rs[0]->table_dev, many);
  return hipGetLastError () == hipSuccess ? GSTAMD_OK : audio_hip_fail (__func__);
}

extern "C" {

void gstamd_audio_resampler_options_init (GstAmdAudioResamplerOptions *options)
{
  if (options)
    audio_options_init (options);
}

void gstamd_audio_resampler_options_set_quality (int method, unsigned quality, int in_rate, int out_rate,
    GstAmdAudioResamplerOptions *options)
{
  audio_options_set_quality (method, quality, in_rate, out_rate, options);
}

GstAmdAudioResampler *gstamd_audio_resampler_new (int method, int flags, int format, int channels, int in_rate, int out_rate,
    const GstAmdAudioResamplerOptions *options, int *status)
{
  GstAmdAudioResampler *r = new GstAmdAudioResampler ();
  std::string err;
  int e = plan_audio_resampler (method, flags, format, channels, in_rate, out_rate, options, &r->plan, &err);
  if (status)
    *status = e;
  if (e != GSTAMD_OK) {
    gstamd_internal_set_error (err.c_str ());
    delete r;
    return nullptr;
  }
  gstamd_audio_resampler_reset (r);
  return r;
}

void gstamd_audio_resampler_free (GstAmdAudioResampler *r)
{
  if (!r)
    return;
  if (r->table_dev)
    (void) hipFree (r->table_dev);
  for (void *h : r->hist)
    if (h)
      (void) hipFree (h);
  delete r;
}

/* gst_audio_resampler_reset (audio-resampler.c:1466-1488): half of the filter is filled with 0 */
void gstamd_audio_resampler_reset (GstAmdAudioResampler *r)
{
  if (!r)
    return;
  std::lock_guard<std::mutex> g (r->lock);
  audio_state_reset (r->plan, &r->st);
  if (r->device_ready) {
    const size_t fbytes = (size_t) r->plan.bps * r->plan.channels;
    /* (null-stream memset: ordered against the streams the resampler has been used on by synchronising around it, see ensure_hist) */
    (void) hipDeviceSynchronize ();
    (void) hipMemset (r->hist[r->cur], 0, (size_t) (r->plan.n_taps / 2) * fbytes);
    (void) hipDeviceSynchronize ();
  }
}

/* gst_audio_resampler_update (audio-resampler.c:1503-1614).  Rare and synchronous: the device is drained, the filter
 * table is replaced and, when the tap count changed, the few history frames are shifted on a host copy. */
int gstamd_audio_resampler_update (GstAmdAudioResampler *r, int in_rate, int out_rate, const GstAmdAudioResamplerOptions *options)
{
  if (!r)
    return GSTAMD_ERR_INVALID;
  std::lock_guard<std::mutex> g (r->lock);
  AudioPlan plan = r->plan;
  AudioState st = r->st;
  const size_t old_avail = st.samples_avail + (size_t) st.samp_index;
  AudioHistoryShift shift;
  std::string err;
  int e = audio_update (&plan, &st, in_rate, out_rate, options, &shift, &err);
  if (e != GSTAMD_OK) {
    gstamd_internal_set_error (err.c_str ());
    return e;
  }
  const bool new_table = plan.table != r->plan.table;
  r->plan = std::move (plan);
  r->st = st;
  if (shift.stale > 0) {
    char text[512];
    snprintf (text, sizeof (text), "gst_audio_resampler_update enlarged the filter by more than the history held: the reference fills %lld frame(s) of the new "
        "history with what its sample buffer held past the valid samples (input of earlier calls; audio-resampler.c:1587-1590, a FIXME there), this library "
        "with silence; output equals the reference's again once those frames have left the filter window. ", shift.stale);
    r->divergence = text;
  }
  if (!r->device_ready)
    return GSTAMD_OK;
  if (hipDeviceSynchronize () != hipSuccess)
    return audio_hip_fail (__func__);
  if (new_table) {
    if (r->table_dev)
      (void) hipFree (r->table_dev);
    r->table_dev = nullptr;
    if (!r->plan.table.empty ()) {
      if (hipMalloc (&r->table_dev, r->plan.table.size ()) != hipSuccess)
        return audio_hip_fail (__func__);
      if (hipMemcpy (r->table_dev, r->plan.table.data (), r->plan.table.size (), hipMemcpyHostToDevice) != hipSuccess)
        return audio_hip_fail (__func__);
    }
  }
  if (shift.changed) {
    const size_t fbytes = (size_t) r->plan.bps * r->plan.channels;
    const size_t have = old_avail < r->hist_cap[r->cur] ? old_avail : r->hist_cap[r->cur];
    std::vector<uint8_t> h (have * fbytes);
    if (have && hipMemcpy (h.data (), r->hist[r->cur], h.size (), hipMemcpyDeviceToHost) != hipSuccess)
      return audio_hip_fail (__func__);
    audio_history_shift (shift, fbytes, &h);
    e = ensure_hist (r, r->cur, h.size () / fbytes + 64);
    if (e != GSTAMD_OK)
      return e;
    if (!h.empty () && hipMemcpy (r->hist[r->cur], h.data (), h.size (), hipMemcpyHostToDevice) != hipSuccess)
      return audio_hip_fail (__func__);
  }
  return GSTAMD_OK;
}

size_t gstamd_audio_resampler_get_out_frames (GstAmdAudioResampler *r, size_t in_frames)
{
  return r ? audio_get_out_frames (r->plan, r->st, in_frames) : 0;
}

size_t gstamd_audio_resampler_get_in_frames (GstAmdAudioResampler *r, size_t out_frames)
{
  return r ? audio_get_in_frames (r->plan, r->st, out_frames) : 0;
}

size_t gstamd_audio_resampler_get_max_latency (GstAmdAudioResampler *r)
{
  return r ? (size_t) (r->plan.n_taps / 2) : 0;
}

static int resample_strided (GstAmdAudioResampler *r, const void *in, size_t in_frames, void *out, size_t out_frames, long long in_stride,
    long long out_stride, void *stream)
{
  if (!r || (out_frames > 0 && !out))
    return GSTAMD_ERR_INVALID;
  std::lock_guard<std::mutex> g (r->lock);
  int e = ensure_device (r);
  if (e != GSTAMD_OK)
    return e;
  switch (r->plan.format) {
    case GSTAMD_AUDIO_FORMAT_S16: return run_resample<int16_t> (r, in, in_frames, out, out_frames, in_stride, out_stride, (hipStream_t) stream);
    case GSTAMD_AUDIO_FORMAT_S32: return run_resample<int32_t> (r, in, in_frames, out, out_frames, in_stride, out_stride, (hipStream_t) stream);
    case GSTAMD_AUDIO_FORMAT_F32: return run_resample<float> (r, in, in_frames, out, out_frames, in_stride, out_stride, (hipStream_t) stream);
    default: return run_resample<double> (r, in, in_frames, out, out_frames, in_stride, out_stride, (hipStream_t) stream);
  }
}

int gstamd_audio_resampler_resample (GstAmdAudioResampler *r, const void *in, size_t in_frames, void *out, size_t out_frames,
    void *stream)
{
  /* non-interleaved sides: the planes follow each other, in_frames / out_frames samples apart */
  return resample_strided (r, in, in_frames, out, out_frames, (long long) in_frames, (long long) out_frames, stream);
}

int gstamd_audio_resampler_resample_planes (GstAmdAudioResampler *r, const void *const in[], size_t in_frames, void *const out[],
    size_t out_frames, void *stream)
{
  if (!r || (out_frames > 0 && (!out || !out[0])))
    return GSTAMD_ERR_INVALID;
  const AudioPlan &pl = r->plan;
  /* plane c must sit at plane 0 + c * stride (what a GstBuffer's GstAudioMeta holds in practice); the stride is free */
  auto stride_of = [&](const void *const *pp, bool planar, long long *stride) {
    *stride = 0;
    if (!planar || pl.channels < 2 || !pp)
      return true;
    const ptrdiff_t d = (const uint8_t *) pp[1] - (const uint8_t *) pp[0];
    if (d <= 0 || d % pl.bps)
      return false;
    for (int c = 2; c < pl.channels; c++)
      if ((const uint8_t *) pp[c] - (const uint8_t *) pp[c - 1] != d)
        return false;
    *stride = d / pl.bps;
    return true;
  };
  long long is = 0, os = 0;
  if (!stride_of (in, pl.in_planar, &is) || !stride_of ((const void *const *) out, pl.out_planar, &os)) {
    gstamd_internal_set_error ("non-interleaved planes must be equally spaced in ascending order");
    return GSTAMD_ERR_UNSUPPORTED;
  }
  if (pl.in_planar && pl.channels < 2)
    is = (long long) in_frames;
  if (pl.out_planar && pl.channels < 2)
    os = (long long) out_frames;
  return resample_strided (r, in ? in[0] : nullptr, in_frames, out ? out[0] : nullptr, out_frames, is, os, stream);
}

/* N independent resamplers, one buffer each, in ONE kernel launch where they share a filter: the same sample type, channel count, layout flags,
 * rates and taps table (resamplers made with the same arguments), full filter mode, at most 2^31 frames of state each.  Results and the
 * resamplers' states are exactly those of n gstamd_audio_resampler_resample calls (which is also what happens, one by one, for a set that
 * does not qualify).  No reference counterpart: gst_audio_resampler_resample (audio-resampler.c:1750) takes one stream. */
int gstamd_audio_resampler_resample_many (int n, GstAmdAudioResampler *const *resamplers, const void *const *in, const size_t *in_frames, void *const *out,
    const size_t *out_frames, void *stream)
{
  if (n < 0 || (n > 0 && (!resamplers || !in_frames || !out || !out_frames)))
    return GSTAMD_ERR_INVALID;
  for (int i = 0; i < n; i++)
    if (!resamplers[i] || (out_frames[i] > 0 && !out[i]))
      return GSTAMD_ERR_INVALID;
  int done = 0;
  while (done < n) {
    /* the longest run from `done` on that one launch can serve */
    GstAmdAudioResampler *r0 = resamplers[done];
    int e = GSTAMD_OK;
    {
      std::lock_guard<std::mutex> g (r0->lock);
      e = ensure_device (r0);
    }
    if (e != GSTAMD_OK)
      return e;
    const AudioPlan &p0 = r0->plan;
    const bool full = p0.method != GSTAMD_AUDIO_RESAMPLER_METHOD_NEAREST && p0.in_rate != p0.out_rate && p0.filter_mode != GSTAMD_AUDIO_FILTER_MODE_INTERPOLATED &&
        !tuning_on ("GSTAMD_NO_FIR_LDS") && !tuning_on ("GSTAMD_NO_FIR_MANY");
    int run = 1;
    while (full && done + run < n && run < GSTAMD_AUDIO_MANY_MAX) {
      GstAmdAudioResampler *r = resamplers[done + run];
      bool dup = false;
      for (int k = 0; k < run; k++)
        dup = dup || resamplers[done + k] == r;          /* the same resampler twice: its second buffer depends on the first */
      if (dup)
        break;
      {
        std::lock_guard<std::mutex> g (r->lock);
        e = ensure_device (r);
      }
      if (e != GSTAMD_OK)
        return e;
      const AudioPlan &p = r->plan;
      if (p.format != p0.format || p.channels != p0.channels || p.in_rate != p0.in_rate || p.out_rate != p0.out_rate || p.n_taps != p0.n_taps ||
          p.taps_stride != p0.taps_stride || p.in_planar != p0.in_planar || p.out_planar != p0.out_planar || p.method != p0.method ||
          p.filter_mode != p0.filter_mode || r->table_hash != r0->table_hash)
        break;
      run++;
    }
    bool fits = full && run > 1;
    /* (a NULL `in` array or a NULL in[i] means silence, as in gstamd_audio_resampler_resample: such sets go one by one - the batched kernel reads in[i]) */
    for (int k = 0; fits && k < run; k++)
      fits = in && in[done + k] && in_frames[done + k] < (1u << 30) && out_frames[done + k] < (1u << 30);
    if (fits) {
      const size_t fbytes = (size_t) p0.bps;
      const size_t lds_need = ((size_t) FIR_LDS_FRAMES * (p0.taps_stride + 4) + (size_t) p0.channels * (((FIR_LDS_FRAMES * (p0.samp_inc + 1) + p0.taps_stride + 2 + 31) & ~31) + 16)) * fbytes +
          2 * FIR_LDS_FRAMES * sizeof (int);
      fits = lds_need <= 64 * 1024;
    }
    if (!fits) {
      e = resample_strided (r0, in ? in[done] : nullptr, in_frames[done], out[done], out_frames[done], (long long) in_frames[done], (long long) out_frames[done], stream);
      if (e != GSTAMD_OK)
        return e;
      done++;
      continue;
    }
    /* every resampler of the run under its own lock for the duration (ascending addresses: no two callers can wait for each other) */
    std::vector<GstAmdAudioResampler *> order (resamplers + done, resamplers + done + run);
    std::sort (order.begin (), order.end ());
    for (GstAmdAudioResampler *r : order)
      r->lock.lock ();
    switch (p0.format) {
      case GSTAMD_AUDIO_FORMAT_S16: e = run_many<int16_t> (run, resamplers + done, in + done, in_frames + done, out + done, out_frames + done, (hipStream_t) stream); break;
      case GSTAMD_AUDIO_FORMAT_S32: e = run_many<int32_t> (run, resamplers + done, in + done, in_frames + done, out + done, out_frames + done, (hipStream_t) stream); break;
      case GSTAMD_AUDIO_FORMAT_F32: e = run_many<float> (run, resamplers + done, in + done, in_frames + done, out + done, out_frames + done, (hipStream_t) stream); break;
      default: e = run_many<double> (run, resamplers + done, in + done, in_frames + done, out + done, out_frames + done, (hipStream_t) stream); break;
    }
    for (GstAmdAudioResampler *r : order)
      r->lock.unlock ();
    if (e != GSTAMD_OK)
      return e;
    done += run;
  }
  return GSTAMD_OK;
}

/* "" while the output is the reference's bit for bit; otherwise what differs and why (set by gstamd_audio_resampler_update, see audio_taps.cpp
 * audio_update; expires once the frames in question have left the filter window, and on reset) */
const char *gstamd_audio_resampler_divergence (GstAmdAudioResampler *r)
{
  if (!r)
    return "";
  std::lock_guard<std::mutex> g (r->lock);
  return r->st.stale_ahead > 0 ? r->divergence.c_str () : "";
}

int gstamd_audio_resampler_debug_get (GstAmdAudioResampler *r, int32_t *out, int max_out)
{
  if (!r)
    return -1;
  const int32_t v[] = {r->plan.n_taps, r->plan.n_phases, r->plan.in_rate, r->plan.oversample, r->plan.filter_mode,
    r->plan.filter_interpolation, r->plan.taps_stride, (int32_t) r->st.samples_avail, (int32_t) r->st.samp_phase, (int32_t) r->st.skip};
  const int n = (int) (sizeof (v) / sizeof (v[0]));
  for (int i = 0; i < n && i < max_out; i++)
    out[i] = v[i];
  return n;
}

long gstamd_audio_resampler_debug_taps (GstAmdAudioResampler *r, double *out, long max_out)
{
  if (!r)
    return -1;
  const AudioPlan &p = r->plan;
  const long n = (long) p.n_phases * p.n_taps;
  if (!out)
    return n;
  long k = 0;
  for (int ph = 0; ph < p.n_phases; ph++)
    for (int t = 0; t < p.n_taps && k < max_out; t++, k++) {
      const size_t idx = (size_t) ph * p.taps_stride + t;
      switch (p.format) {
        case GSTAMD_AUDIO_FORMAT_S16: out[k] = ((const int16_t *) p.table.data ())[idx]; break;
        case GSTAMD_AUDIO_FORMAT_S32: out[k] = ((const int32_t *) p.table.data ())[idx]; break;
        case GSTAMD_AUDIO_FORMAT_F32: out[k] = ((const float *) p.table.data ())[idx]; break;
        default: out[k] = ((const double *) p.table.data ())[idx]; break;
      }
    }
  return n;
}

}  // extern "C"
