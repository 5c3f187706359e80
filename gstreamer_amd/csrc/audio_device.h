// audio_device.h - device body of the polyphase FIR (shared with the host emulator, see video_device.h).
//
// Reference semantics (subprojects/gst-plugins-base/gst-libs/gst/audio/):
//   per-output index/phase walk    audio-resampler.c:475-486 (get_taps_*: samp_index += samp_inc, samp_phase += samp_frac, carry)
//   inner_product_<T>_full_1_c     audio-resampler.c:636-657 (int), :693-707 (float): 4 interleaved partial sums,
//                                  ((r0 + r1) + r2) + r3, int: + (1 << (prec-1)) >> prec, clamp
//   inner_product_<T>_nearest_1_c  audio-resampler.c:606-615
//   resample loop / history        audio-resampler-macros.h:62-100, audio-resampler.c:879-897, 1750-1806
//
// "Logical" input of channel c: the retained history frames followed by the new input frames.
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifdef __HIPCC__
#define GSTAMD_AD __device__ __forceinline__
#else
#define GSTAMD_AD inline
#endif

namespace gstamd {

struct FirParams {
  int channels;
  int n_taps_padded;        // multiple of 4
  int nearest;              // 1: out = in[samp_index] (method nearest or equal rates)
  int samp_inc, samp_frac, out_rate;
  long long samp_index0;    // index of output 0 in the logical stream (includes skip)
  int samp_phase0;
  long long hist_frames;    // frames in the history buffer
  long long total_frames;   // hist_frames + in_frames
  int in_is_null;           // new input is silence
};

template <typename T> struct Acc;
template <> struct Acc<int16_t> { typedef int32_t type; };
template <> struct Acc<int32_t> { typedef int64_t type; };
template <> struct Acc<float> { typedef float type; };
template <> struct Acc<double> { typedef double type; };

template <typename T>
GSTAMD_AD T logical_sample (const FirParams &p, const T *__restrict__ hist, const T *__restrict__ in, long long idx, int c)
{
  if (idx < p.hist_frames)
    return hist[idx * p.channels + c];
  if (idx >= p.total_frames || p.in_is_null)
    return (T) 0;
  return in[(idx - p.hist_frames) * p.channels + c];
}

template <typename T> GSTAMD_AD T fir_finish (typename Acc<T>::type r);
template <> GSTAMD_AD float fir_finish<float> (float r) { return r; }
template <> GSTAMD_AD double fir_finish<double> (double r) { return r; }
template <> GSTAMD_AD int16_t fir_finish<int16_t> (int32_t r)
{
  r = (r + ((int32_t) 1 << 14)) >> 15;
  return (int16_t) (r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
}
template <> GSTAMD_AD int32_t fir_finish<int32_t> (int64_t r)
{
  r = (r + ((int64_t) 1 << 30)) >> 31;
  const int64_t lim = (int64_t) 1 << 31;
  return (int32_t) (r < -lim ? -lim : (r > lim - 1 ? lim - 1 : r));
}

// output frame j, channel c
template <typename T>
GSTAMD_AD T fir_output (const FirParams &p, const T *__restrict__ hist, const T *__restrict__ in, const T *__restrict__ table,
    long long j, int c)
{
  typedef typename Acc<T>::type A;
  const long long t = (long long) p.samp_phase0 + j * (long long) p.samp_frac;
  const long long idx = p.samp_index0 + j * (long long) p.samp_inc + t / p.out_rate;
  const int phase = (int) (t % p.out_rate);
  if (p.nearest)
    return logical_sample<T> (p, hist, in, idx, c);
  const T *__restrict__ taps = table + (size_t) phase * p.n_taps_padded;
  A r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  for (int i = 0; i < p.n_taps_padded; i += 4) {
    r0 += (A) logical_sample<T> (p, hist, in, idx + i + 0, c) * (A) taps[i + 0];
    r1 += (A) logical_sample<T> (p, hist, in, idx + i + 1, c) * (A) taps[i + 1];
    r2 += (A) logical_sample<T> (p, hist, in, idx + i + 2, c) * (A) taps[i + 2];
    r3 += (A) logical_sample<T> (p, hist, in, idx + i + 3, c) * (A) taps[i + 3];
  }
  return fir_finish<T> (r0 + r1 + r2 + r3);
}

// new history: first `keep` frames are logical[src_start + i] for i < moved, else the OLD history frame i
// (memmove semantics of audio-resampler-macros.h:94-96 when fewer frames are moved than are kept)
template <typename T>
GSTAMD_AD T history_sample (const FirParams &p, const T *__restrict__ hist, const T *__restrict__ in, long long src_start,
    long long moved, long long i, int c)
{
  if (i < moved)
    return logical_sample<T> (p, hist, in, src_start + i, c);
  return logical_sample<T> (p, hist, in, i, c);
}

}  // namespace gstamd
