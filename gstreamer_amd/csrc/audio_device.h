// audio_device.h - device body of the polyphase FIR (shared with the host emulator, see video_device.h).
//
// Reference semantics (subprojects/gst-plugins-base/gst-libs/gst/audio/):
//   per-output index/phase walk    audio-resampler.c:475-486 (get_taps_*: samp_index += samp_inc, samp_phase += samp_frac, carry)
//   inner_product_<T>_full_1_c     audio-resampler.c:636-657 (int), :693-707 (float): 4 interleaved partial sums,
//                                  ((r0 + r1) + r2) + r3, int: + (1 << (prec-1)) >> prec, clamp
//   inner_product_<T>_nearest_1_c  audio-resampler.c:606-615
//   resample loop / history        audio-resampler-macros.h:62-100, audio-resampler.c:879-897, 1750-1806
//
// fir_out_index: where output frame j of channel c goes (interleaved, or plane c of a non-interleaved buffer).
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
  int interp;               // 0: FULL table [out_rate][n_taps]; 1 / 2: INTERPOLATED mode, linear / cubic blend of the oversampled table per output
  int oversample;
  long long in_plane_stride;   // 0: interleaved input; else channel c of the new input starts at in + c * in_plane_stride (samples)
  long long out_plane_stride;  // same for the output (GST_AUDIO_RESAMPLER_FLAG_NON_INTERLEAVED_IN / _OUT, audio-resampler.h:177-182)
};

GSTAMD_AD long long fir_out_index (const FirParams &p, long long j, int c)
{
  return p.out_plane_stride ? (long long) c * p.out_plane_stride + j : j * p.channels + c;
}

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
  return p.in_plane_stride ? in[(long long) c * p.in_plane_stride + (idx - p.hist_frames)] : in[(idx - p.hist_frames) * p.channels + c];
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

// make_coeff_<type>_linear / _cubic (audio-resampler.c:321-373): blend weights of the oversampled rows for frac = num / denom
template <typename T> struct FirPrec;
template <> struct FirPrec<int16_t> { static const int prec = 15; };
template <> struct FirPrec<int32_t> { static const int prec = 31; };

GSTAMD_AD void fir_coeff_linear (int num, int denom, float *ic) { const float x = (float) num / denom; ic[0] = ic[2] = x; ic[1] = ic[3] = 1.0f - x; }
GSTAMD_AD void fir_coeff_linear (int num, int denom, double *ic) { const double x = (double) num / denom; ic[0] = ic[2] = x; ic[1] = ic[3] = 1.0 - x; }
GSTAMD_AD void fir_coeff_linear (int num, int denom, int16_t *ic)
{
  const int16_t x = (int16_t) (((int64_t) num << 15) / denom);
  ic[0] = ic[2] = x;
  ic[1] = ic[3] = (int16_t) ((int16_t) (((int32_t) 1 << 15) - 1) - x);
}
GSTAMD_AD void fir_coeff_linear (int num, int denom, int32_t *ic)
{
  const int32_t x = (int32_t) (((int64_t) num << 31) / denom);
  ic[0] = ic[2] = x;
  ic[1] = ic[3] = (int32_t) ((int32_t) (((int64_t) 1 << 31) - 1) - x);
}
GSTAMD_AD void fir_coeff_cubic (int num, int denom, float *ic)
{
  const float x = (float) num / denom, x2 = x * x, x3 = x2 * x;
  ic[0] = 0.16667f * (x3 - x);
  ic[1] = x + 0.5f * (x2 - x3);
  ic[3] = -0.33333f * x + 0.5f * x2 - 0.16667f * x3;
  ic[2] = (float) 1.0 - ic[0] - ic[1] - ic[3];
}
GSTAMD_AD void fir_coeff_cubic (int num, int denom, double *ic)
{
  const double x = (double) num / denom, x2 = x * x, x3 = x2 * x;
  ic[0] = 0.16667f * (x3 - x);
  ic[1] = x + 0.5f * (x2 - x3);
  ic[3] = -0.33333f * x + 0.5f * x2 - 0.16667f * x3;
  ic[2] = (double) 1.0 - ic[0] - ic[1] - ic[3];
}
template <typename T, typename T2, int prec>
GSTAMD_AD void fir_coeff_cubic_int (int num, int denom, T *ic)
{
  const T2 one = ((T2) 1 << prec) - 1;
  const T2 x = (T2) (((int64_t) num << prec) / denom);
  const T2 x2 = (x * x) >> prec;
  const T2 x3 = (x2 * x) >> prec;
  ic[0] = (T) ((((x3 - x) << prec) / 6) >> prec);
  ic[1] = (T) (x + ((x2 - x3) >> 1));
  ic[3] = (T) (-(((x << prec) / 3) >> prec) + (x2 >> 1) - (((x3 << prec) / 6) >> prec));
  ic[2] = (T) (one - ic[0] - ic[1] - ic[3]);
}
GSTAMD_AD void fir_coeff_cubic (int num, int denom, int16_t *ic) { fir_coeff_cubic_int<int16_t, int32_t, 15> (num, denom, ic); }
GSTAMD_AD void fir_coeff_cubic (int num, int denom, int32_t *ic) { fir_coeff_cubic_int<int32_t, int64_t, 31> (num, denom, ic); }

// inner_product_<T>_linear_1_c / _cubic_1_c (audio-resampler.c:636-687 int, 709-755 float): the input window against 2 / 4
// neighbouring rows of the oversampled table, the row sums blended with ic[]
template <typename T>
GSTAMD_AD T fir_interp_finish_linear (typename Acc<T>::type r0, typename Acc<T>::type r1, typename Acc<T>::type r2, typename Acc<T>::type r3, const T *ic);
template <> GSTAMD_AD float fir_interp_finish_linear<float> (float r0, float r1, float r2, float r3, const float *ic)
{
  r0 += r2;
  r1 += r3;
  return (r0 - r1) * ic[0] + r1;
}
template <> GSTAMD_AD double fir_interp_finish_linear<double> (double r0, double r1, double r2, double r3, const double *ic)
{
  r0 += r2;
  r1 += r3;
  return (r0 - r1) * ic[0] + r1;
}
template <typename T, typename T2, int prec>
GSTAMD_AD T fir_interp_finish_linear_int (T2 r0, T2 r1, T2 r2, T2 r3, const T *ic)
{
  const T2 c0 = ic[0], lim = (T2) 1 << prec;
  r0 = (r0 + r2) >> prec;
  r1 = (r1 + r3) >> prec;
  r0 = ((T2) (T) r0 - (T2) (T) r1) * c0 + ((T2) (T) r1 << prec);
  r0 = (r0 + ((T2) 1 << (prec - 1))) >> prec;
  return (T) (r0 < -lim ? -lim : (r0 > lim - 1 ? lim - 1 : r0));
}
template <> GSTAMD_AD int16_t fir_interp_finish_linear<int16_t> (int32_t r0, int32_t r1, int32_t r2, int32_t r3, const int16_t *ic)
{
  return fir_interp_finish_linear_int<int16_t, int32_t, 15> (r0, r1, r2, r3, ic);
}
template <> GSTAMD_AD int32_t fir_interp_finish_linear<int32_t> (int64_t r0, int64_t r1, int64_t r2, int64_t r3, const int32_t *ic)
{
  return fir_interp_finish_linear_int<int32_t, int64_t, 31> (r0, r1, r2, r3, ic);
}

template <typename T>
GSTAMD_AD T fir_interp_finish_cubic (typename Acc<T>::type r0, typename Acc<T>::type r1, typename Acc<T>::type r2, typename Acc<T>::type r3, const T *ic);
template <> GSTAMD_AD float fir_interp_finish_cubic<float> (float r0, float r1, float r2, float r3, const float *ic)
{
  return r0 * ic[0] + r1 * ic[1] + r2 * ic[2] + r3 * ic[3];
}
template <> GSTAMD_AD double fir_interp_finish_cubic<double> (double r0, double r1, double r2, double r3, const double *ic)
{
  return r0 * ic[0] + r1 * ic[1] + r2 * ic[2] + r3 * ic[3];
}
template <typename T, typename T2, int prec>
GSTAMD_AD T fir_interp_finish_cubic_int (T2 r0, T2 r1, T2 r2, T2 r3, const T *ic)
{
  const T2 lim = (T2) 1 << prec;
  T2 r = (T2) (T) (r0 >> prec) * (T2) ic[0] + (T2) (T) (r1 >> prec) * (T2) ic[1] + (T2) (T) (r2 >> prec) * (T2) ic[2] +
      (T2) (T) (r3 >> prec) * (T2) ic[3];
  r = (r + ((T2) 1 << (prec - 1))) >> prec;
  return (T) (r < -lim ? -lim : (r > lim - 1 ? lim - 1 : r));
}
template <> GSTAMD_AD int16_t fir_interp_finish_cubic<int16_t> (int32_t r0, int32_t r1, int32_t r2, int32_t r3, const int16_t *ic)
{
  return fir_interp_finish_cubic_int<int16_t, int32_t, 15> (r0, r1, r2, r3, ic);
}
template <> GSTAMD_AD int32_t fir_interp_finish_cubic<int32_t> (int64_t r0, int64_t r1, int64_t r2, int64_t r3, const int32_t *ic)
{
  return fir_interp_finish_cubic_int<int32_t, int64_t, 31> (r0, r1, r2, r3, ic);
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
  if (p.interp) {
    /* get_taps_<T>_<inter> (audio-resampler.c:567-590): rows `offset` .. of the oversampled table, blend weights from frac */
    const int pos = phase * p.oversample;
    const int offset = (p.oversample - 1) - pos / p.out_rate, frac = pos % p.out_rate;
    const T *__restrict__ c0 = table + (size_t) offset * p.n_taps_padded, *__restrict__ c1 = c0 + p.n_taps_padded;
    T ic[4];
    A r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    if (p.interp == 1) {
      fir_coeff_linear (frac, p.out_rate, ic);
      for (int i = 0; i < p.n_taps_padded; i += 2) {
        const A a0 = (A) logical_sample<T> (p, hist, in, idx + i, c), a1 = (A) logical_sample<T> (p, hist, in, idx + i + 1, c);
        r0 += a0 * (A) c0[i];
        r1 += a0 * (A) c1[i];
        r2 += a1 * (A) c0[i + 1];
        r3 += a1 * (A) c1[i + 1];
      }
      return fir_interp_finish_linear<T> (r0, r1, r2, r3, ic);
    }
    const T *__restrict__ c2 = c1 + p.n_taps_padded, *__restrict__ c3 = c2 + p.n_taps_padded;
    fir_coeff_cubic (frac, p.out_rate, ic);
    for (int i = 0; i < p.n_taps_padded; i++) {
      const A a = (A) logical_sample<T> (p, hist, in, idx + i, c);
      r0 += a * (A) c0[i];
      r1 += a * (A) c1[i];
      r2 += a * (A) c2[i];
      r3 += a * (A) c3[i];
    }
    return fir_interp_finish_cubic<T> (r0, r1, r2, r3, ic);
  }
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

// ------------------------------------------------------------------------------------------------
// LDS-staged form of the FULL-table inner product (k_fir_lds).  A workgroup of 256 lanes owns FIR_LDS_FRAMES consecutive output
// frames: it stages (a) the logical input window under them, deinterleaved, [channel][frame] - the history / input / silence
// decision of logical_sample is taken once per sample instead of once per tap - and (b) the taps row of every frame's phase
// (rows padded to n_taps + 4 words: consecutive frames' rows then start 12 banks apart and the lanes' tap reads are
// conflict-free).  Four lanes share one output frame: lane q accumulates the taps i = q (mod 4) - exactly the partial sum
// res[q] of inner_product_<T>_full_1_c (audio-resampler.c:693-707) in the same order, a * b rounded then added - and the four
// partial sums are combined ((r0 + r1) + r2) + r3 by every lane of the quad.
// ------------------------------------------------------------------------------------------------
#define FIR_LDS_FRAMES 64

struct FirLdsGeom {
  int row_stride;       // words per staged taps row: n_taps_padded + 4
  int win_frames;       // staged frames per channel (multiple of 32, + 16 so that channels start 16 banks apart)
};

GSTAMD_AD void fir_position (const FirParams &p, long long j, long long *idx, int *phase)
{
  const long long t = (long long) p.samp_phase0 + j * (long long) p.samp_frac;
  *idx = p.samp_index0 + j * (long long) p.samp_inc + t / p.out_rate;
  *phase = (int) (t % p.out_rate);
}

// frames of the logical stream the outputs [jb, jb + nj) touch: [*lo, *lo + *span)
GSTAMD_AD void fir_lds_span (const FirParams &p, long long jb, int nj, long long *lo, int *span)
{
  long long a, b;
  int ph;
  fir_position (p, jb, &a, &ph);
  fir_position (p, jb + nj - 1, &b, &ph);
  *lo = a;
  *span = (int) (b - a) + p.n_taps_padded;
}

// position table of the workgroup's frames, filled once (the 64-bit division of fir_position is paid once per frame, not per tap):
// pos[2 fr] = first window sample relative to the staged window, pos[2 fr + 1] = phase
GSTAMD_AD void fir_lds_positions (const FirParams &p, long long jb, int nj, int *pos, int tid, int nthreads)
{
  long long lo, idx;
  int span, phase;
  fir_lds_span (p, jb, nj, &lo, &span);
  for (int fr = tid; fr < nj; fr += nthreads) {
    fir_position (p, jb + fr, &idx, &phase);
    pos[2 * fr] = (int) (idx - lo);
    pos[2 * fr + 1] = phase;
  }
}

// staging, thread tid of nthreads (after fir_lds_positions and a barrier): taps rows of the frames' phases, four taps (one 16-byte
// piece for float) per step, and the input window
template <typename T>
GSTAMD_AD void fir_lds_stage (const FirParams &p, const FirLdsGeom &g, const T *__restrict__ hist, const T *__restrict__ in,
    const T *__restrict__ table, long long jb, int nj, const int *pos, T *rows, T *win, int tid, int nthreads)
{
  struct __attribute__ ((aligned (4 * sizeof (T) > 16 ? 16 : 4 * sizeof (T)))) Q4 { T v[4]; };
  const int n4 = p.n_taps_padded >> 2;
  for (int k = tid; k < nj * n4; k += nthreads) {
    const int fr = k / n4, i4 = k - fr * n4;
    *(Q4 *) (rows + fr * g.row_stride + 4 * i4) = *(const Q4 *) (table + (size_t) pos[2 * fr + 1] * p.n_taps_padded + 4 * i4);
  }
  long long lo;
  int span;
  fir_lds_span (p, jb, nj, &lo, &span);
  const int C = p.channels;
  for (int k = tid; k < span * C; k += nthreads) {
    const int f = k / C, c = k - f * C;          /* interleaved order: neighbouring lanes read neighbouring samples */
    win[c * g.win_frames + f] = logical_sample<T> (p, hist, in, lo + f, c);
  }
}

// partial sum q of output frame fr of the workgroup, channel c
template <typename T>
GSTAMD_AD typename Acc<T>::type fir_lds_partial (const FirParams &p, const FirLdsGeom &g, const int *pos, const T *rows, const T *win, int fr,
    int q, int c)
{
  typedef typename Acc<T>::type A;
  const T *row = rows + fr * g.row_stride + q, *w = win + c * g.win_frames + pos[2 * fr] + q;
  A r = 0;
  for (int i = 0; i < p.n_taps_padded; i += 4)
    r += (A) w[i] * (A) row[i];
  return r;
}

template <typename T>
GSTAMD_AD T fir_lds_combine (typename Acc<T>::type r0, typename Acc<T>::type r1, typename Acc<T>::type r2, typename Acc<T>::type r3)
{
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
