// audio_taps.cpp - host-side set-up of the polyphase FIR: rate reduction, Kaiser/cubic/... parameters and
// the per-phase taps table, following the reference's double/float arithmetic step by step
// (subprojects/gst-plugins-base/gst-libs/gst/audio/audio-resampler.c; lines cited per function).
// Compiled with -ffp-contract=off: every double/float operation rounds exactly as in the C reference.
#include "audio_taps.h"

#include <algorithm>
#include <cmath>
#include <cstring>

#include "ooura_i0.h"

namespace gstamd {

// quality maps (audio-resampler.c:49-88)
static const int kOversampleQualities[] = {4, 4, 4, 8, 8, 16, 16, 16, 16, 32, 32};
struct KaiserQuality { double cutoff, downsample_cutoff_factor, stopband_attenuation, transition_bandwidth; };
static const KaiserQuality kKaiserQualities[] = {
  {0.860, 0.96511, 60, 0.7}, {0.880, 0.96591, 65, 0.29}, {0.910, 0.96923, 70, 0.145}, {0.920, 0.97600, 80, 0.105},
  {0.940, 0.97979, 85, 0.087}, {0.940, 0.98085, 95, 0.077}, {0.945, 0.99471, 100, 0.068}, {0.950, 1.0, 105, 0.055},
  {0.960, 1.0, 110, 0.045}, {0.968, 1.0, 115, 0.039}, {0.975, 1.0, 120, 0.0305}
};
struct BlackmanQuality { int n_taps; double cutoff; };
static const BlackmanQuality kBlackmanQualities[] = {
  {8, 0.5}, {16, 0.6}, {24, 0.72}, {32, 0.8}, {48, 0.85}, {64, 0.90}, {80, 0.92}, {96, 0.933}, {128, 0.950},
  {148, 0.955}, {160, 0.960}
};
enum { DEFAULT_QUALITY = 4 };

static inline bool is_set (double v) { return !std::isnan (v); }
static inline bool is_set (int v) { return v >= 0; }

void audio_options_init (GstAmdAudioResamplerOptions *o)
{
  memset (o, 0, sizeof (*o));
  o->cutoff = o->stop_attenuation = o->transition_bandwidth = o->cubic_b = o->cubic_c = o->max_phase_error = NAN;
  o->n_taps = o->filter_mode = o->filter_mode_threshold = o->filter_interpolation = o->filter_oversample = -1;
}

/* gst_audio_resampler_options_set_quality (audio-resampler.c:1270-1328) */
void audio_options_set_quality (int method, unsigned quality, int in_rate, int out_rate, GstAmdAudioResamplerOptions *o)
{
  if (!o || quality > 10 || in_rate <= 0 || out_rate <= 0)
    return;
  switch (method) {
    case GSTAMD_AUDIO_RESAMPLER_METHOD_LINEAR:
      o->n_taps = 2;
      break;
    case GSTAMD_AUDIO_RESAMPLER_METHOD_CUBIC:
      o->n_taps = 4;
      o->cubic_b = 1.0;
      o->cubic_c = 0.0;
      break;
    case GSTAMD_AUDIO_RESAMPLER_METHOD_BLACKMAN_NUTTALL:
      o->n_taps = kBlackmanQualities[quality].n_taps;
      o->cutoff = kBlackmanQualities[quality].cutoff;
      break;
    case GSTAMD_AUDIO_RESAMPLER_METHOD_KAISER: {
      const KaiserQuality &m = kKaiserQualities[quality];
      double cutoff = m.cutoff;
      if (out_rate < in_rate)
        cutoff *= m.downsample_cutoff_factor;
      o->cutoff = cutoff;
      o->stop_attenuation = m.stopband_attenuation;
      o->transition_bandwidth = m.transition_bandwidth;
      break;
    }
    default:
      break;
  }
  o->filter_oversample = kOversampleQualities[quality];
}

static int gcd_int (int a, int b)
{
  while (b != 0) {
    int t = a;
    a = b;
    b = t % b;
  }
  return a < 0 ? -a : a;
}

// ---- tap functions (audio-resampler.c:168-218) --------------------------------------------------------
static inline double linear_tap (double x, int n_taps) { return ((n_taps + 1) / 2 * 2) / 2 - fabs (x); }

static inline double cubic_tap (double x, int n_taps, double b, double c)
{
  double a = fabs (x * 4.0) / n_taps, a2 = a * a, a3 = a2 * a;
  if (a <= 1.0)
    return ((12.0 - 9.0 * b - 6.0 * c) * a3 + (-18.0 + 12.0 * b + 6.0 * c) * a2 + (6.0 - 2.0 * b)) / 6.0;
  else if (a <= 2.0)
    return ((-b - 6.0 * c) * a3 + (6.0 * b + 30.0 * c) * a2 + (-12.0 * b - 48.0 * c) * a + (8.0 * b + 24.0 * c)) / 6.0;
  return 0.0;
}

static inline double blackman_nuttall_tap (double x, int n_taps, double Fc)
{
  double y = M_PI * x;
  double s = (y == 0.0 ? Fc : sin (y * Fc) / y);
  double w = 2.0 * y / n_taps + M_PI;
  return s * (0.3635819 - 0.4891775 * cos (w) + 0.1365995 * cos (2 * w) - 0.0106411 * cos (3 * w));
}

static inline double kaiser_tap (double x, int n_taps, double Fc, double beta)
{
  double y = M_PI * x;
  double s = (y == 0.0 ? Fc : sin (y * Fc) / y);
  double w = 2.0 * x / n_taps;
  double v = 1 - w * w;
  return s * bessel_i0 (beta * sqrt (v > 0 ? v : 0));
}

// ---- per-type conversions (audio-resampler.c:220-281, 325-375, 377-463) ---------------------------------
template <typename T> struct Traits;
template <> struct Traits<int16_t> { typedef int32_t T2; static const int prec = 15; static const bool is_int = true; };
template <> struct Traits<int32_t> { typedef int64_t T2; static const int prec = 31; static const bool is_int = true; };
template <> struct Traits<float> { typedef float T2; static const int prec = 0; static const bool is_int = false; };
template <> struct Traits<double> { typedef double T2; static const int prec = 0; static const bool is_int = false; };

template <typename T>
static void convert_taps (const double *tmp, T *t, double weight, int n_taps)
{
  if constexpr (!Traits<T>::is_int) {
    for (int i = 0; i < n_taps; i++)
      t[i] = (T) (tmp[i] / weight);
    return;
  } else {
  const int64_t one = (1LL << Traits<T>::prec) - 1;
  const double multiplier = (double) one;
  double l_offset = 0.0, h_offset = 1.0, offset = 0.5;
  for (int i = 0; i < 32; i++) {
    int64_t sum = 0;
    for (int j = 0; j < n_taps; j++)
      sum += (int64_t) floor (offset + tmp[j] * multiplier / weight);
    if (sum == one)
      break;
    if (l_offset == h_offset)
      break;
    if (sum < one) {
      if (offset > l_offset)
        l_offset = offset;
      offset += (h_offset - l_offset) / 2;
    } else {
      if (offset < h_offset)
        h_offset = offset;
      offset -= (h_offset - l_offset) / 2;
    }
  }
  for (int j = 0; j < n_taps; j++)
    t[j] = (T) floor (offset + tmp[j] * multiplier / weight);
  }
}

template <typename T>
static void make_taps (const AudioPlan &p, std::vector<double> &tmp, T *res, double x, int n_taps)
{
  double weight = 0.0;
  switch (p.method) {
    case GSTAMD_AUDIO_RESAMPLER_METHOD_LINEAR:
      for (int i = 0; i < n_taps; i++)
        weight += tmp[i] = linear_tap (x + i, p.n_taps);
      break;
    case GSTAMD_AUDIO_RESAMPLER_METHOD_CUBIC:
      for (int i = 0; i < n_taps; i++)
        weight += tmp[i] = cubic_tap (x + i, p.n_taps, p.b, p.c);
      break;
    case GSTAMD_AUDIO_RESAMPLER_METHOD_BLACKMAN_NUTTALL:
      for (int i = 0; i < n_taps; i++)
        weight += tmp[i] = blackman_nuttall_tap (x + i, p.n_taps, p.cutoff);
      break;
    case GSTAMD_AUDIO_RESAMPLER_METHOD_KAISER:
      for (int i = 0; i < n_taps; i++)
        weight += tmp[i] = kaiser_tap (x + i, p.n_taps, p.cutoff, p.kaiser_beta);
      break;
    default:
      return;
  }
  convert_taps<T> (tmp.data (), res, weight, n_taps);
}

template <typename T>
static void make_coeff_linear (int num, int denom, T *ic)
{
  typedef typename Traits<T>::T2 T2;
  if constexpr (Traits<T>::is_int) {
    T x = (T) (((int64_t) num << Traits<T>::prec) / denom);
    ic[0] = ic[2] = x;
    ic[1] = ic[3] = (T) ((T) ((((T2) 1) << Traits<T>::prec) - 1) - x);
  } else {
    T x = (T) num / denom;
    ic[0] = ic[2] = x;
    ic[1] = ic[3] = (T) 1.0 - x;
  }
}

template <typename T>
static void make_coeff_cubic (int num, int denom, T *ic);

template <>
void make_coeff_cubic<float> (int num, int denom, float *ic)
{
  float x = (float) num / denom, x2 = x * x, x3 = x2 * x;
  ic[0] = 0.16667f * (x3 - x);
  ic[1] = x + 0.5f * (x2 - x3);
  ic[3] = -0.33333f * x + 0.5f * x2 - 0.16667f * x3;
  ic[2] = (float) 1.0 - ic[0] - ic[1] - ic[3];
}

template <>
void make_coeff_cubic<double> (int num, int denom, double *ic)
{
  double x = (double) num / denom, x2 = x * x, x3 = x2 * x;
  ic[0] = 0.16667f * (x3 - x);
  ic[1] = x + 0.5f * (x2 - x3);
  ic[3] = -0.33333f * x + 0.5f * x2 - 0.16667f * x3;
  ic[2] = (double) 1.0 - ic[0] - ic[1] - ic[3];
}

template <typename T, typename T2, int prec>
static void make_coeff_cubic_int (int num, int denom, T *ic)
{
  T2 one = ((T2) 1 << prec) - 1;
  T2 x = (T2) (((int64_t) num << prec) / denom);
  T2 x2 = (x * x) >> prec;
  T2 x3 = (x2 * x) >> prec;
  ic[0] = (T) ((((x3 - x) << prec) / 6) >> prec);
  ic[1] = (T) (x + ((x2 - x3) >> 1));
  ic[3] = (T) (-(((x << prec) / 3) >> prec) + (x2 >> 1) - (((x3 << prec) / 6) >> prec));
  ic[2] = (T) (one - ic[0] - ic[1] - ic[3]);
}

template <>
void make_coeff_cubic<int16_t> (int num, int denom, int16_t *ic) { make_coeff_cubic_int<int16_t, int32_t, 15> (num, denom, ic); }
template <>
void make_coeff_cubic<int32_t> (int num, int denom, int32_t *ic) { make_coeff_cubic_int<int32_t, int64_t, 31> (num, denom, ic); }

/* interpolate_<type>_linear_c / _cubic_c (audio-resampler.c:377-463) over rows `stride` elements apart */
template <typename T>
static void interpolate_rows (T *o, const T *a, int len, const T *ic, int stride, bool cubic)
{
  typedef typename Traits<T>::T2 T2;
  const int prec = Traits<T>::prec;
  if (!cubic) {
    const T *c0 = a, *c1 = a + stride;
    if constexpr (Traits<T>::is_int) {
      for (int i = 0; i < len; i++) {
        T2 tmp = ((T2) c0[i] - (T2) c1[i]) * (T2) ic[0] + (((T2) c1[i]) << prec);
        o[i] = (T) ((tmp + ((T2) 1 << (prec - 1))) >> prec);
      }
    } else {
      for (int i = 0; i < len; i++)
        o[i] = (c0[i] - c1[i]) * ic[0] + c1[i];
    }
    return;
  }
  const T *c0 = a, *c1 = a + stride, *c2 = a + 2 * stride, *c3 = a + 3 * stride;
  if constexpr (Traits<T>::is_int) {
    const T2 lim = (T2) 1 << prec;
    for (int i = 0; i < len; i++) {
      T2 tmp = (T2) c0[i] * (T2) ic[0] + (T2) c1[i] * (T2) ic[1] + (T2) c2[i] * (T2) ic[2] + (T2) c3[i] * (T2) ic[3];
      tmp = (tmp + ((T2) 1 << (prec - 1))) >> prec;
      o[i] = (T) (tmp < -lim ? -lim : (tmp > lim - 1 ? lim - 1 : tmp));
    }
  } else {
    for (int i = 0; i < len; i++)
      o[i] = c0[i] * ic[0] + c1[i] * ic[1] + c2[i] * ic[2] + c3[i] * ic[3];
  }
}

template <typename T>
static void build_table (AudioPlan &p)
{
  const int n_taps = p.n_taps, n_phases = p.n_phases;
  p.taps_stride = (n_taps + 3) / 4 * 4;      /* inner products run in blocks of 4 over zero-padded rows */
  p.table.assign ((size_t) n_phases * p.taps_stride * sizeof (T), 0);
  T *table = (T *) p.table.data ();
  std::vector<double> tmp (n_taps);
  if (p.method == GSTAMD_AUDIO_RESAMPLER_METHOD_NEAREST)
    return;
  if (p.filter_mode == GSTAMD_AUDIO_FILTER_MODE_INTERPOLATED) {
    /* INTERPOLATED mode: the device blends per output sample; the table is the oversampled main table itself
     * (resampler_calculate_taps :1176-1206), rows of taps_stride elements */
    const bool cubic = p.filter_interpolation == GSTAMD_AUDIO_FILTER_INTERPOLATION_CUBIC;
    const int rows = p.oversample + (cubic ? 4 : 2);
    p.table.assign ((size_t) rows * p.taps_stride * sizeof (T), 0);
    table = (T *) p.table.data ();
    for (int i = 0; i < rows; i++) {
      double x = -(n_taps / 2) + i / (double) p.oversample;
      make_taps<T> (p, tmp, table + (size_t) i * p.taps_stride, x, n_taps);
    }
    return;
  }
  if (p.filter_interpolation == GSTAMD_AUDIO_FILTER_INTERPOLATION_NONE) {
    /* get_taps_*_full, INTERPOLATION_NONE branch (:503-525) */
    for (int phase = 0; phase < n_phases; phase++) {
      double x = 1.0 - n_taps / 2 - (double) phase / n_phases;
      make_taps<T> (p, tmp, table + (size_t) phase * p.taps_stride, x, n_taps);
    }
    return;
  }
  /* oversampled main table (resampler_calculate_taps :1176-1206) then per-phase interpolation (:526-552) */
  const bool cubic = p.filter_interpolation == GSTAMD_AUDIO_FILTER_INTERPOLATION_CUBIC;
  const int isize = cubic ? 4 : 2, oversample = p.oversample, rows = oversample + isize;
  std::vector<T> main_tab ((size_t) rows * n_taps);
  for (int i = 0; i < rows; i++) {
    double x = -(n_taps / 2) + i / (double) oversample;
    make_taps<T> (p, tmp, &main_tab[(size_t) i * n_taps], x, n_taps);
  }
  for (int phase = 0; phase < n_phases; phase++) {
    const int pos = phase * oversample;
    const int offset = (oversample - 1) - pos / n_phases, frac = pos % n_phases;
    T ic[4];
    if (cubic)
      make_coeff_cubic<T> (frac, n_phases, ic);
    else
      make_coeff_linear<T> (frac, n_phases, ic);
    interpolate_rows<T> (table + (size_t) phase * p.taps_stride, &main_tab[(size_t) offset * n_taps], n_taps, ic, n_taps, cubic);
  }
}

static void rebuild_table (AudioPlan *p)
{
  switch (p->format) {
    case GSTAMD_AUDIO_FORMAT_S16: build_table<int16_t> (*p); break;
    case GSTAMD_AUDIO_FORMAT_S32: build_table<int32_t> (*p); break;
    case GSTAMD_AUDIO_FORMAT_F32: build_table<float> (*p); break;
    default: build_table<double> (*p); break;
  }
}

// resampler_calculate_taps (audio-resampler.c:1063-1208) for the rates, method and format already in *p, then the table
static int design_filter (AudioPlan *p, const GstAmdAudioResamplerOptions &o, std::string *error)
{
  auto fail = [&](int code, const char *msg) {
    if (error)
      *error = msg;
    return code;
  };
  const int method = p->method;
  /* resampler_calculate_taps (:1063-1208) */
  bool scale = true, sinc_table = false;
  p->cutoff = 0;
  p->kaiser_beta = 0;
  p->b = p->c = 0;
  switch (method) {
    case GSTAMD_AUDIO_RESAMPLER_METHOD_NEAREST:
      p->n_taps = 2;
      scale = false;
      break;
    case GSTAMD_AUDIO_RESAMPLER_METHOD_LINEAR:
      p->n_taps = is_set (o.n_taps) ? o.n_taps : 2;
      break;
    case GSTAMD_AUDIO_RESAMPLER_METHOD_CUBIC:
      p->n_taps = is_set (o.n_taps) ? o.n_taps : 4;
      p->b = is_set (o.cubic_b) ? o.cubic_b : 1.0;
      p->c = is_set (o.cubic_c) ? o.cubic_c : 0.0;
      break;
    case GSTAMD_AUDIO_RESAMPLER_METHOD_BLACKMAN_NUTTALL: {
      const BlackmanQuality &q = kBlackmanQualities[DEFAULT_QUALITY];
      p->n_taps = is_set (o.n_taps) ? o.n_taps : q.n_taps;
      p->cutoff = is_set (o.cutoff) ? o.cutoff : q.cutoff;
      sinc_table = true;
      break;
    }
    case GSTAMD_AUDIO_RESAMPLER_METHOD_KAISER: {
      /* calculate_kaiser_params (:928-965) */
      const KaiserQuality &q = kKaiserQualities[DEFAULT_QUALITY];
      double Fc = q.cutoff;
      if (p->out_rate < p->in_rate)
        Fc *= q.downsample_cutoff_factor;
      Fc = is_set (o.cutoff) ? o.cutoff : Fc;
      double A = is_set (o.stop_attenuation) ? o.stop_attenuation : q.stopband_attenuation;
      double tr_bw = is_set (o.transition_bandwidth) ? o.transition_bandwidth : q.transition_bandwidth;
      double B;
      if (A > 50)
        B = 0.1102 * (A - 8.7);
      else if (A >= 21)
        B = 0.5842 * pow (A - 21, 0.4) + 0.07886 * (A - 21);
      else
        B = 0.0;
      double dw = 2 * M_PI * (tr_bw);
      int n = (int) ((A - 8.0) / (2.285 * dw));
      p->kaiser_beta = B;
      p->n_taps = n + 1;
      p->cutoff = Fc;
      sinc_table = true;
      break;
    }
  }
  if (p->n_taps <= 0)
    return fail (GSTAMD_ERR_INVALID, "n-taps must be positive");
  if (p->out_rate < p->in_rate && scale) {
    p->cutoff = p->cutoff * p->out_rate / p->in_rate;
    p->n_taps = (int) (((uint64_t) p->n_taps * (uint64_t) p->in_rate) / (uint64_t) p->out_rate);   /* gst_util_uint64_scale_int */
  }
  int filter_mode, filter_interpolation;
  unsigned filter_threshold = 1048576;
  if (sinc_table) {
    p->n_taps = (p->n_taps + 7) / 8 * 8;
    filter_mode = is_set (o.filter_mode) ? o.filter_mode : GSTAMD_AUDIO_FILTER_MODE_AUTO;
    filter_threshold = is_set (o.filter_mode_threshold) ? (unsigned) o.filter_mode_threshold : 1048576u;
    filter_interpolation = is_set (o.filter_interpolation) ? o.filter_interpolation : GSTAMD_AUDIO_FILTER_INTERPOLATION_CUBIC;
  } else {
    filter_mode = GSTAMD_AUDIO_FILTER_MODE_FULL;
    filter_interpolation = GSTAMD_AUDIO_FILTER_INTERPOLATION_NONE;
  }
  int oversample;
  if (filter_interpolation != GSTAMD_AUDIO_FILTER_INTERPOLATION_NONE) {
    int mult = 2;
    oversample = is_set (o.filter_oversample) ? o.filter_oversample : 8;
    while (oversample > 1) {
      if (mult * p->out_rate >= p->in_rate)
        break;
      mult *= 2;
      oversample >>= 1;
    }
    if (filter_interpolation == GSTAMD_AUDIO_FILTER_INTERPOLATION_LINEAR)
      oversample *= 11;
  } else {
    oversample = 1;
  }
  p->oversample = oversample;
  if (filter_mode == GSTAMD_AUDIO_FILTER_MODE_AUTO) {
    if (p->out_rate <= oversample && !p->variable_rate)
      filter_mode = GSTAMD_AUDIO_FILTER_MODE_FULL;
    else if ((unsigned) (p->bps * p->n_taps * p->out_rate) < filter_threshold)
      filter_mode = GSTAMD_AUDIO_FILTER_MODE_FULL;
    else
      filter_mode = GSTAMD_AUDIO_FILTER_MODE_INTERPOLATED;
  }
  if (filter_mode != GSTAMD_AUDIO_FILTER_MODE_FULL && filter_interpolation == GSTAMD_AUDIO_FILTER_INTERPOLATION_NONE)
    filter_interpolation = GSTAMD_AUDIO_FILTER_INTERPOLATION_CUBIC;
  p->filter_mode = filter_mode;
  p->filter_interpolation = filter_interpolation;
  if (oversample < 1)
    return fail (GSTAMD_ERR_INVALID, "bad filter-oversample");
  p->n_phases = p->out_rate;
  rebuild_table (p);
  return GSTAMD_OK;
}

/* the rate part of gst_audio_resampler_update (audio-resampler.c:1527-1560): the common divisor of the two rates is
 * taken out as far as the rescaled phase stays within max_error of where it was */
static void reduce_rates (AudioPlan *p, int in_rate, int out_rate, double max_error, long long *samp_phase_io)
{
  int samp_phase = (int) *samp_phase_io;
  int gcd = gcd_int (in_rate, out_rate);
  if (max_error < 1.0e-8) {
    gcd = gcd_int (gcd, samp_phase);
  } else {
    while (gcd > 1) {
      double ph1 = (double) samp_phase / out_rate;
      int factor = 2;
      double ph2 = (double) (samp_phase / gcd) / (out_rate / gcd);
      if (fabs (ph1 - ph2) < max_error)
        break;
      while (gcd % factor != 0)
        factor++;
      gcd /= factor;
    }
  }
  *samp_phase_io = samp_phase / gcd;
  p->in_rate = in_rate / gcd;
  p->out_rate = out_rate / gcd;
  p->samp_inc = p->in_rate / p->out_rate;
  p->samp_frac = p->in_rate % p->out_rate;
}

int plan_audio_resampler (int method, int flags, int format, int channels, int in_rate, int out_rate,
    const GstAmdAudioResamplerOptions *options_in, AudioPlan *p, std::string *error)
{
  auto fail = [&](int code, const char *msg) {
    if (error)
      *error = msg;
    return code;
  };
  /* `format` is a GstAudioFormat as gst_audio_resampler_new takes it (audio-resampler.h:218): the four native-endian ones the
     resampler accepts (audio-resampler.c:1358-1360); the library's own 0 .. 3 of earlier releases mean the same four */
  switch (format) {
    case GSTAMD_AFMT_S16LE: format = GSTAMD_AUDIO_FORMAT_S16; break;
    case GSTAMD_AFMT_S32LE: format = GSTAMD_AUDIO_FORMAT_S32; break;
    case GSTAMD_AFMT_F32LE: format = GSTAMD_AUDIO_FORMAT_F32; break;
    case GSTAMD_AFMT_F64LE: format = GSTAMD_AUDIO_FORMAT_F64; break;
    default: break;
  }
  if (method < 0 || method > 4 || format < 0 || format > 3 || channels <= 0 || in_rate <= 0 || out_rate <= 0)
    return fail (GSTAMD_ERR_INVALID, "bad resampler arguments");
  GstAmdAudioResamplerOptions o;
  if (options_in)
    o = *options_in;
  else {                      /* gst_audio_resampler_new with NULL options (:1414-1419) */
    audio_options_init (&o);
    audio_options_set_quality (GSTAMD_AUDIO_RESAMPLER_METHOD_KAISER, DEFAULT_QUALITY, in_rate, out_rate, &o);
  }
  p->method = method;
  p->format = format;
  p->channels = channels;
  p->bps = format == GSTAMD_AUDIO_FORMAT_S16 ? 2 : (format == GSTAMD_AUDIO_FORMAT_F64 ? 8 : 4);
  p->variable_rate = (flags & 4) != 0;
  p->in_planar = (flags & 1) != 0;
  p->out_planar = (flags & 2) != 0;

  /* gst_audio_resampler_update (:1503-1560): reduce the rates; samp_phase is 0 for a new resampler.
   * NB the reference reads max-phase-error from resampler->options, which is still NULL here, so the
   * default 0.1 applies on creation whatever the caller passed. */
  long long samp_phase = 0;
  reduce_rates (p, in_rate, out_rate, 0.1, &samp_phase);

  p->max_phase_error = is_set (o.max_phase_error) ? o.max_phase_error : 0.1;      /* what a later update () reads back from the options */
  return design_filter (p, o, error);
}

/* gst_audio_resampler_update (audio-resampler.c:1503-1614) on an existing stream.  Rates <= 0 keep the (reduced) current
 * ones.  With options the filter is designed again for the new rates and, when the tap count changes, the history is
 * shifted by half the difference (*shift says how); without options the reference keeps the OLD filter design (cutoff,
 * tap count, oversampled prototype - even when the ratio now calls for another one) and only re-derives the per-phase
 * rows for the new number of phases. */
int audio_update (AudioPlan *p, AudioState *st, int in_rate, int out_rate, const GstAmdAudioResamplerOptions *options,
    AudioHistoryShift *shift, std::string *error)
{
  memset (shift, 0, sizeof (*shift));
  if (in_rate <= 0)
    in_rate = p->in_rate;
  if (out_rate <= 0)
    out_rate = p->out_rate;
  /* gst_util_uint64_scale_int (samp_phase, out_rate, old out_rate) */
  long long samp_phase = p->out_rate > 0 ? (long long) (((unsigned __int128) (uint64_t) st->samp_phase * (uint64_t) out_rate) / (uint64_t) p->out_rate) : 0;
  reduce_rates (p, in_rate, out_rate, p->max_phase_error, &samp_phase);
  st->samp_phase = samp_phase;
  if (options) {
    const int old_n_taps = p->n_taps;
    p->max_phase_error = is_set (options->max_phase_error) ? options->max_phase_error : 0.1;
    int e = design_filter (p, *options, error);
    if (e != GSTAMD_OK)
      return e;
    if (old_n_taps > 0 && old_n_taps != p->n_taps) {
      const long long diff = ((long long) p->n_taps - old_n_taps) / 2;
      long long frames = (long long) st->samples_avail, soff = st->samp_index, doff = st->samp_index;
      if (diff < 0) {
        soff += -diff;
        frames -= -diff;
      } else {
        doff += diff;
      }
      shift->changed = true;
      shift->src_off = soff;
      shift->dst_off = doff;
      shift->frames = frames > 0 ? frames : 0;
      /* "when we enlarge we just leave the old samples in there.  FIXME, probably do something better like mirror or fill with zeroes"
       * (audio-resampler.c:1587-1590): the head of the enlarged history, [samp_index, samp_index + diff), keeps what the sample buffer held there.
       * Up to samples_avail that is the old history's own head (reproduced by audio_history_shift); beyond it, it is whatever EARLIER calls
       * left in the buffer past its valid samples - input of the previous gst_audio_resampler_resample calls that this library never kept.
       * Those frames are silence here (the FIXME's second suggestion), and the resampler says so: gstamd_audio_resampler_divergence. */
      if (diff > 0) {
        const long long past = (long long) st->samp_index + diff - (long long) st->samples_avail;
        shift->stale = past > 0 ? past : 0;
        if (past > 0)
          st->stale_ahead = (long long) st->samp_index + diff;
      }
      st->samples_avail = (size_t) ((long long) st->samples_avail + diff > 0 ? (long long) st->samples_avail + diff : 0);
    }
  } else if (p->filter_mode == GSTAMD_AUDIO_FILTER_MODE_FULL) {
    p->n_phases = p->out_rate;
    rebuild_table (p);
  }
  return GSTAMD_OK;
}

/* the memmove of gst_audio_resampler_update (:1592-1593) on a host copy of the interleaved history; frames the old buffer
 * did not hold read as silence (the reference's buffers are zero-initialised, get_sample_bufs :1439) */
void audio_history_shift (const AudioHistoryShift &s, size_t frame_bytes, std::vector<uint8_t> *hist)
{
  if (!s.changed)
    return;
  const size_t old_frames = hist->size () / frame_bytes;
  std::vector<uint8_t> n (*hist);
  const size_t need = (size_t) (s.dst_off + s.frames);
  if (n.size () < need * frame_bytes)
    n.resize (need * frame_bytes, 0);
  for (long long i = 0; i < s.frames; i++) {
    const size_t si = (size_t) (s.src_off + i);
    uint8_t *d = &n[(size_t) (s.dst_off + i) * frame_bytes];
    if (si < old_frames)
      memcpy (d, hist->data () + si * frame_bytes, frame_bytes);
    else
      memset (d, 0, frame_bytes);
  }
  hist->swap (n);
}

/* gst_audio_resampler_reset (audio-resampler.c:1466-1488) */
void audio_state_reset (const AudioPlan &plan, AudioState *st)
{
  st->stale_ahead = 0;
  st->samp_index = 0;
  st->samples_avail = (size_t) (plan.n_taps / 2 - 1);
}

/* gst_audio_resampler_resample (audio-resampler.c:1750-1806) + MAKE_RESAMPLE_FUNC
 * (audio-resampler-macros.h:62-100): all integer bookkeeping, no samples touched */
AudioStep audio_step (const AudioPlan &pl, AudioState *st, size_t in_frames, size_t out_frames)
{
  AudioStep s;
  memset (&s, 0, sizeof (s));
  if (st->skip >= (long long) in_frames) {
    st->skip -= (long long) in_frames;
    s.skipped_all = true;
    return s;
  }
  st->samp_index += st->skip;
  const size_t hist_frames = st->samples_avail;
  const size_t samples_avail = hist_frames + in_frames;
  st->samples_avail = samples_avail;
  s.samp_index0 = st->samp_index;
  s.samp_phase0 = (int) st->samp_phase;
  s.hist_frames = (long long) hist_frames;
  s.total_frames = (long long) samples_avail;
  s.src_start = 0;
  s.moved = s.keep = (long long) samples_avail;
  const size_t need = (size_t) pl.n_taps + (size_t) st->samp_index;
  if (samples_avail < need || out_frames == 0)
    return s;                           /* not enough samples to start: the input joins the history */
  s.run_fir = true;
  s.n_out = (long long) out_frames;
  const long long tot = st->samp_phase + s.n_out * (long long) pl.samp_frac;
  const long long end_index = st->samp_index + s.n_out * (long long) pl.samp_inc + tot / pl.out_rate;
  const long long consumed = end_index - st->samp_index;
  /* memmove (ip, &ip[samp_index], in_len - samp_index) only when in_len > samp_index */
  if ((long long) samples_avail > end_index) {
    s.src_start = end_index;
    s.moved = (long long) samples_avail - end_index;
  } else {
    s.src_start = 0;
    s.moved = 0;
  }
  st->samp_index = 0;
  st->samp_phase = tot % pl.out_rate;
  if (consumed > 0)
    st->stale_ahead = st->stale_ahead > consumed ? st->stale_ahead - consumed : 0;
  if (consumed > 0) {
    const long long left = (long long) samples_avail - consumed;
    if (left > 0) {
      st->samples_avail = (size_t) left;
    } else {
      st->samples_avail = 0;
      st->skip = -left;
    }
  }
  s.keep = (long long) st->samples_avail;
  return s;
}

size_t audio_get_out_frames (const AudioPlan &pl, const AudioState &st, size_t in_frames)
{
  const size_t need = (size_t) pl.n_taps + (size_t) st.samp_index + (size_t) st.skip;
  const size_t avail = st.samples_avail + in_frames;
  if (avail < need)
    return 0;
  size_t out = (avail - need) * (size_t) pl.out_rate;
  if (out < (size_t) st.samp_phase)
    return 0;
  return ((out - (size_t) st.samp_phase) / (size_t) pl.in_rate) + 1;
}

size_t audio_get_in_frames (const AudioPlan &pl, const AudioState &st, size_t out_frames)
{
  size_t in_frames = ((size_t) st.samp_phase + out_frames * (size_t) pl.samp_frac) / (size_t) pl.out_rate;
  in_frames += out_frames * (size_t) pl.samp_inc;
  return in_frames;
}

}  // namespace gstamd
