// tests/emu/emu_audio.cpp - TEST INFRASTRUCTURE: host loop over the FIR kernel bodies (audio_device.h)
// with the product's host bookkeeping (audio_taps.cpp), so the resampler can be checked against the
// reference in this GPU-less container.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../gstreamer_amd/csrc/audio_device.h"
#include "../../gstreamer_amd/csrc/audio_taps.h"

using namespace gstamd;

struct EmuResampler {
  AudioPlan plan;
  AudioState st;
  std::vector<uint8_t> hist;
};

static int g_fir_lds_runs = 0;
extern "C" int emu_fir_lds_runs (void) { return g_fir_lds_runs; }

template <typename T>
static void emu_run (EmuResampler *r, const void *in, size_t in_frames, void *out, size_t out_frames)
{
  const AudioPlan &pl = r->plan;
  const AudioStep s = audio_step (pl, &r->st, in_frames, out_frames);
  if (s.skipped_all)
    return;
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
  p.in_is_null = in == nullptr;
  p.in_plane_stride = pl.in_planar ? (long long) in_frames : 0;
  p.out_plane_stride = pl.out_planar ? (long long) out_frames : 0;
  const T *hist = (const T *) r->hist.data ();
  bool lds_done = false;
  if (s.run_fir && !p.nearest && !p.interp && getenv ("GSTAMD_NO_FIR_LDS") == nullptr) {
    /* k_fir_lds: workgroups of FIR_LDS_FRAMES frames, staging phase by all 256 threads, then four lanes per frame */
    typedef typename Acc<T>::type A;
    FirLdsGeom g;
    g.row_stride = p.n_taps_padded + 4;
    const int span_max = FIR_LDS_FRAMES * (p.samp_inc + 1) + p.n_taps_padded + 2;
    g.win_frames = ((span_max + 31) & ~31) + 16;
    const size_t words = (size_t) FIR_LDS_FRAMES * g.row_stride + (size_t) pl.channels * g.win_frames;
    if (words * sizeof (T) <= 64 * 1024) {
      std::vector<T> lds (words);
      for (long long jb = 0; jb < s.n_out; jb += FIR_LDS_FRAMES) {
        memset (lds.data (), 0x5a, words * sizeof (T));
        T *rows = lds.data (), *win = rows + FIR_LDS_FRAMES * g.row_stride;
        const int nj = s.n_out - jb < FIR_LDS_FRAMES ? (int) (s.n_out - jb) : FIR_LDS_FRAMES;
        int pos[2 * FIR_LDS_FRAMES];
        for (int tid = 0; tid < 256; tid++)
          fir_lds_positions (p, jb, nj, pos, tid, 256);
        for (int tid = 0; tid < 256; tid++)
          fir_lds_stage<T> (p, g, hist, (const T *) in, (const T *) pl.table.data (), jb, nj, pos, rows, win, tid, 256);
        for (int fr = 0; fr < nj; fr++)
          for (int c = 0; c < pl.channels; c++) {
            A rq[4];
            for (int q = 0; q < 4; q++)
              rq[q] = fir_lds_partial<T> (p, g, pos, rows, win, fr, q, c);
            ((T *) out)[fir_out_index (p, jb + fr, c)] = fir_lds_combine<T> (rq[0], rq[1], rq[2], rq[3]);
          }
      }
      lds_done = true;
      g_fir_lds_runs++;
    }
  }
  if (s.run_fir && !lds_done)
    for (long long j = 0; j < s.n_out; j++)
      for (int c = 0; c < pl.channels; c++)
        ((T *) out)[fir_out_index (p, j, c)] = fir_output<T> (p, hist, (const T *) in, (const T *) pl.table.data (), j, c);
  std::vector<uint8_t> nh ((size_t) (s.keep + 1) * pl.channels * sizeof (T));
  for (long long i = 0; i < s.keep; i++)
    for (int c = 0; c < pl.channels; c++)
      ((T *) nh.data ())[i * pl.channels + c] = history_sample<T> (p, hist, (const T *) in, s.src_start, s.moved, i, c);
  r->hist.swap (nh);
}

extern "C" {

void *emu_audio_new (int method, int flags, int format, int channels, int in_rate, int out_rate,
    const GstAmdAudioResamplerOptions *options, int *status, char *err, int err_len)
{
  EmuResampler *r = new EmuResampler ();
  std::string e;
  int st = plan_audio_resampler (method, flags, format, channels, in_rate, out_rate, options, &r->plan, &e);
  if (status)
    *status = st;
  if (st != GSTAMD_OK) {
    if (err)
      strncpy (err, e.c_str (), err_len - 1);
    delete r;
    return nullptr;
  }
  audio_state_reset (r->plan, &r->st);
  r->hist.assign ((size_t) (r->plan.n_taps + 8) * channels * r->plan.bps, 0);
  return r;
}

void emu_audio_free (void *h) { delete (EmuResampler *) h; }
size_t emu_audio_get_out_frames (void *h, size_t in_frames) { EmuResampler *r = (EmuResampler *) h; return audio_get_out_frames (r->plan, r->st, in_frames); }
int emu_audio_n_taps (void *h) { return ((EmuResampler *) h)->plan.n_taps; }

int emu_audio_update (void *h, int in_rate, int out_rate, const GstAmdAudioResamplerOptions *options)
{
  EmuResampler *r = (EmuResampler *) h;
  AudioHistoryShift shift;
  std::string e;
  const size_t old_avail = r->st.samples_avail + (size_t) r->st.samp_index;
  int st = audio_update (&r->plan, &r->st, in_rate, out_rate, options, &shift, &e);
  if (st != GSTAMD_OK)
    return st;
  const size_t fbytes = (size_t) r->plan.channels * r->plan.bps;
  if (r->hist.size () > old_avail * fbytes)
    r->hist.resize (old_avail * fbytes);
  audio_history_shift (shift, fbytes, &r->hist);
  r->hist.resize (r->hist.size () + 8 * fbytes, 0);
  return st;
}
int emu_audio_stale_ahead (void *h) { return (int) ((EmuResampler *) h)->st.stale_ahead; }
int emu_audio_state (void *h, int which)
{
  EmuResampler *r = (EmuResampler *) h;
  switch (which) {
    case 0: return r->plan.n_taps;
    case 1: return r->plan.in_rate;
    case 2: return r->plan.out_rate;
    case 3: return (int) r->st.samp_phase;
    case 4: return (int) r->st.samples_avail;
    default: return r->plan.filter_mode;
  }
}

void emu_audio_resample (void *h, const void *in, size_t in_frames, void *out, size_t out_frames)
{
  EmuResampler *r = (EmuResampler *) h;
  switch (r->plan.format) {
    case GSTAMD_AUDIO_FORMAT_S16: emu_run<int16_t> (r, in, in_frames, out, out_frames); break;
    case GSTAMD_AUDIO_FORMAT_S32: emu_run<int32_t> (r, in, in_frames, out, out_frames); break;
    case GSTAMD_AUDIO_FORMAT_F32: emu_run<float> (r, in, in_frames, out, out_frames); break;
    default: emu_run<double> (r, in, in_frames, out, out_frames); break;
  }
}

}  // extern "C"

// ---- audio converter (audio_convert_plan.h + audio_convert_device.h): the two kernels' bodies around the emulated resampler ------
#include "../../gstreamer_amd/csrc/audio_convert_plan.h"

struct EmuAConv {
  AConvPlan plan;
  bool resample = false, passthrough = false;
  void *resampler = nullptr;
  AConvDitherState dither = { 0xc2d6038fu, 0u, 0 };
  AConvJump jump;
  std::vector<int32_t> hist = std::vector<int32_t> (8 * GSTAMD_AUDIO_MAX_CHANNELS, 0);
};

extern "C" {

void *emu_aconv_new (int flags, const GstAmdAudioInfo *in, const GstAmdAudioInfo *out, const GstAmdAudioConverterConfig *cfg, char *err, int err_len)
{
  EmuAConv *c = new EmuAConv ();
  std::string e;
  if (aconv_make_plan (flags, in, out, *cfg, &c->plan, &c->resample, &c->passthrough, &e) != GSTAMD_OK) {
    if (err)
      strncpy (err, e.c_str (), err_len - 1);
    delete c;
    return nullptr;
  }
  if (c->resample) {
    GstAmdAudioResamplerOptions ro;
    if (cfg->has_resampler_options)
      ro = cfg->resampler_options;
    else
      audio_options_init (&ro);
    int st = 0;
    c->resampler = emu_audio_new (cfg->resampler_method, (flags & 2) ? 4 : 0, c->plan.mid_in, out->channels, in->rate, out->rate, &ro, &st, err, err_len);
    if (!c->resampler) {
      delete c;
      return nullptr;
    }
  }
  aconv_make_jump (&c->jump);
  return c;
}

void emu_aconv_free (void *h)
{
  EmuAConv *c = (EmuAConv *) h;
  if (c && c->resampler)
    emu_audio_free (c->resampler);
  delete c;
}

size_t emu_aconv_get_out_frames (void *h, size_t in_frames)
{
  EmuAConv *c = (EmuAConv *) h;
  return c->resampler ? emu_audio_get_out_frames (c->resampler, in_frames) : in_frames;
}

void emu_aconv_reset (void *h)
{
  EmuAConv *c = (EmuAConv *) h;          /* converters without a resampler only (the emulated resampler has no reset) */
  std::fill (c->hist.begin (), c->hist.end (), 0);
}

int emu_aconv_is_passthrough (void *h) { return ((EmuAConv *) h)->passthrough ? 1 : 0; }

void emu_aconv_samples (void *h, const uint8_t *in, size_t in_frames, uint8_t *out, size_t out_frames)
{
  EmuAConv *c = (EmuAConv *) h;
  const AConvPlan &p = c->plan;
  if (in_frames == 0)
    return;
  if (c->passthrough) {
    memcpy (out, in, out_frames * (size_t) p.out_ch * (size_t) afmt_bytes (p.out_fmt));
    return;
  }
  const size_t mb = (size_t) amid_bytes (p.mid_in) * (size_t) p.out_ch;
  std::vector<uint8_t> a ((in_frames ? in_frames : 1) * mb), b ((out_frames ? out_frames : 1) * mb);
  if (in)
    for (size_t n = 0; n < in_frames; n++)
      for (int co = 0; co < p.out_ch; co++)
        aconv_pre_sample (p, in, a.data (), n, co);
  const uint8_t *after = a.data ();
  if (c->resampler) {
    emu_audio_resample (c->resampler, in ? a.data () : nullptr, in_frames, b.data (), out_frames);
    after = b.data ();
  }
  const size_t samples = out_frames * (size_t) p.out_ch;
  std::vector<int32_t> qv (samples + 1), qd (samples + 1);
  for (size_t i = 0; i < samples; i++)
    aconv_post_sample (p, c->jump, c->dither, after, out, qv.data (), qd.data (), i);
  if (p.ns && p.quant_shift > 0)
    for (int ch = 0; ch < p.out_ch; ch++)
      aconv_shape_channel (p, qv.data (), qd.data (), c->hist.data (), out, out_frames, ch);
  aconv_dither_advance (p, c->jump, &c->dither, samples);
}

}  // extern "C"
