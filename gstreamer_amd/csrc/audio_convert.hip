// audio_convert.hip - GstAudioConverter on the device: the decision code of gst_audio_converter_new (audio-converter.c:1346-1470) restated
// on the host, two kernels around the resampler of audio_kernels.hip, and the C ABI of include/gstamd_audio.h.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstring>
#include <string>

#include "../../include/gstamd_audio.h"
#include "../../include/gstamd_video.h"
#include "audio_convert_device.h"
#include "audio_convert_plan.h"

using namespace gstamd;

extern "C" void gstamd_internal_set_error (const char *msg);

static int aconv_fail (int code, const std::string &msg)
{
  gstamd_internal_set_error (("audio converter: " + msg).c_str ());
  return code;
}

// ---- kernels: one lane per sample -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__ (256) void k_aconv_pre (AConvPlan p, const uint8_t *__restrict__ in, uint8_t *__restrict__ mid, size_t frames)
{
  const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= frames * (size_t) p.out_ch)
    return;
  aconv_pre_sample (p, in, mid, i / (size_t) p.out_ch, (int) (i % (size_t) p.out_ch));
}

__global__ __launch_bounds__ (256) void k_aconv_post (AConvPlan p, const AConvJump *__restrict__ jump, AConvDitherState ds, const uint8_t *__restrict__ mid,
    uint8_t *__restrict__ out, int32_t *__restrict__ qv, int32_t *__restrict__ qd, size_t samples)
{
  const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= samples)
    return;
  aconv_post_sample (p, *jump, ds, mid, out, qv, qd, i);
}

// noise shaping: the error recurrence of a channel is sequential in time, so one lane walks one channel's frames (the samples and
// dither words were prepared in parallel by k_aconv_post)
__global__ __launch_bounds__ (64) void k_aconv_shape (AConvPlan p, const int32_t *__restrict__ qv, const int32_t *__restrict__ qd, int32_t *__restrict__ hist,
    uint8_t *__restrict__ out, size_t frames)
{
  const int c = (int) threadIdx.x;
  if (c < p.out_ch)
    aconv_shape_channel (p, qv, qd, hist, out, frames, c);
}

struct GstAmdAudioConverter {
  GstAmdAudioInfo in, out;
  GstAmdAudioConverterConfig cfg;
  int flags = 0;
  AConvPlan plan;
  bool passthrough = false;
  GstAmdAudioResampler *resampler = nullptr;
  AConvDitherState dither = { 0xc2d6038fu, 0u, 0 };     /* gst_audio_quantize_setup_dither */
  int32_t *hist = nullptr;                              /* [8][channels] noise shaping error history */
  uint8_t *q_v = nullptr, *q_d = nullptr;               /* S32 samples / dither words of a call with noise shaping */
  size_t q_v_size = 0, q_d_size = 0;
  AConvJump jump_host;
  AConvJump *jump_dev = nullptr;
  uint8_t *mid_a = nullptr, *mid_b = nullptr;           /* before / after the resampler */
  size_t mid_a_size = 0, mid_b_size = 0;
};

extern "C" {

void gstamd_audio_converter_config_init (GstAmdAudioConverterConfig *c)
{
  if (!c)
    return;
  memset (c, 0, sizeof (*c));
  /* DEFAULT_OPT_*, audio-converter.c:291-294 (the audioconvert element sets tpdf itself; the library's default is none) */
  c->dither_method = GSTAMD_AUDIO_DITHER_NONE;
  c->noise_shaping = 0;
  c->dither_threshold = 20;
  c->resampler_method = 3;                              /* GST_AUDIO_RESAMPLER_METHOD_BLACKMAN_NUTTALL */
}

GstAmdAudioConverter *gstamd_audio_converter_new (int flags, const GstAmdAudioInfo *in, const GstAmdAudioInfo *out, const GstAmdAudioConverterConfig *config,
    int *status)
{
  auto fail = [&](int code, const std::string &msg) -> GstAmdAudioConverter * {
    if (status)
      *status = aconv_fail (code, msg);
    else
      aconv_fail (code, msg);
    return nullptr;
  };
  if (!in || !out)
    return fail (GSTAMD_ERR_INVALID, "NULL info");
  GstAmdAudioConverterConfig cfg;
  if (config)
    cfg = *config;
  else
    gstamd_audio_converter_config_init (&cfg);
  GstAmdAudioConverter *c = new GstAmdAudioConverter ();
  c->in = *in;
  c->out = *out;
  c->cfg = cfg;
  c->flags = flags;
  bool resample = false;
  std::string err;
  const int code = aconv_make_plan (flags, in, out, cfg, &c->plan, &resample, &c->passthrough, &err);
  if (code != GSTAMD_OK) {
    delete c;
    return fail (code, err);
  }
  if (resample) {
    GstAmdAudioResamplerOptions ro;
    if (cfg.has_resampler_options)
      ro = cfg.resampler_options;
    else
      gstamd_audio_resampler_options_init (&ro);        /* the converter hands its (empty) config to the resampler: every key at its default */
    int st = 0;
    c->resampler = gstamd_audio_resampler_new (cfg.resampler_method, (flags & 2) ? 4 : 0, c->plan.mid_in, out->channels, in->rate, out->rate, &ro, &st);
    if (!c->resampler) {
      delete c;
      if (status)
        *status = st;
      return nullptr;
    }
  }
  aconv_make_jump (&c->jump_host);
  if (hipMalloc ((void **) &c->jump_dev, sizeof (AConvJump)) != hipSuccess ||
      hipMemcpy (c->jump_dev, &c->jump_host, sizeof (AConvJump), hipMemcpyHostToDevice) != hipSuccess) {
    gstamd_audio_converter_free (c);
    return fail (GSTAMD_ERR_HIP, "jump table upload");
  }
  if (c->plan.ns) {
    const size_t hb = sizeof (int32_t) * 8 * GSTAMD_AUDIO_MAX_CHANNELS;
    /* (a null-stream memset returns before it has run and is not ordered against a non-blocking stream: wait for it - audio_kernels.hip ensure_hist) */
    if (hipMalloc ((void **) &c->hist, hb) != hipSuccess || hipMemset (c->hist, 0, hb) != hipSuccess || hipDeviceSynchronize () != hipSuccess) {
      gstamd_audio_converter_free (c);
      return fail (GSTAMD_ERR_HIP, "error history");
    }
  }
  if (status)
    *status = GSTAMD_OK;
  return c;
}

void gstamd_audio_converter_free (GstAmdAudioConverter *c)
{
  if (!c)
    return;
  if (c->resampler)
    gstamd_audio_resampler_free (c->resampler);
  if (c->jump_dev)
    (void) hipFree (c->jump_dev);
  if (c->mid_a)
    (void) hipFree (c->mid_a);
  if (c->mid_b)
    (void) hipFree (c->mid_b);
  if (c->hist)
    (void) hipFree (c->hist);
  if (c->q_v)
    (void) hipFree (c->q_v);
  if (c->q_d)
    (void) hipFree (c->q_d);
  delete c;
}

void gstamd_audio_converter_reset (GstAmdAudioConverter *c)
{
  /* gst_audio_converter_reset (:1520-1530): the resampler and the quantizer, whose reset (audio-quantize.c:503-509) drops the error
     history and touches neither the random state nor last_random */
  if (c && c->resampler)
    gstamd_audio_resampler_reset (c->resampler);
  if (c && c->hist) {
    (void) hipDeviceSynchronize ();
    (void) hipMemset (c->hist, 0, sizeof (int32_t) * 8 * GSTAMD_AUDIO_MAX_CHANNELS);
    (void) hipDeviceSynchronize ();
  }
}

size_t gstamd_audio_converter_get_out_frames (GstAmdAudioConverter *c, size_t in_frames)
{
  return c && c->resampler ? gstamd_audio_resampler_get_out_frames (c->resampler, in_frames) : in_frames;
}

size_t gstamd_audio_converter_get_in_frames (GstAmdAudioConverter *c, size_t out_frames)
{
  return c && c->resampler ? gstamd_audio_resampler_get_in_frames (c->resampler, out_frames) : out_frames;
}

size_t gstamd_audio_converter_get_max_latency (GstAmdAudioConverter *c)
{
  return c && c->resampler ? gstamd_audio_resampler_get_max_latency (c->resampler) : 0;
}

int gstamd_audio_converter_is_passthrough (GstAmdAudioConverter *c) { return c && c->passthrough ? 1 : 0; }

int gstamd_audio_converter_get_mix_matrix (GstAmdAudioConverter *c, float *matrix, int max)
{
  if (!c || !matrix)
    return -1;
  int n = 0;
  for (int ci = 0; ci < c->in.channels; ci++)
    for (int co = 0; co < c->out.channels; co++, n++)
      if (n < max)
        matrix[n] = c->plan.m[ci][co];
  return n;
}

static int ensure (uint8_t **buf, size_t *size, size_t need)
{
  if (*size >= need)
    return GSTAMD_OK;
  if (*buf)
    (void) hipFree (*buf);
  *buf = nullptr;
  *size = 0;
  if (hipMalloc ((void **) buf, need) != hipSuccess)
    return aconv_fail (GSTAMD_ERR_HIP, "hipMalloc(intermediate samples)");
  *size = need;
  return GSTAMD_OK;
}

int gstamd_audio_converter_samples (GstAmdAudioConverter *c, int flags, const void *in, size_t in_frames, void *out, size_t out_frames, void *stream_)
{
  (void) flags;
  if (!c || (!out && out_frames))
    return aconv_fail (GSTAMD_ERR_INVALID, "NULL converter or output");
  hipStream_t stream = (hipStream_t) stream_;
  const AConvPlan &p = c->plan;
  if (in_frames == 0)                           /* gst_audio_converter_samples :1618-1621: "skipping empty buffer" */
    return GSTAMD_OK;
  if (c->passthrough) {
    if (!in)
      return aconv_fail (GSTAMD_ERR_INVALID, "NULL input");
    const size_t bytes = out_frames * (size_t) p.out_ch * (size_t) afmt_bytes (p.out_fmt);
    if (in != out && hipMemcpyAsync (out, in, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess)
      return aconv_fail (GSTAMD_ERR_HIP, "copy");
    return GSTAMD_OK;
  }
  if (!c->resampler && in_frames != out_frames)
    return aconv_fail (GSTAMD_ERR_INVALID, "in_frames != out_frames without a resampler");
  const size_t mid_bytes_in = (size_t) amid_bytes (p.mid_in) * (size_t) p.out_ch;
  int r;
  const uint8_t *after = nullptr;
  if (in) {
    if ((r = ensure (&c->mid_a, &c->mid_a_size, (in_frames ? in_frames : 1) * mid_bytes_in)) != GSTAMD_OK)
      return r;
    const size_t n = in_frames * (size_t) p.out_ch;
    if (n)
      hipLaunchKernelGGL (k_aconv_pre, dim3 ((unsigned) ((n + 255) / 256)), dim3 (256), 0, stream, p, (const uint8_t *) in, c->mid_a, in_frames);
    after = c->mid_a;
  } else if (!c->resampler) {
    return aconv_fail (GSTAMD_ERR_INVALID, "NULL input");
  }
  if (c->resampler) {
    if ((r = ensure (&c->mid_b, &c->mid_b_size, (out_frames ? out_frames : 1) * mid_bytes_in)) != GSTAMD_OK)
      return r;
    r = gstamd_audio_resampler_resample (c->resampler, in ? c->mid_a : nullptr, in_frames, c->mid_b, out_frames, stream_);
    if (r != GSTAMD_OK)
      return r;
    after = c->mid_b;
  }
  const size_t samples = out_frames * (size_t) p.out_ch;
  if (samples == 0)                             /* the resampler only took input into its history */
    return GSTAMD_OK;
  if (p.ns && p.quant_shift > 0) {
    if ((r = ensure (&c->q_v, &c->q_v_size, samples * 4)) != GSTAMD_OK || (r = ensure (&c->q_d, &c->q_d_size, samples * 4)) != GSTAMD_OK)
      return r;
  }
  hipLaunchKernelGGL (k_aconv_post, dim3 ((unsigned) ((samples + 255) / 256)), dim3 (256), 0, stream, p, c->jump_dev, c->dither, after, (uint8_t *) out,
      (int32_t *) c->q_v, (int32_t *) c->q_d, samples);
  if (p.ns && p.quant_shift > 0)
    hipLaunchKernelGGL (k_aconv_shape, dim3 (1), dim3 (64), 0, stream, p, (const int32_t *) c->q_v, (const int32_t *) c->q_d, c->hist, (uint8_t *) out, out_frames);
  if (hipGetLastError () != hipSuccess)
    return aconv_fail (GSTAMD_ERR_HIP, "kernel launch");
  /* the generator moves on by the draws of this call (setup_dither_buf draws for every sample of the block) */
  aconv_dither_advance (p, c->jump_host, &c->dither, samples);
  return GSTAMD_OK;
}

}  // extern "C"
