// audio_convert_plan.h - the decision code of gst_audio_converter_new (audio-converter.c:1346-1470) restated: which stages a conversion
// has and on which intermediate format they run.  Host only; shared by audio_convert.hip and the host emulation of tests/emu.
#pragma once
#include <cstring>
#include <string>

#include "../../include/gstamd_video.h"
#include "audio_convert_device.h"

// ---- the plan: gst_audio_converter_new's chain ---------------------------------------------------------------------------------------
namespace gstamd {

struct AFmtInfo {
  bool known, integer;
  int depth;
};

static AFmtInfo afmt_info (int fmt)
{
  switch (fmt) {
    case GSTAMD_AFMT_S8: case GSTAMD_AFMT_U8: return {true, true, 8};
    case GSTAMD_AFMT_S16LE: return {true, true, 16};
    case GSTAMD_AFMT_S24LE: case GSTAMD_AFMT_S24_32LE: return {true, true, 24};
    case GSTAMD_AFMT_S32LE: return {true, true, 32};
    case GSTAMD_AFMT_F32LE: return {true, false, 32};
    case GSTAMD_AFMT_F64LE: return {true, false, 64};
    default: return {false, false, 0};
  }
}

static bool afmt_is_intermediate (int fmt)
{
  return fmt == GSTAMD_AFMT_S16LE || fmt == GSTAMD_AFMT_S32LE || fmt == GSTAMD_AFMT_F32LE || fmt == GSTAMD_AFMT_F64LE;
}

static int afmt_mid (int fmt)
{
  return fmt == GSTAMD_AFMT_S16LE ? AMID_S16 : fmt == GSTAMD_AFMT_S32LE ? AMID_S32 : fmt == GSTAMD_AFMT_F32LE ? AMID_F32 : AMID_F64;
}

// the mixing matrix (gst_audio_channel_mixer_fill_matrix, audio-channel-mixer.c:731-805), as far as built: the mono <-> stereo
// special cases, unpositioned input (identity), and layouts whose output positions all exist in the input (fill_identical; nothing
// left for fill_compatible / fill_others to add, fill_normalize a no-op).  Everything else wants an explicit mix-matrix.
static bool default_mix_matrix (const GstAmdAudioInfo &in, const GstAmdAudioInfo &out, float m[GSTAMD_AUDIO_MAX_CHANNELS][GSTAMD_AUDIO_MAX_CHANNELS], std::string *why)
{
  const int FL = 0, FR = 1, MONO = -1;
  memset (m, 0, sizeof (float) * GSTAMD_AUDIO_MAX_CHANNELS * GSTAMD_AUDIO_MAX_CHANNELS);
  if (in.channels == 2 && out.channels == 1 && !in.unpositioned && !out.unpositioned &&
      ((in.position[0] == FL && in.position[1] == FR) || (in.position[0] == FR && in.position[1] == FL)) && out.position[0] == MONO) {
    m[0][0] = m[1][0] = 0.5f;
    return true;
  }
  if (in.channels == 1 && out.channels == 2 && !in.unpositioned && !out.unpositioned &&
      ((out.position[0] == FL && out.position[1] == FR) || (out.position[0] == FR && out.position[1] == FL)) && in.position[0] == MONO) {
    m[0][0] = m[0][1] = 1.0f;
    return true;
  }
  if (in.unpositioned) {
    for (int co = 0; co < out.channels; co++)
      for (int ci = 0; ci < in.channels; ci++)
        m[ci][co] = ci == co ? 1.0f : 0.0f;
    return true;
  }
  if (out.unpositioned) {
    *why = "positioned input into unpositioned output";
    return false;
  }
  /* all-mono or alternating left / right inputs of more than one / two channels are "virtual inputs" (:684-729): not built */
  bool all_mono = in.channels >= 2, alternate = in.channels > 2;
  for (int i = 0; i < in.channels; i++) {
    all_mono = all_mono && in.position[i] == MONO;
    alternate = alternate && in.position[i] == (i & 1);
  }
  if (all_mono || alternate) {
    *why = "virtual mono / stereo input layouts";
    return false;
  }
  if (in.channels != out.channels) {
    *why = "channel layouts other than mono <-> stereo need the position-based down / up-mix rules (fill_compatible / fill_others)";
    return false;
  }
  for (int co = 0; co < out.channels; co++) {
    int found = 0;
    for (int ci = 0; ci < in.channels; ci++)
      if (in.position[ci] == out.position[co]) {
        m[ci][co] = 1.0f;
        found++;
      }
    if (found != 1) {
      *why = "output channel positions that the input does not have";
      return false;
    }
  }
  return true;
}


// The whole plan.  *resample: a resampler (on p->mid_in, out->channels) sits between the two kernels; *passthrough: the bytes
// themselves.  Returns GSTAMD_OK or an error code with *err set.
inline int aconv_make_plan (int flags, const GstAmdAudioInfo *in, const GstAmdAudioInfo *out, const GstAmdAudioConverterConfig &cfg, AConvPlan *plan,
    bool *resample, bool *passthrough, std::string *err)
{
  const AFmtInfo fi = afmt_info (in->format), fo = afmt_info (out->format);
  if (!fi.known || !fo.known) {
    *err = "sample format not built on the GPU path (S8, U8, S16LE, S24LE, S24_32LE, S32LE, F32LE, F64LE are)";
    return GSTAMD_ERR_UNSUPPORTED;
  }
  if (in->channels < 1 || out->channels < 1 || in->channels > GSTAMD_AUDIO_MAX_CHANNELS || out->channels > GSTAMD_AUDIO_MAX_CHANNELS) {
    *err = "1 .. 8 channels";
    return GSTAMD_ERR_UNSUPPORTED;
  }
  if (in->layout != 0 || out->layout != 0) {
    *err = "non-interleaved layouts are not built into the converter yet";
    return GSTAMD_ERR_UNSUPPORTED;
  }
  if (in->rate <= 0 || out->rate <= 0) {
    *err = "bad rate";
    return GSTAMD_ERR_INVALID;
  }
  /* gst_audio_converter_new :1370-1378 */
  if (!cfg.has_mix_matrix && in->channels != out->channels && (in->unpositioned || out->unpositioned)) {
    *err = "unpositioned channels with different channel counts and no mix-matrix";
    return GSTAMD_ERR_INVALID;
  }
  AConvPlan &p = *plan;
  memset (&p, 0, sizeof (p));
  p.in_fmt = in->format;
  p.out_fmt = out->format;
  p.in_ch = in->channels;
  p.out_ch = out->channels;
  /* chain_unpack :708-740 */
  const bool same_format = in->format == out->format;
  int cur = (same_format && afmt_is_intermediate (in->format)) ? afmt_mid (in->format) : (fi.integer ? AMID_S32 : AMID_F64);
  /* chain_convert_in :742-762 */
  if (fi.integer && !fo.integer) {
    p.convert_in = 1;
    cur = AMID_F64;
  }
  p.mid_in = cur;
  /* chain_mix :849-902 */
  if (cfg.has_mix_matrix) {
    for (int ci = 0; ci < in->channels; ci++)
      for (int co = 0; co < out->channels; co++)
        p.m[ci][co] = cfg.mix_matrix[co][ci];           /* mix_matrix_from_g_value: the option is [out][in] */
  } else {
    std::string why;
    if (!default_mix_matrix (*in, *out, p.m, &why)) {
      *err = "no GPU-side rule for this channel conversion (" + why + "); pass a mix-matrix";
      return GSTAMD_ERR_UNSUPPORTED;
    }
  }
  for (int ci = 0; ci < in->channels; ci++)
    for (int co = 0; co < out->channels; co++) {
      const float tmp = p.m[ci][co] * (float) (1 << 10);        /* gst_audio_channel_mixer_setup_matrix_int */
      p.mi[ci][co] = (int) tmp;
    }
  bool mix_passthrough = in->channels == out->channels;
  for (int i = 0; i < in->channels && mix_passthrough; i++)
    for (int j = 0; j < out->channels && mix_passthrough; j++)
      mix_passthrough = p.m[i][j] == (i == j ? 1.0f : 0.0f);
  p.mix = mix_passthrough ? 0 : 1;
  /* chain_resample :904-943 */
  *resample = in->rate != out->rate || (flags & 2) != 0;
  /* chain_convert_out :945-966 */
  if (!fi.integer && fo.integer) {
    p.convert_out = 1;
    cur = AMID_S32;
  }
  p.mid_out = cur;
  /* chain_quantize :968-1022 */
  {
    const int in_depth = cur == AMID_S16 ? 16 : cur == AMID_F64 ? 64 : 32;
    const bool in_int = cur == AMID_S16 || cur == AMID_S32;
    int dither = cfg.dither_method, ns = cfg.noise_shaping;
    if ((unsigned) fo.depth > cfg.dither_threshold || (in_int && fo.depth >= in_depth)) {
      dither = GSTAMD_AUDIO_DITHER_NONE;
      ns = 0;
    } else if (ns > 1 && out->rate < 32000) {
      ns = 1;
    }
    if (fo.integer && fo.depth < 32 && cur == AMID_S32) {
      if (ns != 0) {
        *err = "noise shaping (error feedback per channel, audio-quantize.c:200-290) is sequential; not built on the GPU path";
        return GSTAMD_ERR_UNSUPPORTED;
      }
      if (dither == GSTAMD_AUDIO_DITHER_TPDF_HF) {
        *err = "tpdf-hf dither (depends on the previous sample's random value) is not built on the GPU path";
        return GSTAMD_ERR_UNSUPPORTED;
      }
      p.quant_shift = 32 - fo.depth;                    /* quantizer 1 << (32 - depth): count_power */
      p.dither = dither;
    }
  }
  /* "optimize" :1404-1453: same format, passthrough mixing, no resampler -> the bytes themselves */
  *passthrough = mix_passthrough && same_format && !*resample;
  return GSTAMD_OK;
}

}  // namespace gstamd

