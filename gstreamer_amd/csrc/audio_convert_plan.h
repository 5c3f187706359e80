// audio_convert_plan.h - the decision code of gst_audio_converter_new (audio-converter.c:1346-1470) restated: which stages a conversion
// has and on which intermediate format they run.  Host only; shared by audio_convert.hip and the host emulation of tests/emu.
#pragma once
#include <cmath>
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

// ---- the default mixing matrix: gst_audio_channel_mixer_fill_matrix (audio-channel-mixer.c:731-805) restated --------------------------
// Positions are GstAudioChannelPosition values (audio-channels.h:101-133).
enum APos : int {
  APOS_NONE = -3, APOS_MONO = -2, APOS_INVALID = -1, APOS_FL = 0, APOS_FR = 1, APOS_FC = 2, APOS_LFE1 = 3, APOS_RL = 4, APOS_RR = 5, APOS_FLOC = 6, APOS_FROC = 7,
  APOS_RC = 8, APOS_LFE2 = 9, APOS_SL = 10, APOS_SR = 11
};

typedef float AMixMatrix[GSTAMD_AUDIO_MAX_CHANNELS][GSTAMD_AUDIO_MAX_CHANNELS];

// fill_compatible (:163-253): mono <-> stereo pairs of the front, the centre pair and the rear
static void amix_fill_compatible (AMixMatrix m, int in_ch, const int *ip, int out_ch, const int *op)
{
  static const int conv[3][3] = { {APOS_FL, APOS_FR, APOS_MONO}, {APOS_FLOC, APOS_FROC, APOS_FC}, {APOS_RL, APOS_RR, APOS_RC} };
  for (int c = 0; c < 3; c++) {
    int a0 = -1, a1 = -1, a2 = -1, b0 = -1, b1 = -1, b2 = -1;
    for (int n = 0; n < in_ch; n++) {
      if (ip[n] == conv[c][0]) a0 = n;
      else if (ip[n] == conv[c][1]) a1 = n;
      else if (ip[n] == conv[c][2]) a2 = n;
    }
    for (int n = 0; n < out_ch; n++) {
      if (op[n] == conv[c][0]) b0 = n;
      else if (op[n] == conv[c][1]) b1 = n;
      else if (op[n] == conv[c][2]) b2 = n;
    }
    /* left -> centre, right -> centre */
    if (a0 != -1 && a2 == -1 && b0 == -1 && b2 != -1) m[a0][b2] = 1.0f;
    else if (a0 != -1 && a2 != -1 && b0 == -1 && b2 != -1) m[a0][b2] = 0.5f;
    else if (a0 != -1 && a2 == -1 && b0 != -1 && b2 != -1) m[a0][b2] = 1.0f;
    if (a1 != -1 && a2 == -1 && b1 == -1 && b2 != -1) m[a1][b2] = 1.0f;
    else if (a1 != -1 && a2 != -1 && b1 == -1 && b2 != -1) m[a1][b2] = 0.5f;
    else if (a1 != -1 && a2 == -1 && b1 != -1 && b2 != -1) m[a1][b2] = 1.0f;
    /* centre -> left, centre -> right */
    if (a2 != -1 && a0 == -1 && b2 == -1 && b0 != -1) m[a2][b0] = 1.0f;
    else if (a2 != -1 && a0 != -1 && b2 == -1 && b0 != -1) m[a2][b0] = 0.5f;
    else if (a2 != -1 && a0 == -1 && b2 != -1 && b0 != -1) m[a2][b0] = 1.0f;
    if (a2 != -1 && a1 == -1 && b2 == -1 && b1 != -1) m[a2][b1] = 1.0f;
    else if (a2 != -1 && a1 != -1 && b2 == -1 && b1 != -1) m[a2][b1] = 0.5f;
    else if (a2 != -1 && a1 == -1 && b2 != -1 && b1 != -1) m[a2][b1] = 1.0f;
  }
}

struct AMixGroups {             // detect_pos (:264-327): [0] left, [1] centre-ish, [2] right of the five families
  int f[3] = { -1, -1, -1 }, c[3] = { -1, -1, -1 }, r[3] = { -1, -1, -1 }, s[3] = { -1, -1, -1 }, b[3] = { -1, -1, -1 };
  bool has_f = false, has_c = false, has_r = false, has_s = false, has_b = false;
};

static void amix_detect (int ch, const int *pos, AMixGroups &g)
{
  for (int n = 0; n < ch; n++)
    switch (pos[n]) {
      case APOS_MONO: g.f[1] = n; g.has_f = true; break;
      case APOS_FL: g.f[0] = n; g.has_f = true; break;
      case APOS_FR: g.f[2] = n; g.has_f = true; break;
      case APOS_FC: g.c[1] = n; g.has_c = true; break;
      case APOS_FLOC: g.c[0] = n; g.has_c = true; break;
      case APOS_FROC: g.c[2] = n; g.has_c = true; break;
      case APOS_RC: g.r[1] = n; g.has_r = true; break;
      case APOS_RL: g.r[0] = n; g.has_r = true; break;
      case APOS_RR: g.r[2] = n; g.has_r = true; break;
      case APOS_SL: g.s[0] = n; g.has_s = true; break;
      case APOS_SR: g.s[2] = n; g.has_s = true; break;
      case APOS_LFE1: g.b[1] = n; g.has_b = true; break;
      default: break;
    }
}

// fill_one_other (:329-378); `ratio` is a gfloat there and 0.5 * ratio is rounded to float on the store
static void amix_one_other (AMixMatrix m, const int *from, const int *to, float ratio)
{
  const float half = (float) (0.5 * (double) ratio);
  if (from[1] != -1 && to[1] != -1) m[from[1]][to[1]] = ratio;
  if (from[0] != -1 && to[0] != -1) m[from[0]][to[0]] = ratio;
  if (from[2] != -1 && to[2] != -1) m[from[2]][to[2]] = ratio;
  if (from[0] != -1 && to[1] != -1) m[from[0]][to[1]] = from[1] != -1 ? half : ratio;
  if (from[2] != -1 && to[1] != -1) m[from[2]][to[1]] = from[1] != -1 ? half : ratio;
  if (from[1] != -1 && to[0] != -1) m[from[1]][to[0]] = from[0] != -1 ? half : ratio;
  if (from[1] != -1 && to[2] != -1) m[from[1]][to[2]] = from[2] != -1 ? half : ratio;
}

// fill_others (:398-590): a family one side lacks goes to / comes from the nearest family the other side has
static void amix_fill_others (AMixMatrix m, int in_ch, const int *ip, int out_ch, const int *op)
{
  AMixGroups i, o;
  amix_detect (in_ch, ip, i);
  amix_detect (out_ch, op, o);
  const double R2 = 1.0 / sqrt (2.0), R8 = 1.0 / sqrt (8.0);
  const double CENTER_FRONT = R2, CENTER_SIDE = 0.5, CENTER_REAR = R8, FRONT_SIDE = R2, FRONT_REAR = 0.5, SIDE_REAR = R2;
  const double CENTER_BASS = R2, FRONT_BASS = 1.0, SIDE_BASS = R2, REAR_BASS = R2;
  auto go = [&](const int *from, const int *to, double ratio) { amix_one_other (m, from, to, (float) ratio); };
  /* centre <-> front / side / rear */
  if (!i.has_c && i.has_f && o.has_c) go (i.f, o.c, CENTER_FRONT);
  else if (!i.has_c && !i.has_f && i.has_s && o.has_c) go (i.s, o.c, CENTER_SIDE);
  else if (!i.has_c && !i.has_f && !i.has_s && i.has_r && o.has_c) go (i.r, o.c, CENTER_REAR);
  else if (i.has_c && !o.has_c && o.has_f) go (i.c, o.f, CENTER_FRONT);
  else if (i.has_c && !o.has_c && !o.has_f && o.has_s) go (i.c, o.s, CENTER_SIDE);
  else if (i.has_c && !o.has_c && !o.has_f && !o.has_s && o.has_r) go (i.c, o.r, CENTER_REAR);
  /* front <-> centre / side / rear */
  if (!i.has_f && i.has_c && !i.has_s && o.has_f) go (i.c, o.f, CENTER_FRONT);
  else if (!i.has_f && !i.has_c && i.has_s && o.has_f) go (i.s, o.f, FRONT_SIDE);
  else if (!i.has_f && i.has_c && i.has_s && o.has_f) { go (i.c, o.f, 0.5 * CENTER_FRONT); go (i.s, o.f, 0.5 * FRONT_SIDE); }
  else if (!i.has_f && !i.has_c && !i.has_s && i.has_r && o.has_f) go (i.r, o.f, FRONT_REAR);
  else if (i.has_f && o.has_c && !o.has_s && !o.has_f) go (i.f, o.c, CENTER_FRONT);
  else if (i.has_f && !o.has_c && o.has_s && !o.has_f) go (i.f, o.s, FRONT_SIDE);
  else if (i.has_f && o.has_c && o.has_s && !o.has_f) { go (i.f, o.c, 0.5 * CENTER_FRONT); go (i.f, o.s, 0.5 * FRONT_SIDE); }
  else if (i.has_f && !o.has_c && !o.has_s && !o.has_f && o.has_r) go (i.f, o.r, FRONT_REAR);
  /* side <-> centre / front / rear */
  if (!i.has_s && i.has_f && !i.has_r && o.has_s) go (i.f, o.s, FRONT_SIDE);
  else if (!i.has_s && !i.has_f && i.has_r && o.has_s) go (i.r, o.s, SIDE_REAR);
  else if (!i.has_s && i.has_f && i.has_r && o.has_s) { go (i.f, o.s, 0.5 * FRONT_SIDE); go (i.r, o.s, 0.5 * SIDE_REAR); }
  else if (!i.has_s && !i.has_f && !i.has_r && i.has_c && o.has_s) go (i.c, o.s, CENTER_SIDE);
  else if (i.has_s && o.has_f && !o.has_r && !o.has_s) go (i.s, o.f, FRONT_SIDE);
  else if (i.has_s && !o.has_f && o.has_r && !o.has_s) go (i.s, o.r, SIDE_REAR);
  else if (i.has_s && o.has_f && o.has_r && !o.has_s) { go (i.s, o.f, 0.5 * FRONT_SIDE); go (i.s, o.r, 0.5 * SIDE_REAR); }
  else if (i.has_s && !o.has_f && !o.has_r && o.has_c && !o.has_s) go (i.s, o.c, CENTER_SIDE);
  /* rear <-> centre / front / side */
  if (!i.has_r && i.has_s && o.has_r) go (i.s, o.r, SIDE_REAR);
  else if (!i.has_r && !i.has_s && i.has_f && o.has_r) go (i.f, o.r, FRONT_REAR);
  else if (!i.has_r && !i.has_s && !i.has_f && i.has_c && o.has_r) go (i.c, o.r, CENTER_REAR);
  else if (i.has_r && !o.has_r && o.has_s) go (i.r, o.s, SIDE_REAR);
  else if (i.has_r && !o.has_r && !o.has_s && o.has_f) go (i.r, o.f, FRONT_REAR);
  else if (i.has_r && !o.has_r && !o.has_s && !o.has_f && o.has_c) go (i.r, o.c, CENTER_REAR);
  /* bass <-> any */
  if (i.has_b && !o.has_b) {
    if (o.has_c) go (i.b, o.c, CENTER_BASS);
    if (o.has_f) go (i.b, o.f, FRONT_BASS);
    if (o.has_s) go (i.b, o.s, SIDE_BASS);
    if (o.has_r) go (i.b, o.r, REAR_BASS);
  } else if (!i.has_b && o.has_b) {
    if (i.has_c) go (i.c, o.b, CENTER_BASS);
    if (i.has_f) go (i.f, o.b, FRONT_BASS);
    if (i.has_s) go (i.s, o.b, REAR_BASS);              /* the reference uses the rear ratio for the sides here (:581) */
    if (i.has_r) go (i.r, o.b, REAR_BASS);
  }
}

// fill_normalize (:596-626): float sums of |m| per output channel, every entry divided by the largest
static void amix_normalize (AMixMatrix m, int in_ch, int out_ch)
{
  float top = 0;
  for (int j = 0; j < out_ch; j++) {
    float sum = 0.0f;
    for (int i = 0; i < in_ch; i++)
      sum = (float) ((double) sum + fabs ((double) m[i][j]));
    if (sum > top)
      top = sum;
  }
  if (top == 0.0f)
    return;
  for (int j = 0; j < out_ch; j++)
    for (int i = 0; i < in_ch; i++)
      m[i][j] /= top;
}

static void default_mix_matrix (const GstAmdAudioInfo &in, const GstAmdAudioInfo &out, AMixMatrix m)
{
  memset (m, 0, sizeof (AMixMatrix));
  const int *ip = in.position, *op = out.position;
  const int in_ch = in.channels, out_ch = out.channels;
  /* fill_special (:628-656) */
  if (in_ch == 2 && out_ch == 1 && ((ip[0] == APOS_FL && ip[1] == APOS_FR) || (ip[0] == APOS_FR && ip[1] == APOS_FL)) && op[0] == APOS_MONO) {
    m[0][0] = m[1][0] = 0.5f;
    return;
  }
  if (in_ch == 1 && out_ch == 2 && ((op[0] == APOS_FL && op[1] == APOS_FR) || (op[0] == APOS_FR && op[1] == APOS_FL)) && ip[0] == APOS_MONO) {
    m[0][0] = m[0][1] = 1.0f;
    return;
  }
  /* virtual inputs (:684-729): all-mono inputs count as one mono channel, alternating left / right ones as one stereo pair */
  int in_size = in_ch, virt = 0;
  if (in_ch >= 2) {
    bool mono = true, alt = true;
    for (int i = 0; i < in_ch; i++) {
      mono = mono && ip[i] == APOS_MONO;
      alt = alt && ip[i] == (i & 1 ? APOS_FR : APOS_FL);
    }
    if (mono) { virt = 1; in_size = 1; }
    else if (alt && in_ch > 2) { virt = 2; in_size = 2; }
  }
  /* fill_identical (:130-155) */
  for (int co = 0; co < out_ch; co++)
    for (int ci = 0; ci < in_size; ci++) {
      if (in.unpositioned)
        m[ci][co] = ci == co ? 1.0f : 0.0f;
      else if (ip[ci] == op[co])
        m[ci][co] = 1.0f;
    }
  if (!in.unpositioned) {
    amix_fill_compatible (m, in_size, ip, out_ch, op);
    amix_fill_others (m, in_size, ip, out_ch, op);
    amix_normalize (m, in_size, out_ch);
  }
  if (virt == 1) {
    for (int o = 0; o < out_ch; o++)
      m[0][o] /= (float) in_ch;
    for (int i = 1; i < in_ch; i++)
      memcpy (m[i], m[0], sizeof (float) * (size_t) out_ch);
  } else if (virt == 2) {
    const int right = in_ch >> 1, left = right + (in_ch % 2);
    for (int o = 0; o < out_ch; o++) {
      m[0][o] /= (float) left;
      m[1][o] /= (float) right;
    }
    for (int i = 2; i < in_ch; i++)
      memcpy (m[i], m[i % 2], sizeof (float) * (size_t) out_ch);
  }
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
    default_mix_matrix (*in, *out, p.m);
  }
  /* gst_audio_channel_mixer_build_sparse_matrix (:1063-1122): with fewer than half of the coefficients above 1e-6 the mixer walks a
     list of those only - the others are not summed at all (which matters for float samples: inf * 0) */
  {
    int pairs = 0;
    for (int ci = 0; ci < in->channels; ci++)
      for (int co = 0; co < out->channels; co++)
        if (fabsf (p.m[ci][co]) > 1e-6f)
          pairs++;
    p.sparse = (double) pairs / (double) (in->channels * out->channels) < 0.5 ? 1 : 0;
    for (int co = 0; co < out->channels; co++) {
      p.use[co] = 0;
      for (int ci = 0; ci < in->channels; ci++)
        if (!p.sparse || fabsf (p.m[ci][co]) > 1e-6f)
          p.use[co] |= 1u << ci;
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
      /* gst_audio_quantize_setup_noise_shaping :344-373 */
      static const double ns_simple[] = { -0.5, 1.0 };
      static const double ns_medium[] = { 0.6149, -1.590, 1.959, -2.165, 2.033 };
      static const double ns_high[] = { -0.340122, 0.876066, -1.72008, 2.61339, -3.31399, 3.27918, -2.92975, 2.08484 };
      const double *cf = ns == 2 ? ns_simple : ns == 3 ? ns_medium : ns_high;
      p.ns = ns;
      p.n_coeffs = ns == 2 ? 2 : ns == 3 ? 5 : ns == 4 ? 8 : 0;
      for (int i = 0; i < p.n_coeffs; i++)
        p.coeffs[i] = (int32_t) floor (cf[i] * 1024.0 + 0.5);
      if (ns < 0 || ns > 4 || dither < 0 || dither > 3) {
        *err = "unknown dither / noise shaping method";
        return GSTAMD_ERR_INVALID;
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

