// video_gamma.h - the linear-light part of GstVideoConverter with gamma-mode = remap, per pixel (bodies shared with tests/emu):
//   do_convert_to_RGB_lines  video-converter.c:3070-3093   8-bit matrix to R'G'B' (prepare_matrix's table / matrix8 / AYUV_ARGB forms),
//                                                          then gamma_convert_u8_u16 (:1444-1460): table[component], alpha a << 8 | a
//   do_convert_lines         :3095-3140                    video_converter_matrix16 with the primaries matrix (:1296-1320)
//   do_alpha_lines           :3142-3160                    convert_set_alpha_u16 / convert_mult_alpha_u16 (:1882-1908)
//   do_convert_to_YUV_lines  :3162-3190                    gamma_convert_u16_u8 (:1462-1478): table[component], alpha >> 8; then the 8-bit
//                                                          matrix to Y'CbCr
// An ARGB64 pixel is two words as in video_deep.h: {A | c1 << 16, c2 | c3 << 16}.
#pragma once
#include "video_deep.h"

namespace gstamd {

// GAMMA_STAGE_DEC16 / _ENC16: the same stages when the unpack / pack format is a 16-bit one (round 3): video_converter_matrix16 with the
// to-R'G'B' matrix prepared for 16 bits, then gamma_convert_u16_u16 (:1480-1494: table[component] on 65536 entries, alpha copied) - and on
// the way out the 16 -> 16 encode table followed by matrix16 to Y'CbCr; source / destination are AYUV64 / ARGB64 images then.
enum { GAMMA_STAGE_DEC = 1, GAMMA_STAGE_MID = 2, GAMMA_STAGE_ENC = 4, GAMMA_STAGE_DEC16 = 8, GAMMA_STAGE_ENC16 = 16 };

struct GammaDev {
  MatrixParams to_rgb, to_yuv;
  Deep16Params prim;
  int alpha_kind;
  unsigned alpha_value;
  const uint16_t *dec;          // [256]
  const uint8_t *enc;           // [65536]
  const uint8_t *comp;          // [256] enc[dec[v]] or NULL (GammaPlan::comp)
  Deep16Params to_rgb16, to_yuv16;
  const uint16_t *dec16;        // [65536]
  const uint16_t *enc16;        // [65536]
};

// video_converter_matrix16 (video-converter.c:1296-1320) on the colour components of an ARGB64 pixel
// rows of video_converter_matrix16 on 16-bit samples: v_mad_i32_i24 when every coefficient is a 24-bit operand (wave-uniform test), the 32-bit multiply
// otherwise - the same low 32 bits either way
GSTAMD_HD bool matrix16_fits24 (const Deep16Params &m)
{
  bool fits = true;
  for (int k = 0; k < 3; k++)
    for (int j = 0; j < 3; j++)
      fits = fits && m.im[k][j] > -(1 << 23) && m.im[k][j] < (1 << 23);
  return fits;
}

GSTAMD_HD int matrix16_row (const Deep16Params &m, bool fits24, int k, int r, int g, int b)
{
  const int s = fits24 ? mul24s (m.im[k][0], r) + mul24s (m.im[k][1], g) + mul24s (m.im[k][2], b) : m.im[k][0] * r + m.im[k][1] * g + m.im[k][2] * b;
  return clampi ((s + m.im[k][3]) >> 8, 0, 65535);
}

GSTAMD_HD uint2 gamma_matrix16 (const Deep16Params &m, uint2 px)
{
  if (!m.has_matrix)
    return px;
  const int r = (int) (px.x >> 16), gg = (int) (px.y & 0xffffu), b = (int) (px.y >> 16);
  const bool f24 = matrix16_fits24 (m);
  const int c1 = matrix16_row (m, f24, 0, r, gg, b), c2 = matrix16_row (m, f24, 1, r, gg, b), c3 = matrix16_row (m, f24, 2, r, gg, b);
  uint2 o;
  o.x = (px.x & 0xffffu) | ((uint32_t) c1 << 16);
  o.y = (uint32_t) c2 | ((uint32_t) c3 << 16);
  return o;
}

GSTAMD_HD uint2 gamma_lut16 (const uint16_t *t, uint2 px)
{
  uint2 o;
  o.x = (px.x & 0xffffu) | ((uint32_t) t[px.x >> 16] << 16);
  o.y = (uint32_t) t[px.y & 0xffffu] | ((uint32_t) t[px.y >> 16] << 16);
  return o;
}

GSTAMD_HD uint2 gamma_dec_px (const GammaDev &g, uint32_t px)
{
  px = apply_matrix (g.to_rgb, px);
  const uint32_t a = px & 0xff;
  uint2 r;
  r.x = ((a << 8) | a) | ((uint32_t) g.dec[(px >> 8) & 0xff] << 16);
  r.y = (uint32_t) g.dec[(px >> 16) & 0xff] | ((uint32_t) g.dec[px >> 24] << 16);
  return r;
}

GSTAMD_HD uint2 gamma_mid_px (const GammaDev &g, uint2 px)
{
  int a = (int) (px.x & 0xffffu), c1 = (int) (px.x >> 16), c2 = (int) (px.y & 0xffffu), c3 = (int) (px.y >> 16);
  if (g.prim.has_matrix) {
    const int r = c1, gg = c2, b = c3;
    const bool f24 = matrix16_fits24 (g.prim);
    c1 = matrix16_row (g.prim, f24, 0, r, gg, b);
    c2 = matrix16_row (g.prim, f24, 1, r, gg, b);
    c3 = matrix16_row (g.prim, f24, 2, r, gg, b);
  }
  if (g.alpha_kind == ALPHA_SET) {
    const unsigned v = g.alpha_value < 255u ? g.alpha_value : 255u;
    a = (int) ((v | (v << 8)) & 0xffffu);
  } else if (g.alpha_kind == ALPHA_MULT) {
    a = clampi ((int) (((unsigned) a * g.alpha_value) / 255u), 0, 65535);
  }
  uint2 r;
  r.x = (uint32_t) a | ((uint32_t) c1 << 16);
  r.y = (uint32_t) c2 | ((uint32_t) c3 << 16);
  return r;
}

GSTAMD_HD uint32_t gamma_enc_px (const GammaDev &g, uint2 px)
{
  const uint32_t p = ((px.x & 0xffffu) >> 8) | ((uint32_t) g.enc[px.x >> 16] << 8) | ((uint32_t) g.enc[px.y & 0xffffu] << 16) |
      ((uint32_t) g.enc[px.y >> 16] << 24);
  return apply_matrix (g.to_yuv, p);
}

// GammaPlan::lut_direct: the composed table on the colour bytes of a finished 4-byte RGB pixel (`keep`: the byte that is alpha / filler)
GSTAMD_HD uint32_t gamma_lut3_px (const uint8_t *comp, uint32_t px, int keep)
{
  uint32_t r = 0;
#pragma unroll
  for (int b = 0; b < 4; b++) {
    const uint32_t v = (px >> (8 * b)) & 0xffu;
    r |= (b == keep ? v : (uint32_t) comp[v]) << (8 * b);
  }
  return r;
}

// the whole per-pixel chain as the step convert_body takes between its colour stage and its packer (k_convert_gamma)
struct GammaChainFn {
  GammaDev g;
  GSTAMD_HD uint32_t operator() (uint32_t px) const
  {
    if (g.comp) {               /* decode table, nothing, encode table: one 256-entry table per component; alpha (a << 8 | a) >> 8 = a */
      px = apply_matrix (g.to_rgb, px);
      px = (px & 0xffu) | ((uint32_t) g.comp[(px >> 8) & 0xff] << 8) | ((uint32_t) g.comp[(px >> 16) & 0xff] << 16) | ((uint32_t) g.comp[px >> 24] << 24);
      return apply_matrix (g.to_yuv, px);
    }
    return gamma_enc_px (g, gamma_mid_px (g, gamma_dec_px (g, px)));
  }
};

// one pixel of a stage launch: the stages of `mask` in order; the source is an 8-bit image when the mask starts with the decode, the
// destination an 8-bit image when it ends with the encode
GSTAMD_HD void gamma_stage_px (const GammaDev &g, int mask, const uint8_t *src, int sstride, uint8_t *dst, int dstride, int x, int y)
{
  uint2 v;
  if (mask & GAMMA_STAGE_DEC)
    v = gamma_dec_px (g, *(const uint32_t *) (src + (ptrdiff_t) y * sstride + 4 * (ptrdiff_t) x));
  else
    v = *(const uint2 *) (src + (ptrdiff_t) y * sstride + 8 * (ptrdiff_t) x);
  if (mask & GAMMA_STAGE_DEC16)
    v = gamma_lut16 (g.dec16, gamma_matrix16 (g.to_rgb16, v));
  if (mask & GAMMA_STAGE_MID)
    v = gamma_mid_px (g, v);
  if (mask & GAMMA_STAGE_ENC16)
    v = gamma_matrix16 (g.to_yuv16, gamma_lut16 (g.enc16, v));
  if (mask & GAMMA_STAGE_ENC)
    *(uint32_t *) (dst + (ptrdiff_t) y * dstride + 4 * (ptrdiff_t) x) = gamma_enc_px (g, v);
  else
    *(uint2 *) (dst + (ptrdiff_t) y * dstride + 8 * (ptrdiff_t) x) = v;
}

}  // namespace gstamd
