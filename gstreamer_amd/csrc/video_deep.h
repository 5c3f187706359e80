// video_deep.h - the 16-bit chain of GstVideoConverter for 10-bit sources into an 8-bit 4-byte destination (scaled or not):
//   unpack_I420_10LE / unpack_P010_10LE  -> AYUV64   (video-format.c:3836-3873, 5331-5400: 10 bits widened to 16 by bit replication)
//   video_chroma_up_h2_u16 / _h2_cs_u16 / _v2_u16    (video-chroma.c:277-327, 687-699, instantiated for guint16 at :469-471, 796:
//                                                     the same (3a+b+2)>>2 and (a+b+1)>>1 on 16-bit values, same line pairing)
//   video_converter_matrix16                         (video-converter.c:1296-1320: (im . px + offset) >> 8, clamped to 0..65535)
//   video_orc_convert_u16_to_u8                      (do_convert_lines :3133: the high byte of every component)
//   alpha / pack as in the 8-bit chain
// First version: one lane = 4 pixels of one row, per-pixel loads (correctness and coverage first, like video_planes.h).
#pragma once
#include "video_device.h"
#include "video_dither.h"

namespace gstamd {

// a stored 16-bit little-endian word -> the unpacked 16-bit value (FormatDesc::hi_depth: 1 / 4 = 10 / 12 bits in the low bits,
// 2 / 5 = in the high bits, 6 = all 16)
GSTAMD_HD int deep_widen (int hi_depth, int v)
{
  if (hi_depth_be (hi_depth)) {           /* GST_READ_UINT16_BE, then the little-endian form's arithmetic */
    v = bswap16i (v);
    hi_depth = hi_depth_le (hi_depth);
  }
  if (hi_depth == 6)
    return v;                           // P016_LE, Y444_16LE
  const int bits = hi_depth_bits (hi_depth);
  if (hi_depth == 1 || hi_depth == 4) { // I420_10LE: Y = v << 6 (as guint16), Y |= Y >> 10; I420_12LE: << 4, >> 12
    const int t = (v << (16 - bits)) & 0xffff;
    return t | (t >> bits);
  }
  return v | (v >> bits);               // P010_10LE / P012_LE: value in the high bits
}

// the stored word of a UNPACK_Y410 format as the little-endian one the fields are cut from (hi_depth 27, r210: GST_READ_UINT32_BE)
GSTAMD_HD uint32_t y410_word (int hi_depth, uint32_t w)
{
  return hi_depth == 27 ? (w >> 24) | ((w >> 8) & 0xff00u) | ((w << 8) & 0xff0000u) | (w << 24) : w;
}

// sample n of a row that holds three 10-bit samples per little-endian 32-bit word (unpack_GRAY10_LE32 / _NV12_10LE32, video-format.c:5490-5660), widened
// like every 10-bit sample: (v << 6) | (v >> 4)
GSTAMD_HD int le32_sample (const uint8_t *row, int n)
{
  const uint32_t w = ((const uint32_t *) row)[n / 3];
  const int t = (int) ((w >> (10 * (n % 3))) & 0x3ffu) << 6;
  return t | (t >> 10);
}

// sample n of a row that is a little-endian stream of 10-bit samples (unpack_NV12_10LE40 video-format.c:6023-6098: four samples in five bytes), widened
GSTAMD_HD int le40_sample (const uint8_t *row, int n)
{
  const int bit = 10 * n;
  const uint32_t v = (uint32_t) row[bit >> 3] | ((uint32_t) row[(bit >> 3) + 1] << 8);
  const int t = (int) ((v >> (bit & 7)) & 0x3ffu) << 6;
  return t | (t >> 10);
}

// field idx (0 U, 1 Y0, 2 V, 3 Y1) of a UYVP macropixel - 40 bits, big endian, ten per sample (unpack_UYVP video-format.c:2044-2085) -, widened
GSTAMD_HD int uyvp_field (const uint8_t *m, int idx)
{
  const unsigned long long bits = ((unsigned long long) m[0] << 32) | ((unsigned long long) m[1] << 24) | ((unsigned long long) m[2] << 16) | ((unsigned long long) m[3] << 8) | m[4];
  const int t = (int) ((bits >> (30 - 10 * idx)) & 0x3ffu) << 6;
  return t | (t >> 10);
}

// one 10-bit field of a Y410 word (unpack_Y410 video-format.c:863-896): (field << 6) | (field >> 4)
GSTAMD_HD int y410_field (uint32_t w, int shift)
{
  const int t = (int) ((w >> shift) & 0x3ffu) << 6;
  return t | (t >> 10);
}

GSTAMD_HD UV deep_load_uv (const FrontParams &f, const Planes &pl, int crow, int k)
{
  UV r;
  if (f.kind == UNPACK_P422_16) {        // unpack_Y210 / _Y212_LE: U, V of macropixel k, widened like P010's samples
    const uint16_t *p = (const uint16_t *) (pl.p[0] + (ptrdiff_t) crow * pl.stride[0]) + 4 * k;
    r.u = deep_widen (f.hi_depth, p[f.pos[2]]);
    r.v = deep_widen (f.hi_depth, p[f.pos[3]]);
  } else if (f.kind == UNPACK_Y410) {
    const uint32_t w = y410_word (f.hi_depth, ((const uint32_t *) (pl.p[0] + (ptrdiff_t) crow * pl.stride[0]))[k]);
    r.u = y410_field (w, f.pos[2]);          /* FormatDesc::pos of these formats: bit of the 10-bit field of component c1, c2, c3 */
    r.v = y410_field (w, f.pos[3]);
  } else if (f.kind == UNPACK_PACKED64) { // unpack_RGBA64_LE & co (video-format.c:2470-2815): word pos[c] of the pixel, read little / big endian
    const uint16_t *p = (const uint16_t *) (pl.p[0] + (ptrdiff_t) crow * pl.stride[0]) + 4 * k;
    r.u = px16_load (f.hi_depth, p[f.pos[2]]);
    r.v = px16_load (f.hi_depth, p[f.pos[3]]);
  } else if (f.kind == UNPACK_GRAY16 || f.kind == UNPACK_GRAY_LE32) {
    r.u = r.v = 0x8000;
  } else if (f.kind == UNPACK_SEMI_LE32) { // the UV plane's samples run U0 V0 U1 V1 ...: pair k at samples 2 k, 2 k + 1
    const uint8_t *row = pl.p[1] + (ptrdiff_t) crow * pl.stride[1];
    r.u = le32_sample (row, 2 * k);
    r.v = le32_sample (row, 2 * k + 1);
  } else if (f.kind == UNPACK_P422_UYVP) {
    const uint8_t *m = pl.p[0] + (ptrdiff_t) crow * pl.stride[0] + 5 * (ptrdiff_t) k;
    r.u = uyvp_field (m, 0);
    r.v = uyvp_field (m, 2);
  } else if (f.kind == UNPACK_SEMI_LE40_TILED) {      // the pair's two samples in its UV tile row (4 x 4 tiles: two pairs a row)
    const uint8_t *row = pl.p[1] + tiled_uv_row (f.pos, pl.stride[1], 5, k, crow);
    const int j = ((2 * k) & ((1 << f.pos[1]) - 1)) >> 1;
    r.u = le40_sample (row, 2 * j);
    r.v = le40_sample (row, 2 * j + 1);
  } else if (f.kind == UNPACK_SEMI_LE40) {
    const uint8_t *row = pl.p[1] + (ptrdiff_t) crow * pl.stride[1];
    r.u = le40_sample (row, 2 * k);
    r.v = le40_sample (row, 2 * k + 1);
  } else if (f.kind == UNPACK_V210) {     // unpack_v210 (video-format.c:560-649): chroma pair k of the line, pair k % 3 of group k / 3
    const uint32_t *g = (const uint32_t *) (pl.p[0] + (ptrdiff_t) crow * pl.stride[0]) + 4 * (k / 3);
    const int j = k % 3;
    r.u = j == 0 ? y410_field (g[0], 0) : (j == 1 ? y410_field (g[1], 10) : y410_field (g[2], 20));
    r.v = j == 0 ? y410_field (g[0], 20) : (j == 1 ? y410_field (g[2], 0) : y410_field (g[3], 10));
  } else if (f.kind == UNPACK_SEMI) {
    const uint16_t *p = (const uint16_t *) (pl.p[1] + (ptrdiff_t) crow * pl.stride[1]) + 2 * k;
    r.u = deep_widen (f.hi_depth, p[f.u_plane ? 0 : 1]);
    r.v = deep_widen (f.hi_depth, p[f.u_plane ? 1 : 0]);
  } else {
    r.u = deep_widen (f.hi_depth, ((const uint16_t *) (pl.p[f.u_plane] + (ptrdiff_t) crow * pl.stride[f.u_plane]))[k]);
    r.v = deep_widen (f.hi_depth, ((const uint16_t *) (pl.p[f.v_plane] + (ptrdiff_t) crow * pl.stride[f.v_plane]))[k]);
  }
  return r;
}

// horizontally filtered chroma of chroma row `crow` at luma position x (chroma_h_at on 16-bit samples)
GSTAMD_HD UV deep_chroma_h_at (const FrontParams &f, const Planes &pl, int crow, int x)
{
  if (f.w_sub == 0)
    return deep_load_uv (f, pl, crow, x);
  const int k = x >> 1, w = f.width;
  UV c = deep_load_uv (f, pl, crow, k);
  if (f.chroma_h == CHROMA_H_H2_CS) {
    if ((x & 1) && x < w - 1) {
      const UV n = deep_load_uv (f, pl, crow, k + 1);
      c.u = (c.u + n.u + 1) >> 1;
      c.v = (c.v + n.v + 1) >> 1;
    }
  } else if (f.chroma_h == CHROMA_H_H2) {
    if ((x & 1) && x < w - 1) {
      const UV n = deep_load_uv (f, pl, crow, k + 1);
      c.u = (3 * c.u + n.u + 2) >> 2;
      c.v = (3 * c.v + n.v + 2) >> 2;
    } else if (!(x & 1) && x >= 2) {
      const UV pv = deep_load_uv (f, pl, crow, k - 1);
      c.u = (pv.u + 3 * c.u + 2) >> 2;
      c.v = (pv.v + 3 * c.v + 2) >> 2;
    }
  }
  return c;
}

// pixel (x, y) after unpack + chroma upsampling: the AYUV64 pixel as two words, {A | c1 << 16, c2 | c3 << 16} (memory order A, c1, c2, c3)
GSTAMD_HD uint2 deep_front_px (const FrontParams &f, const Planes &pl, const int *__restrict__ vpair, int x, int y)
{
  int c1, a = 0xffff;
  if (f.kind == UNPACK_P422_16) {
    /* unpack_Y210 (video-format.c:783-806): the pair loop widens Y0, U and V - not Y1, which keeps its low bits clear; the last pixel of an
       odd-width line is a Y0 */
    const int raw = ((const uint16_t *) (pl.p[0] + (ptrdiff_t) y * pl.stride[0]))[4 * (x >> 1) + f.pos[1] + 2 * (x & 1)];
    c1 = (x & 1) ? (hi_depth_be (f.hi_depth) ? bswap16i (raw) : raw) : deep_widen (f.hi_depth, raw);
  } else if (f.kind == UNPACK_Y410) {
    const uint32_t w = y410_word (f.hi_depth, ((const uint32_t *) (pl.p[0] + (ptrdiff_t) y * pl.stride[0]))[x]);
    c1 = y410_field (w, f.pos[1]);
    const int t = (int) (w >> 30) << 14;                /* A: two bits, A |= A >> 10 */
    a = f.hi_depth == 27 ? 0xffff : t | (t >> 10);       /* (r210 has none) */
  } else if (f.kind == UNPACK_PACKED64) {
    const uint16_t *p = (const uint16_t *) (pl.p[0] + (ptrdiff_t) y * pl.stride[0]) + 4 * x;
    c1 = px16_load (f.hi_depth, p[f.pos[1]]);
    a = px16_load (f.hi_depth, p[f.pos[0]]);
  } else if (f.kind == UNPACK_GRAY16) {
    const int raw = ((const uint16_t *) (pl.p[0] + (ptrdiff_t) y * pl.stride[0]))[x];
    c1 = f.hi_depth == 9 || f.hi_depth == 10 ? px16_word (f.hi_depth, raw) : deep_widen (f.hi_depth, raw);         /* GRAY16_LE / _BE, GRAY10_LE16 */
  } else if (f.kind == UNPACK_P422_UYVP) {
    c1 = uyvp_field (pl.p[0] + (ptrdiff_t) y * pl.stride[0] + 5 * (ptrdiff_t) (x >> 1), 1 + 2 * (x & 1));
  } else if (f.kind == UNPACK_SEMI_LE40_TILED) {
    c1 = le40_sample (pl.p[0] + tiled_luma_row (f.pos, pl.stride[0], 5, x, y), x & ((1 << f.pos[1]) - 1));
  } else if (f.kind == UNPACK_SEMI_LE40) {
    c1 = le40_sample (pl.p[0] + (ptrdiff_t) y * pl.stride[0], x);
  } else if (GSTAMD_KIND_LE32 (f.kind)) {
    c1 = le32_sample (pl.p[0] + (ptrdiff_t) y * pl.stride[0], x);
  } else if (f.kind == UNPACK_V210) {   // luma j of the group: words 0 1 1 2 3 3 at bit 10 0 20 10 0 20
    const uint32_t *g = (const uint32_t *) (pl.p[0] + (ptrdiff_t) y * pl.stride[0]) + 4 * (x / 6);
    const int j = x % 6;
    c1 = y410_field (g[j < 1 ? 0 : (j < 3 ? 1 : (j < 4 ? 2 : 3))], j == 0 || j == 3 ? 10 : (j == 1 || j == 4 ? 0 : 20));
  } else {
    c1 = deep_widen (f.hi_depth, ((const uint16_t *) (pl.p[0] + (ptrdiff_t) y * pl.stride[0]))[x]);
    if (f.kind == UNPACK_PLANAR_A)        /* unpack_A420_16 / _A422_16 / _A444_16 (video-format.c:4645-5040), unpack_GBRA_10LE :3240: the alpha plane widened like the others */
      a = deep_widen (f.hi_depth, ((const uint16_t *) (pl.p[3] + (ptrdiff_t) y * pl.stride[3]))[x]);
  }
  UV c;
  if (f.chroma_v2) {
    const VPairW t = vpair_get (vpair, y, f.chroma_v2);
    const int ra = t.ra, rb = t.rb;
    const UV a = deep_chroma_h_at (f, pl, ra, x);
    if (ra == rb) {
      c = a;
    } else {
      const UV b = deep_chroma_h_at (f, pl, rb, x);          // (3 a + b + 2) >> 2 = (6 a + 2 b + 4) >> 3: vpair_get
      c.u = (t.wa * a.u + (8 - t.wa) * b.u + 4) >> 3;
      c.v = (t.wa * a.v + (8 - t.wa) * b.v + 4) >> 3;
    }
  } else {
    c = deep_chroma_h_at (f, pl, y >> f.h_sub, x);
  }
  uint2 r;
  r.x = (uint32_t) a | ((uint32_t) c1 << 16);
  r.y = (uint32_t) c.u | ((uint32_t) c.v << 16);
  return r;
}

// ---- the same for four neighbouring pixels x0 .. x0+3 of one row with the loads shared (planar / semi-planar sources with horizontally subsampled
// chroma, x0 a multiple of 4, x0 + 4 <= width): the four lumas in one 8-byte load, the chroma samples k0-1 .. k0+2 of a chroma row once for all four
// pixels.  Same arithmetic as deep_front_px / deep_chroma_h_at, value for value.
// deep_widen with the format decoded once: t = (v << sh) & 0xffff; t | t >> bits  (sh = 16 - bits for samples in the low bits, 0 otherwise; the 16-bit
// formats come out unchanged)
struct Widen { int sh, bits; };
GSTAMD_HD Widen deep_widen_params (int hi_depth)
{
  Widen w;
  w.bits = hi_depth_bits (hi_depth);
  w.sh = hi_depth == 1 || hi_depth == 4 ? 16 - w.bits : 0;
  return w;
}
GSTAMD_HD int deep_widen_w (const Widen &w, int v)
{
  const int t = (v << w.sh) & 0xffff;
  return t | (t >> w.bits);
}

// a * b on 24-bit operands (the matrix coefficients checked by the caller, samples have 16 bits): v_mul_i32_i24 / v_mad_i32_i24 at full rate, the low
// 32 bits of the product - what the 32-bit multiply of video_converter_matrix16 keeps
GSTAMD_HD int mul24s (int a, int b)
{
#ifdef __HIPCC__
  return __mul24 (a, b);
#else
  return (int) ((uint32_t) a * (uint32_t) b);
#endif
}

struct UV4 { int u[4], v[4]; };         // samples k0-1, k0, k0+1, k0+2 of a chroma row (indices clamped into the row: the formulas never use a clamped one)

GSTAMD_HD UV4 deep_load_uv4 (const FrontParams &f, const Planes &pl, const Widen &wd, int crow, int k0, int cw)
{
  UV4 r;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int k = k0 - 1 + i;
    k = k < 0 ? 0 : (k > cw - 1 ? cw - 1 : k);
    if (f.kind == UNPACK_SEMI) {
      const uint32_t w = ((const uint32_t *) (pl.p[1] + (ptrdiff_t) crow * pl.stride[1]))[k];
      const int lo = (int) (w & 0xffffu), hi = (int) (w >> 16);
      r.u[i] = deep_widen_w (wd, f.u_plane ? lo : hi);
      r.v[i] = deep_widen_w (wd, f.u_plane ? hi : lo);
    } else {
      r.u[i] = deep_widen_w (wd, ((const uint16_t *) (pl.p[f.u_plane] + (ptrdiff_t) crow * pl.stride[f.u_plane]))[k]);
      r.v[i] = deep_widen_w (wd, ((const uint16_t *) (pl.p[f.v_plane] + (ptrdiff_t) crow * pl.stride[f.v_plane]))[k]);
    }
  }
  return r;
}

// deep_chroma_h_at for pixel x0 + i (i = 0 .. 3) from the row's four samples
GSTAMD_HD UV deep_chroma_h4 (const FrontParams &f, const UV4 &s, int x0, int i)
{
  const int x = x0 + i, j = 1 + (i >> 1), w = f.width;           /* s.u[j] is sample k = x >> 1 */
  UV c;
  c.u = s.u[j], c.v = s.v[j];
  if (f.chroma_h == CHROMA_H_H2_CS) {
    if ((x & 1) && x < w - 1) {
      c.u = (s.u[j] + s.u[j + 1] + 1) >> 1;
      c.v = (s.v[j] + s.v[j + 1] + 1) >> 1;
    }
  } else if (f.chroma_h == CHROMA_H_H2) {
    if ((x & 1) && x < w - 1) {
      c.u = (3 * s.u[j] + s.u[j + 1] + 2) >> 2;
      c.v = (3 * s.v[j] + s.v[j + 1] + 2) >> 2;
    } else if (!(x & 1) && x >= 2) {
      c.u = (s.u[j - 1] + 3 * s.u[j] + 2) >> 2;
      c.v = (s.v[j - 1] + 3 * s.v[j] + 2) >> 2;
    }
  }
  return c;
}

GSTAMD_HD bool deep_front4_usable (const FrontParams &f, int x0)
{
  return kind_has_planes (f.kind) && !hi_depth_be (f.hi_depth) && f.w_sub == 1 && (x0 & 3) == 0 && x0 + 4 <= f.width;
}

GSTAMD_HD void deep_front4 (const FrontParams &f, const Planes &pl, const int *__restrict__ vpair, int x0, int y, uint2 *out)
{
  struct __attribute__ ((aligned (4))) L4 { uint16_t v[4]; };
  const L4 l = *(const L4 *) (pl.p[0] + (ptrdiff_t) y * pl.stride[0] + 2 * (ptrdiff_t) x0);
  const int cw = (f.width + 1) >> 1, k0 = x0 >> 1;
  int ra = y >> f.h_sub, rb = ra, wa = 6;
  if (f.chroma_v2) {
    const VPairW t = vpair_get (vpair, y, f.chroma_v2);
    ra = t.ra, rb = t.rb, wa = t.wa;
  }
  const Widen wd = deep_widen_params (f.hi_depth);
  const UV4 a = deep_load_uv4 (f, pl, wd, ra, k0, cw);
  UV4 b = a;
  if (ra != rb)
    b = deep_load_uv4 (f, pl, wd, rb, k0, cw);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    UV c = deep_chroma_h4 (f, a, x0, i);
    if (ra != rb) {
      const UV d = deep_chroma_h4 (f, b, x0, i);
      c.u = (wa * c.u + (8 - wa) * d.u + 4) >> 3;
      c.v = (wa * c.v + (8 - wa) * d.v + 4) >> 3;
    }
    out[i].x = 0xffffu | ((uint32_t) deep_widen_w (wd, l.v[i]) << 16);
    out[i].y = (uint32_t) c.u | ((uint32_t) c.v << 16);
  }
}

// an AYUV64 pixel through video_converter_matrix16 and video_orc_convert_u16_to_u8: the 8-bit unpack-order word
GSTAMD_HD uint32_t deep_finish_px (const Deep16Params &d, uint2 px)
{
  const int a = (int) (px.x & 0xffffu);
  int c1 = (int) (px.x >> 16), c2 = (int) (px.y & 0xffffu), c3 = (int) (px.y >> 16);
  if (d.has_matrix) {
    const int r = c1, g = c2, b = c3;
    c1 = clampi ((d.im[0][0] * r + d.im[0][1] * g + d.im[0][2] * b + d.im[0][3]) >> 8, 0, 65535);
    c2 = clampi ((d.im[1][0] * r + d.im[1][1] * g + d.im[1][2] * b + d.im[1][3]) >> 8, 0, 65535);
    c3 = clampi ((d.im[2][0] * r + d.im[2][1] * g + d.im[2][2] * b + d.im[2][3]) >> 8, 0, 65535);
  }
  return (uint32_t) (a >> 8) | ((uint32_t) (c1 >> 8) << 8) | ((uint32_t) (c2 >> 8) << 16) | ((uint32_t) (c3 >> 8) << 24);
}

// pixel (x, y) through the whole unscaled chain
GSTAMD_HD uint32_t deep_pixel (const FrontParams &f, const Planes &pl, const int *__restrict__ vpair, const Deep16Params &d, int x, int y)
{
  return deep_finish_px (d, deep_front_px (f, pl, vpair, x, y));
}

// ---- scaling on 16-bit lines (the source is scaled BEFORE the convert stage when the picture shrinks, chain_scale
// video-converter.c:1685-1717, i.e. by the u16 scalers on AYUV64 lines) -------------------------------------------------------
// (Deep16Image, an AYUV64 image in HBM at 8 bytes per pixel: video_types.h)
GSTAMD_HD uint2 deep_img_at (const Deep16Image &im, int x, int y)
{
  x = x < 0 ? 0 : (x >= im.width ? im.width - 1 : x);
  y = y < 0 ? 0 : (y >= im.height ? im.height - 1 : y);
  return *(const uint2 *) (im.p + (ptrdiff_t) y * im.stride + 8 * (ptrdiff_t) x);
}

GSTAMD_HD int deep_comp (uint2 px, int c) { return (int) (((c & 2) ? px.y : px.x) >> (16 * (c & 1))) & 0xffff; }

GSTAMD_HD uint2 deep_pack4 (const int *v)
{
  uint2 r;
  r.x = (uint32_t) v[0] | ((uint32_t) v[1] << 16);
  r.y = (uint32_t) v[2] | ((uint32_t) v[3] << 16);
  return r;
}

// video_orc_resample_scaletaps_u16 (video-orc.orc:2507): (sum + 4095) >> 12, saturated to 16 unsigned bits
GSTAMD_HD int deep_scaletaps (int acc) { return clampi ((acc + 4095) >> 12, 0, 65535); }

// one output pixel of a pass: video_scale_h_near_u64 / h_ntap_u16 (2 taps: video_orc_resample_h_2tap_u16, + 4096) and
// video_scale_v_near_u16 / v_2tap_u16 / v_ntap_u16 (video-scaler.c:546-606, 762-826, 1040-1106); taps at 12 fractional bits
GSTAMD_HD uint2 deep_scale_px (const Deep16Image &im, const ScaleDev &sd, bool horizontal, int x, int y)
{
  const int o = horizontal ? x : y;
  const int off = (int) sd.offset[o];
  if (sd.kind == SCALE_NEAREST)
    return horizontal ? deep_img_at (im, off, y) : deep_img_at (im, x, off);
  const int16_t *t = sd.taps + (size_t) o * sd.n_taps;
  int v[4];
  if (sd.kind == SCALE_2TAP) {
    const uint2 a = horizontal ? deep_img_at (im, off, y) : deep_img_at (im, x, off);
    const uint2 b = horizontal ? deep_img_at (im, off + 1, y) : deep_img_at (im, x, off + 1);
    for (int c = 0; c < 4; c++) {
      const int s1 = deep_comp (a, c), s2 = deep_comp (b, c);
      if (horizontal)
        v[c] = clampi ((int) ((uint32_t) s1 * (uint32_t) (int) t[0] + (uint32_t) s2 * (uint32_t) (int) t[1] + 4096u) >> 12, 0, 65535);
      else                      /* l1 + (((l2 - l1) * p1 + 4096) >> 12), p1 read as an unsigned 16-bit parameter */
        v[c] = clampi (s1 + ((int) ((uint32_t) (s2 - s1) * (uint32_t) (uint16_t) t[1] + 4096u) >> 12), 0, 65535);
    }
    return deep_pack4 (v);
  }
  uint32_t acc[4] = {0, 0, 0, 0};       /* mulll / addl: 32-bit wrapping sums */
  for (int l = 0; l < sd.n_taps; l++) {
    const uint2 p = horizontal ? deep_img_at (im, off + l, y) : deep_img_at (im, x, off + l);
    const uint32_t tp = (uint32_t) (int) t[l];
    for (int c = 0; c < 4; c++)
      acc[c] += (uint32_t) deep_comp (p, c) * tp;
  }
  for (int c = 0; c < 4; c++)
    v[c] = deep_scaletaps ((int) acc[c]);
  return deep_pack4 (v);
}

// ---- deep_front4 with the layout (SEMI: interleaved UV plane / separate planes) and the horizontal chroma filter (CH) fixed at compile time, every
// other decision arithmetic: no wave-uniform branch trees (the general kernel's listing is 20 000 instructions of them - it was bound by its SCALAR
// unit).  x0 a multiple of 4, x0 + 4 <= width.  Value for value deep_front_px: the vertical blend as (wa a + wb b + 2) >> 2 with (3, 1) / (1, 3)
// by the pair's role ((4 a + 2) >> 2 = a when both rows are the same), the edge rules of deep_chroma_h_at as selects.
template <int SEMI, int CH>
GSTAMD_HD void deep_front4_t (const FrontParams &f, const Planes &pl, const int *__restrict__ vpair, int x0, int y, uint2 *out)
{
  struct __attribute__ ((aligned (4))) L4 { uint16_t v[4]; };
  const L4 l = *(const L4 *) (pl.p[0] + (ptrdiff_t) y * pl.stride[0] + 2 * (ptrdiff_t) x0);
  const Widen wd = deep_widen_params (f.hi_depth);
  const int w = f.width, cw = (w + 1) >> 1, k0 = x0 >> 1;
  int ra = y >> f.h_sub, rb = ra, wa = 6;          /* weights over 8: vpair_get */
  if (f.chroma_v2) {
    const VPairW t = vpair_get (vpair, y, f.chroma_v2);
    ra = t.ra, rb = t.rb, wa = t.wa;
  }
  const int wb = 8 - wa;
  int u[2][4], v[2][4];
  /* the chroma positions k0 - 1 .. k0 + 2 (clamped into the row) of a chroma row with ONE load per plane where the row has four positions:
     position by position the kernel issued eight loads per line and lane and sat on the vector-memory issue rate (P010 4K -> BGRA 25 us).  The
     window starts at k0 - 1; the row's first lane starts it at 0, the last at cw - 4, and both take their clamped positions by selects */
  const bool left = k0 == 0, right = k0 + 2 > cw - 1, wide_c = cw >= 4;
  const int cbase = left ? 0 : (right ? cw - 4 : k0 - 1);
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int crow = r ? rb : ra;
    int su[4], sv[4];
    if (wide_c) {
      uint32_t e[4];                  /* {U | V << 16} of the four positions */
      if (SEMI) {
        struct __attribute__ ((aligned (4))) C4 { uint32_t w[4]; };
        const C4 c = *(const C4 *) (pl.p[1] + (ptrdiff_t) crow * pl.stride[1] + 4 * (ptrdiff_t) cbase);
        const uint32_t w0 = f.u_plane ? c.w[0] : (c.w[0] >> 16) | (c.w[0] << 16), w1 = f.u_plane ? c.w[1] : (c.w[1] >> 16) | (c.w[1] << 16);
        const uint32_t w2 = f.u_plane ? c.w[2] : (c.w[2] >> 16) | (c.w[2] << 16), w3 = f.u_plane ? c.w[3] : (c.w[3] >> 16) | (c.w[3] << 16);
        e[0] = right ? w1 : w0, e[1] = left ? w0 : (right ? w2 : w1), e[2] = left ? w1 : (right ? w3 : w2), e[3] = left ? w2 : w3;
      } else {
        struct __attribute__ ((aligned (2))) C4 { uint16_t h[4]; };
        const C4 cu = *(const C4 *) (pl.p[f.u_plane] + (ptrdiff_t) crow * pl.stride[f.u_plane] + 2 * (ptrdiff_t) cbase);
        const C4 cv = *(const C4 *) (pl.p[f.v_plane] + (ptrdiff_t) crow * pl.stride[f.v_plane] + 2 * (ptrdiff_t) cbase);
        const uint32_t w0 = (uint32_t) cu.h[0] | ((uint32_t) cv.h[0] << 16), w1 = (uint32_t) cu.h[1] | ((uint32_t) cv.h[1] << 16);
        const uint32_t w2 = (uint32_t) cu.h[2] | ((uint32_t) cv.h[2] << 16), w3 = (uint32_t) cu.h[3] | ((uint32_t) cv.h[3] << 16);
        e[0] = right ? w1 : w0, e[1] = left ? w0 : (right ? w2 : w1), e[2] = left ? w1 : (right ? w3 : w2), e[3] = left ? w2 : w3;
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
        su[i] = (int) (e[i] & 0xffffu), sv[i] = (int) (e[i] >> 16);
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        int k = k0 - 1 + i;
        k = k < 0 ? 0 : (k > cw - 1 ? cw - 1 : k);
        if (SEMI) {
          const uint32_t t = ((const uint32_t *) (pl.p[1] + (ptrdiff_t) crow * pl.stride[1]))[k];
          const int lo = (int) (t & 0xffffu), hi = (int) (t >> 16);
          su[i] = f.u_plane ? lo : hi;
          sv[i] = f.u_plane ? hi : lo;
        } else {
          su[i] = ((const uint16_t *) (pl.p[f.u_plane] + (ptrdiff_t) crow * pl.stride[f.u_plane]))[k];
          sv[i] = ((const uint16_t *) (pl.p[f.v_plane] + (ptrdiff_t) crow * pl.stride[f.v_plane]))[k];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      u[r][i] = deep_widen_w (wd, su[i]);
      v[r][i] = deep_widen_w (wd, sv[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int x = x0 + i, j = 1 + (i >> 1);
    int cu[2], cv[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
      int a = u[r][j], b = v[r][j];
      if (CH == CHROMA_H_H2_CS) {
        if (i & 1) {
          const bool in = x < w - 1;
          a = in ? (u[r][j] + u[r][j + 1] + 1) >> 1 : a;
          b = in ? (v[r][j] + v[r][j + 1] + 1) >> 1 : b;
        }
      } else if (CH == CHROMA_H_H2) {
        if (i & 1) {
          const bool in = x < w - 1;
          a = in ? (3 * u[r][j] + u[r][j + 1] + 2) >> 2 : a;
          b = in ? (3 * v[r][j] + v[r][j + 1] + 2) >> 2 : b;
        } else {
          const bool in = x >= 2;
          a = in ? (u[r][j - 1] + 3 * u[r][j] + 2) >> 2 : a;
          b = in ? (v[r][j - 1] + 3 * v[r][j] + 2) >> 2 : b;
        }
      }
      cu[r] = a, cv[r] = b;
    }
    const int fu = (wa * cu[0] + wb * cu[1] + 4) >> 3, fv = (wa * cv[0] + wb * cv[1] + 4) >> 3;
    out[i].x = 0xffffu | ((uint32_t) deep_widen_w (wd, l.v[i]) << 16);
    out[i].y = (uint32_t) fu | ((uint32_t) fv << 16);
  }
}

// which instantiation serves a front (-1: none): planes with horizontally subsampled chroma, one of the three filters
GSTAMD_VP int deep_front4_variant (const FrontParams &f)
{
  if (!kind_has_planes (f.kind) || f.w_sub != 1 || hi_depth_be (f.hi_depth))
    return -1;
  const int ch = f.chroma_h == CHROMA_H_NONE ? 0 : (f.chroma_h == CHROMA_H_H2 ? 1 : (f.chroma_h == CHROMA_H_H2_CS ? 2 : -1));
  return ch < 0 ? -1 : (f.kind == UNPACK_SEMI ? 3 : 0) + ch;
}

GSTAMD_HD void deep_front4_any (const FrontParams &f, const Planes &pl, const int *__restrict__ vpair, int x0, int y, uint2 *out)
{
  switch (deep_front4_variant (f)) {
    case 0: deep_front4_t<0, CHROMA_H_NONE> (f, pl, vpair, x0, y, out); break;
    case 1: deep_front4_t<0, CHROMA_H_H2> (f, pl, vpair, x0, y, out); break;
    case 2: deep_front4_t<0, CHROMA_H_H2_CS> (f, pl, vpair, x0, y, out); break;
    case 3: deep_front4_t<1, CHROMA_H_NONE> (f, pl, vpair, x0, y, out); break;
    case 4: deep_front4_t<1, CHROMA_H_H2> (f, pl, vpair, x0, y, out); break;
    case 5: deep_front4_t<1, CHROMA_H_H2_CS> (f, pl, vpair, x0, y, out); break;
    default: deep_front4 (f, pl, vpair, x0, y, out); break;
  }
}

// four unpacked pixels through the convert stage (matrix16 on 24-bit operands when the coefficients allow), narrowing, alpha and pack, stored as a row piece
GSTAMD_HD void deep_finish_store4 (const Deep16Params &d, const PostParams &post, const uint2 *px, uint32_t *out)
{
  uint32_t o[4];
  bool fits24 = true;                 /* every coefficient a 24-bit operand (wave-uniform; the offsets im[k][3] are addends) */
  for (int k = 0; k < 3; k++)
    for (int j = 0; j < 3; j++)
      fits24 = fits24 && d.im[k][j] > -(1 << 23) && d.im[k][j] < (1 << 23);
  uint32_t sel = 0;                   /* destination byte pos[c] <- component c, as a v_perm_b32 selector */
  for (int c = 0; c < 4; c++)
    sel |= (uint32_t) c << (8 * post.pack_pos[c]);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint32_t w8;
    if (d.has_matrix && fits24) {
      const int a = (int) (px[i].x & 0xffffu), r = (int) (px[i].x >> 16), g = (int) (px[i].y & 0xffffu), b = (int) (px[i].y >> 16);
      const int c1 = clampi ((mul24s (d.im[0][0], r) + mul24s (d.im[0][1], g) + mul24s (d.im[0][2], b) + d.im[0][3]) >> 8, 0, 65535);
      const int c2 = clampi ((mul24s (d.im[1][0], r) + mul24s (d.im[1][1], g) + mul24s (d.im[1][2], b) + d.im[1][3]) >> 8, 0, 65535);
      const int c3 = clampi ((mul24s (d.im[2][0], r) + mul24s (d.im[2][1], g) + mul24s (d.im[2][2], b) + d.im[2][3]) >> 8, 0, 65535);
      w8 = (uint32_t) (a >> 8) | ((uint32_t) (c1 >> 8) << 8) | ((uint32_t) (c2 >> 8) << 16) | ((uint32_t) (c3 >> 8) << 24);
    } else {
      w8 = deep_finish_px (d, px[i]);
    }
    w8 = apply_alpha (post.alpha_kind, (unsigned) post.alpha_value, w8);
#ifdef __HIPCC__
    o[i] = __builtin_amdgcn_perm (0u, w8, sel);
#else
    o[i] = pack_px (post.pack_pos, w8);
#endif
  }
  /* one 16-byte store whatever the row's alignment (a dword-aligned vector type: the hardware takes it; the two-way form "aligned ? one store :
     four" came out of the compiler as four stores on both sides) */
#ifdef __HIPCC__
  typedef uint32_t u32x4_a4 __attribute__ ((ext_vector_type (4), aligned (4)));
  u32x4_a4 v;
  v.x = o[0], v.y = o[1], v.z = o[2], v.w = o[3];
  *(u32x4_a4 *) out = v;
#else
  for (int i = 0; i < 4; i++)
    out[i] = o[i];
#endif
}

// pixels x0 .. x0+3 of row y
GSTAMD_HD void convert16_lane4 (const FrontParams &f, const Planes &pl, const int *__restrict__ vpair, const Deep16Params &d, const PostParams &post,
    uint8_t *dst, int dstride, int x0, int y)
{
  if (x0 >= f.width || y >= f.height)
    return;
  uint32_t *out = (uint32_t *) (dst + (size_t) y * dstride + 4 * (size_t) x0);
  if (deep_front4_usable (f, x0)) {
    uint2 px[4];
    deep_front4_any (f, pl, vpair, x0, y, px);
    deep_finish_store4 (d, post, px, out);
    return;
  }
  for (int i = 0; i < 4 && x0 + i < f.width; i++) {
    const uint32_t px = apply_alpha (post.alpha_kind, (unsigned) post.alpha_value, deep_pixel (f, pl, vpair, d, x0 + i, y));
    out[i] = pack_px (post.pack_pos, px);
  }
}

// the same for rows y0 and y0 + 1: both rows' loads go out before either row's arithmetic (twice the memory parallelism per lane, half the waves)
GSTAMD_HD void convert16_rows2 (const FrontParams &f, const Planes &pl, const int *__restrict__ vpair, const Deep16Params &d, const PostParams &post,
    uint8_t *dst, int dstride, int x0, int y0)
{
  if (x0 < f.width && y0 + 1 < f.height && deep_front4_usable (f, x0)) {
    uint2 pa[4], pb[4];
    deep_front4_any (f, pl, vpair, x0, y0, pa);
    deep_front4_any (f, pl, vpair, x0, y0 + 1, pb);
    deep_finish_store4 (d, post, pa, (uint32_t *) (dst + (size_t) y0 * dstride + 4 * (size_t) x0));
    deep_finish_store4 (d, post, pb, (uint32_t *) (dst + (size_t) (y0 + 1) * dstride + 4 * (size_t) x0));
    return;
  }
  convert16_lane4 (f, pl, vpair, d, post, dst, dstride, x0, y0);
  convert16_lane4 (f, pl, vpair, d, post, dst, dstride, x0, y0 + 1);
}

// the specialised kernels' lane: rows y0, y0 + 1 of a frame whose width is a multiple of 4
template <int SEMI, int CH>
GSTAMD_HD void convert16_fast_rows2 (const FrontParams &f, const Planes &pl, const int *__restrict__ vpair, const Deep16Params &d, const PostParams &post,
    uint8_t *dst, int dstride, int x0, int y0)
{
  if (x0 + 4 > f.width || y0 >= f.height)
    return;
  uint2 pa[4], pb[4];
  const int y1 = y0 + 1 < f.height ? y0 + 1 : y0;
  deep_front4_t<SEMI, CH> (f, pl, vpair, x0, y0, pa);
  deep_front4_t<SEMI, CH> (f, pl, vpair, x0, y1, pb);
  deep_finish_store4 (d, post, pa, (uint32_t *) (dst + (size_t) y0 * dstride + 4 * (size_t) x0));
  if (y1 != y0)
    deep_finish_store4 (d, post, pb, (uint32_t *) (dst + (size_t) y1 * dstride + 4 * (size_t) x0));
}

// ---- the front fused into the FIRST, horizontal u16 pass (a 10-bit source that shrinks: chain_scale ahead of the convert stage): the pass evaluates the
// source pixels under its taps itself - no full-size AYUV64 image is written and read back (66 MB each way for a 4K frame).  One output pixel per lane;
// deep_front_px's value for a single pixel with layout and filter fixed at compile time, deep_scale_px's horizontal arithmetic on it.
struct FrontRow {       // what every pixel of a row shares
  Widen wd;
  int ra, rb, wa, wb, cw;
};

GSTAMD_HD FrontRow deep_front_row (const FrontParams &f, const int *__restrict__ vpair, int y)
{
  FrontRow r;
  r.wd = deep_widen_params (f.hi_depth);
  r.cw = (f.width + 1) >> 1;
  r.ra = y >> f.h_sub, r.rb = r.ra, r.wa = 6;          /* weights over 8: vpair_get */
  if (f.chroma_v2) {
    const VPairW t = vpair_get (vpair, y, f.chroma_v2);
    r.ra = t.ra, r.rb = t.rb, r.wa = t.wa;
  }
  r.wb = 8 - r.wa;
  return r;
}

template <int SEMI, int CH>
GSTAMD_HD uint2 deep_front1_t (const FrontParams &f, const Planes &pl, const FrontRow &fr, int x, int y)
{
  const int w = f.width, k = x >> 1, odd = x & 1;
  int kn = odd ? k + 1 : k - 1;                         /* the neighbour sample the filter of this pixel may reach */
  kn = kn < 0 ? 0 : (kn > fr.cw - 1 ? fr.cw - 1 : kn);
  const bool in = odd ? x < w - 1 : x >= 2;
  int cu[2], cv[2];
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int crow = r ? fr.rb : fr.ra;
    int su, sv, nu, nv;
    if (SEMI) {
      const uint32_t *q = (const uint32_t *) (pl.p[1] + (ptrdiff_t) crow * pl.stride[1]);
      const uint32_t t = q[k], tn = CH == CHROMA_H_NONE ? t : q[kn];
      su = (int) (f.u_plane ? t & 0xffffu : t >> 16), sv = (int) (f.u_plane ? t >> 16 : t & 0xffffu);
      nu = (int) (f.u_plane ? tn & 0xffffu : tn >> 16), nv = (int) (f.u_plane ? tn >> 16 : tn & 0xffffu);
    } else {
      const uint16_t *qu = (const uint16_t *) (pl.p[f.u_plane] + (ptrdiff_t) crow * pl.stride[f.u_plane]);
      const uint16_t *qv = (const uint16_t *) (pl.p[f.v_plane] + (ptrdiff_t) crow * pl.stride[f.v_plane]);
      su = qu[k], sv = qv[k];
      nu = CH == CHROMA_H_NONE ? su : qu[kn], nv = CH == CHROMA_H_NONE ? sv : qv[kn];
    }
    su = deep_widen_w (fr.wd, su), sv = deep_widen_w (fr.wd, sv), nu = deep_widen_w (fr.wd, nu), nv = deep_widen_w (fr.wd, nv);
    int a = su, b = sv;
    if (CH == CHROMA_H_H2_CS) {
      if (odd && in)
        a = (su + nu + 1) >> 1, b = (sv + nv + 1) >> 1;
    } else if (CH == CHROMA_H_H2) {
      if (in) {
        a = odd ? (3 * su + nu + 2) >> 2 : (nu + 3 * su + 2) >> 2;
        b = odd ? (3 * sv + nv + 2) >> 2 : (nv + 3 * sv + 2) >> 2;
      }
    }
    cu[r] = a, cv[r] = b;
  }
  const int fu = (fr.wa * cu[0] + fr.wb * cu[1] + 4) >> 3, fv = (fr.wa * cv[0] + fr.wb * cv[1] + 4) >> 3;
  const int c1 = deep_widen_w (fr.wd, ((const uint16_t *) (pl.p[0] + (ptrdiff_t) y * pl.stride[0]))[x]);
  uint2 px;
  px.x = 0xffffu | ((uint32_t) c1 << 16);
  px.y = (uint32_t) fu | ((uint32_t) fv << 16);
  return px;
}

template <int SEMI, int CH>
GSTAMD_HD void front_hscale16_lane (const FrontParams &f, const Planes &pl, const int *__restrict__ vpair, const ScaleDev &sd, uint8_t *dst, int dstride, int ow,
    int x, int y)
{
  if (x >= ow || y >= f.height)
    return;
  const FrontRow fr = deep_front_row (f, vpair, y);
  const int w = f.width, off = (int) sd.offset[x];
  auto at = [&](int sx) { return deep_front1_t<SEMI, CH> (f, pl, fr, sx < 0 ? 0 : (sx > w - 1 ? w - 1 : sx), y); };          /* deep_img_at's clamp */
  uint2 o;
  if (sd.kind == SCALE_NEAREST) {
    o = at (off);
  } else {
    const int16_t *t = sd.taps + (size_t) x * sd.n_taps;
    int v[4];
    if (sd.kind == SCALE_2TAP) {
      const uint2 a = at (off), b = at (off + 1);
      for (int c = 0; c < 4; c++)
        v[c] = clampi ((int) ((uint32_t) deep_comp (a, c) * (uint32_t) (int) t[0] + (uint32_t) deep_comp (b, c) * (uint32_t) (int) t[1] + 4096u) >> 12, 0, 65535);
    } else {
      uint32_t acc[4] = {0, 0, 0, 0};
      for (int l = 0; l < sd.n_taps; l++) {
        const uint2 p = at (off + l);
        const uint32_t tp = (uint32_t) (int) t[l];
        for (int c = 0; c < 4; c++)
          acc[c] += (uint32_t) deep_comp (p, c) * tp;
      }
      for (int c = 0; c < 4; c++)
        v[c] = deep_scaletaps ((int) acc[c]);
    }
    o = deep_pack4 (v);
  }
  *(uint2 *) (dst + (size_t) y * dstride + 8 * (size_t) x) = o;
}

// front_hscale16_lane of the variant deep_front4_variant names (host emulator; the device launcher instantiates the kernels)
GSTAMD_HD void front_hscale16_any (int variant, const FrontParams &f, const Planes &pl, const int *__restrict__ vpair, const ScaleDev &sd, uint8_t *dst, int dstride,
    int ow, int x, int y)
{
  switch (variant) {
    case 0: front_hscale16_lane<0, CHROMA_H_NONE> (f, pl, vpair, sd, dst, dstride, ow, x, y); break;
    case 1: front_hscale16_lane<0, CHROMA_H_H2> (f, pl, vpair, sd, dst, dstride, ow, x, y); break;
    case 2: front_hscale16_lane<0, CHROMA_H_H2_CS> (f, pl, vpair, sd, dst, dstride, ow, x, y); break;
    case 3: front_hscale16_lane<1, CHROMA_H_NONE> (f, pl, vpair, sd, dst, dstride, ow, x, y); break;
    case 4: front_hscale16_lane<1, CHROMA_H_H2> (f, pl, vpair, sd, dst, dstride, ow, x, y); break;
    default: front_hscale16_lane<1, CHROMA_H_H2_CS> (f, pl, vpair, sd, dst, dstride, ow, x, y); break;
  }
}

// the specialised front kernels' lane: pixels x0 .. x0+3 of row y into an AYUV64 image (width a multiple of 4, the image rows 16-byte aligned)
template <int SEMI, int CH>
GSTAMD_HD void front16_fast_lane4 (const FrontParams &f, const Planes &pl, const int *__restrict__ vpair, uint8_t *img, int istride, int x0, int y)
{
  if (x0 + 4 > f.width || y >= f.height)
    return;
  uint2 px[4];
  deep_front4_t<SEMI, CH> (f, pl, vpair, x0, y, px);
  uint4 *out = (uint4 *) (img + (size_t) y * istride + 8 * (size_t) x0);
  out[0] = gstamd_make_uint4 (px[0].x, px[0].y, px[1].x, px[1].y);
  out[1] = gstamd_make_uint4 (px[2].x, px[2].y, px[3].x, px[3].y);
}

// front only: pixels x0 .. x0+3 of row y into an AYUV64 image
GSTAMD_HD void front16_lane4 (const FrontParams &f, const Planes &pl, const int *__restrict__ vpair, uint8_t *img, int istride, int x0, int y)
{
  if (x0 >= f.width || y >= f.height)
    return;
  uint2 *out = (uint2 *) (img + (size_t) y * istride + 8 * (size_t) x0);
  if (deep_front4_usable (f, x0)) {
    uint2 px[4];
    deep_front4_any (f, pl, vpair, x0, y, px);
    if ((((uintptr_t) out) & 15) == 0) {
      ((uint4 *) out)[0] = gstamd_make_uint4 (px[0].x, px[0].y, px[1].x, px[1].y);
      ((uint4 *) out)[1] = gstamd_make_uint4 (px[2].x, px[2].y, px[3].x, px[3].y);
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++)
        out[i] = px[i];
    }
    return;
  }
  for (int i = 0; i < 4 && x0 + i < f.width; i++)
    out[i] = deep_front_px (f, pl, vpair, x0 + i, y);
}

// one pass AYUV64 image -> AYUV64 image
GSTAMD_HD void scale16_lane (const Deep16Image &im, const ScaleDev &sd, bool horizontal, uint8_t *dst, int dstride, int ow, int oh, int x, int y)
{
  if (x >= ow || y >= oh)
    return;
  *(uint2 *) (dst + (size_t) y * dstride + 8 * (size_t) x) = deep_scale_px (im, sd, horizontal, x, y);
}

// last pass fused with the convert stage: scaled AYUV64 pixel -> matrix16 -> 8 bits -> alpha -> pack
GSTAMD_HD void scale16_final_lane (const Deep16Image &im, const ScaleDev &sd, bool horizontal, const Deep16Params &d, const PostParams &post, uint8_t *dst,
    int dstride, int ow, int oh, int x, int y)
{
  if (x >= ow || y >= oh)
    return;
  const uint32_t px = apply_alpha (post.alpha_kind, (unsigned) post.alpha_value, deep_finish_px (d, deep_scale_px (im, sd, horizontal, x, y)));
  *(uint32_t *) (dst + (size_t) y * dstride + 4 * (size_t) x) = pack_px (post.pack_pos, px);
}

// ---- 10-bit destinations: chroma downsample + dither + pack on the final AYUV64 image -------------------------------------------------
//   video_orc_chroma_down_v2_u16 (video-orc.orc:2769-2780), video_orc_chroma_down_h2_u16, video_chroma_down_h2_cs_u16 (video-chroma.c:740-762
//   instantiated for guint16 :796): the arithmetic of video_pack.h on 16-bit values
//   dither_ordered_u16_mask (video-dither.c:272-280, video_orc_dither_ordered_4u16_mask video-orc.orc:2926-2935): addusw of
//   bayer >> (8 - shift), then andnw with the mask - on every component of every pixel, after the downsampler, before the packer
//   pack_I420_10LE (video-format.c:3876-3916): value >> 6;  pack_P010_10LE (:5402-5450): value & 0xffc0
struct DstPlanes16 {
  uint8_t *p[3];
  int stride[3];
};

GSTAMD_HD int dither16_comp (const DitherParams &d, int comp, int v, int x, int y)
{
  if (!d.on)
    return v;
  const int sh = d.shift[comp];
  const int b = dither_bayer_value (x, y + d.y0);          /* do_dither_lines passes out_line = i + out_y */
  const int e = sh < 8 ? b >> (8 - sh) : b;
  int p = v + e;
  p = p > 65535 ? 65535 : p;                            /* addusw */
  return p & ~((1 << sh) - 1) & 0xffff;                 /* andnw */
}

// the dither stage ahead of pack_ARGB64 / pack_AYUV64 (copies): every component of pixel (x, y) of a finished 16-bit frame, in place
GSTAMD_HD void dither16_image_px (const DitherParams &d, uint8_t *img, int stride, int w, int h, int x, int y)
{
  if (x >= w || y >= h)
    return;
  uint2 *p = (uint2 *) (img + (size_t) y * stride + 8 * (size_t) x);
  const uint2 v = *p;
  uint2 o;
  o.x = (uint32_t) dither16_comp (d, 0, (int) (v.x & 0xffffu), x, y) | ((uint32_t) dither16_comp (d, 1, (int) (v.x >> 16), x, y) << 16);
  o.y = (uint32_t) dither16_comp (d, 2, (int) (v.y & 0xffffu), x, y) | ((uint32_t) dither16_comp (d, 3, (int) (v.y >> 16), x, y) << 16);
  *p = o;
}

// the same with the matrix value already at hand
GSTAMD_HD int dither16_with (const DitherParams &d, int comp, int v, int b)
{
  if (!d.on)
    return v;
  const int sh = d.shift[comp];
  const int e = sh < 8 ? b >> (8 - sh) : b;
  int p = v + e;
  p = p > 65535 ? 65535 : p;
  return p & ~((1 << sh) - 1) & 0xffff;
}

GSTAMD_HD uint16_t pack16_sample (int hi_depth, int v)
{
  const int le = hi_depth_le (hi_depth);
  int w = v;
  if (le != 6) {
    const int drop = 16 - hi_depth_bits (hi_depth);
    w = le == 1 || le == 4 ? v >> drop : v & ~((1 << drop) - 1);
  }
  if (hi_depth_be (hi_depth))
    w = bswap16i (w);                     /* GST_WRITE_UINT16_BE */
  return (uint16_t) w;      /* pack_I420_10LE >> 6, pack_P010_10LE & 0xffc0, ... */
}

GSTAMD_HD void pack16_chroma_h (const PackPlanarParams &pk, const uint2 *row, int w, int x, int *u, int *v);

// The chroma downsamplers IN PLACE on the AYUV64 image, the way the reference's run on its lines (video_chroma_down_v2_u16 writes the
// pair's first line, the horizontal ones the even pixels; everything else keeps its values): what an error-diffusion dither stage has to
// see - it runs over every component of every pixel AFTER them, also the ones no packer picks up (video_pack.h pack_down_v_px / _h_px
// are the 8-bit twins).  x = pixel, yb = line pair (h_sub) / line.
GSTAMD_HD void pack16_down_v_px (const PackPlanarParams &pk, uint8_t *img, int stride, int x, int yb)
{
  const int w = pk.width, h = pk.height, y0 = yb << pk.h_sub;
  if (x >= w || y0 >= h || !pk.down_v)
    return;
  uint2 *ra = (uint2 *) (img + (size_t) y0 * stride);
  const uint2 *rb = (const uint2 *) (img + (size_t) (y0 + 1 < h ? y0 + 1 : h - 1) * stride);
  const uint32_t a = ra[x].y, b = rb[x].y;
  const uint32_t u = ((a & 0xffffu) + (b & 0xffffu) + 1u) >> 1, v = ((a >> 16) + (b >> 16) + 1u) >> 1;          /* avguw */
  ra[x].y = u | (v << 16);
}

GSTAMD_HD void pack16_down_h_px (const PackPlanarParams &pk, uint8_t *img, int stride, int x, int yb)
{
  const int w = pk.width, h = pk.height, y0 = yb << pk.h_sub;
  if (x >= w || y0 >= h || pk.w_sub != 1 || !pk.down_h || (x & 1))
    return;
  uint2 *ra = (uint2 *) (img + (size_t) y0 * stride);
  int u, v;
  pack16_chroma_h (pk, ra, w, x, &u, &v);          /* reads the odd neighbours only, which no lane writes */
  ra[x].y = (uint32_t) u | ((uint32_t) v << 16);
}

// GRAY10_LE32 / NV12_10LE32 / NV16_10LE32 destinations (pack_GRAY10_LE32 / _NV12_10LE32 / _NV16_10LE32, video-format.c:5539-5868): a lane per SIX pixels
// (two luma words, the chroma pairs 3 u .. 3 u + 2 = the two UV words U V U | V U V) of the lines (yb << h_sub) ..  The stages are pack16_body's.  Which
// words the reference writes at a line's end: the first UV word of a group whenever the group has a pixel, the second when it has four or more (with three
// the pixel's V stays in the packer's local - that word is not written); missing samples are zero bits.
GSTAMD_HD void pack16_le32_body (const PackPlanarParams &pk, const DitherParams &dt, const uint8_t *__restrict__ src, int sstride, const DstPlanes16 &d, int unit, int yb)
{
  const int w = pk.width, h = pk.height, x0 = 6 * unit, y0 = yb << pk.h_sub;
  if (x0 >= w || y0 >= h)
    return;
  const int nlines = 1 << pk.h_sub;
  const uint2 *ra = (const uint2 *) (src + (size_t) y0 * sstride);
  const uint2 *rb = (const uint2 *) (src + (size_t) (y0 + 1 < h ? y0 + 1 : h - 1) * sstride);
  for (int r = 0; r < nlines; r++) {
    const int y = y0 + r;
    if (y >= h)
      break;
    const uint2 *pr = r ? rb : ra;
    uint32_t *dy = (uint32_t *) (d.p[0] + (size_t) y * d.stride[0]) + 2 * unit;
    for (int wd = 0; wd < 2 && x0 + 3 * wd < w; wd++) {
      uint32_t Y = 0;
      for (int c = 0; c < 3 && x0 + 3 * wd + c < w; c++)
        Y |= ((uint32_t) dither16_comp (dt, 1, (int) (pr[x0 + 3 * wd + c].x >> 16), x0 + 3 * wd + c, y) >> 6) << (10 * c);
      dy[wd] = Y;
    }
  }
  if (pk.kind == UNPACK_GRAY_LE32)
    return;
  auto cu = [&](int i) {
    i = i < 0 ? 0 : (i > w - 1 ? w - 1 : i);
    int u = (int) (ra[i].y & 0xffffu);
    return pk.down_v ? (u + (int) (rb[i].y & 0xffffu) + 1) >> 1 : u;
  };
  auto cv = [&](int i) {
    i = i < 0 ? 0 : (i > w - 1 ? w - 1 : i);
    int v = (int) (ra[i].y >> 16);
    return pk.down_v ? (v + (int) (rb[i].y >> 16) + 1) >> 1 : v;
  };
  uint32_t s[6] = {0, 0, 0, 0, 0, 0};           // U0 V0 U1 V1 U2 V2 of the group, ten bits each
  for (int j = 0; j < 3; j++) {
    const int x = x0 + 2 * j;
    if (x >= w)
      break;
    int u = cu (x), v = cv (x);
    if (pk.down_h == 1) {
      if (x + 1 < w) {
        u = (cu (x) + cu (x + 1) + 1) >> 1;
        v = (cv (x) + cv (x + 1) + 1) >> 1;
      }
    } else if (pk.down_h == 2 && w >= 2) {
      if (x == 0) {
        u = (3 * cu (x) + cu (x + 1) + 2) >> 2;
        v = (3 * cv (x) + cv (x + 1) + 2) >> 2;
      } else if (x < w - 2) {
        u = (cu (x - 1) + 2 * cu (x) + cu (x + 1) + 2) >> 2;
        v = (cv (x - 1) + 2 * cv (x) + cv (x + 1) + 2) >> 2;
      } else {
        u = (cu (x - 1) + 3 * cu (x) + 2) >> 2;
        v = (cv (x - 1) + 3 * cv (x) + 2) >> 2;
      }
    }
    s[2 * j] = (uint32_t) dither16_comp (dt, 2, u, x, y0) >> 6;
    s[2 * j + 1] = (uint32_t) dither16_comp (dt, 3, v, x, y0) >> 6;
  }
  uint32_t *duv = (uint32_t *) (d.p[1] + (size_t) yb * d.stride[1]) + 2 * unit;
  duv[0] = s[0] | (s[1] << 10) | (s[2] << 20);
  if (w - x0 >= 4)
    duv[1] = s[3] | (s[4] << 10) | (s[5] << 20);
}

// NV12_10LE40 / NV16_10LE40 destinations (pack_NV12_10LE40 / _NV16_10LE40, video-format.c:5946-6200): the four-pixel block is five bytes of the luma row and
// (two chroma pairs = four samples) five bytes of the UV row; a line's last block holds what its pixels need, zero bits above them (m samples: the
// bytes their 10 m bits reach into)
GSTAMD_HD void pack16_le40_body (const PackPlanarParams &pk, const DitherParams &dt, const uint8_t *__restrict__ src, int sstride, const DstPlanes16 &d, int x0, int yb)
{
  const int w = pk.width, h = pk.height, y0 = yb << pk.h_sub;
  if (x0 >= w || y0 >= h)
    return;
  const int nlines = 1 << pk.h_sub;
  const uint2 *ra = (const uint2 *) (src + (size_t) y0 * sstride);
  const uint2 *rb = (const uint2 *) (src + (size_t) (y0 + 1 < h ? y0 + 1 : h - 1) * sstride);
  auto put = [](uint8_t *q, unsigned long long bits, int n_samples) {
    const int nb = (10 * n_samples + 7) >> 3;
    for (int i = 0; i < nb; i++)
      q[i] = (uint8_t) (bits >> (8 * i));
  };
  for (int r = 0; r < nlines; r++) {
    const int y = y0 + r;
    if (y >= h)
      break;
    const uint2 *pr = r ? rb : ra;
    unsigned long long bits = 0;
    int m = 0;
    for (; m < 4 && x0 + m < w; m++)
      bits |= (unsigned long long) ((uint32_t) dither16_comp (dt, 1, (int) (pr[x0 + m].x >> 16), x0 + m, y) >> 6) << (10 * m);
    put (pk.kind == UNPACK_SEMI_LE40_TILED ? d.p[0] + tiled_luma_row (pk.pos, d.stride[0], 5, x0, y) : d.p[0] + (size_t) y * d.stride[0] + 5 * (size_t) (x0 >> 2), bits, m);
  }
  auto cu = [&](int i) {
    i = i < 0 ? 0 : (i > w - 1 ? w - 1 : i);
    int u = (int) (ra[i].y & 0xffffu);
    return pk.down_v ? (u + (int) (rb[i].y & 0xffffu) + 1) >> 1 : u;
  };
  auto cv = [&](int i) {
    i = i < 0 ? 0 : (i > w - 1 ? w - 1 : i);
    int v = (int) (ra[i].y >> 16);
    return pk.down_v ? (v + (int) (rb[i].y >> 16) + 1) >> 1 : v;
  };
  unsigned long long bits = 0;
  int ns = 0;
  for (int j = 0; j < 2; j++) {
    const int x = x0 + 2 * j;
    if (x >= w)
      break;
    int u = cu (x), v = cv (x);
    if (pk.down_h == 1) {
      if (x + 1 < w) {
        u = (cu (x) + cu (x + 1) + 1) >> 1;
        v = (cv (x) + cv (x + 1) + 1) >> 1;
      }
    } else if (pk.down_h == 2 && w >= 2) {
      if (x == 0) {
        u = (3 * cu (x) + cu (x + 1) + 2) >> 2;
        v = (3 * cv (x) + cv (x + 1) + 2) >> 2;
      } else if (x < w - 2) {
        u = (cu (x - 1) + 2 * cu (x) + cu (x + 1) + 2) >> 2;
        v = (cv (x - 1) + 2 * cv (x) + cv (x + 1) + 2) >> 2;
      } else {
        u = (cu (x - 1) + 3 * cu (x) + 2) >> 2;
        v = (cv (x - 1) + 3 * cv (x) + 2) >> 2;
      }
    }
    bits |= (unsigned long long) ((uint32_t) dither16_comp (dt, 2, u, x, y0) >> 6) << (10 * ns);
    bits |= (unsigned long long) ((uint32_t) dither16_comp (dt, 3, v, x, y0) >> 6) << (10 * ns + 10);
    ns += 2;
  }
  put (pk.kind == UNPACK_SEMI_LE40_TILED ? d.p[1] + tiled_uv_row (pk.pos, d.stride[1], 5, x0 >> 1, yb) : d.p[1] + (size_t) yb * d.stride[1] + 5 * (size_t) (x0 >> 2), bits, ns);
}

// pack16_body from the block's pixels at hand: pa / pb = pixels x0 - 1 .. x0 + 4 (clamped into the row) of the block's two lines (planar / semi-planar kinds;
// k_deep_scale_pack16 hands over what its lanes made, video_deep_pack.h)
GSTAMD_HD void pack16_block (const PackPlanarParams &pk, int hi_depth, const DitherParams &dt, const uint2 *pa, const uint2 *pb, bool whole, const DstPlanes16 &d,
    int x0, int yb);

// per-lane block as pack_planar_body: pixels x0 .. x0+3 of the lines (yb << h_sub) ..; planar and semi-planar kinds
GSTAMD_HD void pack16_body (const PackPlanarParams &pk, int hi_depth, const DitherParams &dt, const uint8_t *__restrict__ src, int sstride,
    const DstPlanes16 &d, int x0, int yb)
{
  const int w = pk.width, h = pk.height;
  const int y0 = yb << pk.h_sub;
  if (pk.kind == UNPACK_SEMI_LE40 || pk.kind == UNPACK_SEMI_LE40_TILED) {
    pack16_le40_body (pk, dt, src, sstride, d, x0, yb);
    return;
  }
  if (GSTAMD_KIND_LE32 (pk.kind)) {          /* this grid's lane x0 / 4 is group x0 / 4 of those formats (more lanes than groups) */
    pack16_le32_body (pk, dt, src, sstride, d, x0 >> 2, yb);
    return;
  }
  if (x0 >= w || y0 >= h)
    return;
  const int nlines = 1 << pk.h_sub;
  const uint2 *ra = (const uint2 *) (src + (size_t) y0 * sstride);
  const uint2 *rb = (const uint2 *) (src + (size_t) (y0 + 1 < h ? y0 + 1 : h - 1) * sstride);
  /* a whole block inside the row, image rows on 16 bytes: the block's four pixels of a row in two 16-byte loads (a lane's pixels are 32 contiguous
     bytes; pixel by pixel every load instruction of a wave touched 2 KB for 256 useful bytes), the two neighbours the cosited filter reaches singly */
  uint2 pa[6], pb[6];                                   // pixels x0-1 .. x0+4 of the two rows, clamped into the row
  const bool whole = x0 + 4 <= w && (((uintptr_t) src | (uintptr_t) sstride) & 15) == 0;
  if (whole) {
    const uint4 a0 = *(const uint4 *) (ra + x0), a1 = *(const uint4 *) (ra + x0 + 2), b0 = *(const uint4 *) (rb + x0), b1 = *(const uint4 *) (rb + x0 + 2);
    pa[1].x = a0.x, pa[1].y = a0.y, pa[2].x = a0.z, pa[2].y = a0.w, pa[3].x = a1.x, pa[3].y = a1.y, pa[4].x = a1.z, pa[4].y = a1.w;
    pb[1].x = b0.x, pb[1].y = b0.y, pb[2].x = b0.z, pb[2].y = b0.w, pb[3].x = b1.x, pb[3].y = b1.y, pb[4].x = b1.z, pb[4].y = b1.w;
    const int xl = x0 > 0 ? x0 - 1 : 0, xr = x0 + 4 < w ? x0 + 4 : w - 1;
    pa[0] = ra[xl], pb[0] = rb[xl], pa[5] = ra[xr], pb[5] = rb[xr];
  } else {
    for (int i = 0; i < 6; i++) {
      int x = x0 - 1 + i;
      x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
      pa[i] = ra[x], pb[i] = rb[x];
    }
  }
  pack16_block (pk, hi_depth, dt, pa, pb, whole, d, x0, yb);
}

GSTAMD_HD void pack16_block (const PackPlanarParams &pk, int hi_depth, const DitherParams &dt, const uint2 *pa, const uint2 *pb, bool whole, const DstPlanes16 &d,
    int x0, int yb)
{
  const int w = pk.width, h = pk.height;
  const int y0 = yb << pk.h_sub;
  const int nlines = 1 << pk.h_sub;
  /* (every loop with constant bounds and every array with constant indices: run-time bounds / a pointer chosen between the two lines put pa, pb and the
     planes into scratch memory - 112 bytes a lane in k_deep_scale_pack16) */
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int y = y0 + r;
    if (r >= nlines || y >= h)
      break;
    uint16_t *dy = (uint16_t *) (d.p[0] + (size_t) y * d.stride[0]) + x0;
    uint16_t o[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (x0 + i < w)
        o[i] = pack16_sample (hi_depth, dither16_comp (dt, 1, (int) ((r ? pb[i + 1].x : pa[i + 1].x) >> 16), x0 + i, y));
    if (whole && (((uintptr_t) dy) & 7) == 0) {
      uint2 st;
      st.x = (uint32_t) o[0] | ((uint32_t) o[1] << 16);
      st.y = (uint32_t) o[2] | ((uint32_t) o[3] << 16);
      *(uint2 *) dy = st;
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (x0 + i < w)
          dy[i] = o[i];
    }
  }
  int cu[6], cv[6];                                     // U, V of pixels x0-1 .. x0+4
#pragma unroll
  for (int i = 0; i < 6; i++) {
    int u = (int) (pa[i].y & 0xffffu), v = (int) (pa[i].y >> 16);
    if (pk.down_v) {
      u = (u + (int) (pb[i].y & 0xffffu) + 1) >> 1;     /* avguw */
      v = (v + (int) (pb[i].y >> 16) + 1) >> 1;
    }
    cu[i] = u;
    cv[i] = v;
  }
  const int step = 1 << pk.w_sub;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int x = x0 + i;
    if ((i & (step - 1)) != 0 || x >= w)
      continue;
    int u = cu[i + 1], v = cv[i + 1];
    if (pk.w_sub == 1) {
      if (pk.down_h == 1) {
        if (x + 1 < w) {
          u = (cu[i + 1] + cu[i + 2] + 1) >> 1;
          v = (cv[i + 1] + cv[i + 2] + 1) >> 1;
        }
      } else if (pk.down_h == 2 && w >= 2) {
        if (x == 0) {
          u = (3 * cu[i + 1] + cu[i + 2] + 2) >> 2;
          v = (3 * cv[i + 1] + cv[i + 2] + 2) >> 2;
        } else if (x < w - 2) {
          u = (cu[i] + 2 * cu[i + 1] + cu[i + 2] + 2) >> 2;
          v = (cv[i] + 2 * cv[i + 1] + cv[i + 2] + 2) >> 2;
        } else {
          u = (cu[i] + 3 * cu[i + 1] + 2) >> 2;
          v = (cv[i] + 3 * cv[i + 1] + 2) >> 2;
        }
      }
    }
    const uint16_t pu = pack16_sample (hi_depth, dither16_comp (dt, 2, u, x, y0)), pv = pack16_sample (hi_depth, dither16_comp (dt, 3, v, x, y0));
    const int k = x >> pk.w_sub;
    if (pk.kind == UNPACK_SEMI) {
      uint16_t *duv = (uint16_t *) (d.p[1] + (size_t) yb * d.stride[1]) + 2 * k;
      const uint16_t c0 = pk.u_plane ? pu : pv, c1 = pk.u_plane ? pv : pu;
      if ((((uintptr_t) duv) & 3) == 0) {
        *(uint32_t *) duv = (uint32_t) c0 | ((uint32_t) c1 << 16);
      } else {
        duv[0] = c0;
        duv[1] = c1;
      }
    } else {
      const bool u1 = pk.u_plane == 1;
      ((uint16_t *) ((u1 ? d.p[1] : d.p[2]) + (size_t) yb * (u1 ? d.stride[1] : d.stride[2])))[k] = pu;
      ((uint16_t *) ((u1 ? d.p[2] : d.p[1]) + (size_t) yb * (u1 ? d.stride[2] : d.stride[1])))[k] = pv;
    }
  }
}

// the fourth plane of a 10 / 12 / 16-bit destination with an alpha plane (pack_A420_16 & co, pack_GBRA_10LE): pixels x0 .. x0 + 3 of line y
GSTAMD_HD void pack16_alpha_plane_body (const PackPlanarParams &pk, int hi_depth, const DitherParams &dt, const uint8_t *__restrict__ src, int sstride,
    uint8_t *__restrict__ da, int dstride, int x0, int y)
{
  if (x0 >= pk.width || y >= pk.height)
    return;
  const uint2 *row = (const uint2 *) (src + (size_t) y * sstride);
  uint16_t *d = (uint16_t *) (da + (size_t) y * dstride);
  for (int i = 0; i < 4 && x0 + i < pk.width; i++)
    d[x0 + i] = pack16_sample (hi_depth, dither16_comp (dt, 0, (int) (row[x0 + i].x & 0xffffu), x0 + i, y));
}

// ---- packed 10 / 12-bit destinations (Y210, Y212_LE, Y410): one lane per stored unit - a macropixel of two pixels / a pixel.  The stages are
// the planar packer's: chroma downsampled on the line (h only, these formats have no vertical subsampling), every component of every pixel
// dithered at its own position, then pack_Y210 (video-format.c:835-861: & 0xffc0, Y1 of an odd-width line's last macropixel = its Y0) /
// pack_Y410 (:898-921)
GSTAMD_HD void pack16_chroma_h (const PackPlanarParams &pk, const uint2 *row, int w, int x, int *u, int *v)
{
  auto cu = [&](int i) { return (int) (row[i < 0 ? 0 : (i > w - 1 ? w - 1 : i)].y & 0xffffu); };
  auto cv = [&](int i) { return (int) (row[i < 0 ? 0 : (i > w - 1 ? w - 1 : i)].y >> 16); };
  *u = cu (x), *v = cv (x);
  if (pk.w_sub != 1)
    return;
  if (pk.down_h == 1) {
    if (x + 1 < w) {
      *u = (cu (x) + cu (x + 1) + 1) >> 1;
      *v = (cv (x) + cv (x + 1) + 1) >> 1;
    }
  } else if (pk.down_h == 2 && w >= 2) {
    if (x == 0) {
      *u = (3 * cu (x) + cu (x + 1) + 2) >> 2;
      *v = (3 * cv (x) + cv (x + 1) + 2) >> 2;
    } else if (x < w - 2) {
      *u = (cu (x - 1) + 2 * cu (x) + cu (x + 1) + 2) >> 2;
      *v = (cv (x - 1) + 2 * cv (x) + cv (x + 1) + 2) >> 2;
    } else {
      *u = (cu (x - 1) + 3 * cu (x) + 2) >> 2;
      *v = (cv (x - 1) + 3 * cv (x) + 2) >> 2;
    }
  }
}

GSTAMD_VP int pack16_units (const PackPlanarParams &pk)
{
  if (pk.kind == UNPACK_V210)
    return ((pk.frame_on ? pk.frame_w : pk.width) + 5) / 6;          /* groups of the frame line */
  return pk.kind == UNPACK_P422_16 || pk.kind == UNPACK_P422_UYVP ? (pk.width + 1) / 2 : pk.width;
}
// rows of k_pack16_packed's grid (PackPlanarParams::frame_on 2: the frame's)
GSTAMD_VP int pack16_rows (const PackPlanarParams &pk) { return pk.kind == UNPACK_V210 && pk.frame_on == 2 ? pk.frame_h : pk.height; }

GSTAMD_HD void pack16_packed_body (const PackPlanarParams &pk, int hi_depth, const DitherParams &dt, const uint8_t *__restrict__ src, int sstride,
    uint8_t *__restrict__ dst, int dstride, int unit, int y)
{
  const int w = pk.width;
  if (y >= pack16_rows (pk) || unit >= pack16_units (pk))
    return;
  if (pk.kind == UNPACK_V210 && pk.frame_on) {
    /* the frame line's group `unit` (PackPlanarParams::frame_on): pixels left and right of the picture - and every pixel of the rows above and below
       it - are the border's; dst is the frame's first row (2) or the rectangle's (1) */
    const int yp = pk.frame_on == 2 ? y - pk.frame_y : y;
    const bool in_row = yp >= 0 && yp < pk.height;
    const int x_lo = 6 * unit - pk.frame_x;                 /* picture pixel of the group's first sample */
    if (pk.frame_on == 1 && (x_lo + 6 <= 0 || x_lo >= w))
      return;
    const uint2 *prow = (const uint2 *) (src + (size_t) (in_row ? yp : 0) * sstride);
    uint32_t yy[6], uu[3], vv[3];
    for (int j = 0; j < 6; j++) {
      const int x = x_lo + j;
      const bool in_frame = 6 * unit + j < pk.frame_w, in_pic = in_row && x >= 0 && x < w;
      yy[j] = !in_frame ? 0u : (in_pic ? (uint32_t) dither16_comp (dt, 1, (int) (prow[x].x >> 16), x, yp) >> 6 : pk.border10[0]);
      if (!(j & 1)) {
        uu[j >> 1] = in_frame ? pk.border10[1] : 0u, vv[j >> 1] = in_frame ? pk.border10[2] : 0u;
        if (in_frame && in_pic) {
          int u, v;
          pack16_chroma_h (pk, prow, w, x, &u, &v);
          uu[j >> 1] = (uint32_t) dither16_comp (dt, 2, u, x, yp) >> 6;
          vv[j >> 1] = (uint32_t) dither16_comp (dt, 3, v, x, yp) >> 6;
        }
      }
    }
    uint32_t *d = (uint32_t *) (dst + (size_t) y * dstride) + 4 * unit;
    d[0] = uu[0] | (yy[0] << 10) | (vv[0] << 20);
    d[1] = yy[1] | (uu[1] << 10) | (yy[2] << 20);
    d[2] = vv[1] | (yy[3] << 10) | (uu[2] << 20);
    d[3] = yy[4] | (vv[2] << 10) | (yy[5] << 20);
    return;
  }
  const uint2 *row = (const uint2 *) (src + (size_t) y * sstride);
  if (pk.kind == UNPACK_Y410) {
    const uint2 px = row[unit];
    const uint32_t a = (uint32_t) dither16_comp (dt, 0, (int) (px.x & 0xffffu), unit, y) & 0xc000u;
    const uint32_t yy = (uint32_t) dither16_comp (dt, 1, (int) (px.x >> 16), unit, y) & 0xffc0u;
    const uint32_t u = (uint32_t) dither16_comp (dt, 2, (int) (px.y & 0xffffu), unit, y) & 0xffc0u;
    const uint32_t v = (uint32_t) dither16_comp (dt, 3, (int) (px.y >> 16), unit, y) & 0xffc0u;
    /* pack_Y410 / pack_rgb10a2_le / pack_bgr10a2_le (video-format.c:898-921, 6243-6330): the top ten bits of each component at its field */
    const uint32_t word = ((yy >> 6) << pk.pos[1]) | ((u >> 6) << pk.pos[2]) | ((v >> 6) << pk.pos[3]) | (hi_depth == 27 ? 0u : a << 16);         /* (pack_r210 leaves the top two bits 0) */
    ((uint32_t *) (dst + (size_t) y * dstride))[unit] = y410_word (hi_depth, word);
    return;
  }
  if (pk.kind == UNPACK_PACKED64 || pk.kind == UNPACK_GRAY16) {
    /* pack_RGBA64_LE & co (video-format.c:2548-2815), pack_GRAY16_LE / _BE (:1250-1299): the components' 16 bits at their words, in the format's
       byte order */
    const uint2 px = row[unit];
    if (pk.kind == UNPACK_GRAY16) {
      const int yd = dither16_comp (dt, 1, (int) (px.x >> 16), unit, y);
      ((uint16_t *) (dst + (size_t) y * dstride))[unit] = hi_depth == 9 || hi_depth == 10 ? (uint16_t) px16_word (hi_depth, yd) : pack16_sample (hi_depth, yd);
      return;
    }
    const int yy = px16_store (hi_depth, dither16_comp (dt, 1, (int) (px.x >> 16), unit, y));
    uint16_t *d = (uint16_t *) (dst + (size_t) y * dstride) + 4 * unit;
    d[pk.pos[0]] = (uint16_t) px16_store (hi_depth, dither16_comp (dt, 0, (int) (px.x & 0xffffu), unit, y));
    d[pk.pos[1]] = (uint16_t) yy;
    d[pk.pos[2]] = (uint16_t) px16_store (hi_depth, dither16_comp (dt, 2, (int) (px.y & 0xffffu), unit, y));
    d[pk.pos[3]] = (uint16_t) px16_store (hi_depth, dither16_comp (dt, 3, (int) (px.y >> 16), unit, y));
    return;
  }
  if (pk.kind == UNPACK_V210) {
    /* pack_v210 (video-format.c:651-708): the group's six lumas and the chroma of its even pixels at 10 bits (>> 6), samples past the line's end 0 */
    uint32_t yy[6], uu[3], vv[3];
    for (int j = 0; j < 6; j++) {
      const int x = 6 * unit + j;
      yy[j] = x < w ? (uint32_t) dither16_comp (dt, 1, (int) (row[x].x >> 16), x, y) >> 6 : 0u;
      if (!(j & 1)) {
        uu[j >> 1] = vv[j >> 1] = 0;
        if (x < w) {
          int u, v;
          pack16_chroma_h (pk, row, w, x, &u, &v);
          uu[j >> 1] = (uint32_t) dither16_comp (dt, 2, u, x, y) >> 6;
          vv[j >> 1] = (uint32_t) dither16_comp (dt, 3, v, x, y) >> 6;
        }
      }
    }
    uint32_t *d = (uint32_t *) (dst + (size_t) y * dstride) + 4 * unit;
    d[0] = uu[0] | (yy[0] << 10) | (vv[0] << 20);
    d[1] = yy[1] | (uu[1] << 10) | (yy[2] << 20);
    d[2] = vv[1] | (yy[3] << 10) | (uu[2] << 20);
    d[3] = yy[4] | (vv[2] << 10) | (yy[5] << 20);
    return;
  }
  const int x = 2 * unit;
  int u, v;
  pack16_chroma_h (pk, row, w, x, &u, &v);
  if (pk.kind == UNPACK_P422_UYVP) {       /* pack_UYVP (video-format.c:2087-2118): the samples' top ten bits, U Y0 V Y1, big endian; an odd line's last Y1 = its Y0 */
    const uint32_t Y0 = (uint32_t) dither16_comp (dt, 1, (int) (row[x].x >> 16), x, y) >> 6;
    const uint32_t Y1 = x + 1 < w ? (uint32_t) dither16_comp (dt, 1, (int) (row[x + 1].x >> 16), x + 1, y) >> 6 : Y0;
    const uint32_t U = (uint32_t) dither16_comp (dt, 2, u, x, y) >> 6, V = (uint32_t) dither16_comp (dt, 3, v, x, y) >> 6;
    uint8_t *m = dst + (size_t) y * dstride + 5 * (size_t) unit;
    m[0] = (uint8_t) (U >> 2);
    m[1] = (uint8_t) (((U & 3u) << 6) | (Y0 >> 4));
    m[2] = (uint8_t) (((Y0 & 0xfu) << 4) | (V >> 6));
    m[3] = (uint8_t) (((V & 0x3fu) << 2) | (Y1 >> 8));
    m[4] = (uint8_t) (Y1 & 0xffu);
    return;
  }
  const uint16_t y0 = pack16_sample (hi_depth, dither16_comp (dt, 1, (int) (row[x].x >> 16), x, y));
  const uint16_t y1 = x + 1 < w ? pack16_sample (hi_depth, dither16_comp (dt, 1, (int) (row[x + 1].x >> 16), x + 1, y)) : y0;
  uint16_t *d = (uint16_t *) (dst + (size_t) y * dstride) + 4 * unit;
  d[pk.pos[1]] = y0;
  if (x + 1 < w || pk.tail_swap != 2)          /* (2: the second luma of the picture's last macropixel is the border's - planner.cpp, border_picture_positions) */
    d[pk.pos[1] + 2] = y1;
  d[pk.pos[2]] = pack16_sample (hi_depth, dither16_comp (dt, 2, u, x, y));
  d[pk.pos[3]] = pack16_sample (hi_depth, dither16_comp (dt, 3, v, x, y));
}

// ---- plane to plane (GammaPlan::planes_fast): every destination sample is one source sample widened to 16 bits (unpack), dithered when the
// destination has 10 bits, packed / narrowed.  One lane = 8 consecutive samples of a luma row, or 4 chroma positions (U and V) of a chroma row.
struct DeepPlanesPtrs {
  const uint8_t *in[3];
  int in_stride[3];
  uint8_t *out[3];
  int out_stride[3];
  int vec;                      // every pointer and pitch is a multiple of 16: whole groups go through one vector access each
};

// N (4 or 8) consecutive samples from index idx0 of a row, widened to 16 bits (8-bit unpack into the 16-bit chain:
// video_orc_convert_u8_to_u16; 10-bit: deep_widen)
template <int N>
GSTAMD_HD void deep_planes_load (int hi, const uint8_t *row, int idx0, int n, int vec, int *v)
{
  if (vec && n == N) {
    if (!hi) {
      uint32_t w[N / 4];
      if (N == 8)
        *(uint2 *) w = *(const uint2 *) (row + idx0);
      else
        w[0] = *(const uint32_t *) (row + idx0);
#pragma unroll
      for (int i = 0; i < N; i++)
        v[i] = (int) ((w[i >> 2] >> (8 * (i & 3))) & 0xffu) * 257;
    } else {
      uint32_t w[N / 2];
      if (N == 8)
        *(uint4 *) w = *(const uint4 *) (row + 2 * (size_t) idx0);
      else
        *(uint2 *) w = *(const uint2 *) (row + 2 * (size_t) idx0);
#pragma unroll
      for (int i = 0; i < N; i++)
        v[i] = deep_widen (hi, (int) ((w[i >> 1] >> (16 * (i & 1))) & 0xffffu));
    }
    return;
  }
  for (int i = 0; i < N; i++)
    if (i < n)
      v[i] = hi ? deep_widen (hi, ((const uint16_t *) row)[idx0 + i]) : (int) row[idx0 + i] * 257;
}

// N consecutive samples of component comp to index idx0 of a row; sample i sits at picture position (x0 + i * xstep, y) for the dither
template <int N>
GSTAMD_HD void deep_planes_store (const DeepPlanesParams &d, uint8_t *row, int idx0, int n, int vec, int comp, const int *v, int x0, int xstep, int y)
{
  if (d.out_hi) {
    uint32_t w[N / 2];
#pragma unroll
    for (int i = 0; i < N / 2; i++)
      w[i] = 0;
    /* the lane's positions x0 + i * xstep stay inside one 8-column group of the matrix row (x0 is a multiple of 8 for luma rows,
       of 4 * xstep for chroma rows with xstep 1 or 2) */
    const uint2 brow = d.dither.on ? dither_bayer_row8 (x0 & ~7, y) : uint2{0, 0};
#pragma unroll
    for (int i = 0; i < N; i++) {
      const int col = (x0 + i * xstep) & 7;
      const int bv = (int) (((col & 4) ? brow.y : brow.x) >> (8 * (col & 3))) & 0xff;
      const uint32_t o = pack16_sample (d.out_hi, dither16_with (d.dither, comp, v[i], bv));
      w[i >> 1] |= o << (16 * (i & 1));
    }
    if (vec && n == N) {
      if (N == 8)
        *(uint4 *) (row + 2 * (size_t) idx0) = *(const uint4 *) w;
      else
        *(uint2 *) (row + 2 * (size_t) idx0) = *(const uint2 *) w;
    } else {
      for (int i = 0; i < N; i++)
        if (i < n)
          ((uint16_t *) row)[idx0 + i] = (uint16_t) (w[i >> 1] >> (16 * (i & 1)));
    }
    return;
  }
  uint32_t w[N / 4];
#pragma unroll
  for (int i = 0; i < N / 4; i++)
    w[i] = 0;
#pragma unroll
  for (int i = 0; i < N; i++)
    w[i >> 2] |= (uint32_t) ((v[i] >> 8) & 0xff) << (8 * (i & 3));              /* video_orc_convert_u16_to_u8 */
  if (vec && n == N) {
    if (N == 8)
      *(uint2 *) (row + idx0) = *(const uint2 *) w;
    else
      *(uint32_t *) (row + idx0) = w[0];
  } else {
    for (int i = 0; i < N; i++)
      if (i < n)
        row[idx0 + i] = (uint8_t) (w[i >> 2] >> (8 * (i & 3)));
  }
}

// row < height: luma row `row`, 8 samples per lane; otherwise chroma row `row - height`, 4 chroma positions (U and V) per lane
GSTAMD_HD void deep_planes_body (const DeepPlanesParams &d, const DeepPlanesPtrs &pp, int lane_x, int row, long long ds = 0, long long dd = 0)
{
  if (row < d.height) {
    const int x0 = lane_x * 8;
    if (x0 >= d.width)
      return;
    const int n = d.width - x0 < 8 ? d.width - x0 : 8;
    int v[8];
    deep_planes_load<8> (d.in_hi, pp.in[0] + ds + (size_t) row * pp.in_stride[0], x0, n, pp.vec, v);
    deep_planes_store<8> (d, pp.out[0] + dd + (size_t) row * pp.out_stride[0], x0, n, pp.vec, 1, v, x0, 1, row);
    return;
  }
  const int cr = row - d.height;
  const int cw = (d.width + (1 << d.w_sub) - 1) >> d.w_sub, ch = (d.height + (1 << d.h_sub) - 1) >> d.h_sub;
  const int k0 = lane_x * 4;
  if (cr >= ch || k0 >= cw)
    return;
  const int n = cw - k0 < 4 ? cw - k0 : 4;
  const int y = cr << d.h_sub, x0 = k0 << d.w_sub, xstep = 1 << d.w_sub;
  int u[4], v[4];
  if (d.in_kind == UNPACK_SEMI) {
    int t[8];
    deep_planes_load<8> (d.in_hi, pp.in[1] + ds + (size_t) cr * pp.in_stride[1], 2 * k0, 2 * n, pp.vec, t);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      u[i] = d.in_u ? t[2 * i] : t[2 * i + 1];
      v[i] = d.in_u ? t[2 * i + 1] : t[2 * i];
    }
  } else {
    deep_planes_load<4> (d.in_hi, pp.in[d.in_u] + ds + (size_t) cr * pp.in_stride[d.in_u], k0, n, pp.vec, u);
    deep_planes_load<4> (d.in_hi, pp.in[d.in_v] + ds + (size_t) cr * pp.in_stride[d.in_v], k0, n, pp.vec, v);
  }
  if (d.out_kind == UNPACK_SEMI) {
    /* interleave after the per-component dither: U with component 2, V with component 3, both at the position of the chroma sample */
    DeepPlanesParams du = d;
    int t[8];
    uint8_t *q = pp.out[1] + dd + (size_t) cr * pp.out_stride[1];
    if (d.out_hi) {
      const uint2 brow = d.dither.on ? dither_bayer_row8 (x0 & ~7, y) : uint2{0, 0};
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int col = (x0 + i * xstep) & 7;
        const int bv = (int) (((col & 4) ? brow.y : brow.x) >> (8 * (col & 3))) & 0xff;
        const int pu = dither16_with (d.dither, 2, u[i], bv), pv = dither16_with (d.dither, 3, v[i], bv);
        t[2 * i] = d.out_u ? pu : pv;
        t[2 * i + 1] = d.out_u ? pv : pu;
      }
      du.dither.on = 0;
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        t[2 * i] = d.out_u ? u[i] : v[i];
        t[2 * i + 1] = d.out_u ? v[i] : u[i];
      }
    }
    deep_planes_store<8> (du, q, 2 * k0, 2 * n, pp.vec, 2, t, 0, 0, 0);
  } else {
    deep_planes_store<4> (d, pp.out[d.out_u] + dd + (size_t) cr * pp.out_stride[d.out_u], k0, n, pp.vec, 2, u, x0, xstep, y);
    deep_planes_store<4> (d, pp.out[d.out_v] + dd + (size_t) cr * pp.out_stride[d.out_v], k0, n, pp.vec, 3, v, x0, xstep, y);
  }
}

// ---- k_deep_planes16: the plain depth changes, sixteen samples of a plane row per lane ---------------------------------------------------
// deep_planes_body is written per sample with every format switch at run time (some 200 VALU instructions for a lane's 8 samples: P010 ->
// NV12 at 4K ran 11 us, NV12 -> P010 15 us, against 4.7 us of HBM traffic).  When both formats have the same plane layout and exactly one of
// them is deep, a plane row is a row of samples in and a row of samples out, and the per-sample chain collapses:
//   deep -> 8 bits: widen to 16 (bit replication below the stored bits), keep the high byte (video_orc_convert_u16_to_u8) = the high byte
//     of the stored word for the formats that keep their bits on top (P010, P012, P016), (word >> (bits - 8)) & 0xff for the others;
//   8 bits -> deep: byte * 257 (video_orc_convert_u8_to_u16) = the byte in both halves of the word, + the dither matrix value of the
//     sample's picture position >> (8 - shift), saturated, & ~((1 << shift) - 1) (gst_video_dither_line, ordered dither on 16-bit lines),
//     then the packer's & ~((1 << drop) - 1) or >> drop.  The matrix is 16 columns wide and a lane starts at a multiple of 16 samples:
//     the eight pairs of dither values are the same for every lane of a row.
// A lane loads 32 / 16 bytes and stores 16 / 32; two samples per register in the packed 16-bit instructions.
GSTAMD_VP bool deep_planes16_ok (const DeepPlanesParams &d)
{
  if ((d.in_kind != UNPACK_PLANAR && d.in_kind != UNPACK_SEMI) || (d.out_kind != UNPACK_PLANAR && d.out_kind != UNPACK_SEMI) || (d.in_hi == 0 && d.out_hi == 0) || d.w_sub > 1)
    return false;
  const auto le_deep = [](int hi) { return hi == 1 || hi == 2 || hi == 4 || hi == 5 || hi == 6; };
  if (d.in_hi && d.out_hi) {            /* deep -> deep (P010 -> I420_10LE: a 10-bit decoder's frames for a 10-bit three-plane encoder; deep_planes16_dd_body) */
    if (!le_deep (d.in_hi) || !le_deep (d.out_hi))
      return false;
    const int cwd = (d.width + (1 << d.w_sub) - 1) >> d.w_sub;
    const bool mixed = d.in_kind != d.out_kind;
    if ((d.width % 16) != 0 || (mixed && d.w_sub != 1) || (((mixed || d.in_kind == UNPACK_SEMI) ? 2 * cwd : cwd) % 16) != 0)
      return false;
    if (!mixed && d.in_kind == UNPACK_SEMI && (d.in_u != d.out_u || d.w_sub != 1))
      return false;
    if (d.dither.on && (d.dither.shift[1] != d.dither.shift[2] || d.dither.shift[1] != d.dither.shift[3] || d.dither.shift[1] > 8 || d.dither.shift[1] < 0))
      return false;
    return true;
  }
  /* the other plane layout on the way down to 8 bits (P010 -> I420 / YV12, I420_10LE -> NV12 / NV21: a decoder's frames for an 8-bit encoder): the chroma rows
     are taken apart / put together with byte permutations (deep_planes16_body "mixed") */
  if (d.in_kind != d.out_kind && d.w_sub != 1)
    return false;
  const int deep = d.in_hi ? d.in_hi : d.out_hi;
  if (deep != 1 && deep != 2 && deep != 4 && deep != 5 && deep != 6)
    return false;
  const int cw = (d.width + (1 << d.w_sub) - 1) >> d.w_sub;
  if ((d.width % 16) != 0 || ((d.in_kind == UNPACK_SEMI || d.out_kind == UNPACK_SEMI ? 2 * cw : cw) % 16) != 0)
    return false;
  if (d.in_kind == UNPACK_SEMI && d.out_kind == UNPACK_SEMI && (d.in_u != d.out_u || d.w_sub != 1))
    return false;
  if (d.out_hi && d.dither.on && (d.dither.shift[1] != d.dither.shift[2] || d.dither.shift[1] != d.dither.shift[3] || d.dither.shift[1] > 8 || d.dither.shift[1] < 0))
    return false;
  return true;
}

GSTAMD_HD uint32_t pk_adds16 (uint32_t a, uint32_t b)            // v_pk_add_u16 clamp: per 16-bit lane, saturating
{
#ifdef __HIPCC__
  typedef unsigned short us2 __attribute__ ((ext_vector_type (2)));
  return __builtin_bit_cast (uint32_t, __builtin_elementwise_add_sat (__builtin_bit_cast (us2, a), __builtin_bit_cast (us2, b)));
#else
  const uint32_t lo = (a & 0xffffu) + (b & 0xffffu), hi = (a >> 16) + (b >> 16);
  return (lo > 0xffffu ? 0xffffu : lo) | ((hi > 0xffffu ? 0xffffu : hi) << 16);
#endif
}

GSTAMD_VP int deep_planes16_rows (const DeepPlanesParams &d)
{
  const int ch = (d.height + (1 << d.h_sub) - 1) >> d.h_sub;
  return d.height + (d.in_kind == UNPACK_SEMI || d.out_kind == UNPACK_SEMI ? ch : 2 * ch);          /* (mixed layouts: a lane of a chroma row serves both planes) */
}

// sixteen stored words of a deep plane row -> their sixteen bytes (the high byte of the widened value)
GSTAMD_HD uint4 deep_narrow16 (int in_hi, const uint4 &a, const uint4 &b)
{
  uint4 o;
  if (in_hi == 1 || in_hi == 4) {           /* the value in the low bits: (word >> (bits - 8)) & 0xff */
    const int sh = hi_depth_bits (in_hi) - 8;
    o.x = bperm (a.y >> sh, a.x >> sh, 0x06040200u), o.y = bperm (a.w >> sh, a.z >> sh, 0x06040200u);
    o.z = bperm (b.y >> sh, b.x >> sh, 0x06040200u), o.w = bperm (b.w >> sh, b.z >> sh, 0x06040200u);
  } else {
    o.x = bperm (a.y, a.x, 0x07050301u), o.y = bperm (a.w, a.z, 0x07050301u);
    o.z = bperm (b.y, b.x, 0x07050301u), o.w = bperm (b.w, b.z, 0x07050301u);
  }
  return o;
}

// the sixteen dither values of row y as the eight pairs a lane's sample pairs meet: mode 0 - sample j at column j; 1 - U and V of chroma position k (samples
// 2 k, 2 k + 1) at column 2 k; 2 - chroma sample j at column 2 j (mod 16)
struct DeepPairs8 { uint32_t v[8]; };
GSTAMD_HD DeepPairs8 deep_dither_pairs16 (const DeepPlanesParams &d, int mode, int y)
{
  DeepPairs8 r;
  uint32_t *e = r.v;
  const int sh = d.dither.on ? d.dither.shift[1] : 0;
  if (d.dither.on && sh > 0) {
    const uint2 lo = dither_bayer_row8 (0, y), hi = dither_bayer_row8 (8, y);
    uint32_t eb[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
    for (int i = 0; i < 4; i++)
      eb[i] = (eb[i] >> (8 - sh)) & ((0xffu >> (8 - sh)) * 0x01010101u);
#pragma unroll
    for (int p = 0; p < 8; p++) {
      if (mode == 0)
        e[p] = bperm (0u, eb[p / 2], (p & 1) ? 0x0c030c02u : 0x0c010c00u);
      else if (mode == 1)
        e[p] = bperm (0u, eb[p / 2], (p & 1) ? 0x0c020c02u : 0x0c000c00u);
      else
        e[p] = bperm (0u, eb[p & 3], 0x0c020c00u);
    }
  } else {
#pragma unroll
    for (int p = 0; p < 8; p++)
      e[p] = 0;
  }
  return r;
}

// what deep_planes16_dd_body does to a pair of stored words: widen, dither, mask, pack
struct DeepFin {
  int wsh, wbits, drop, low;
  uint32_t m1, m2, keep, lowm;
};
GSTAMD_HD uint32_t deep_fin (const DeepFin &f, uint32_t v, uint32_t e)
{
  const uint32_t t = (v << f.wsh) & f.m1;
  uint32_t q = pk_adds16 (t | ((t >> f.wbits) & f.m2), e) & f.keep;
  if (f.low)
    q = (q >> f.drop) & f.lowm;
  return q;
}

// deep -> deep, sixteen samples a lane: widen the stored words to 16 bits (deep_widen on both halves of a word), ordered dither of the destination's depth
// (saturating add, mask), the packer's mask or shift - gst_video_dither_line on 16-bit lines between unpack_P010_10LE & co and pack_I420_10LE & co.  The
// other plane layout (P010 -> I420_10LE, I420_10LE -> P010): a lane of a chroma row works on the interleaved pairs / on eight words of each plane and
// sorts the halves with byte permutations.
GSTAMD_HD void deep_planes16_dd_body (const DeepPlanesParams &d, const DeepPlanesPtrs &pp, int lane, int row, long long ds, long long dd)
{
  const int ch = (d.height + (1 << d.h_sub) - 1) >> d.h_sub, cw = (d.width + (1 << d.w_sub) - 1) >> d.w_sub;
  /* (the planes as scalars before anything chooses between them: a choice between two members of the argument struct is a choice between two ADDRESSES to
     the compiler, and the struct moves to scratch memory for it) */
  const uint8_t *const in0 = pp.in[0], *const in1 = pp.in[1], *const in2 = pp.in[2];
  uint8_t *const out0 = pp.out[0], *const out1 = pp.out[1], *const out2 = pp.out[2];
  const int is0 = pp.in_stride[0], is1 = pp.in_stride[1], is2 = pp.in_stride[2], os0 = pp.out_stride[0], os1 = pp.out_stride[1], os2 = pp.out_stride[2];
  const Widen wdn = deep_widen_params (d.in_hi);
  const int sh = d.dither.on ? d.dither.shift[1] : 0;
  DeepFin F;
  F.wsh = wdn.sh, F.wbits = wdn.bits;
  F.m1 = ((0xffffu << wdn.sh) & 0xffffu) * 0x10001u, F.m2 = (0xffffu >> wdn.bits) * 0x10001u;
  F.drop = d.out_hi == 6 ? 0 : 16 - hi_depth_bits (d.out_hi);
  F.low = d.out_hi == 1 || d.out_hi == 4;
  F.keep = ((0xffffu & ~((1u << sh) - 1u)) * 0x00010001u) & (F.low ? 0xffffffffu : (0xffffu & ~((1u << F.drop) - 1u)) * 0x00010001u);
  F.lowm = (0xffffu >> F.drop) * 0x00010001u;
#define fin(v, e) deep_fin (F, (v), (e))
  const int s0 = 16 * lane;
  if (row < d.height || d.in_kind == d.out_kind) {
    int r = row, n = d.width, mode = 0, y = row;
    const uint8_t *sb = in0;          /* (two-way choices only, made where the row's kind is known: a chain of three became a table in scratch memory) */
    uint8_t *db = out0;
    int sst = is0, dst_ = os0;
    if (row >= d.height) {
      r = row - d.height;
      if (d.in_kind == UNPACK_SEMI) {
        sb = in1, db = out1, sst = is1, dst_ = os1, n = 2 * cw, mode = 1;
      } else {
        const int second = r >= ch ? 1 : 0;
        r -= second * ch;
        const bool i1 = (second ? d.in_v : d.in_u) == 1, o1 = (second ? d.out_v : d.out_u) == 1;
        sb = i1 ? in1 : in2, sst = i1 ? is1 : is2;
        db = o1 ? out1 : out2, dst_ = o1 ? os1 : os2;
        n = cw, mode = d.w_sub ? 2 : 0;
      }
      if (r >= ch)
        return;
      y = r << d.h_sub;
    }
    if (s0 >= n)
      return;
    const uint8_t *src = sb + ds + (size_t) r * sst;
    uint8_t *dst = db + dd + (size_t) r * dst_;
    const DeepPairs8 E = deep_dither_pairs16 (d, mode, y);
    const uint4 a = *(const uint4 *) (src + 2 * (size_t) s0), b = *(const uint4 *) (src + 2 * (size_t) s0 + 16);
    uint4 *dq = (uint4 *) (dst + 2 * (size_t) s0);
    dq[0] = gstamd_make_uint4 (fin (a.x, E.v[0]), fin (a.y, E.v[1]), fin (a.z, E.v[2]), fin (a.w, E.v[3]));
    dq[1] = gstamd_make_uint4 (fin (b.x, E.v[4]), fin (b.y, E.v[5]), fin (b.z, E.v[6]), fin (b.w, E.v[7]));
    return;
  }
  /* mixed layouts, a chroma row: the lane's eight chroma positions */
  const int r = row - d.height;
  if (r >= ch || s0 >= 2 * cw)
    return;
  const int y = r << d.h_sub;
  if (d.in_kind == UNPACK_SEMI) {
    const DeepPairs8 E = deep_dither_pairs16 (d, 1, y);
    const uint8_t *src = in1 + ds + (size_t) r * is1 + 2 * (size_t) s0;
    const uint4 a = *(const uint4 *) src, b = *(const uint4 *) (src + 16);
    const uint32_t q0 = fin (a.x, E.v[0]), q1 = fin (a.y, E.v[1]), q2 = fin (a.z, E.v[2]), q3 = fin (a.w, E.v[3]), q4 = fin (b.x, E.v[4]), q5 = fin (b.y, E.v[5]),
        q6 = fin (b.z, E.v[6]), q7 = fin (b.w, E.v[7]);
    const uint4 first = gstamd_make_uint4 (bperm (q1, q0, 0x05040100u), bperm (q3, q2, 0x05040100u), bperm (q5, q4, 0x05040100u), bperm (q7, q6, 0x05040100u));
    const uint4 second = gstamd_make_uint4 (bperm (q1, q0, 0x07060302u), bperm (q3, q2, 0x07060302u), bperm (q5, q4, 0x07060302u), bperm (q7, q6, 0x07060302u));
    const bool u1 = d.out_u == 1;
    uint8_t *du = (u1 ? out1 : out2) + dd + (size_t) r * (u1 ? os1 : os2) + s0;
    uint8_t *dv = (u1 ? out2 : out1) + dd + (size_t) r * (u1 ? os2 : os1) + s0;
    *(uint4 *) (d.in_u ? du : dv) = first;          /* (semi-planar: in_u is the U-first flag; the choice between the POINTERS - between the values it is a table in scratch) */
    *(uint4 *) (d.in_u ? dv : du) = second;
  } else {
    const DeepPairs8 E = deep_dither_pairs16 (d, 2, y);
    const bool u1 = d.in_u == 1;
    const uint8_t *su = (u1 ? in1 : in2) + ds + (size_t) r * (u1 ? is1 : is2) + s0;
    const uint8_t *sv = (u1 ? in2 : in1) + ds + (size_t) r * (u1 ? is2 : is1) + s0;
    const uint4 a = *(const uint4 *) su, b = *(const uint4 *) sv;
    const bool uf = d.out_u != 0;          /* (semi-planar: out_u is the U-first flag) */
    const uint32_t f0 = fin (uf ? a.x : b.x, E.v[0]), f1 = fin (uf ? a.y : b.y, E.v[1]), f2 = fin (uf ? a.z : b.z, E.v[2]), f3 = fin (uf ? a.w : b.w, E.v[3]);
    const uint32_t g0 = fin (uf ? b.x : a.x, E.v[0]), g1 = fin (uf ? b.y : a.y, E.v[1]), g2 = fin (uf ? b.z : a.z, E.v[2]), g3 = fin (uf ? b.w : a.w, E.v[3]);
    uint4 *dq = (uint4 *) (out1 + dd + (size_t) r * os1 + 2 * (size_t) s0);
    dq[0] = gstamd_make_uint4 (bperm (g0, f0, 0x05040100u), bperm (g0, f0, 0x07060302u), bperm (g1, f1, 0x05040100u), bperm (g1, f1, 0x07060302u));
    dq[1] = gstamd_make_uint4 (bperm (g2, f2, 0x05040100u), bperm (g2, f2, 0x07060302u), bperm (g3, f3, 0x05040100u), bperm (g3, f3, 0x07060302u));
  }
#undef fin
}

// lane: samples 16 * lane .. + 15 of plane row `row` (luma rows, then the chroma rows: of the interleaved plane, or of U and then of V)
// TO_HI: 0 deep -> 8 bits, 1 8 bits -> deep, 2 deep -> deep (deep_planes16_dd_body)
template <int TO_HI>
GSTAMD_HD void deep_planes16_body (const DeepPlanesParams &d, const DeepPlanesPtrs &pp, int lane, int row, long long ds = 0, long long dd = 0)
{
  if (TO_HI == 2) {
    deep_planes16_dd_body (d, pp, lane, row, ds, dd);
    return;
  }
  const int ch = (d.height + (1 << d.h_sub) - 1) >> d.h_sub, cw = (d.width + (1 << d.w_sub) - 1) >> d.w_sub;
  int ip = 0, op = 0, r = row, n = d.width, mode = 0, y = row;
  if (!TO_HI && row >= d.height && d.in_kind != d.out_kind) {
    /* mixed layouts, a chroma row: the lane's eight chroma positions - sixteen interleaved words of a semi-planar row, or eight words of each plane */
    r = row - d.height;
    const int s0 = 16 * lane;
    if (r >= ch || s0 >= 2 * cw)
      return;
    if (d.in_kind == UNPACK_SEMI) {
      const uint8_t *src = pp.in[1] + ds + (size_t) r * pp.in_stride[1];
      uint8_t *const out1 = pp.out[1], *const out2 = pp.out[2];         /* (scalars before the choice: deep_planes16_dd_body) */
      const int os1 = pp.out_stride[1], os2 = pp.out_stride[2];
      const uint4 o = deep_narrow16 (d.in_hi, *(const uint4 *) (src + 2 * (size_t) s0), *(const uint4 *) (src + 2 * (size_t) s0 + 16));
      uint2 first, second;          /* the pairs' first and second samples */
      first.x = bperm (o.y, o.x, 0x06040200u), first.y = bperm (o.w, o.z, 0x06040200u);
      second.x = bperm (o.y, o.x, 0x07050301u), second.y = bperm (o.w, o.z, 0x07050301u);
      const bool u1 = d.out_u == 1;
      uint8_t *du = (u1 ? out1 : out2) + dd + (size_t) r * (u1 ? os1 : os2) + s0 / 2;
      uint8_t *dv = (u1 ? out2 : out1) + dd + (size_t) r * (u1 ? os2 : os1) + s0 / 2;
      *(uint2 *) du = d.in_u ? first : second;          /* (semi-planar: in_u is the U-first flag) */
      *(uint2 *) dv = d.in_u ? second : first;
    } else {
      const bool u1 = d.in_u == 1;
      const uint8_t *const in1 = pp.in[1], *const in2 = pp.in[2];
      const int is1 = pp.in_stride[1], is2 = pp.in_stride[2];
      const uint8_t *su = (u1 ? in1 : in2) + ds + (size_t) r * (u1 ? is1 : is2) + s0;
      const uint8_t *sv = (u1 ? in2 : in1) + ds + (size_t) r * (u1 ? is2 : is1) + s0;
      const uint4 z = gstamd_make_uint4 (0, 0, 0, 0);
      const uint4 nu = deep_narrow16 (d.in_hi, *(const uint4 *) su, z), nv = deep_narrow16 (d.in_hi, *(const uint4 *) sv, z);          /* eight bytes each: .x .y */
      const uint32_t fx = d.out_u ? nu.x : nv.x, fy = d.out_u ? nu.y : nv.y, gx = d.out_u ? nv.x : nu.x, gy = d.out_u ? nv.y : nu.y;
      uint4 o;
      o.x = bperm (gx, fx, 0x05010400u), o.y = bperm (gx, fx, 0x07030602u), o.z = bperm (gy, fy, 0x05010400u), o.w = bperm (gy, fy, 0x07030602u);
      *(uint4 *) (pp.out[1] + dd + (size_t) r * pp.out_stride[1] + s0) = o;
    }
    return;
  }
  if (TO_HI == 1 && row >= d.height && d.in_kind != d.out_kind) {
    /* 8 bits -> deep across plane layouts (NV12 -> I420_10LE, I420 -> P010): a chroma row, the lane's eight chroma positions */
    r = row - d.height;
    const int s0 = 16 * lane;
    if (r >= ch || s0 >= 2 * cw)
      return;
    const int sh = d.dither.on ? d.dither.shift[1] : 0;
    DeepFin F;
    F.wsh = 0, F.wbits = 0, F.m1 = 0xffffffffu, F.m2 = 0;
    F.drop = d.out_hi == 6 ? 0 : 16 - hi_depth_bits (d.out_hi);
    F.low = d.out_hi == 1 || d.out_hi == 4;
    F.keep = ((0xffffu & ~((1u << sh) - 1u)) * 0x00010001u) & (F.low ? 0xffffffffu : (0xffffu & ~((1u << F.drop) - 1u)) * 0x00010001u);
    F.lowm = (0xffffu >> F.drop) * 0x00010001u;
    const uint8_t *const in1 = pp.in[1], *const in2 = pp.in[2];
    uint8_t *const out1 = pp.out[1], *const out2 = pp.out[2];
    const int is1 = pp.in_stride[1], is2 = pp.in_stride[2], os1 = pp.out_stride[1], os2 = pp.out_stride[2];
#define B2(w, hi) bperm (0u, (w), (hi) ? 0x03030202u : 0x01010000u)          /* two bytes * 257 */
    if (d.in_kind == UNPACK_SEMI) {
      const DeepPairs8 E = deep_dither_pairs16 (d, 1, r << d.h_sub);
      const uint4 w = *(const uint4 *) (in1 + ds + (size_t) r * is1 + s0);
      const uint32_t q0 = deep_fin (F, B2 (w.x, 0), E.v[0]), q1 = deep_fin (F, B2 (w.x, 1), E.v[1]), q2 = deep_fin (F, B2 (w.y, 0), E.v[2]), q3 = deep_fin (F, B2 (w.y, 1), E.v[3]),
          q4 = deep_fin (F, B2 (w.z, 0), E.v[4]), q5 = deep_fin (F, B2 (w.z, 1), E.v[5]), q6 = deep_fin (F, B2 (w.w, 0), E.v[6]), q7 = deep_fin (F, B2 (w.w, 1), E.v[7]);
      const uint4 first = gstamd_make_uint4 (bperm (q1, q0, 0x05040100u), bperm (q3, q2, 0x05040100u), bperm (q5, q4, 0x05040100u), bperm (q7, q6, 0x05040100u));
      const uint4 second = gstamd_make_uint4 (bperm (q1, q0, 0x07060302u), bperm (q3, q2, 0x07060302u), bperm (q5, q4, 0x07060302u), bperm (q7, q6, 0x07060302u));
      const bool u1 = d.out_u == 1;
      uint8_t *du = (u1 ? out1 : out2) + dd + (size_t) r * (u1 ? os1 : os2) + s0;
      uint8_t *dv = (u1 ? out2 : out1) + dd + (size_t) r * (u1 ? os2 : os1) + s0;
      *(uint4 *) (d.in_u ? du : dv) = first;
      *(uint4 *) (d.in_u ? dv : du) = second;
    } else {
      const DeepPairs8 E = deep_dither_pairs16 (d, 2, r << d.h_sub);
      const bool u1 = d.in_u == 1;
      const uint2 a = *(const uint2 *) ((u1 ? in1 : in2) + ds + (size_t) r * (u1 ? is1 : is2) + s0 / 2);
      const uint2 b = *(const uint2 *) ((u1 ? in2 : in1) + ds + (size_t) r * (u1 ? is2 : is1) + s0 / 2);
      const bool uf = d.out_u != 0;
      const uint32_t fx = uf ? a.x : b.x, fy = uf ? a.y : b.y, gx = uf ? b.x : a.x, gy = uf ? b.y : a.y;
      const uint32_t f0 = deep_fin (F, B2 (fx, 0), E.v[0]), f1 = deep_fin (F, B2 (fx, 1), E.v[1]), f2 = deep_fin (F, B2 (fy, 0), E.v[2]), f3 = deep_fin (F, B2 (fy, 1), E.v[3]);
      const uint32_t g0 = deep_fin (F, B2 (gx, 0), E.v[0]), g1 = deep_fin (F, B2 (gx, 1), E.v[1]), g2 = deep_fin (F, B2 (gy, 0), E.v[2]), g3 = deep_fin (F, B2 (gy, 1), E.v[3]);
      uint4 *dq = (uint4 *) (out1 + dd + (size_t) r * os1 + 2 * (size_t) s0);
      dq[0] = gstamd_make_uint4 (bperm (g0, f0, 0x05040100u), bperm (g0, f0, 0x07060302u), bperm (g1, f1, 0x05040100u), bperm (g1, f1, 0x07060302u));
      dq[1] = gstamd_make_uint4 (bperm (g2, f2, 0x05040100u), bperm (g2, f2, 0x07060302u), bperm (g3, f3, 0x05040100u), bperm (g3, f3, 0x07060302u));
    }
#undef B2
    return;
  }
  if (row >= d.height) {
    r = row - d.height;
    if (d.in_kind == UNPACK_SEMI) {
      ip = op = 1, n = 2 * cw, mode = 1;
    } else {
      const int second = r >= ch ? 1 : 0;
      r -= second * ch;
      ip = second ? d.in_v : d.in_u, op = second ? d.out_v : d.out_u;
      n = cw, mode = d.w_sub ? 2 : 0;
    }
    if (r >= ch)
      return;
    y = r << d.h_sub;
  }
  const int s0 = 16 * lane;
  if (s0 >= n)
    return;
  const uint8_t *src = pp.in[ip] + ds + (size_t) r * pp.in_stride[ip];
  uint8_t *dst = pp.out[op] + dd + (size_t) r * pp.out_stride[op];
  if (!TO_HI) {
    *(uint4 *) (dst + s0) = deep_narrow16 (d.in_hi, *(const uint4 *) (src + 2 * (size_t) s0), *(const uint4 *) (src + 2 * (size_t) s0 + 16));
    return;
  }
  /* the row's sixteen dither values, as the eight pairs the lane's sample pairs meet */
  uint32_t e[8];
  const int sh = d.dither.on ? d.dither.shift[1] : 0;
  if (d.dither.on && sh > 0) {
    const uint2 lo = dither_bayer_row8 (0, y), hi = dither_bayer_row8 (8, y);
    uint32_t eb[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
    for (int i = 0; i < 4; i++)
      eb[i] = (eb[i] >> (8 - sh)) & ((0xffu >> (8 - sh)) * 0x01010101u);
#pragma unroll
    for (int p = 0; p < 8; p++) {
      if (mode == 0)            /* sample j at column j */
        e[p] = bperm (0u, eb[p / 2], (p & 1) ? 0x0c030c02u : 0x0c010c00u);
      else if (mode == 1)       /* U and V of chroma position k (samples 2k, 2k + 1) at column 2k */
        e[p] = bperm (0u, eb[p / 2], (p & 1) ? 0x0c020c02u : 0x0c000c00u);
      else                      /* chroma sample j at column 2j (mod 16) */
        e[p] = bperm (0u, eb[p & 3], 0x0c020c00u);
    }
  } else {
#pragma unroll
    for (int p = 0; p < 8; p++)
      e[p] = 0;
  }
  const uint32_t keep = (0xffffu & ~((1u << sh) - 1u)) * 0x00010001u;
  const int drop = d.out_hi == 6 ? 0 : 16 - hi_depth_bits (d.out_hi);
  const bool low = d.out_hi == 1 || d.out_hi == 4;
  const uint32_t keep2 = low ? 0xffffffffu : (0xffffu & ~((1u << drop) - 1u)) * 0x00010001u;
  const uint4 w = *(const uint4 *) (src + s0);
  const uint32_t wv[4] = {w.x, w.y, w.z, w.w};
  uint32_t o[8];
#pragma unroll
  for (int p = 0; p < 8; p++) {
    const uint32_t v2 = bperm (0u, wv[p / 2], (p & 1) ? 0x03030202u : 0x01010000u);           /* byte * 257, two samples */
    uint32_t q = pk_adds16 (v2, e[p]) & keep & keep2;
    if (low)
      q = (q >> drop) & ((0xffffu >> drop) * 0x00010001u);
    o[p] = q;
  }
  uint4 *dq = (uint4 *) (dst + 2 * (size_t) s0);
  dq[0] = gstamd_make_uint4 (o[0], o[1], o[2], o[3]);
  dq[1] = gstamd_make_uint4 (o[4], o[5], o[6], o[7]);
}

// ---- k_encode16: 4-byte 8-bit pixels -> 10 / 12 / 16-bit planar or semi-planar 4:2:0 / 4:2:2 YUV in one kernel ----------------------------
// The composite plan (deep_out) makes three images in HBM on the way: the 8-bit unpack-order image of the sub-conversion, the AYUV64 image
// of k_gamma_stage (widen + matrix16), and k_pack16 reads that one (BGRA 4K -> P010: 87 us for 58 MB of frame bytes).  When nothing scales,
// the same integers come from the source pixels directly, a 4 x 2 (4 x 1 for 4:2:2) pixel block per lane as in video_encode_fast.h:
//   unpack: byte positions (FormatDesc::pos); widen: byte * 257 (video_orc_convert_u8_to_u16), folded into the coefficients (im * (257 c)
//   = (257 im) * c); video_converter_matrix16 (:1296-1320): (row . px + offset) >> 8, clamped to 0 .. 65535; chroma downsampling on the
//   16-bit values (video_orc_chroma_down_v2_u16 = avguw of the line pair, then _h2_u16 / video_chroma_down_h2_cs_u16: pack16_body's
//   rules); ordered dither per stored sample at its picture position (dither16_comp); pack (pack16_sample).
struct Enc16Params {
  int width, height;
  int sh[3];                    // bit position of the source byte of unpacked components 1, 2, 3 (R G B / Y U V)
  int has_matrix;
  int cf[3][3], off[3];         // 257 * im[k][j] (24-bit operands), im[k][3]
  int dot;                      // every |im[k][j]| <= 255: the rows as byte dot products on the pixel word
  uint32_t cpos[3], cneg[3];    // |im[k][j]| of the positive / negative entries at the source byte of component j
  int neg_mask;                 // bit k: row k has negative entries
  int hi_depth;
  PackPlanarParams pk;
  DitherParams dt;
};

// host: does the plan go this way?  (`p` is a deep_out plan whose pack16 finishes it)
inline bool enc16_params (const VideoPlan &p, Enc16Params *ep)
{
  const GammaPlan &g = p.gamma;
  if (!g.on || !g.pack16 || g.src16 || g.src64 || g.store64 || g.planes_fast || g.fused || !p.passes.empty () || !g.dec16.empty () || !g.enc16.empty ())
    return false;
  if (p.fin->kind != UNPACK_PACKED4 || p.fin->hi_depth != 0 || g.to_rgb.kind != 0 || g.to_yuv.kind != 0)
    return false;
  if (g.mid_in.width != g.mid_out.width || g.mid_in.height != g.mid_out.height || p.in_info.width != p.out_info.width || p.in_info.height != p.out_info.height)
    return false;
  if ((g.pack.kind != UNPACK_PLANAR && g.pack.kind != UNPACK_SEMI) || hi_depth_be (g.pack_hi_depth) || g.pack.w_sub != 1 || (g.pack.width % 4) != 0 || g.pack.width < 4 || g.pack.virtual_line)
    return false;
  if (g.dither16.on && (g.dither16.method != GSTAMD_DITHER_BAYER || g.dither16.shift[1] != g.dither16.shift[2] || g.dither16.shift[1] != g.dither16.shift[3]))
    return false;
  memset (ep, 0, sizeof (*ep));
  ep->width = g.pack.width, ep->height = g.pack.height;
  for (int j = 0; j < 3; j++)
    ep->sh[j] = 8 * p.fin->pos[j + 1];
  ep->has_matrix = g.prim.has_matrix;
  for (int k = 0; k < 3; k++) {
    for (int j = 0; j < 3; j++) {
      const long long c = 257ll * g.prim.im[k][j];
      if (c <= -(1 << 23) || c >= (1 << 23))
        return false;
      ep->cf[k][j] = (int) c;
    }
    ep->off[k] = g.prim.im[k][3];
  }
  ep->dot = 1;
  for (int k = 0; k < 3; k++)
    for (int j = 0; j < 3; j++) {
      const int cf = g.prim.im[k][j];
      ep->dot = ep->dot && cf >= -255 && cf <= 255;
      if (cf >= 0)
        ep->cpos[k] |= ((uint32_t) cf & 0xffu) << ep->sh[j];
      else {
        ep->cneg[k] |= ((uint32_t) (-cf) & 0xffu) << ep->sh[j];
        ep->neg_mask |= 1 << k;
      }
    }
  ep->hi_depth = g.pack_hi_depth;
  ep->pk = g.pack;
  ep->dt = g.dither16;
  return true;
}

GSTAMD_HD uint32_t dot4_u8_16 (uint32_t a, uint32_t b, uint32_t c)         // sum of the four byte products + c: v_dot4_u32_u8
{
#ifdef __HIPCC__
  return __builtin_amdgcn_udot4 (a, b, c, false);
#else
  uint32_t r = c;
  for (int i = 0; i < 4; i++)
    r += ((a >> (8 * i)) & 0xffu) * ((b >> (8 * i)) & 0xffu);
  return r;
#endif
}

// unpacked component K (0, 1, 2 = Y U V / what the matrix makes of R G B) of a source pixel, 16 bits.  Coefficients of at most 8 bits
// (every Y'CbCr matrix between the usual ranges): the row is two byte dot products on the pixel word, positive and negative entries apart
// (video_encode_fast.h), then (257 s + offset) >> 8; larger ones: three 24-bit multiplies on the unpacked bytes
template <int K>
GSTAMD_HD int enc16_comp (const Enc16Params &ep, uint32_t px)
{
  if (!ep.has_matrix)
    return (int) ((px >> ep.sh[K]) & 0xffu) * 257;
  int s;
  if (ep.dot) {
    s = (int) dot4_u8_16 (px, ep.cpos[K], 0u) - (ep.neg_mask & (1 << K) ? (int) dot4_u8_16 (px, ep.cneg[K], 0u) : 0);
    s = mul24s (s, 257) + ep.off[K];
  } else {
    const int c0 = (int) ((px >> ep.sh[0]) & 0xffu), c1 = (int) ((px >> ep.sh[1]) & 0xffu), c2 = (int) ((px >> ep.sh[2]) & 0xffu);
    s = mul24s (ep.cf[K][0], c0) + mul24s (ep.cf[K][1], c1) + mul24s (ep.cf[K][2], c2) + ep.off[K];
  }
  const int v = s >> 8;
  return v < 0 ? 0 : (v > 65535 ? 65535 : v);
}

// the dither values of columns c0, c1 of a matrix row's word, as the 16-bit pair a sample pair takes (>> (8 - shift), dither16_with)
GSTAMD_HD uint32_t enc16_dither_pair (const Enc16Params &ep, uint32_t bw, uint32_t sel)
{
  if (!ep.dt.on)
    return 0u;
  const int sh = ep.dt.shift[1];
  const uint32_t eb = sh < 8 ? (bw >> (8 - sh)) & ((0xffu >> (8 - sh)) * 0x01010101u) : bw;
  return bperm (0u, eb, sel);
}

// two samples (v0 | v1 << 16) through the dither (saturating add, & ~(quantiser - 1)) and the packer (& ~((1 << drop) - 1), or >> drop)
GSTAMD_HD uint32_t enc16_finish_pair (const Enc16Params &ep, uint32_t v2, uint32_t e2)
{
  uint32_t q = v2;
  if (ep.dt.on)
    q = pk_adds16 (v2, e2) & ((0xffffu & ~((1u << ep.dt.shift[1]) - 1u)) * 0x00010001u);
  if (ep.hi_depth == 6)
    return q;
  const int drop = 16 - hi_depth_bits (ep.hi_depth);
  if (ep.hi_depth == 1 || ep.hi_depth == 4)
    return (q >> drop) & ((0xffffu >> drop) * 0x00010001u);
  return q & ((0xffffu & ~((1u << drop) - 1u)) * 0x00010001u);
}

// one lane: pixels x0 .. x0 + 4 NB - 1 (x0 % (4 NB) == 0) of the lines (yb << h_sub) ..; source rows on 16 bytes, destination rows on 8 NB
template <int SEMI, int NB>
GSTAMD_HD void enc16_block (const Enc16Params &ep, const uint8_t *__restrict__ src, int sstride, const DstPlanes16 &d, int x0, int yb, long long ds = 0, long long dd = 0)
{
  const int NPX = 4 * NB;
  const int w = ep.width, h = ep.height, h_sub = ep.pk.h_sub;
  const int y0 = yb << h_sub;
  if (x0 >= w || y0 >= h)
    return;
  const int y1 = h_sub && y0 + 1 < h ? y0 + 1 : y0;
  const uint8_t *row0 = src + ds + (size_t) y0 * sstride, *row1 = src + ds + (size_t) y1 * sstride;
  uint32_t pa[NPX + 1], pb[NPX + 1];                  /* pixels x0 - 1 .. x0 + NPX - 1 of the two lines */
#pragma unroll
  for (int g = 0; g < NB; g++) {
    const uint4 a = *(const uint4 *) (row0 + 4 * (size_t) (x0 + 4 * g));
    pa[4 * g + 1] = a.x, pa[4 * g + 2] = a.y, pa[4 * g + 3] = a.z, pa[4 * g + 4] = a.w;
  }
#pragma unroll
  for (int g = 0; g < NB; g++) {
    if (h_sub) {
      const uint4 b = *(const uint4 *) (row1 + 4 * (size_t) (x0 + 4 * g));
      pb[4 * g + 1] = b.x, pb[4 * g + 2] = b.y, pb[4 * g + 3] = b.z, pb[4 * g + 4] = b.w;
    } else {
      pb[4 * g + 1] = pa[4 * g + 1], pb[4 * g + 2] = pa[4 * g + 2], pb[4 * g + 3] = pa[4 * g + 3], pb[4 * g + 4] = pa[4 * g + 4];
    }
  }
  pa[0] = pa[1], pb[0] = pb[1];
  if (ep.pk.down_h == 2 && x0 > 0) {
    pa[0] = *(const uint32_t *) (row0 + 4 * (size_t) (x0 - 1));
    pb[0] = *(const uint32_t *) (row1 + 4 * (size_t) (x0 - 1));
  }
  /* luma */
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int y = r ? y1 : y0;
    if (r && y1 == y0)
      break;
    const uint2 brow = ep.dt.on ? dither_bayer_row8 (x0 & ~7, y + ep.dt.y0) : gstamd_make_uint2 (0, 0);
    uint32_t st[2 * NB];
#pragma unroll
    for (int g = 0; g < NB; g++) {
      const uint32_t bw = NB == 2 ? (g ? brow.y : brow.x) : ((x0 & 4) ? brow.y : brow.x);
      const uint32_t *px = (r ? pb : pa) + 4 * g + 1;
      const uint32_t v01 = (uint32_t) enc16_comp<0> (ep, px[0]) | ((uint32_t) enc16_comp<0> (ep, px[1]) << 16);
      const uint32_t v23 = (uint32_t) enc16_comp<0> (ep, px[2]) | ((uint32_t) enc16_comp<0> (ep, px[3]) << 16);
      st[2 * g] = enc16_finish_pair (ep, v01, enc16_dither_pair (ep, bw, 0x0c010c00u));
      st[2 * g + 1] = enc16_finish_pair (ep, v23, enc16_dither_pair (ep, bw, 0x0c030c02u));
    }
    uint8_t *q = d.p[0] + dd + (size_t) y * d.stride[0] + 2 * (size_t) x0;
    if (NB == 2)
      *(uint4 *) q = gstamd_make_uint4 (st[0], st[1], st[2], st[2 * NB - 1]);
    else
      *(uint2 *) q = gstamd_make_uint2 (st[0], st[1]);
  }
  /* chroma of pixels x0 - 1 .. x0 + NPX - 1, the line pair averaged first */
  int cu[NPX + 1], cv[NPX + 1];
#pragma unroll
  for (int i = 0; i < NPX + 1; i++) {
    const bool needed = ep.pk.down_h == 2 ? true : (ep.pk.down_h == 1 ? i >= 1 : (i & 1) == 1);
    if (!needed) {
      cu[i] = cv[i] = 0;
      continue;
    }
    int u = enc16_comp<1> (ep, pa[i]), v = enc16_comp<2> (ep, pa[i]);
    if (ep.pk.down_v) {
      u = (u + enc16_comp<1> (ep, pb[i]) + 1) >> 1;          /* avguw */
      v = (v + enc16_comp<2> (ep, pb[i]) + 1) >> 1;
    }
    cu[i] = u, cv[i] = v;
  }
  const uint2 crow = ep.dt.on ? dither_bayer_row8 (x0 & ~7, y0 + ep.dt.y0) : gstamd_make_uint2 (0, 0);
  uint32_t pu[NB], pv[NB];                            /* the group's two chroma positions as a pair */
#pragma unroll
  for (int g = 0; g < NB; g++) {
    int uu[2], vv[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int i = 4 * g + 2 * t, x = x0 + i;
      int u = cu[i + 1], v = cv[i + 1];
      if (ep.pk.down_h == 1) {
        u = (cu[i + 1] + cu[i + 2] + 1) >> 1;
        v = (cv[i + 1] + cv[i + 2] + 1) >> 1;
      } else if (ep.pk.down_h == 2) {
        if (x == 0) {
          u = (3 * cu[i + 1] + cu[i + 2] + 2) >> 2;
          v = (3 * cv[i + 1] + cv[i + 2] + 2) >> 2;
        } else if (x < w - 2) {
          u = (cu[i] + 2 * cu[i + 1] + cu[i + 2] + 2) >> 2;
          v = (cv[i] + 2 * cv[i + 1] + cv[i + 2] + 2) >> 2;
        } else {
          u = (cu[i] + 3 * cu[i + 1] + 2) >> 2;
          v = (cv[i] + 3 * cv[i + 1] + 2) >> 2;
        }
      }
      uu[t] = u, vv[t] = v;
    }
    const uint32_t bw = NB == 2 ? (g ? crow.y : crow.x) : ((x0 & 4) ? crow.y : crow.x);
    const uint32_t e = enc16_dither_pair (ep, bw, 0x0c020c00u);
    pu[g] = enc16_finish_pair (ep, (uint32_t) uu[0] | ((uint32_t) uu[1] << 16), e);
    pv[g] = enc16_finish_pair (ep, (uint32_t) vv[0] | ((uint32_t) vv[1] << 16), e);
  }
  if (SEMI) {
    uint32_t st[2 * NB];
#pragma unroll
    for (int g = 0; g < NB; g++) {
      const uint32_t c0 = ep.pk.u_plane ? pu[g] : pv[g], c1 = ep.pk.u_plane ? pv[g] : pu[g];
      st[2 * g] = bperm (c1, c0, 0x05040100u), st[2 * g + 1] = bperm (c1, c0, 0x07060302u);
    }
    uint8_t *q = d.p[1] + dd + (size_t) yb * d.stride[1] + 2 * (size_t) x0;
    if (NB == 2)
      *(uint4 *) q = gstamd_make_uint4 (st[0], st[1], st[2], st[2 * NB - 1]);
    else
      *(uint2 *) q = gstamd_make_uint2 (st[0], st[1]);
  } else {
    uint8_t *qu = d.p[ep.pk.u_plane] + dd + (size_t) yb * d.stride[ep.pk.u_plane] + (size_t) x0;
    uint8_t *qv = d.p[ep.pk.v_plane] + dd + (size_t) yb * d.stride[ep.pk.v_plane] + (size_t) x0;
    if (NB == 2) {
      *(uint2 *) qu = gstamd_make_uint2 (pu[0], pu[NB - 1]);
      *(uint2 *) qv = gstamd_make_uint2 (pv[0], pv[NB - 1]);
    } else {
      *(uint32_t *) qu = pu[0];
      *(uint32_t *) qv = pv[0];
    }
  }
}

}  // namespace gstamd
