// video_scale_fast.h - wave-tile scalers: the speed path of the GstVideoScaler passes.
//
// One wave (a 64-lane workgroup of its own) produces `tile_w` (<= 256) consecutive output pixels of one output
// row, 4 per lane.  It first evaluates the source pixels under the tile ONCE into its LDS slice - unpack + chroma
// upsample of the source frame in packed 16-bit arithmetic, 8 pixels per lane per step - and then every lane
// filters from LDS and stores 16 bytes.  Same arithmetic as the per-pixel bodies in video_device.h (fetch_front,
// hscale_px, v2tap_px), restructured for VALU issue:
//
//   * chroma samples travel as {U, V} in the two 16-bit halves of one register; the (a+b+1)>>1 and
//     (3a+b+2)>>2 filters of video-chroma.c:277-327, 687-699 are one add/multiply-add + one packed shift for
//     both components (values <= 1022, no carry between the halves);
//   * the AYUV word is assembled with one v_perm_b32 per pixel;
//   * pixels are split into even / odd byte pairs (0x00CC00AA, 0x00DD00BB) so that a 2-tap blend or an N-tap
//     multiply-accumulate is one packed 16-bit operation per two channels - exactly the 16-bit wrapping
//     arithmetic of the ORC programs (video-orc.orc:2208-2224, 2388-2480).
//
// Reference: video-scaler.c:546-760 (horizontal), 829-1072 (vertical); video-orc-dist.c:26162-26195 (ldreslinl).
#pragma once
#include "video_fast.h"

namespace gstamd {

// ------------------------------------------------------------------------------------------------
// packed front: 8 pixels x0 .. x0+7 of line y (x0 % 8 == 0, x0 + 8 <= width, w_sub == 1, planes aligned as
// front_vec_ok checks)
// ------------------------------------------------------------------------------------------------
struct ChromaP6 {
  uint32_t c[6];        // samples k0-1 .. k0+4, each U | V << 16
};

GSTAMD_HD void load_chroma6_packed (const FrontParams &f, const Planes &pl, int crow, int k0, int cw, ChromaP6 &c)
{
  const int km = k0 > 0 ? k0 - 1 : 0, kp = k0 + 4 < cw ? k0 + 4 : cw - 1;
  if (f.kind == UNPACK_SEMI) {
    const uint8_t *row = pl.p[1] + (ptrdiff_t) crow * pl.stride[1];
    const uint2 mid = *(const uint2 *) (row + 2 * k0);           // samples k0..k0+3
    const uint32_t pm = *(const uint16_t *) (row + 2 * km), pp = *(const uint16_t *) (row + 2 * kp);
    // byte pair (b0, b1) -> U | V << 16; NV12 (u_plane != 0): U first
    const uint32_t sel_lo = f.u_plane ? 0x0c010c00u : 0x0c000c01u, sel_hi = f.u_plane ? 0x0c030c02u : 0x0c020c03u;
    c.c[0] = bperm (0, pm, sel_lo);
    c.c[1] = bperm (0, mid.x, sel_lo);
    c.c[2] = bperm (0, mid.x, sel_hi);
    c.c[3] = bperm (0, mid.y, sel_lo);
    c.c[4] = bperm (0, mid.y, sel_hi);
    c.c[5] = bperm (0, pp, sel_lo);
  } else {
    const uint8_t *ru = pl.p[f.u_plane] + (ptrdiff_t) crow * pl.stride[f.u_plane];
    const uint8_t *rv = pl.p[f.v_plane] + (ptrdiff_t) crow * pl.stride[f.v_plane];
    const uint32_t mu = *(const uint32_t *) (ru + k0), mv = *(const uint32_t *) (rv + k0);
    c.c[0] = (uint32_t) ru[km] | ((uint32_t) rv[km] << 16);
    c.c[1] = bperm (mv, mu, 0x0c040c00u);
    c.c[2] = bperm (mv, mu, 0x0c050c01u);
    c.c[3] = bperm (mv, mu, 0x0c060c02u);
    c.c[4] = bperm (mv, mu, 0x0c070c03u);
    c.c[5] = (uint32_t) ru[kp] | ((uint32_t) rv[kp] << 16);
  }
}

// horizontally filtered chroma under the 8 pixels (same cases as hfilter8)
template <int CH>
GSTAMD_HD void hfilter8_packed (const ChromaP6 &c, int x0, int w, uint32_t *o)
{
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int x = x0 + i, j = (i >> 1) + 1;
    uint32_t cc = c.c[j];
    if (CH == CHROMA_H_H2_CS) {
      if ((i & 1) && x < w - 1)
        cc = pk_shr<1> (c.c[j] + c.c[j + 1] + 0x00010001u);
    } else if (CH == CHROMA_H_H2) {
      if ((i & 1) && x < w - 1)
        cc = pk_shr<2> (3u * c.c[j] + c.c[j + 1] + 0x00020002u);
      else if (!(i & 1) && x >= 2)
        cc = pk_shr<2> (c.c[j - 1] + 3u * c.c[j] + 0x00020002u);
    }
    o[i] = cc;
  }
}

// luma words and the upsampled chroma (U | V << 16 per pixel) of the 8 pixels
template <int CH>
GSTAMD_HD void front_chroma8_packed (const FrontParams &f, const Planes &pl, const int *__restrict__ vpair, int x0, int y, uint2 &yy,
    uint32_t *ca)
{
  const int w = f.width;
  yy = *(const uint2 *) (pl.p[0] + (size_t) y * pl.stride[0] + x0);
  const int cw = (w + 1) >> 1, k0 = x0 >> 1;
  int ra, rb, wa = 6;
  if (f.chroma_v2) {
    const VPairW t = vpair_get (vpair, y, f.chroma_v2);
    ra = t.ra, rb = t.rb, wa = t.wa;
  } else {
    ra = rb = y >> f.h_sub;
  }
  {
    ChromaP6 c;
    load_chroma6_packed (f, pl, ra, k0, cw, c);
    hfilter8_packed<CH> (c, x0, w, ca);
  }
  if (ra != rb) {                         // uniform over the row
    ChromaP6 c;
    uint32_t cb[8];
    load_chroma6_packed (f, pl, rb, k0, cw, c);
    hfilter8_packed<CH> (c, x0, w, cb);
    /* weights over 8 (vpair_get): (3 a + b + 2) >> 2 = (6 a + 2 b + 4) >> 3; 8 * 255 + 4 stays inside the 16-bit halves */
    const uint32_t ua = (uint32_t) wa, ub = (uint32_t) (8 - wa);
#pragma unroll
    for (int i = 0; i < 8; i++)
      ca[i] = pk_shr<3> (ua * ca[i] + ub * cb[i] + 0x00040004u);
  }
}

GSTAMD_HD void front_chroma8_packed_any (const FrontParams &f, const Planes &pl, const int *__restrict__ vpair, int x0, int y, uint2 &yy,
    uint32_t *ca)
{
  if (f.chroma_h == CHROMA_H_H2_CS)
    front_chroma8_packed<CHROMA_H_H2_CS> (f, pl, vpair, x0, y, yy, ca);
  else if (f.chroma_h == CHROMA_H_H2)
    front_chroma8_packed<CHROMA_H_H2> (f, pl, vpair, x0, y, yy, ca);
  else
    front_chroma8_packed<CHROMA_H_NONE> (f, pl, vpair, x0, y, yy, ca);
}

template <int CH>
GSTAMD_HD void front_span8_packed (const FrontParams &f, const Planes &pl, const int *__restrict__ vpair, int x0, int y, uint32_t *out)
{
  uint2 yy;
  uint32_t ca[8];
  front_chroma8_packed<CH> (f, pl, vpair, x0, y, yy, ca);
  // byte0 = 0xff, byte1 = Y, byte2 = U (chroma byte 0), byte3 = V (chroma byte 2)
  static const uint32_t ysel[4] = {0x0604000du, 0x0604010du, 0x0604020du, 0x0604030du};
#pragma unroll
  for (int i = 0; i < 8; i++)
    out[i] = bperm (ca[i], i < 4 ? yy.x : yy.y, ysel[i & 3]);
}

GSTAMD_HD void front_span8_packed_any (const FrontParams &f, const Planes &pl, const int *__restrict__ vpair, int x0, int y, uint32_t *out)
{
  if (f.chroma_h == CHROMA_H_H2_CS)
    front_span8_packed<CHROMA_H_H2_CS> (f, pl, vpair, x0, y, out);
  else if (f.chroma_h == CHROMA_H_H2)
    front_span8_packed<CHROMA_H_H2> (f, pl, vpair, x0, y, out);
  else
    front_span8_packed<CHROMA_H_NONE> (f, pl, vpair, x0, y, out);
}

// can the 8 pixels starting at x0 take the word-load path?
GSTAMD_HD bool src_span8_ok (const SrcFront &s, int x0) { return s.vec_ok && x0 + 8 <= s.f.width; }
GSTAMD_HD bool src_span8_ok (const SrcImage &s, int x0) { return x0 + 8 <= s.width; }

// 8 source pixels x0 .. x0+7 (x0 % 8 == 0, src_span8_ok) of line y
GSTAMD_HD void src_span8 (const SrcFront &s, int x0, int y, uint32_t *px)
{
  front_span8_packed_any (s.f, s.pl, s.vpair, x0, y, px);
}

GSTAMD_HD void src_span8 (const SrcImage &s, int x0, int y, uint32_t *px)
{
  struct __attribute__ ((aligned (4))) W8 { uint32_t v[8]; };
  const W8 v = *(const W8 *) (s.p + (size_t) y * s.stride + 4 * (size_t) x0);
#pragma unroll
  for (int i = 0; i < 8; i++)
    px[i] = v.v[i];
}

// a 4-byte packed 8-bit source (BGRA & co) whose rows are word aligned: eight pixels with word loads issued together, the unpack permutation
// and the colour stage ahead of the scaler on each - instead of eight dependent trips through fetch_front's switch (the horizontal pass of a
// BGRA 4K -> 1080p linear downscale spent 33 us there, profiles/r05/survey_item6_kernel_split_start.log)
GSTAMD_HD bool src_p4 (const SrcFront &s)
{
  return s.f.kind == UNPACK_PACKED4 && s.f.hi_depth == 0 && (s.pl.stride[0] & 3) == 0 && ((size_t) s.pl.p[0] & 3) == 0;
}
GSTAMD_HD bool src_p4 (const SrcImage &) { return false; }
GSTAMD_HD void src_span8_p4 (const SrcFront &s, int x0, int y, uint32_t *px)
{
  struct __attribute__ ((aligned (4))) W8 { uint32_t v[8]; };
  const W8 v = *(const W8 *) (s.pl.p[0] + (size_t) y * s.pl.stride[0] + 4 * (size_t) x0);
  const uint32_t sel = (uint32_t) s.f.pos[0] | ((uint32_t) s.f.pos[1] << 8) | ((uint32_t) s.f.pos[2] << 16) | ((uint32_t) s.f.pos[3] << 24);
#pragma unroll
  for (int i = 0; i < 8; i++)
    px[i] = apply_color (s.pre, bperm (0u, v.v[i], sel));
}
GSTAMD_HD void src_span8_p4 (const SrcImage &, int, int, uint32_t *) {}

// the same for a packed 4:2:2 source (YUY2 & co): pixels x0 .. x0 + 7 are macropixels k0 .. k0 + 3, the horizontal chroma upsampler also reads
// k0 - 1 and k0 + 4 - six words loaded together, then fetch_front's arithmetic (load_uv, chroma_h_at) on registers.  Inside the picture only:
// x0 >= 2, x0 + 9 <= width (every odd pixel has a right neighbour), no swapped tail macropixel in reach, rows the unpacker does not clamp.
GSTAMD_HD bool src_p422 (const SrcFront &s) { return s.f.kind == UNPACK_PACKED422 && (s.pl.stride[0] & 3) == 0 && ((size_t) s.pl.p[0] & 3) == 0; }
GSTAMD_HD bool src_p422 (const SrcImage &) { return false; }
GSTAMD_HD bool src_span8_p422_ok (const SrcFront &s, int x0, int y)
{
  return x0 >= 2 && x0 + 9 <= s.f.width && y <= s.f.luma_last && (s.f.swap_k < 0 || (x0 >> 1) + 4 < s.f.swap_k);
}
GSTAMD_HD bool src_span8_p422_ok (const SrcImage &, int, int) { return false; }
// one pixel of a packed 4:2:2 line as fetch_front makes it (A = 0xff, Y, horizontally upsampled U, V: load_uv + chroma_h_at of video_device.h), x inside the line
GSTAMD_HD uint32_t p422_px (const uint8_t *row, int x, int src_w, int pos1, int pos2, int pos3, int chroma_h, int swap_k)
{
  const int k = x >> 1;
  const uint32_t m = load_macropixel (row + 4 * k);
  const int su = 8 * (k == swap_k ? pos3 : pos2), sv = 8 * (k == swap_k ? pos2 : pos3);
  const uint32_t Y = (m >> (8 * (pos1 + 2 * (x & 1)))) & 0xffu;
  uint32_t u = (m >> su) & 0xffu, v = (m >> sv) & 0xffu;
  /* chroma_h_at (video_device.h) */
  int kn = -1, wa = 0;
  if (chroma_h == CHROMA_H_H2_CS) {
    if ((x & 1) && x < src_w - 1)
      kn = k + 1, wa = 1;
  } else if (chroma_h == CHROMA_H_H2) {
    if ((x & 1) && x < src_w - 1)
      kn = k + 1, wa = 3;
    else if (!(x & 1) && x >= 2)
      kn = k - 1, wa = 3;
  }
  if (kn >= 0) {
    const uint32_t n = load_macropixel (row + 4 * kn);
    const uint32_t un = (n >> (8 * (kn == swap_k ? pos3 : pos2))) & 0xffu, vn = (n >> (8 * (kn == swap_k ? pos2 : pos3))) & 0xffu;
    if (wa == 1) {
      u = (u + un + 1) >> 1;
      v = (v + vn + 1) >> 1;
    } else {
      u = (3 * u + un + 2) >> 2;
      v = (3 * v + vn + 2) >> 2;
    }
  }
  return 0xffu | (Y << 8) | (u << 16) | (v << 24);
}

GSTAMD_HD void span8_p422_raw (const uint8_t *row, int x0, int pos1, int pos2, int pos3, int chroma_h, uint32_t *px)
{
  struct __attribute__ ((aligned (4))) W6 { uint32_t v[6]; };
  const W6 m = *(const W6 *) (row + 2 * (size_t) x0 - 4);
  const int by = 8 * pos1, bu = 8 * pos2, bv = 8 * pos3;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t cur = m.v[1 + (i >> 1)];
    const int Y = (int) ((cur >> (by + 16 * (i & 1))) & 0xff);
    int u = (int) ((cur >> bu) & 0xff), v = (int) ((cur >> bv) & 0xff);
    if (chroma_h == CHROMA_H_H2_CS) {
      if (i & 1) {
        const uint32_t n = m.v[2 + (i >> 1)];
        u = (u + (int) ((n >> bu) & 0xff) + 1) >> 1;
        v = (v + (int) ((n >> bv) & 0xff) + 1) >> 1;
      }
    } else if (chroma_h == CHROMA_H_H2) {
      const uint32_t n = m.v[(i & 1) ? 2 + (i >> 1) : (i >> 1)];
      u = (3 * u + (int) ((n >> bu) & 0xff) + 2) >> 2;
      v = (3 * v + (int) ((n >> bv) & 0xff) + 2) >> 2;
    }
    px[i] = 0xffu | ((uint32_t) Y << 8) | ((uint32_t) u << 16) | ((uint32_t) v << 24);
  }
}
GSTAMD_HD void src_span8_p422 (const SrcFront &s, int x0, int y, uint32_t *px)
{
  span8_p422_raw (s.pl.p[0] + (size_t) y * s.pl.stride[0], x0, s.f.pos[1], s.f.pos[2], s.f.pos[3], s.f.chroma_h, px);
#pragma unroll
  for (int i = 0; i < 8; i++)
    px[i] = apply_color (s.pre, px[i]);
}
GSTAMD_HD void src_span8_p422 (const SrcImage &, int, int, uint32_t *) {}

// ------------------------------------------------------------------------------------------------
// SrcLean: a packed 8-bit frame (4-byte pixels in any byte order, or packed 4:2:2) with no colour step ahead of the scaler - what the wave-tile
// kernels need of SrcFront in 48 bytes of kernel arguments.  A kernel instantiated on SrcFront carries the whole per-pixel front (every unpack kind,
// the chroma pair table, two colour matrices: 330 spilled SGPRs in k_scale2x2_wave) and loads a kilobyte of arguments at the start of every one of
// its ~17 000 short waves: the horizontal 4-tap pass of a 4K frame took 43.4 us from UYVY through SrcFront and 18.9 us from AYUV through SrcImage
// (profiles/r05/item6_case11 / case16).
// ------------------------------------------------------------------------------------------------
struct SrcLean {
  const uint8_t *p;
  int stride, width;
  int p422;             // 0: 4-byte pixels, `sel` the unpack permutation (memory bytes -> A, c1, c2, c3); 1: packed 4:2:2
  uint32_t sel;
  int pos1, pos2, pos3, chroma_h, swap_k;       // FormatDesc::pos, ChromaH, FrontParams::swap_k of a packed 4:2:2 source
  GSTAMD_HD uint32_t at (int x, int y) const
  {
    const uint8_t *row = p + (size_t) y * stride;
    if (!p422)
      return bperm (0u, *(const uint32_t *) (row + 4 * (size_t) x), sel);
    return p422_px (row, x, width, pos1, pos2, pos3, chroma_h, swap_k);
  }
};
GSTAMD_HD bool src_span8_ok (const SrcLean &, int) { return false; }
GSTAMD_HD void src_span8 (const SrcLean &, int, int, uint32_t *) {}
GSTAMD_HD bool src_p4 (const SrcLean &s) { return !s.p422; }
GSTAMD_HD void src_span8_p4 (const SrcLean &s, int x0, int y, uint32_t *px)
{
  struct __attribute__ ((aligned (4))) W8 { uint32_t v[8]; };
  const W8 v = *(const W8 *) (s.p + (size_t) y * s.stride + 4 * (size_t) x0);
#pragma unroll
  for (int i = 0; i < 8; i++)
    px[i] = bperm (0u, v.v[i], s.sel);
}
GSTAMD_HD bool src_p422 (const SrcLean &s) { return s.p422 != 0; }
GSTAMD_HD bool src_span8_p422_ok (const SrcLean &s, int x0, int)
{
  return x0 >= 2 && x0 + 9 <= s.width && (s.swap_k < 0 || (x0 >> 1) + 4 < s.swap_k);
}
GSTAMD_HD void src_span8_p422 (const SrcLean &s, int x0, int y, uint32_t *px)
{
  span8_p422_raw (s.p + (size_t) y * s.stride, x0, s.pos1, s.pos2, s.pos3, s.chroma_h, px);
}

// ------------------------------------------------------------------------------------------------
// wave tiles
// ------------------------------------------------------------------------------------------------
// stage pixels [xa, x_hi) (xa % 8 == 0) of source line y into lds[0 ..): lane handles 8-pixel groups lane, lane+64, ..
// `packed` (wave-uniform, from the launcher): the source allows the word-load path and needs no colour step before
// the scaler; groups that cannot take it (right edge) and all other sources go pixel by pixel, in a rolled loop so
// that the per-pixel front is instantiated once.
template <class SRC>
GSTAMD_HD void tile_stage_row (const SRC &src, uint32_t *lds, int xa, int x_hi, int y, int lane, int packed)
{
  for (int x0 = xa + 8 * lane; x0 < x_hi; x0 += 8 * 64) {
    if (src_p4 (src) && x0 + 8 <= x_hi) {
      uint32_t px[8];
      src_span8_p4 (src, x0, y, px);
      uint4 *d = (uint4 *) (lds + (x0 - xa));
      d[0] = gstamd_make_uint4 (px[0], px[1], px[2], px[3]);
      d[1] = gstamd_make_uint4 (px[4], px[5], px[6], px[7]);
    } else if (src_p422 (src) && x0 + 8 <= x_hi && src_span8_p422_ok (src, x0, y)) {
      uint32_t px[8];
      src_span8_p422 (src, x0, y, px);
      uint4 *d = (uint4 *) (lds + (x0 - xa));
      d[0] = gstamd_make_uint4 (px[0], px[1], px[2], px[3]);
      d[1] = gstamd_make_uint4 (px[4], px[5], px[6], px[7]);
    } else if (packed && src_span8_ok (src, x0)) {
      uint32_t px[8];
      src_span8 (src, x0, y, px);
      uint4 *d = (uint4 *) (lds + (x0 - xa));
      d[0] = gstamd_make_uint4 (px[0], px[1], px[2], px[3]);
      d[1] = gstamd_make_uint4 (px[4], px[5], px[6], px[7]);
    } else {
      const int xe = x0 + 8 < x_hi ? x0 + 8 : x_hi;
#pragma unroll 1
      for (int x = x0; x < xe; x++)
        lds[x - xa] = src.at (x, y);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// packed 16-bit arithmetic on pixel halves (e = bytes 0,2; o = bytes 1,3; one byte per 16-bit lane)
// ------------------------------------------------------------------------------------------------
GSTAMD_HD uint32_t pk_sub16 (uint32_t a, uint32_t b)
{
#ifdef __HIPCC__
  typedef unsigned short us2 __attribute__ ((ext_vector_type (2)));
  return __builtin_bit_cast (uint32_t, (us2) (__builtin_bit_cast (us2, a) - __builtin_bit_cast (us2, b)));
#else
  return (((a & 0xffffu) - (b & 0xffffu)) & 0xffffu) | (((a >> 16) - (b >> 16)) << 16);
#endif
}

// a * b + c per 16-bit lane, wrapping (v_pk_mad_u16): mullw + addw of the ORC programs
GSTAMD_HD uint32_t pk_mad16 (uint32_t a, uint32_t b, uint32_t c)
{
#ifdef __HIPCC__
  typedef unsigned short us2 __attribute__ ((ext_vector_type (2)));
  return __builtin_bit_cast (uint32_t, (us2) (__builtin_bit_cast (us2, a) * __builtin_bit_cast (us2, b) + __builtin_bit_cast (us2, c)));
#else
  const uint32_t lo = ((a & 0xffffu) * (b & 0xffffu) + (c & 0xffffu)) & 0xffffu;
  const uint32_t hi = ((a >> 16) * (b >> 16) + (c >> 16)) & 0xffffu;
  return lo | (hi << 16);
#endif
}

GSTAMD_HD uint32_t umul24 (uint32_t a, uint32_t b)
{
#ifdef __HIPCC__
  uint32_t r;
  asm ("v_mul_u32_u24 %0, %1, %2" : "=v" (r) : "v" (a), "v" (b));
  return r;
#else
  return a * b;
#endif
}

// ldreslinl on two source pixels (video-orc-dist.c:26162-26195): per channel (a * (256 - f) + b * f) >> 8, never above
// 255 * 256, so two channels share one 24-bit multiply-add
GSTAMD_HD void h2tap_eo (uint32_t a, uint32_t b, uint32_t fr, uint32_t &e, uint32_t &o)
{
  const uint32_t nf = 256u - fr;
  e = pk_shr<8> (umul24 (a & 0x00ff00ffu, nf) + umul24 (b & 0x00ff00ffu, fr));
  o = pk_shr<8> (umul24 (pk_shr<8> (a), nf) + umul24 (pk_shr<8> (b), fr));
}

// video_orc_resample_v_2tap_u8_lq on two channels: s1 + (((s2 - s1) * p1 + 128) >> 8), 16-bit wrapping, low byte kept
GSTAMD_HD uint32_t v2tap_pk (uint32_t s1, uint32_t s2, uint32_t p1_splat)
{
  const uint32_t m = pk_mad16 (pk_sub16 (s2, s1), p1_splat, 0x00800080u);
  return (pk_shr<8> (m) + s1) & 0x00ff00ffu;
}

// last stage of a converter whose post step is the AYUV->ARGB matrix with provably no 16-bit wrap, no alpha
// operation and an alpha channel known to be 0xff (planner: VideoPlan::fast_post): fast_pixel of video_fast.h
struct PostFast {
  int use;
  FastParams fp;
};

GSTAMD_HD uint32_t post_px (const Dst &dst, const PostFast &pf, uint32_t px)
{
  if (pf.use) {
    const uint32_t z = px ^ 0x80808080u;
    return fast_pixel (pf.fp, z, 0x0c01010cu, z, 0x0c02020cu, 0x0c03030cu);
  }
  return dst.final ? pack_px (dst.pack_pos, apply_color (dst.post, px)) : px;
}

GSTAMD_HD void store_px (const Dst &dst, int x, int y, uint32_t v)
{
  uint32_t *p = (uint32_t *) (dst.p + (size_t) y * dst.stride + 4 * (size_t) x);
#ifdef __HIPCC__
  if (dst.final)
    __builtin_nontemporal_store (v, p);     // write-once output
  else
    *p = v;                                 // intermediate image: the next pass reads it straight back (L2 / Infinity Cache)
#else
  *p = v;
#endif
}

// lane part of the fused nearest/2-tap scaler: outputs t0 + lane + 64 * i (i < 4) of row y from the staged rows.
// Neighbouring lanes take neighbouring outputs, so their LDS reads fall into neighbouring banks.
GSTAMD_HD void scale2x2_tile_lane (const uint32_t *la, const uint32_t *lb, int xa, const ScaleDev &sh, const ScaleDev &sv, int h_first,
    const Dst &dst, const PostFast &pf, int t0, int t1, int y, int lane)
{
  const bool v2 = sv.kind == SCALE_2TAP, h2 = sh.kind == SCALE_2TAP;
  const uint32_t p1s = v2 ? ((uint32_t) (uint16_t) sv.taps[(size_t) y * 2 + 1]) * 0x00010001u : 0u;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int x = t0 + lane + 64 * i;
    if (x >= t1)
      break;
    uint32_t e, o;
    if (h2) {
      const int tmp = x * sh.inc;
      const int idx = (tmp >> 16) - xa;
      const uint32_t fr = (uint32_t) (tmp >> 8) & 0xffu;
      if (!v2) {
        h2tap_eo (la[idx], la[idx + 1], fr, e, o);
      } else if (h_first) {
        uint32_t e2, o2;
        h2tap_eo (la[idx], la[idx + 1], fr, e, o);
        h2tap_eo (lb[idx], lb[idx + 1], fr, e2, o2);
        e = v2tap_pk (e, e2, p1s);
        o = v2tap_pk (o, o2, p1s);
      } else {                            // vertical first: filter the two source columns, then blend them
        const uint32_t a1 = la[idx], a2 = lb[idx], b1 = la[idx + 1], b2 = lb[idx + 1];
        const uint32_t ae = v2tap_pk (a1 & 0x00ff00ffu, a2 & 0x00ff00ffu, p1s), ao = v2tap_pk (pk_shr<8> (a1), pk_shr<8> (a2), p1s);
        const uint32_t be = v2tap_pk (b1 & 0x00ff00ffu, b2 & 0x00ff00ffu, p1s), bo = v2tap_pk (pk_shr<8> (b1), pk_shr<8> (b2), p1s);
        const uint32_t nf = 256u - fr;
        e = pk_shr<8> (umul24 (ae, nf) + umul24 (be, fr));
        o = pk_shr<8> (umul24 (ao, nf) + umul24 (bo, fr));
      }
    } else {
      const int idx = (int) sh.offset[x] - xa;
      const uint32_t a1 = la[idx];
      e = a1 & 0x00ff00ffu;
      o = pk_shr<8> (a1);
      if (v2) {
        const uint32_t a2 = lb[idx];
        e = v2tap_pk (e, a2 & 0x00ff00ffu, p1s);
        o = v2tap_pk (o, pk_shr<8> (a2), p1s);
      }
    }
    store_px (dst, x, y, post_px (dst, pf, e | (o << 8)));
  }
}

// ------------------------------------------------------------------------------------------------
// The fused nearest / 2-tap scaler for 4-byte packed sources without a colour step in front of it (BGRA -> BGRA `videoscale`, the
// scaler of a converted frame on its way to a display size): a lane owns FOUR neighbouring outputs of `rows` consecutive rows, reads
// the source pixels it needs straight from memory (neighbouring lanes share them through L1; at 1080p -> 4K a lane's 4 outputs sit on
// 3 source pixels) and leaves one 16-byte streaming store per row.  The wave-tile form above spends a workgroup and two staged source
// rows on 256 outputs of ONE row - 70 us for a 4K destination; this one is bound by the destination's bytes.  Same integers: the
// arithmetic is scale2x2_tile_lane's, line by line.
// ------------------------------------------------------------------------------------------------
struct Bil4Params {
  const uint8_t *src;
  int sstride, src_w, src_h;
  uint32_t sel_in;              // v_perm selector: memory bytes -> A, c1, c2, c3 (the unpack of the 4-byte format)
  ScaleDev sh, sv;
  int h_first;
  int out_w, out_h, rows;
  // the packed 4:2:2 form (bilinear4_rows_lane<1>): `src` is a YUY2 / UYVY / YVYU / VYUY frame, a fetch unpacks one pixel of it - fetch_front's
  // A = 0xff, Y, and the horizontally upsampled U, V (chroma_h_at); byte positions, ChromaH and FrontParams::swap_k of the source
  int pos1, pos2, pos3, chroma_h, swap_k;
};

template <int P422>
GSTAMD_HD uint32_t bil4_fetch (const Bil4Params &b, const uint8_t *row, int x)
{
  x = x < b.src_w - 1 ? x : b.src_w - 1;
  if (!P422)
    return swizzle4_px (*(const uint32_t *) (row + 4 * (size_t) x), b.sel_in);
  return p422_px (row, x, b.src_w, b.pos1, b.pos2, b.pos3, b.chroma_h, b.swap_k);
}

template <int P422 = 0>
GSTAMD_HD void bilinear4_rows_lane (const Bil4Params &b, const Dst &dst, const PostFast &pf, int x0, int y0)
{
  if (x0 >= b.out_w)
    return;
  const bool v2 = b.sv.kind == SCALE_2TAP, h2 = b.sh.kind == SCALE_2TAP;
  int idx[4];
  uint32_t fr[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int x = x0 + i < b.out_w ? x0 + i : b.out_w - 1;
    if (h2) {
      const int tmp = x * b.sh.inc;
      idx[i] = tmp >> 16;
      fr[i] = (uint32_t) (tmp >> 8) & 0xffu;
    } else {
      idx[i] = (int) b.sh.offset[x];
      fr[i] = 0;
    }
  }
  const int y1 = y0 + b.rows < b.out_h ? y0 + b.rows : b.out_h;
  for (int y = y0; y < y1; y++) {
    const int ya = (int) b.sv.offset[y];
    const uint32_t p1s = v2 ? ((uint32_t) (uint16_t) b.sv.taps[(size_t) y * 2 + 1]) * 0x00010001u : 0u;
    const uint8_t *ra = b.src + (size_t) ya * b.sstride;
    const uint8_t *rb = b.src + (size_t) (ya + 1 < b.src_h ? ya + 1 : b.src_h - 1) * b.sstride;
    uint32_t out[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      uint32_t e, o;
      if (h2) {
        if (!v2) {
          h2tap_eo (bil4_fetch<P422> (b, ra, idx[i]), bil4_fetch<P422> (b, ra, idx[i] + 1), fr[i], e, o);
        } else if (b.h_first) {
          uint32_t e2, o2;
          h2tap_eo (bil4_fetch<P422> (b, ra, idx[i]), bil4_fetch<P422> (b, ra, idx[i] + 1), fr[i], e, o);
          h2tap_eo (bil4_fetch<P422> (b, rb, idx[i]), bil4_fetch<P422> (b, rb, idx[i] + 1), fr[i], e2, o2);
          e = v2tap_pk (e, e2, p1s);
          o = v2tap_pk (o, o2, p1s);
        } else {
          const uint32_t a1 = bil4_fetch<P422> (b, ra, idx[i]), a2 = bil4_fetch<P422> (b, rb, idx[i]);
          const uint32_t b1 = bil4_fetch<P422> (b, ra, idx[i] + 1), b2 = bil4_fetch<P422> (b, rb, idx[i] + 1);
          const uint32_t ae = v2tap_pk (a1 & 0x00ff00ffu, a2 & 0x00ff00ffu, p1s), ao = v2tap_pk (pk_shr<8> (a1), pk_shr<8> (a2), p1s);
          const uint32_t be = v2tap_pk (b1 & 0x00ff00ffu, b2 & 0x00ff00ffu, p1s), bo = v2tap_pk (pk_shr<8> (b1), pk_shr<8> (b2), p1s);
          const uint32_t nf = 256u - fr[i];
          e = pk_shr<8> (umul24 (ae, nf) + umul24 (be, fr[i]));
          o = pk_shr<8> (umul24 (ao, nf) + umul24 (bo, fr[i]));
        }
      } else {
        const uint32_t a1 = bil4_fetch<P422> (b, ra, idx[i]);
        e = a1 & 0x00ff00ffu;
        o = pk_shr<8> (a1);
        if (v2) {
          const uint32_t a2 = bil4_fetch<P422> (b, rb, idx[i]);
          e = v2tap_pk (e, a2 & 0x00ff00ffu, p1s);
          o = v2tap_pk (o, pk_shr<8> (a2), p1s);
        }
      }
      out[i] = post_px (dst, pf, e | (o << 8));
    }
    uint8_t *d = dst.p + (size_t) y * dst.stride + 4 * (size_t) x0;
    if (x0 + 4 <= b.out_w && (((uintptr_t) d) & 15) == 0) {
#ifdef __HIPCC__
      typedef unsigned int u32x4 __attribute__ ((ext_vector_type (4)));
      const u32x4 v = {out[0], out[1], out[2], out[3]};
      __builtin_nontemporal_store (v, (u32x4 *) d);
#else
      for (int i = 0; i < 4; i++)
        ((uint32_t *) d)[i] = out[i];
#endif
    } else {
      for (int i = 0; i < 4 && x0 + i < b.out_w; i++)
        ((uint32_t *) d)[i] = out[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same scaler for plans that filter horizontally first with two taps both ways (what an enlargement of 4-byte pixels is: BGRA 1080p -> 4K,
// the second stage of NV12 1080p -> BGRA 4K), with the source lines CARRIED DOWN the rows: bilinear4_rows_lane fetched 16 separate pixels per
// output row (17 vector-memory instructions per four outputs - the chip issues one per ~37 clocks and CU whatever its width, so a 4K destination
// could not go below ~38 us) and filtered both source lines again for every output row.  Here a lane walks a strip of rows with the ldreslinl
// results of source lines ya and ya + 1 of its four outputs in registers: a source line is fetched once (one 8-byte load per output: the pixel
// pair idx, idx + 1) and filtered once, however many output rows blend it - at 2x that is two loads and one 16-byte store per output row.
// The line the NEXT row will need is requested before this row's vertical pass.  Run-time forms of the post stage are decided once per lane,
// not per pixel.  Same integers: h2tap_eo, v2tap_pk, post_px.
// ------------------------------------------------------------------------------------------------
struct Bil4Line {
  uint32_t e[4], o[4];          // h2tap_eo of the lane's four outputs on one source line
};

#ifdef GSTAMD_EMU_BOUNDS
static const uint8_t *g_bil4_lo, *g_bil4_hi;     // the emulator's bounds check of the pair loads (tests/emu)
#endif
GSTAMD_HD uint2 bil4_pair (const uint8_t *__restrict__ row, uint32_t off)
{
#ifdef __HIPCC__
  typedef unsigned int u32x2a __attribute__ ((ext_vector_type (2), aligned (4)));
  const u32x2a v = *(const u32x2a *) (row + off);
  return make_uint2 (v.x, v.y);
#else
#ifdef GSTAMD_EMU_BOUNDS
  if (row + off < g_bil4_lo || row + off + 8 > g_bil4_hi) {
    fprintf (stderr, "bil4_pair: read of 8 bytes at %+ld, frame of %ld bytes\n", (long) (row + off - g_bil4_lo), (long) (g_bil4_hi - g_bil4_lo));
    abort ();
  }
#endif
  uint2 v;
  __builtin_memcpy (&v, row + off, 8);
  return v;
#endif
}

// can the plan take bilinear4_up_lane?  (host)
inline bool bilinear4_up_ok (const Bil4Params &b)
{
  return b.sh.kind == SCALE_2TAP && b.sv.kind == SCALE_2TAP && b.h_first && b.src_w >= 2;
}

// the post stage as one byte permutation, when it is nothing else: an intermediate image, or a final pack without matrix / alpha stage
inline bool bilinear4_plain_sel (const Dst &dst, const PostFast &pf, uint32_t *sel)
{
  if (pf.use || (dst.final && (dst.post.matrix.kind != MATRIX_NONE || dst.post.alpha_kind != ALPHA_NONE)))
    return false;
  *sel = 0x03020100u;
  if (dst.final) {
    *sel = 0;
    for (int j = 0; j < 4; j++)
      *sel |= (uint32_t) j << (8 * dst.pack_pos[j]);
  }
  return true;
}

// TAB: void (int y, int *ya, uint32_t *p1) -> first source line and second tap of output row y (the kernel keeps a strip's entries in a lane
// table and reads them with v_readlane, the emulator takes them from the plan)
template <int PLAIN, class TAB>
GSTAMD_HD void bilinear4_up_lane (const Bil4Params &b, const Dst &dst, const PostFast &pf, uint32_t plain_sel, int x0, int y0, int y1, TAB tab)
{
  /* lanes right of the picture stay (on the first four outputs, storing nothing): the kernel's row table lives in the LANES of the wave and is read
   * with v_readlane - with an early exit here the compiler sinks the table loads below it, lanes that left never load their entry, and a picture
   * narrower than 4 x (rows per strip) outputs reads garbage line numbers (found by the device fuzz: a 25-pixel-wide frame) */
  const bool active = x0 < b.out_w;
  x0 = active ? x0 : 0;
  uint32_t off[4], fr[4], sa[4];
  const uint32_t sb = b.sel_in + 0x04040404u;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int x = x0 + i < b.out_w ? x0 + i : b.out_w - 1;
    const int tmp = x * b.sh.inc;
    const int idx = tmp >> 16;
    fr[i] = (uint32_t) (tmp >> 8) & 0xffu;
    /* the pair (idx, idx + 1); on the row's last pixel the pair before it, read as (second, second): the clamp of bil4_fetch */
    const bool edge = idx >= b.src_w - 1;
    off[i] = 4u * (uint32_t) (edge ? b.src_w - 2 : idx);
    sa[i] = edge ? sb : b.sel_in;
  }
  Bil4Line A, B;
  uint2 P[4];
  int id_a = -1, id_b = -1, id_p = -1;
  const auto fetch = [&] (int line, uint2 *p) {
    const uint8_t *__restrict__ row = b.src + (size_t) line * b.sstride;
#pragma unroll
    for (int i = 0; i < 4; i++)
      p[i] = bil4_pair (row, off[i]);
  };
  const auto hline = [&] (const uint2 *p, Bil4Line &l) {
#pragma unroll
    for (int i = 0; i < 4; i++)
      h2tap_eo (bperm (p[i].y, p[i].x, sa[i]), bperm (p[i].y, p[i].x, sb), fr[i], l.e[i], l.o[i]);
  };
  for (int y = y0; y < y1; y++) {
    int ya;
    uint32_t p1;
    tab (y, &ya, &p1);
    const int yb = ya + 1 < b.src_h ? ya + 1 : b.src_h - 1;
    /* wave-uniform from here to the vertical pass: line numbers only */
    if (id_a != ya) {
      if (id_b == ya) {
        A = B;
      } else if (id_p == ya) {
        hline (P, A);
      } else {
        uint2 t[4];
        fetch (ya, t);
        hline (t, A);
      }
      id_a = ya;
    }
    if (id_b != yb) {
      if (id_a == yb) {
        B = A;
      } else if (id_p == yb) {
        hline (P, B);
      } else {
        uint2 t[4];
        fetch (yb, t);
        hline (t, B);
      }
      id_b = yb;
    }
    if (y + 1 < y1) {
      int yn;
      uint32_t pn;
      tab (y + 1, &yn, &pn);
      const int ynb = yn + 1 < b.src_h ? yn + 1 : b.src_h - 1;
      const int want = (yn != id_a && yn != id_b) ? yn : ynb;
      if (want != id_a && want != id_b && want != id_p) {
        fetch (want, P);                /* in flight during this row's vertical pass and store */
        id_p = want;
      }
    }
    const uint32_t p1s = (p1 & 0xffffu) * 0x00010001u;
    uint32_t out[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t e = v2tap_pk (A.e[i], B.e[i], p1s), o = v2tap_pk (A.o[i], B.o[i], p1s);
      const uint32_t px = e | (o << 8);
      out[i] = PLAIN ? bperm (0u, px, plain_sel) : post_px (dst, pf, px);
    }
    uint8_t *d = dst.p + (size_t) y * dst.stride + 4 * (size_t) x0;
    if (!active)
      continue;
    if (x0 + 4 <= b.out_w && (((uintptr_t) d) & 15) == 0) {
#ifdef __HIPCC__
      typedef unsigned int u32x4 __attribute__ ((ext_vector_type (4)));
      const u32x4 v = {out[0], out[1], out[2], out[3]};
      if (dst.final)
        __builtin_nontemporal_store (v, (u32x4 *) d);
      else
        *(u32x4 *) d = v;
#else
      for (int i = 0; i < 4; i++)
        ((uint32_t *) d)[i] = out[i];
#endif
    } else {
      for (int i = 0; i < 4 && x0 + i < b.out_w; i++)
        ((uint32_t *) d)[i] = out[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// N-tap horizontal pass as byte dot products (v_dot4_i32_i8).  The wave stages Y, U, V of the source span as three
// byte planes, each byte XOR 0x80 (= value - 128 as int8); one output channel is
//     sum (px * tap) = sum ((px - 128) * tap) + 128 * sum (tap),      sum (tap) = 64 (checked by the planner)
// i.e. nw dot4 instructions instead of 4 * n_taps multiply-adds; the 16-bit wrap of the ORC program
// (video-orc.orc:2388-2480) is applied to the exact sum afterwards, which is the same thing.  Alpha is 0xff on both
// sides (opaque source, taps sum to 1.0).
// ------------------------------------------------------------------------------------------------
GSTAMD_HD int dot4_i8 (uint32_t a, uint32_t b, int acc)
{
#ifdef __HIPCC__
  return __builtin_amdgcn_sdot4 ((int) a, (int) b, acc, false);
#else
  for (int k = 0; k < 4; k++)
    acc += (int) (int8_t) (a >> (8 * k)) * (int) (int8_t) (b >> (8 * k));
  return acc;
#endif
}

// stage source pixels [xa, x_hi) of line y as byte planes py/pu/pv (byte i = pixel xa + i, XOR 0x80)
GSTAMD_HD void tile_stage_row_planes (const SrcFront &src, uint32_t *py, uint32_t *pu, uint32_t *pv, int xa, int x_hi, int y, int lane,
    int packed)
{
  for (int x0 = xa + 8 * lane; x0 < x_hi; x0 += 8 * 64) {
    const int w0 = (x0 - xa) >> 2;
    if (src_p422 (src) && x0 + 8 <= x_hi && src_span8_p422_ok (src, x0, y)) {
      /* packed 4:2:2 (alpha 0xff like the planar sources): the eight pixels of the span from its macropixel words, split into the three byte planes */
      uint32_t px[8];
      src_span8_p422 (src, x0, y, px);
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const uint32_t a = px[4 * h], b = px[4 * h + 1], c = px[4 * h + 2], d = px[4 * h + 3];
        py[w0 + h] = (((a >> 8) & 0xffu) | (((b >> 8) & 0xffu) << 8) | (((c >> 8) & 0xffu) << 16) | (((d >> 8) & 0xffu) << 24)) ^ 0x80808080u;
        pu[w0 + h] = (((a >> 16) & 0xffu) | (((b >> 16) & 0xffu) << 8) | (((c >> 16) & 0xffu) << 16) | (((d >> 16) & 0xffu) << 24)) ^ 0x80808080u;
        pv[w0 + h] = ((a >> 24) | ((b >> 24) << 8) | ((c >> 24) << 16) | ((d >> 24) << 24)) ^ 0x80808080u;
      }
    } else if (packed && src_span8_ok (src, x0)) {
      uint2 yy;
      uint32_t ca[8];
      front_chroma8_packed_any (src.f, src.pl, src.vpair, x0, y, yy, ca);
      py[w0] = yy.x ^ 0x80808080u;
      py[w0 + 1] = yy.y ^ 0x80808080u;
      // ca = [U, 0, V, 0]: two pixels -> [U0, U1, V0, V1], then two of those -> four U bytes / four V bytes
      const uint32_t t01 = ca[0] | (ca[1] << 8), t23 = ca[2] | (ca[3] << 8), t45 = ca[4] | (ca[5] << 8), t67 = ca[6] | (ca[7] << 8);
      pu[w0] = bperm (t23, t01, 0x05040100u) ^ 0x80808080u;
      pv[w0] = bperm (t23, t01, 0x07060302u) ^ 0x80808080u;
      pu[w0 + 1] = bperm (t67, t45, 0x05040100u) ^ 0x80808080u;
      pv[w0 + 1] = bperm (t67, t45, 0x07060302u) ^ 0x80808080u;
    } else {
      const int xe = x0 + 8 < x_hi ? x0 + 8 : x_hi;
#pragma unroll 1
      for (int x = x0; x < xe; x++) {
        const uint32_t px = src.at (x, y) ^ 0x80808080u;
        ((uint8_t *) py)[x - xa] = (uint8_t) (px >> 8);
        ((uint8_t *) pu)[x - xa] = (uint8_t) (px >> 16);
        ((uint8_t *) pv)[x - xa] = (uint8_t) (px >> 24);
      }
    }
  }
}

// per-lane filter data of the (up to) 4 outputs t0 + lane + 64 * i: requested before the staging step so that the
// table reads overlap it
template <int NW>
struct Dot4Taps {
  int w0[4];                          // first LDS word of the aligned filter window
  uint32_t t[4][NW > 0 ? NW : 1];
};

template <int NW>
GSTAMD_HD void hscale_dot4_fetch (const ScaleDev &sd, int xa, int t0, int t1, int lane, Dot4Taps<NW> &ft)
{
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int x = t0 + lane + 64 * i;
    const int xc = x < t1 ? x : t1 - 1;
    ft.w0[i] = ((int) sd.offset[xc] - xa) >> 2;
    if (NW > 0) {
      const uint32_t *tw = sd.tapw + (size_t) xc * sd.nw4;
#pragma unroll
      for (int k = 0; k < NW; k++)
        ft.t[i][k] = tw[k];
    }
  }
}

template <int NW>
GSTAMD_HD void hscale_dot4_lane (const uint32_t *py, const uint32_t *pu, const uint32_t *pv, const Dot4Taps<NW> &ft, const ScaleDev &sd, int nw,
    const Dst &dst, const PostFast &pf, int t0, int t1, int y, int lane)
{
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int x = t0 + lane + 64 * i;
    if (x >= t1)
      break;
    const int w0 = ft.w0[i];
    int ay = 128 * 64, au = 128 * 64, av = 128 * 64;
    if (NW > 0) {
#pragma unroll
      for (int k = 0; k < NW; k++) {
        const uint32_t t = ft.t[i][k];
        ay = dot4_i8 (py[w0 + k], t, ay);
        au = dot4_i8 (pu[w0 + k], t, au);
        av = dot4_i8 (pv[w0 + k], t, av);
      }
    } else {
      const uint32_t *tw = sd.tapw + (size_t) x * sd.nw4;
      for (int k = 0; k < nw; k++) {
        const uint32_t t = tw[k];
        ay = dot4_i8 (py[w0 + k], t, ay);
        au = dot4_i8 (pu[w0 + k], t, au);
        av = dot4_i8 (pv[w0 + k], t, av);
      }
    }
    const uint32_t px = 0xffu | ((uint32_t) lq_round (ay) << 8) | ((uint32_t) lq_round (au) << 16) | ((uint32_t) lq_round (av) << 24);
    store_px (dst, x, y, post_px (dst, pf, px));
  }
}

// ------------------------------------------------------------------------------------------------
// vertical N-tap pass, 4 pixels per lane, packed 16-bit accumulators (video_orc_resample_v_multaps*_u8_lq:
// mullw / addw accumulate with 16-bit wrap, then (acc + 32) >> 6 arithmetic and unsigned saturation)
// ------------------------------------------------------------------------------------------------
// per 16-bit lane: clamp (((int16) (x + 32)) >> 6, 0, 255)
GSTAMD_HD uint32_t pk_lq_finish (uint32_t acc)
{
#ifdef __HIPCC__
  typedef short s2 __attribute__ ((ext_vector_type (2)));
  s2 v = __builtin_bit_cast (s2, acc) + (s2) (short) 32;
  v = v >> (short) 6;
  v = __builtin_elementwise_min (__builtin_elementwise_max (v, (s2) (short) 0), (s2) (short) 255);
  return __builtin_bit_cast (uint32_t, v);
#else
  uint32_t r = 0;
  for (int h = 0; h < 2; h++) {
    int v = (int) (int16_t) (uint16_t) (((acc >> (16 * h)) & 0xffffu) + 32u);
    v >>= 6;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    r |= (uint32_t) v << (16 * h);
  }
  return r;
#endif
}

// outputs x0 .. x0+3 (x0 + 4 <= width) of row y from an AYUV image
GSTAMD_HD void vscale_ntap_lane4 (const SrcImage &src, const ScaleDev &sd, const Dst &dst, const PostFast &pf, int x0, int y)
{
  struct __attribute__ ((aligned (4))) W4 { uint32_t v[4]; };
  const int off = (int) sd.offset[y];
  const int16_t *t = sd.taps + (size_t) y * sd.n_taps;
  uint32_t ae[4] = {0, 0, 0, 0}, ao[4] = {0, 0, 0, 0};
  const uint8_t *p = src.p + (size_t) off * src.stride + 4 * (size_t) x0;
#ifndef GSTAMD_VSCALE_UNROLL8
#pragma unroll 16
#else
#pragma unroll 8
#endif
  for (int l = 0; l < sd.n_taps; l++) {
    const W4 w = *(const W4 *) (p + (size_t) l * src.stride);
    const uint32_t ts = (uint32_t) (uint16_t) t[l] * 0x00010001u;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      ae[i] = pk_mad16 (w.v[i] & 0x00ff00ffu, ts, ae[i]);
      ao[i] = pk_mad16 (pk_shr<8> (w.v[i]), ts, ao[i]);
    }
  }
  uint32_t o[4];
#pragma unroll
  for (int i = 0; i < 4; i++)
    o[i] = post_px (dst, pf, pk_lq_finish (ae[i]) | (pk_lq_finish (ao[i]) << 8));
  uint8_t *d = dst.p + (size_t) y * dst.stride + 4 * (size_t) x0;
#ifdef __HIPCC__
  typedef unsigned int u32x4_a4 __attribute__ ((ext_vector_type (4), aligned (4)));
  const u32x4_a4 v = {o[0], o[1], o[2], o[3]};
  __builtin_nontemporal_store (v, (u32x4_a4 *) d);
#else
  W4 v = {{o[0], o[1], o[2], o[3]}};
  *(W4 *) d = v;
#endif
}

// the same for R consecutive output rows y0 .. y0+R-1 at once: every source row under the union of their tap windows is
// loaded and unpacked ONCE and multiply-accumulated into the rows it belongs to (a 4:1 reduction with 16 taps shares 12 of
// the 16 rows between neighbouring outputs: 28 row loads instead of 64 for R = 4).  The 16-bit accumulators wrap like the ORC
// program's, so the order of the additions is immaterial.  Which rows a source line feeds is wave-uniform.
template <int R>
GSTAMD_HD void vscale_ntap_lane4_rows (const SrcImage &src, const ScaleDev &sd, const Dst &dst, const PostFast &pf, int x0, int y0, int out_h)
{
  struct __attribute__ ((aligned (4))) W4 { uint32_t v[4]; };
  int off[R], lo = 0x7fffffff, hi = -0x7fffffff;
  const int16_t *t[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int yc = y0 + r < out_h ? y0 + r : out_h - 1;
    off[r] = (int) sd.offset[yc];
    t[r] = sd.taps + (size_t) yc * sd.n_taps;
    lo = off[r] < lo ? off[r] : lo;
    hi = off[r] + sd.n_taps > hi ? off[r] + sd.n_taps : hi;
  }
  uint32_t ae[R][4], ao[R][4];
#pragma unroll
  for (int r = 0; r < R; r++)
#pragma unroll
    for (int i = 0; i < 4; i++)
      ae[r][i] = ao[r][i] = 0;
  const uint8_t *p = src.p + 4 * (size_t) x0;
#pragma unroll 2
  for (int l = lo; l < hi; l++) {
    const W4 w = *(const W4 *) (p + (size_t) l * src.stride);
    uint32_t e[4], o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      e[i] = w.v[i] & 0x00ff00ffu;
      o[i] = pk_shr<8> (w.v[i]);
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int k = l - off[r];
      if (y0 + r < out_h && k >= 0 && k < sd.n_taps) {
        const uint32_t ts = (uint32_t) (uint16_t) t[r][k] * 0x00010001u;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          ae[r][i] = pk_mad16 (e[i], ts, ae[r][i]);
          ao[r][i] = pk_mad16 (o[i], ts, ao[r][i]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; r++) {
    if (y0 + r >= out_h)
      break;
    uint32_t q[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
      q[i] = post_px (dst, pf, pk_lq_finish (ae[r][i]) | (pk_lq_finish (ao[r][i]) << 8));
    uint8_t *d = dst.p + (size_t) (y0 + r) * dst.stride + 4 * (size_t) x0;
#ifdef __HIPCC__
    typedef unsigned int u32x4_a4 __attribute__ ((ext_vector_type (4), aligned (4)));
    const u32x4_a4 v = {q[0], q[1], q[2], q[3]};
    if (dst.final)
      __builtin_nontemporal_store (v, (u32x4_a4 *) d);
    else
      *(u32x4_a4 *) d = v;
#else
    W4 v = {{q[0], q[1], q[2], q[3]}};
    *(W4 *) d = v;
#endif
  }
}

// lane of the multi-row vertical kernel: 4 pixels from x0 of R rows from y0; a ragged last lane goes row by row
template <int R>
GSTAMD_HD void vscale_pk_rows_lane (const SrcImage &src, const ScaleDev &sd, const Dst &dst, const PostFast &pf, int width, int out_h, int x0, int y0)
{
  if (x0 >= width || y0 >= out_h)
    return;
  if (x0 + 4 <= width) {
    vscale_ntap_lane4_rows<R> (src, sd, dst, pf, x0, y0, out_h);
  } else {
#pragma unroll 1
    for (int y = y0; y < y0 + R && y < out_h; y++)
#pragma unroll 1
      for (int x = x0; x < width; x++)
        vscale_body<SrcImage> (src, sd, dst, width, out_h, x, y);
  }
}

// lane of the vertical pass kernel: 4 pixels from x0, the last lane of a row finishes pixel by pixel
GSTAMD_HD void vscale_pk_lane (const SrcImage &src, const ScaleDev &sd, const Dst &dst, const PostFast &pf, int width, int out_h, int x0, int y)
{
  if (x0 >= width || y >= out_h)
    return;
  if (x0 + 4 <= width) {
    vscale_ntap_lane4 (src, sd, dst, pf, x0, y);
  } else {
#pragma unroll 1
    for (int x = x0; x < width; x++)
      vscale_body<SrcImage> (src, sd, dst, width, out_h, x, y);
  }
}

// lane part of a horizontal pass (any kind): outputs t0 + lane + 64 * i
GSTAMD_HD void hscale_tile_lane (const uint32_t *lds, int xa, const ScaleDev &sd, const Dst &dst, const PostFast &pf, int t0, int t1, int y,
    int lane)
{
  const RowOfLds row = {lds, xa};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int x = t0 + lane + 64 * i;
    if (x >= t1)
      break;
    store_px (dst, x, y, post_px (dst, pf, hscale_px (row, sd, x)));
  }
}

}  // namespace gstamd
