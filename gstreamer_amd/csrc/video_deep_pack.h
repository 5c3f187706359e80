// video_deep_pack.h - a 10-bit planar / semi-planar source that SHRINKS into an 8-bit planar / semi-planar (or 3-byte, packed 4:2:2, gray) destination,
// in ONE kernel: the 16-bit front, both u16 scaler passes, the narrowing and the chroma downsampler + packer per 4 x 2 block of the destination.
//
// What the reference runs for P010_10LE 3840x2160 -> NV12 1920x1080 (video-converter.c chain_unpack_line :1406, chain_upsample :1436,
// chain_scale :1685-1717 ahead of chain_convert :1719 because the picture shrinks, chain_downsample :2040, pack): unpack_P010_10LE -> AYUV64 lines,
// video_chroma_up_* on u16, video_scale_h_2tap_u16 / v_2tap_u16 (or the n-tap / nearest forms), video_orc_convert_u16_to_u8, the 8-bit chroma
// downsamplers, pack_NV12.  The composite plan (GammaPlan src16, capi_video.cpp convert_gamma) ran that as k_front_hscale16 -> k_scale16 -> k_gamma_stage
// (narrow) -> k_convert_pack with three images in HBM in between (33 + 17 + 8 MB written and read back for 28 MB of frame bytes: 60 us).  Here the
// packer's pixel source (the SRC / ROWS interface of video_pack.h) IS that chain: pixel (x, y) of the scaled, narrowed picture is evaluated from the
// source planes with the arithmetic of deep_front1_t, front_hscale16_lane and deep_scale_px, value for value.
// One lane makes a block of four pixels (two lines of a 4:2:0 destination) when both passes are 2-tap and the horizontal one reads source pixels 2 x,
// 2 x + 1 for every output x (an exact 2:1: `hx2`): the eight lumas of a source line in one 16-byte load, a chroma row's samples once for all of them.
// Other shapes stay with the composite's launches.
#pragma once
#include "video_deep.h"
#include "video_pack.h"
#include "video_dither_ed.h"
#include "video_gamma.h"

namespace gstamd {

struct DeepPackParams {
  FrontParams f;        // the 10-bit source (crop size)
  Planes pl;            // its planes at the crop origin
  const int *vpair;
  ScaleDev sh, sv;      // horizontal pass (runs first), vertical pass
  int out_w, out_h;     // size after both passes
  int hx2;              // both passes 2-tap, sh.offset[x] == 2 x for every x, taps 0 .. 4096, f.width >= 2 out_w
  Deep16Params mx;      // the convert stage on the scaled 16-bit pixels (video_converter_matrix16: other colorimetry on the two sides - a bt2020 decoder's frames
                        // for a bt709 encoder), ahead of the narrowing / the 16-bit pack; has_matrix 0: none.  (k_deep_scale4's is its own argument.)
};

// one component through the horizontal 2-tap (video_orc_resample_h_2tap_u16) and the vertical one (video_scale_v_2tap_u16): deep_scale_px's lines
// ---- arithmetic -------------------------------------------------------------------------------------------------------------------------------
// Chroma travels as the pair {c1 | c2 << 16} (the two components in the source's storage order: a semi-planar sample as it lies in memory, plane 1 |
// plane 2 of a planar one) and the upsamplers run on both halves at once:
//   (a + b + 1) >> 1 = (a | b) - ((a ^ b) >> 1)          video_chroma_up_h2_cs_u16's odd pixels
//   (a + b) >> 1     = (a & b) + ((a ^ b) >> 1)
//   (3 a + b + 2) >> 2 = ((a + ((a + b) >> 1)) + 1) >> 1  video_chroma_up_h2_u16, video_chroma_up_v2_u16 (deep_front1_t's (6 a + 2 b + 4) >> 3)
// (the last one: with s = a + b even both sides are (2 a + s + 2) >> 2; with s odd the right side is (2 a + s + 1) >> 2 and 2 a + s + 2 is odd, so
// the floor does not move).  No half ever carries into the other: each intermediate is at most max (a, b).
GSTAMD_HD uint32_t pk_half (uint32_t v)          // both halves >> 1 (v_pk_lshrrev_b16: no mask to apply)
{
#ifdef __HIPCC__
  typedef unsigned short u2 __attribute__ ((ext_vector_type (2)));
  u2 t = __builtin_bit_cast (u2, v);
  t >>= 1;
  return __builtin_bit_cast (uint32_t, t);
#else
  return (v >> 1) & 0x7fff7fffu;
#endif
}
GSTAMD_HD uint32_t pk_avgc (uint32_t a, uint32_t b) { return (a | b) - pk_half (a ^ b); }
GSTAMD_HD uint32_t pk_avgf (uint32_t a, uint32_t b) { return (a & b) + pk_half (a ^ b); }
// {lo (a) | lo (b) << 16} and {hi (a) | hi (b) << 16}: one v_perm_b32 each
GSTAMD_HD uint32_t pk_lo2 (uint32_t a, uint32_t b)
{
#ifdef __HIPCC__
  return __builtin_amdgcn_perm (b, a, 0x05040100u);
#else
  return (a & 0xffffu) | (b << 16);
#endif
}
GSTAMD_HD uint32_t pk_hi2 (uint32_t a, uint32_t b)
{
#ifdef __HIPCC__
  return __builtin_amdgcn_perm (b, a, 0x07060302u);
#else
  return (a >> 16) | (b & 0xffff0000u);
#endif
}
GSTAMD_HD uint32_t pk_f31 (uint32_t a, uint32_t b) { return pk_avgc (a, pk_avgf (a, b)); }

// deep_widen_w on both halves of a word
struct PkWiden { int sh, bits; uint32_t m1, m2; };
GSTAMD_HD PkWiden pk_widen_params (int hi_depth)
{
  const Widen w = deep_widen_params (hi_depth);
  PkWiden p;
  p.sh = w.sh, p.bits = w.bits;
  p.m1 = ((0xffffu << w.sh) & 0xffffu) * 0x10001u;
  p.m2 = (0xffffu >> w.bits) * 0x10001u;
  return p;
}
GSTAMD_HD uint32_t pk_widen (const PkWiden &w, uint32_t v)
{
  const uint32_t t = (v << w.sh) & w.m1;
  return t | ((t >> w.bits) & w.m2);
}

// lo (pair) * lo (taps) + hi (pair) * hi (taps) + acc on unsigned 16-bit halves: the horizontal 2-tap (video_orc_resample_h_2tap_u16) on the pair
// {first source pixel | second << 16} with its taps as they lie in the table - taps of a 2-tap pass are 0 .. 4096, the sum stays below 2^29
GSTAMD_HD uint32_t pk_udot2 (uint32_t pair, uint32_t taps, uint32_t acc)
{
#ifdef __HIPCC__
  typedef unsigned short u2 __attribute__ ((ext_vector_type (2)));
  return __builtin_amdgcn_udot2 (__builtin_bit_cast (u2, pair), __builtin_bit_cast (u2, taps), acc, true);       /* (clamp: keeps the three-operand form, bilh_dot2) */
#else
  return (pair & 0xffffu) * (taps & 0xffffu) + (pair >> 16) * (taps >> 16) + acc;
#endif
}
GSTAMD_HD int deep_h2tap_pk (uint32_t pair, uint32_t taps)
{
  const uint32_t v = pk_udot2 (pair, taps, 4096u) >> 12;
  return (int) (v > 65535u ? 65535u : v);
}
// video_scale_v_2tap_u16: l1 + (((l2 - l1) * p1 + 4096) >> 12), clamped to 16 bits (deep_scale_px) = (l1 (4096 - p1) + l2 p1 + 4096) >> 12 for
// p1 = 0 .. 4096 (l1 is an integer: it moves inside the floor; nothing wraps, the sum stays below 2^29): one dot product on {l1 | l2 << 16} with
// {4096 - p1 | p1 << 16}.  The sum before its shift; deep_v16: the 16-bit value, deep_v8: its high byte (min (x >> 12, 65535) >> 8 = min (x >> 20, 255))
GSTAMD_HD uint32_t deep_v2tap_sum (int s1, int s2, uint32_t vt) { return pk_udot2 ((uint32_t) s1 | ((uint32_t) s2 << 16), vt, 4096u); }
GSTAMD_HD uint32_t deep_v16 (int s1, int s2, uint32_t vt)
{
  const uint32_t v = deep_v2tap_sum (s1, s2, vt) >> 12;
  return v > 65535u ? 65535u : v;
}
GSTAMD_HD uint32_t deep_v8 (int s1, int s2, uint32_t vt)
{
  const uint32_t v = deep_v2tap_sum (s1, s2, vt) >> 20;
  return v > 255u ? 255u : v;
}

// a source line after the horizontal pass at the block's four positions: luma, and the two chroma components
struct DeepHLine {
  int y[4], c1[4], c2[4], al[4];        // (al: the alpha of AYUV64 lines, 0xffff through the pass - read by the 4-byte destinations only)
};

// The packer's row source (video_pack.h pack_planar_block4's ROWS): pixels x0 .. x0 + 3 of line y of the scaled, narrowed picture from the source planes.
// Source pixels 2 x0 .. 2 x0 + 7, chroma samples x0 - 1 .. x0 + 4.  The chroma of the pixels left and right of the block (the cosited downsampler's
// neighbours) is what the neighbouring LANES made: the wave trades them (ds_bpermute), and the kernel lays the blocks over the lanes so that every block
// that stores has both neighbours in its wave (k_deep_scale_pack: 62 storing lanes between two that only compute).  On the host (tests/emu) the
// neighbour blocks are evaluated in place.
// SH: bit 0 - semi-planar source (0: three planes); bit 1 - the stored words hold their bits on top (P010 / P012 / P016: widening is `v | v >> bits`, no shift and
// mask ahead of it - two instructions a word less, and a lane widens 44 words)
template <int SH, int CH>
struct DeepScaledSrc {
  static constexpr int SEMI = SH & 1, TOP = SH >> 1;
  DeepPackParams d;

  GSTAMD_HD uint32_t widen (const PkWiden &w, uint32_t v) const { return TOP ? (v | ((v >> w.bits) & w.m2)) : pk_widen (w, v); }

  GSTAMD_HD bool ok4 (int, int y) const { return y < d.out_h; }

  // chroma samples x0 - 1 .. x0 + 4 of chroma row crow (indices clamped into the row) widened, then the horizontal upsampler at source pixels
  // 2 x0 .. 2 x0 + 7 (pixel j: sample index 1 + (j >> 1)): deep_front1_t's cu / cv, as pairs
  GSTAMD_HD void crow8 (const PkWiden &wd, int cw, int crow, int x0, uint32_t *o) const
  {
    const int km1 = x0 >= 1 ? x0 - 1 : 0, kp4 = x0 + 4 < cw ? x0 + 4 : cw - 1;
    uint32_t s[6];
    s[0] = s[5] = 0;
    if (SEMI) {
      const uint8_t *row = d.pl.p[1] + (ptrdiff_t) crow * d.pl.stride[1];
      const uint32_t *q = (const uint32_t *) row;
      const uint4 m = *(const uint4 *) (row + 4 * (ptrdiff_t) x0);
      s[1] = m.x, s[2] = m.y, s[3] = m.z, s[4] = m.w;
      if (CH == CHROMA_H_H2)
        s[0] = q[km1];
      if (CH != CHROMA_H_NONE)
        s[5] = q[kp4];
    } else {
      const uint8_t *ra = d.pl.p[1] + (ptrdiff_t) crow * d.pl.stride[1], *rb = d.pl.p[2] + (ptrdiff_t) crow * d.pl.stride[2];
      const uint16_t *qa = (const uint16_t *) ra, *qb = (const uint16_t *) rb;
      const uint2 ma = *(const uint2 *) (ra + 2 * (ptrdiff_t) x0), mb = *(const uint2 *) (rb + 2 * (ptrdiff_t) x0);
      s[1] = (ma.x & 0xffffu) | (mb.x << 16), s[2] = (ma.x >> 16) | (mb.x & 0xffff0000u);
      s[3] = (ma.y & 0xffffu) | (mb.y << 16), s[4] = (ma.y >> 16) | (mb.y & 0xffff0000u);
      if (CH == CHROMA_H_H2)
        s[0] = (uint32_t) qa[km1] | ((uint32_t) qb[km1] << 16);
      if (CH != CHROMA_H_NONE)
        s[5] = (uint32_t) qa[kp4] | ((uint32_t) qb[kp4] << 16);
    }
#pragma unroll
    for (int i = 0; i < 6; i++)
      s[i] = widen (wd, s[i]);
    const int w = d.f.width;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = 1 + (j >> 1), odd = j & 1, sx = 2 * x0 + j;
      uint32_t a = s[k];
      if (CH == CHROMA_H_H2_CS) {
        if (odd) {
          const uint32_t f = pk_avgc (a, s[k + 1]);
          a = j == 7 ? (sx < w - 1 ? f : a) : f;          /* (only a line's last pixel can be the one without a right neighbour) */
        }
      } else if (CH == CHROMA_H_H2) {
        const uint32_t f = pk_f31 (a, s[odd ? k + 1 : k - 1]);
        a = j == 7 ? (sx < w - 1 ? f : a) : (j == 0 ? (sx >= 2 ? f : a) : f);
      }
      o[j] = a;
    }
  }

  // crow8 through the two rows the call before kept (the keys are wave-uniform: a branch, not a select)
  mutable uint32_t ka[8], kb[8];
  mutable int keep_x0 = -1, keep_a = -1, keep_b = -1;
  GSTAMD_HD void crow_get (const PkWiden &wd, int cw, int crow, int x0, uint32_t *o) const
  {
    if (x0 == keep_x0 && crow == keep_a) {
#pragma unroll
      for (int i = 0; i < 8; i++)
        o[i] = ka[i];
    } else if (x0 == keep_x0 && crow == keep_b) {
#pragma unroll
      for (int i = 0; i < 8; i++)
        o[i] = kb[i];
    } else {
      crow8 (wd, cw, crow, x0, o);
    }
  }

  // source line L at the block's positions: luma through the horizontal pass; the line's two upsampled chroma rows blended by its weights (`first` has
  // the weight 3 of 3 : 1), then the pass (deep_front1_t + front_hscale16_lane, value for value)
  GSTAMD_HD void hline4 (const PkWiden &wd, const uint32_t *first, const uint32_t *second, int x0, int L, DeepHLine *o) const
  {
    const uint4 q = *(const uint4 *) (d.pl.p[0] + (ptrdiff_t) L * d.pl.stride[0] + 4 * (ptrdiff_t) x0);
    const uint4 tq = *(const uint4 *) (d.sh.taps + 2 * (size_t) x0);
    const uint32_t lw[4] = {q.x, q.y, q.z, q.w}, tw[4] = {tq.x, tq.y, tq.z, tq.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      o->y[i] = deep_h2tap_pk (widen (wd, lw[i]), tw[i]);
      o->al[i] = deep_h2tap_pk (0xffffffffu, tw[i]);
      const uint32_t e = pk_f31 (first[2 * i], second[2 * i]), g = pk_f31 (first[2 * i + 1], second[2 * i + 1]);        /* even and odd source pixel */
      o->c1[i] = deep_h2tap_pk (pk_lo2 (e, g), tw[i]);
      o->c2[i] = deep_h2tap_pk (pk_hi2 (e, g), tw[i]);
    }
  }

  // destination line y at the block's positions: its two source lines after the horizontal pass, the vertical pass's parameter.  The lines' four chroma
  // rows are three in a picture that halves (rows y - 1, y | y, y + 1: the second line's first row is the first line's second - it goes through the
  // horizontal upsampler once; the keys are wave-uniform)
  GSTAMD_HD uint32_t lines2 (int x0, int y, DeepHLine *a, DeepHLine *b) const
  {
    const int off = (int) d.sv.offset[y];
    const int h = d.f.height;
    const int La = off < 0 ? 0 : (off > h - 1 ? h - 1 : off), Lb = off + 1 > h - 1 ? h - 1 : off + 1;          /* deep_img_at's clamp */
    const FrontRow fa = deep_front_row (d.f, d.vpair, La), fb = deep_front_row (d.f, d.vpair, Lb);
    const PkWiden wd = pk_widen_params (d.f.hi_depth);
    uint32_t c0[8], c1[8];
    crow_get (wd, fa.cw, fa.ra, x0, c0);
    crow_get (wd, fa.cw, fa.rb, x0, c1);
    if (fa.wa == 6)
      hline4 (wd, c0, c1, x0, La, a);
    else
      hline4 (wd, c1, c0, x0, La, a);
    if (fb.ra != fa.rb)
      crow_get (wd, fb.cw, fb.ra, x0, c1);
    crow_get (wd, fb.cw, fb.rb, x0, c0);
    if (fb.wa == 6)
      hline4 (wd, c1, c0, x0, Lb, b);
    else
      hline4 (wd, c0, c1, x0, Lb, b);
    /* the second line's rows are the next destination line's first rows in a picture that halves (lines 4 j + 1 and 4 j + 2 share their pair): kept */
    keep_x0 = x0, keep_a = fb.ra, keep_b = fb.rb;
#pragma unroll
    for (int i = 0; i < 8; i++)
      ka[i] = c1[i], kb[i] = c0[i];
    const uint32_t p1 = (uint32_t) (uint16_t) d.sv.taps[2 * (size_t) y + 1];
    return (4096u - p1) | (p1 << 16);
  }

  GSTAMD_HD bool c1_is_u () const { return SEMI ? d.f.u_plane != 0 : d.f.u_plane == 1; }

  // the block's pixels of destination line y, narrowed (video_orc_convert_u16_to_u8): A Y U V bytes
  GSTAMD_HD uint4 core4 (int x0, int y) const
  {
    if (d.mx.has_matrix) {              /* (wave-uniform) the 16-bit pixels through the convert stage, then the high bytes */
      uint2 p16[4];
      core4_16 (x0, y, p16);
      uint4 r;
      uint32_t o[4];
#pragma unroll
      for (int i = 0; i < 4; i++)
        o[i] = 0xffu | ((p16[i].x >> 24) << 8) | (((p16[i].y >> 8) & 0xffu) << 16) | ((p16[i].y >> 24) << 24);
      r.x = o[0], r.y = o[1], r.z = o[2], r.w = o[3];
      return r;
    }
    DeepHLine a, b;
    const uint32_t vt = lines2 (x0, y, &a, &b);
    const bool usw = c1_is_u ();
    uint32_t px[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t v1 = deep_v8 (a.c1[i], b.c1[i], vt), v2 = deep_v8 (a.c2[i], b.c2[i], vt);
      px[i] = 0xffu | (deep_v8 (a.y[i], b.y[i], vt) << 8) | ((usw ? v1 : v2) << 16) | ((usw ? v2 : v1) << 24);
    }
    uint4 r;
    r.x = px[0], r.y = px[1], r.z = px[2], r.w = px[3];
    return r;
  }

  // the same as AYUV64 pixels {A | Y << 16, U | V << 16}: what the convert stage of a 4-byte destination takes (deep_finish_store4)
  GSTAMD_HD void core4_16 (int x0, int y, uint2 *px) const
  {
    DeepHLine a, b;
    const uint32_t vt = lines2 (x0, y, &a, &b);
    const bool usw = c1_is_u ();
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t v1 = deep_v16 (a.c1[i], b.c1[i], vt), v2 = deep_v16 (a.c2[i], b.c2[i], vt);
      px[i].x = deep_v16 (a.al[i], b.al[i], vt) | (deep_v16 (a.y[i], b.y[i], vt) << 16);
      px[i].y = (usw ? v1 : v2) | ((usw ? v2 : v1) << 16);
      px[i] = gamma_matrix16 (d.mx, px[i]);         /* (has_matrix 0: as it is) */
    }
  }

  GSTAMD_HD uint4 row4n (int x0, int y, bool edges, int, int, uint32_t &em, uint32_t &ep) const
  {
    const uint4 r = core4 (x0, y);
    if (edges) {                /* the pixels left and right of the block; at the picture's edge the block's own first / last pixel (xm, xp) */
#ifdef __HIPCC__
      const uint32_t left = (uint32_t) __shfl_up ((int) r.w, 1), right = (uint32_t) __shfl_down ((int) r.x, 1);
#else
      const uint32_t left = x0 > 0 ? core4 (x0 - 4, y).w : 0, right = x0 + 4 < d.out_w ? core4 (x0 + 4, y).x : 0;
#endif
      em = x0 > 0 ? left : r.x;
      ep = x0 + 4 < d.out_w ? right : r.w;
    }
    return r;
  }
};

// one lane of k_deep_scale_pack: the 4 x (1 << h_sub) block at (x0, yb << h_sub); store false: a lane that works for its neighbours only
template <int SEMI, int CH>
GSTAMD_HD void deep_scale_pack_lane (const PackPlanarParams &pk, const DeepPackParams &dp, const DstPlanes &dst, int x0, int yb, long long dd = 0, bool store = true)
{
  DeepScaledSrc<SEMI, CH> src;
  src.d = dp;
  (void) pack_planar_block4 (pk, src, dst, x0, yb, dd, store);
}

// one lane of k_deep_scale_pack16: the block at (x0, yb << h_sub) of a 10 / 12 / 16-bit planar or semi-planar destination - the chain stays on 16-bit values to
// the end: chroma downsamplers on u16, ordered dither, pack (pack16_block: what k_pack16 does after reading the block from an image).  The neighbours the
// cosited downsampler reads are traded between lanes like k_deep_scale_pack's.
template <int SH, int CH>
GSTAMD_HD void deep_scale_pack16_lane (const PackPlanarParams &pk, int hi_depth, const DitherParams &dt, const DeepPackParams &dp, const DstPlanes16 &d, int x0, int yb,
    bool store = true)
{
  DeepScaledSrc<SH, CH> src;
  src.d = dp;
  const int w = pk.width, h = pk.height, y0 = yb << pk.h_sub, y1 = y0 + 1 < h ? y0 + 1 : h - 1;
  uint2 pa[6], pb[6];
  src.core4_16 (x0, y0, pa + 1);
  const bool two = pk.h_sub || pk.down_v;
  if (two) {
    src.core4_16 (x0, y1, pb + 1);
  } else {
#pragma unroll
    for (int i = 1; i < 5; i++)
      pb[i] = pa[i];
  }
  pa[0] = pa[1], pa[5] = pa[4], pb[0] = pb[1], pb[5] = pb[4];         /* pack16_body's clamps at the picture's edges */
  if (pk.w_sub == 1 && pk.down_h == 2) {
#ifdef __HIPCC__
    uint2 l, r, lb, rb;
    l.x = (uint32_t) __shfl_up ((int) pa[4].x, 1), l.y = (uint32_t) __shfl_up ((int) pa[4].y, 1);
    r.x = (uint32_t) __shfl_down ((int) pa[1].x, 1), r.y = (uint32_t) __shfl_down ((int) pa[1].y, 1);
    lb.x = (uint32_t) __shfl_up ((int) pb[4].x, 1), lb.y = (uint32_t) __shfl_up ((int) pb[4].y, 1);
    rb.x = (uint32_t) __shfl_down ((int) pb[1].x, 1), rb.y = (uint32_t) __shfl_down ((int) pb[1].y, 1);
    if (x0 > 0)
      pa[0] = l, pb[0] = lb;
    if (x0 + 4 < w)
      pa[5] = r, pb[5] = rb;
#else
    uint2 t[4];
    if (x0 > 0) {
      src.core4_16 (x0 - 4, y0, t), pa[0] = t[3];
      if (two)
        src.core4_16 (x0 - 4, y1, t);
      pb[0] = t[3];
    }
    if (x0 + 4 < w) {
      src.core4_16 (x0 + 4, y0, t), pa[5] = t[0];
      if (two)
        src.core4_16 (x0 + 4, y1, t);
      pb[5] = t[0];
    }
#endif
  }
  if (store)
    pack16_block (pk, hi_depth, dt, pa, pb, true, d, x0, yb);
}

GSTAMD_HD void deep_scale_pack16_any (int variant, const PackPlanarParams &pk, int hi_depth, const DitherParams &dt, const DeepPackParams &dp, const DstPlanes16 &d,
    int x0, int yb)
{
  switch (variant) {
    case 0: deep_scale_pack16_lane<0, CHROMA_H_NONE> (pk, hi_depth, dt, dp, d, x0, yb); break;
    case 1: deep_scale_pack16_lane<0, CHROMA_H_H2> (pk, hi_depth, dt, dp, d, x0, yb); break;
    case 2: deep_scale_pack16_lane<0, CHROMA_H_H2_CS> (pk, hi_depth, dt, dp, d, x0, yb); break;
    case 3: deep_scale_pack16_lane<1, CHROMA_H_NONE> (pk, hi_depth, dt, dp, d, x0, yb); break;
    case 4: deep_scale_pack16_lane<1, CHROMA_H_H2> (pk, hi_depth, dt, dp, d, x0, yb); break;
    case 5: deep_scale_pack16_lane<1, CHROMA_H_H2_CS> (pk, hi_depth, dt, dp, d, x0, yb); break;
    case 6: deep_scale_pack16_lane<2, CHROMA_H_NONE> (pk, hi_depth, dt, dp, d, x0, yb); break;
    case 7: deep_scale_pack16_lane<2, CHROMA_H_H2> (pk, hi_depth, dt, dp, d, x0, yb); break;
    case 8: deep_scale_pack16_lane<2, CHROMA_H_H2_CS> (pk, hi_depth, dt, dp, d, x0, yb); break;
    case 9: deep_scale_pack16_lane<3, CHROMA_H_NONE> (pk, hi_depth, dt, dp, d, x0, yb); break;
    case 10: deep_scale_pack16_lane<3, CHROMA_H_H2> (pk, hi_depth, dt, dp, d, x0, yb); break;
    default: deep_scale_pack16_lane<3, CHROMA_H_H2_CS> (pk, hi_depth, dt, dp, d, x0, yb); break;
  }
}

// one lane of k_deep_scale4: pixels x0 .. x0 + 3 of line y of a 4-byte destination - the same chain up to the vertical pass, then the convert stage on
// 16-bit values (video_converter_matrix16), narrowing, alpha and pack (deep_finish_store4: what k_scale16_final does after its pass)
template <int SEMI, int CH>
GSTAMD_HD void deep_scale4_lane (const DeepPackParams &dp, const Deep16Params &dd, const PostParams &post, uint8_t *dst, int dstride, int x0, int y)
{
  if (x0 + 4 > dp.out_w || y >= dp.out_h)
    return;
  DeepScaledSrc<SEMI, CH> src;
  src.d = dp;
  uint2 px[4];
  src.core4_16 (x0, y, px);
  deep_finish_store4 (dd, post, px, (uint32_t *) (dst + (size_t) y * dstride + 4 * (size_t) x0));
}

// the instantiation of a front: deep_front4_variant's number (+ 6 when the stored words hold their bits on top)
GSTAMD_VP int deep_pack_variant (const FrontParams &f)
{
  const int v = deep_front4_variant (f);
  return v < 0 ? v : v + (f.hi_depth == 1 || f.hi_depth == 4 ? 0 : 6);          /* (deep_widen_params: those two keep their bits at the bottom) */
}

GSTAMD_HD void deep_scale4_any (int variant, const DeepPackParams &dp, const Deep16Params &dd, const PostParams &post, uint8_t *dst, int dstride, int x0, int y)
{
  switch (variant) {
    case 0: deep_scale4_lane<0, CHROMA_H_NONE> (dp, dd, post, dst, dstride, x0, y); break;
    case 1: deep_scale4_lane<0, CHROMA_H_H2> (dp, dd, post, dst, dstride, x0, y); break;
    case 2: deep_scale4_lane<0, CHROMA_H_H2_CS> (dp, dd, post, dst, dstride, x0, y); break;
    case 3: deep_scale4_lane<1, CHROMA_H_NONE> (dp, dd, post, dst, dstride, x0, y); break;
    case 4: deep_scale4_lane<1, CHROMA_H_H2> (dp, dd, post, dst, dstride, x0, y); break;
    case 5: deep_scale4_lane<1, CHROMA_H_H2_CS> (dp, dd, post, dst, dstride, x0, y); break;
    case 6: deep_scale4_lane<2, CHROMA_H_NONE> (dp, dd, post, dst, dstride, x0, y); break;
    case 7: deep_scale4_lane<2, CHROMA_H_H2> (dp, dd, post, dst, dstride, x0, y); break;
    case 8: deep_scale4_lane<2, CHROMA_H_H2_CS> (dp, dd, post, dst, dstride, x0, y); break;
    case 9: deep_scale4_lane<3, CHROMA_H_NONE> (dp, dd, post, dst, dstride, x0, y); break;
    case 10: deep_scale4_lane<3, CHROMA_H_H2> (dp, dd, post, dst, dstride, x0, y); break;
    default: deep_scale4_lane<3, CHROMA_H_H2_CS> (dp, dd, post, dst, dstride, x0, y); break;
  }
}

GSTAMD_HD void deep_scale_pack_any (int variant, const PackPlanarParams &pk, const DeepPackParams &dp, const DstPlanes &dst, int x0, int yb)
{
  switch (variant) {
    case 0: deep_scale_pack_lane<0, CHROMA_H_NONE> (pk, dp, dst, x0, yb); break;
    case 1: deep_scale_pack_lane<0, CHROMA_H_H2> (pk, dp, dst, x0, yb); break;
    case 2: deep_scale_pack_lane<0, CHROMA_H_H2_CS> (pk, dp, dst, x0, yb); break;
    case 3: deep_scale_pack_lane<1, CHROMA_H_NONE> (pk, dp, dst, x0, yb); break;
    case 4: deep_scale_pack_lane<1, CHROMA_H_H2> (pk, dp, dst, x0, yb); break;
    case 5: deep_scale_pack_lane<1, CHROMA_H_H2_CS> (pk, dp, dst, x0, yb); break;
    case 6: deep_scale_pack_lane<2, CHROMA_H_NONE> (pk, dp, dst, x0, yb); break;
    case 7: deep_scale_pack_lane<2, CHROMA_H_H2> (pk, dp, dst, x0, yb); break;
    case 8: deep_scale_pack_lane<2, CHROMA_H_H2_CS> (pk, dp, dst, x0, yb); break;
    case 9: deep_scale_pack_lane<3, CHROMA_H_NONE> (pk, dp, dst, x0, yb); break;
    case 10: deep_scale_pack_lane<3, CHROMA_H_H2> (pk, dp, dst, x0, yb); break;
    default: deep_scale_pack_lane<3, CHROMA_H_H2_CS> (pk, dp, dst, x0, yb); break;
  }
}

// host: two passes, horizontal first, both 2-tap, the horizontal one reading source pixels 2 x, 2 x + 1 for output x, taps 0 .. 4096 (pk_udot2), a front
// the lanes know (planes with horizontally subsampled chroma, frame weights 3 : 1)
inline bool deep_passes_halve (const VideoPlan &p)
{
  if (p.passes.size () != 2 || !p.passes[0].horizontal || p.passes[1].horizontal || deep_front4_variant (p.front) < 0 || p.front.chroma_v2 == 2)
    return false;
  const int ow = p.passes[0].out_size;
  bool hx2 = p.passes[0].kind == SCALE_2TAP && p.passes[1].kind == SCALE_2TAP && p.front.width >= 2 * ow && ow >= 8;
  for (int x = 0; hx2 && x < ow; x++)
    hx2 = (int) p.passes[0].offset[x] == 2 * x && p.passes[0].taps[2 * (size_t) x] >= 0 && p.passes[0].taps[2 * (size_t) x + 1] >= 0 &&
        (int) p.passes[0].taps[2 * (size_t) x] + (int) p.passes[0].taps[2 * (size_t) x + 1] <= 8192;
  for (int y = 0; hx2 && y < p.passes[1].out_size; y++)          /* deep_v2tap_sum: the vertical parameter is a weight of 0 .. 4096 */
    hx2 = p.passes[1].taps[2 * (size_t) y + 1] >= 0 && p.passes[1].taps[2 * (size_t) y + 1] <= 4096;
  return hx2;
}

// host: is `p` (a 10-bit source into a 4-byte 8-bit destination that shrinks: VideoPlan::deep16 with the scalers ahead of the convert stage) one
// k_deep_scale4 serves?  Fills everything of `dp` but the pointers.
inline bool deep_scale4_plan_ok (const VideoPlan &p, DeepPackParams *dp)
{
  if (!p.deep16 || p.gamma.on || p.matrix_before_scale || p.interlaced || p.field || p.out_planar || p.plane_mode || p.fout->kind != UNPACK_PACKED4 ||
      p.fout->hi_depth != 0 || !deep_passes_halve (p))
    return false;
  const int ow = p.passes[0].out_size, oh = p.passes[1].out_size;
  if (ow != p.out_info.width || oh != p.out_info.height || (ow % 4) != 0)
    return false;
  memset ((void *) dp, 0, sizeof (*dp));
  dp->f = p.front;
  dp->out_w = ow, dp->out_h = oh;
  dp->sh.kind = p.passes[0].kind, dp->sh.n_taps = p.passes[0].n_taps, dp->sh.inc = p.passes[0].inc;
  dp->sv.kind = p.passes[1].kind, dp->sv.n_taps = p.passes[1].n_taps, dp->sv.inc = p.passes[1].inc;
  dp->hx2 = 1;
  return true;
}

// host: is `p` (a composite plan with a 10-bit source AND a 10 / 12 / 16-bit planar or semi-planar destination: GammaPlan::src16 + pack16) one k_deep_scale_pack16
// serves?  Fills everything of `dp` but the pointers.
inline bool deep_scale_pack16_plan_ok (const VideoPlan &p, DeepPackParams *dp)
{
  const GammaPlan &g = p.gamma;
  if (!g.on || !g.src16 || !g.pack16 || g.store64 || g.planes_fast || g.fused || !g.shrink || !g.dec16.empty () || !g.enc16.empty () ||
      g.alpha_kind != ALPHA_NONE || p.interlaced || p.field || !deep_passes_halve (p) || dither_is_diffusion (g.dither16) ||
      (g.pack.kind != UNPACK_PLANAR && g.pack.kind != UNPACK_SEMI) || g.pack.frame_on || g.pack.tail_swap)
    return false;
  const int ow = p.passes[0].out_size, oh = p.passes[1].out_size;
  if (g.pack.width != ow || g.pack.height != oh || (ow % 4) != 0)
    return false;
  memset ((void *) dp, 0, sizeof (*dp));
  dp->f = p.front;
  dp->out_w = ow, dp->out_h = oh;
  dp->sh.kind = p.passes[0].kind, dp->sh.n_taps = p.passes[0].n_taps, dp->sh.inc = p.passes[0].inc;
  dp->sv.kind = p.passes[1].kind, dp->sv.n_taps = p.passes[1].n_taps, dp->sv.inc = p.passes[1].inc;
  dp->hx2 = 1;
  dp->mx = g.prim;
  return true;
}

// host: is `p` (a composite plan with a 10-bit source, GammaPlan::src16) with the sub-conversion `sub` (8-bit unpack-format image -> destination) one
// this kernel serves?  Fills everything of `dp` but the pointers.
inline bool deep_scale_pack_plan_ok (const VideoPlan &p, const VideoPlan &sub, DeepPackParams *dp)
{
  const GammaPlan &g = p.gamma;
  if (!g.on || !g.src16 || g.planes_fast || g.fused || g.pack16 || g.store64 || !g.shrink || !g.dec16.empty () || !g.enc16.empty () ||
      g.alpha_kind != ALPHA_NONE || g.to_yuv.kind != MATRIX_NONE || p.interlaced || p.field)
    return false;
  if (p.passes.size () != 2 || !p.passes[0].horizontal || p.passes[1].horizontal || deep_front4_variant (p.front) < 0)
    return false;
  if (p.front.chroma_v2 == 2)           /* (a field's chroma weights are not 3 : 1) */
    return false;
  /* the sub-conversion must be the plain pack of the narrowed image: no scaler, no colour or alpha stage, no dither, no alpha plane */
  if (sub.gamma.on || sub.interlaced || sub.field || !sub.out_planar || sub.plane_mode || !sub.passes.empty () || sub.deep16 || sub.dither.on || sub.pack.dither.on ||
      sub.matrix.kind != MATRIX_NONE || sub.post.alpha_kind != ALPHA_NONE || sub.pack.virtual_line || sub.pack.frame_on || sub.front.kind != UNPACK_PACKED4 ||
      sub.front.hi_depth != 0 || GSTAMD_KIND_ALPHA_PLANE (sub.fout->kind) >= 0 || sub.rect.in_x || sub.rect.in_y)
    return false;
  for (int i = 0; i < 4; i++)
    if (sub.front.pos[i] != i || sub.post.pack_pos[i] != i)
      return false;
  const int ow = p.passes[0].out_size, oh = p.passes[1].out_size;
  if (sub.pack.width != ow || sub.pack.height != oh || sub.front.width != ow || sub.front.height != oh)
    return false;
  memset ((void *) dp, 0, sizeof (*dp));
  dp->f = p.front;
  dp->out_w = ow, dp->out_h = oh;
  dp->sh.kind = p.passes[0].kind, dp->sh.n_taps = p.passes[0].n_taps, dp->sh.inc = p.passes[0].inc;
  dp->sv.kind = p.passes[1].kind, dp->sv.n_taps = p.passes[1].n_taps, dp->sv.inc = p.passes[1].inc;
  const bool hx2 = deep_passes_halve (p);
  dp->hx2 = hx2;
  dp->mx = g.prim;
  /* whole blocks only (a per-pixel form of the same chain was tried as the path of every other shape: as a whole frame's path it loses to the
     multi-launch composite - P010 4K -> NV12 1080p 4-tap 547 us against 85) */
  return hx2 && (ow % 4) == 0 && (sub.pack.kind == UNPACK_PLANAR || sub.pack.kind == UNPACK_SEMI) && !sub.pack.tail_swap;
}

}  // namespace gstamd
