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
//   at (x, y):  any pixel, any filter kinds (nearest / 2-tap / n-tap), edge clamps included - picture edges, odd shapes, every pack kind.
//   row4n ():   four neighbouring pixels (+ the chroma of the pixels left and right of them) of a whole block when both passes are
//               2-tap and the horizontal one reads source pixels 2 x, 2 x + 1 for every output x (an exact 2:1: `hx2`): the eight lumas of a
//               source line in one 16-byte load, the chroma samples of a chroma row once for all of them.
#pragma once
#include "video_deep.h"
#include "video_pack.h"

namespace gstamd {

struct DeepPackParams {
  FrontParams f;        // the 10-bit source (crop size)
  Planes pl;            // its planes at the crop origin
  const int *vpair;
  ScaleDev sh, sv;      // horizontal pass (runs first), vertical pass
  int out_w, out_h;     // size after both passes
  int hx2;              // both passes 2-tap, sh.offset[x] == 2 x for every x, f.width >= 2 out_w
  int vec;              // source rows allow the wide loads of row4n (planes and pitches on 16 bytes)
};

// one component through the horizontal 2-tap (video_orc_resample_h_2tap_u16) and the vertical one (video_scale_v_2tap_u16): deep_scale_px's lines
// (mul24s: samples have 16 bits, taps are int16 / an unsigned 16-bit parameter, differences 17 bits - the low 32 bits of the 24-bit product are the 32-bit
// product's, and v_mul_i32_i24 runs at full rate where v_mul_lo_u32 takes four passes)
GSTAMD_HD int deep_h2tap (int s1, int s2, int t0, int t1)
{
  return clampi ((int) ((uint32_t) mul24s (s1, t0) + (uint32_t) mul24s (s2, t1) + 4096u) >> 12, 0, 65535);
}
GSTAMD_HD int deep_v2tap (int s1, int s2, uint32_t p1)
{
  return clampi (s1 + ((int) ((uint32_t) mul24s (s2 - s1, (int) p1) + 4096u) >> 12), 0, 65535);
}

// what the block form keeps of a source line after the horizontal pass: luma of the block's four pixels, the two chroma components (in the source's
// storage order: c1 the first sample of a semi-planar pair / plane 1) of the pixels x0 - 1 .. x0 + 4
struct DeepHLine {
  int y[4], c1[6], c2[6];
};
// a chroma row through the horizontal upsampler at source pixels 2 x0 - 2 .. 2 x0 + 9
struct DeepCRow {
  int c1[12], c2[12];
};
template <int SEMI, int CH>
struct DeepScaledSrc {
  DeepPackParams d;

  // output pixel x of the horizontal pass over source line L (front_hscale16_lane)
  GSTAMD_HD uint2 hpx (const FrontRow &fr, int x, int L) const
  {
    const int w = d.f.width, off = (int) d.sh.offset[x];
    auto at1 = [&](int sx) { return deep_front1_t<SEMI, CH> (d.f, d.pl, fr, sx < 0 ? 0 : (sx > w - 1 ? w - 1 : sx), L); };
    if (d.sh.kind == SCALE_NEAREST)
      return at1 (off);
    const int16_t *t = d.sh.taps + (size_t) x * d.sh.n_taps;
    int v[4];
    if (d.sh.kind == SCALE_2TAP) {
      const uint2 a = at1 (off), b = at1 (off + 1);
      for (int c = 0; c < 4; c++)
        v[c] = deep_h2tap (deep_comp (a, c), deep_comp (b, c), (int) t[0], (int) t[1]);
    } else {
      uint32_t acc[4] = {0, 0, 0, 0};
      for (int l = 0; l < d.sh.n_taps; l++) {
        const uint2 p = at1 (off + l);
        const uint32_t tp = (uint32_t) (int) t[l];
        for (int c = 0; c < 4; c++)
          acc[c] += (uint32_t) deep_comp (p, c) * tp;
      }
      for (int c = 0; c < 4; c++)
        v[c] = deep_scaletaps ((int) acc[c]);
    }
    return deep_pack4 (v);
  }

  GSTAMD_HD uint2 hline_px (int x, int L) const
  {
    L = L < 0 ? 0 : (L > d.f.height - 1 ? d.f.height - 1 : L);          /* deep_img_at's clamp */
    return hpx (deep_front_row (d.f, d.vpair, L), x, L);
  }

  // pixel (x, y) of the scaled picture, narrowed: the A, Y, U, V bytes the sub-conversion's packer reads (deep_scale_px's vertical arithmetic, then
  // video_orc_convert_u16_to_u8)
  GSTAMD_HD uint32_t at (int x, int y) const
  {
    const int off = (int) d.sv.offset[y];
    int v[4];
    if (d.sv.kind == SCALE_NEAREST) {
      const uint2 a = hline_px (x, off);
      for (int c = 0; c < 4; c++)
        v[c] = deep_comp (a, c);
    } else if (d.sv.kind == SCALE_2TAP) {
      const int16_t *t = d.sv.taps + (size_t) y * d.sv.n_taps;
      const uint2 a = hline_px (x, off), b = hline_px (x, off + 1);
      for (int c = 0; c < 4; c++)
        v[c] = deep_v2tap (deep_comp (a, c), deep_comp (b, c), (uint32_t) (uint16_t) t[1]);
    } else {
      const int16_t *t = d.sv.taps + (size_t) y * d.sv.n_taps;
      uint32_t acc[4] = {0, 0, 0, 0};
      for (int l = 0; l < d.sv.n_taps; l++) {
        const uint2 p = hline_px (x, off + l);
        const uint32_t tp = (uint32_t) (int) t[l];
        for (int c = 0; c < 4; c++)
          acc[c] += (uint32_t) deep_comp (p, c) * tp;
      }
      for (int c = 0; c < 4; c++)
        v[c] = deep_scaletaps ((int) acc[c]);
    }
    return (uint32_t) (v[0] >> 8) | ((uint32_t) (v[1] >> 8) << 8) | ((uint32_t) (v[2] >> 8) << 16) | ((uint32_t) (v[3] >> 8) << 24);
  }

  // ---- the block form -------------------------------------------------------------------------------------------------------------------------
  // every block of four pixels inside the picture (pack_planar_block4 asks for x0 + 4 <= width): source pixels 2 x0 - 2 .. 2 x0 + 9 under the six
  // positions x0 - 1 .. x0 + 4, chroma samples x0 - 2 .. x0 + 5; at the picture's edges the positions outside it are computed on clamped addresses
  // and replaced by the edge pixel's values (pack_planar_block4's xm / xp), the upsampler's edge rules are selects
  GSTAMD_HD bool ok4 (int, int y) const { return d.hx2 && d.vec && y < d.out_h; }

  // chroma samples x0 - 2 .. x0 + 5 of chroma row crow (indices clamped into the row), widened, through the horizontal upsampler at the twelve source
  // pixels (sample index 1 + (j >> 1) for pixel j): deep_front1_t's cu / cv
  GSTAMD_HD void crow12 (const Widen &wd, int cw, int crow, int x0, bool edges, DeepCRow *o) const
  {
    const int km2 = x0 >= 2 ? x0 - 2 : 0, km1 = x0 >= 1 ? x0 - 1 : 0, kp4 = x0 + 4 < cw ? x0 + 4 : cw - 1, kp5 = x0 + 5 < cw ? x0 + 5 : cw - 1;
    int s1[8], s2[8];
    if (SEMI) {
      const uint8_t *row = d.pl.p[1] + (ptrdiff_t) crow * d.pl.stride[1];
      const uint32_t *q = (const uint32_t *) row;
      const uint4 m = *(const uint4 *) (row + 4 * (ptrdiff_t) x0);
      const uint32_t t[8] = {q[km2], q[km1], m.x, m.y, m.z, m.w, q[kp4], q[kp5]};
#pragma unroll
      for (int i = 0; i < 8; i++)
        s1[i] = deep_widen_w (wd, (int) (t[i] & 0xffffu)), s2[i] = deep_widen_w (wd, (int) (t[i] >> 16));
    } else {
      const uint8_t *ra = d.pl.p[1] + (ptrdiff_t) crow * d.pl.stride[1], *rb = d.pl.p[2] + (ptrdiff_t) crow * d.pl.stride[2];
      const uint16_t *qa = (const uint16_t *) ra, *qb = (const uint16_t *) rb;
      const uint2 ma = *(const uint2 *) (ra + 2 * (ptrdiff_t) x0), mb = *(const uint2 *) (rb + 2 * (ptrdiff_t) x0);
      const int ta[8] = {qa[km2], qa[km1], (int) (ma.x & 0xffffu), (int) (ma.x >> 16), (int) (ma.y & 0xffffu), (int) (ma.y >> 16), qa[kp4], qa[kp5]};
      const int tb[8] = {qb[km2], qb[km1], (int) (mb.x & 0xffffu), (int) (mb.x >> 16), (int) (mb.y & 0xffffu), (int) (mb.y >> 16), qb[kp4], qb[kp5]};
#pragma unroll
      for (int i = 0; i < 8; i++)
        s1[i] = deep_widen_w (wd, ta[i]), s2[i] = deep_widen_w (wd, tb[i]);
    }
    const int w = d.f.width;
#pragma unroll
    for (int j = 0; j < 12; j++) {
      if (!edges && (j < 2 || j >= 10))
        continue;
      const int k = 1 + (j >> 1), odd = j & 1, sx = 2 * (x0 - 1) + j;
      const bool in = odd ? sx < w - 1 : sx >= 2;
      int a = s1[k], b = s2[k];
      if (CH == CHROMA_H_H2_CS) {
        if (odd && in)
          a = (a + s1[k + 1] + 1) >> 1, b = (b + s2[k + 1] + 1) >> 1;
      } else if (CH == CHROMA_H_H2) {
        const int n = odd ? k + 1 : k - 1;
        if (in)
          a = (3 * a + s1[n] + 2) >> 2, b = (3 * b + s2[n] + 2) >> 2;
      }
      o->c1[j] = a, o->c2[j] = b;
    }
  }

  // source line L (clamped by the caller) at the block's positions: luma through the horizontal pass; the line's two upsampled chroma rows blended by
  // its weights, then the pass (deep_front1_t + front_hscale16_lane, value for value)
  GSTAMD_HD void hline4 (const FrontRow &fr, const DeepCRow &ca, const DeepCRow &cb, int x0, int L, bool edges, DeepHLine *o) const
  {
    const uint4 q = *(const uint4 *) (d.pl.p[0] + (ptrdiff_t) L * d.pl.stride[0] + 4 * (ptrdiff_t) x0);
    const uint4 tq = *(const uint4 *) (d.sh.taps + 2 * (size_t) x0);
    const uint32_t lw[4] = {q.x, q.y, q.z, q.w}, tw[4] = {tq.x, tq.y, tq.z, tq.w};
#pragma unroll
    for (int i = 0; i < 4; i++)
      o->y[i] = deep_h2tap (deep_widen_w (fr.wd, (int) (lw[i] & 0xffffu)), deep_widen_w (fr.wd, (int) (lw[i] >> 16)), (int) (int16_t) (tw[i] & 0xffffu),
          (int) (int16_t) (tw[i] >> 16));
    const int xm = x0 > 0 ? x0 - 1 : 0, xp = x0 + 4 < d.out_w ? x0 + 4 : d.out_w - 1;
    uint32_t tm = 0, tp = 0;
    if (edges)
      tm = *(const uint32_t *) (d.sh.taps + 2 * (size_t) xm), tp = *(const uint32_t *) (d.sh.taps + 2 * (size_t) xp);
#pragma unroll
    for (int p = 0; p < 6; p++) {
      if (!edges && (p == 0 || p == 5))
        continue;
      int f1[2], f2[2];         /* even and odd source pixel of position p */
#pragma unroll
      for (int odd = 0; odd < 2; odd++) {
        f1[odd] = (mul24s (fr.wa, ca.c1[2 * p + odd]) + mul24s (fr.wb, cb.c1[2 * p + odd]) + 4) >> 3;
        f2[odd] = (mul24s (fr.wa, ca.c2[2 * p + odd]) + mul24s (fr.wb, cb.c2[2 * p + odd]) + 4) >> 3;
      }
      const uint32_t t = p == 0 ? tm : (p == 5 ? tp : tw[p == 0 || p == 5 ? 0 : p - 1]);
      const int t0 = (int) (int16_t) (t & 0xffffu), t1 = (int) (int16_t) (t >> 16);
      o->c1[p] = deep_h2tap (f1[0], f1[1], t0, t1);
      o->c2[p] = deep_h2tap (f2[0], f2[1], t0, t1);
    }
  }

  // the block's pixels of destination line y: two source lines, whose four chroma rows are three in a picture that halves (rows y - 1, y | y, y + 1: the
  // second line's first row is the first line's second - it goes through the horizontal upsampler once; the keys are wave-uniform)
  GSTAMD_HD uint4 row4n (int x0, int y, bool edges, int, int, uint32_t &em, uint32_t &ep) const
  {
    const int off = (int) d.sv.offset[y];
    const uint32_t p1 = (uint32_t) (uint16_t) d.sv.taps[2 * (size_t) y + 1];
    const int h = d.f.height;
    const int La = off < 0 ? 0 : (off > h - 1 ? h - 1 : off), Lb = off + 1 > h - 1 ? h - 1 : off + 1;          /* deep_img_at's clamp */
    const FrontRow fa = deep_front_row (d.f, d.vpair, La), fb = deep_front_row (d.f, d.vpair, Lb);
    DeepHLine a, b;
    {
      DeepCRow c0, c1;
      crow12 (fa.wd, fa.cw, fa.ra, x0, edges, &c0);
      crow12 (fa.wd, fa.cw, fa.rb, x0, edges, &c1);
      hline4 (fa, c0, c1, x0, La, edges, &a);
      if (fb.ra != fa.rb)
        crow12 (fb.wd, fb.cw, fb.ra, x0, edges, &c1);
      crow12 (fb.wd, fb.cw, fb.rb, x0, edges, &c0);
      hline4 (fb, c1, c0, x0, Lb, edges, &b);
    }
    const bool usw = SEMI ? d.f.u_plane != 0 : d.f.u_plane == 1;          /* c1 is U */
    uint32_t px[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t v1 = (uint32_t) (deep_v2tap (a.c1[i + 1], b.c1[i + 1], p1) >> 8), v2 = (uint32_t) (deep_v2tap (a.c2[i + 1], b.c2[i + 1], p1) >> 8);
      px[i] = 0xffu | ((uint32_t) (deep_v2tap (a.y[i], b.y[i], p1) >> 8) << 8) | ((usw ? v1 : v2) << 16) | ((usw ? v2 : v1) << 24);
    }
    if (edges) {                /* the pixels left and right of the block; at the picture's edge the block's own first / last pixel (xm, xp) */
      uint32_t e[2];
#pragma unroll
      for (int s = 0; s < 2; s++) {
        const int p = s ? 5 : 0;
        const uint32_t v1 = (uint32_t) (deep_v2tap (a.c1[p], b.c1[p], p1) >> 8), v2 = (uint32_t) (deep_v2tap (a.c2[p], b.c2[p], p1) >> 8);
        e[s] = 0xffu | ((usw ? v1 : v2) << 16) | ((usw ? v2 : v1) << 24);
      }
      em = x0 > 0 ? e[0] : px[0];
      ep = x0 + 4 < d.out_w ? e[1] : px[3];
    }
    uint4 r;
    r.x = px[0], r.y = px[1], r.z = px[2], r.w = px[3];
    return r;
  }
};

// one lane of k_deep_scale_pack: the 4 x (1 << h_sub) block at (x0, yb << h_sub)
template <int SEMI, int CH>
GSTAMD_HD void deep_scale_pack_lane (const PackPlanarParams &pk, const DeepPackParams &dp, const DstPlanes &dst, int wide, int x0, int yb, long long dd = 0)
{
  DeepScaledSrc<SEMI, CH> src;
  src.d = dp;
  if (wide && pack_planar_block4 (pk, src, dst, x0, yb, dd))
    return;
  pack_planar_body (pk, src, dst, x0, yb, dd);
}

GSTAMD_HD void deep_scale_pack_any (int variant, const PackPlanarParams &pk, const DeepPackParams &dp, const DstPlanes &dst, int wide, int x0, int yb)
{
  switch (variant) {
    case 0: deep_scale_pack_lane<0, CHROMA_H_NONE> (pk, dp, dst, wide, x0, yb); break;
    case 1: deep_scale_pack_lane<0, CHROMA_H_H2> (pk, dp, dst, wide, x0, yb); break;
    case 2: deep_scale_pack_lane<0, CHROMA_H_H2_CS> (pk, dp, dst, wide, x0, yb); break;
    case 3: deep_scale_pack_lane<1, CHROMA_H_NONE> (pk, dp, dst, wide, x0, yb); break;
    case 4: deep_scale_pack_lane<1, CHROMA_H_H2> (pk, dp, dst, wide, x0, yb); break;
    default: deep_scale_pack_lane<1, CHROMA_H_H2_CS> (pk, dp, dst, wide, x0, yb); break;
  }
}

// host: is `p` (a composite plan with a 10-bit source, GammaPlan::src16) with the sub-conversion `sub` (8-bit unpack-format image -> destination) one
// this kernel serves?  Fills everything of `dp` but the pointers.
inline bool deep_scale_pack_plan_ok (const VideoPlan &p, const VideoPlan &sub, DeepPackParams *dp)
{
  const GammaPlan &g = p.gamma;
  if (!g.on || !g.src16 || g.planes_fast || g.fused || g.pack16 || g.store64 || !g.shrink || !g.dec16.empty () || !g.enc16.empty () || g.prim.has_matrix ||
      g.alpha_kind != ALPHA_NONE || g.to_yuv.kind != MATRIX_NONE || p.interlaced || p.field)
    return false;
  if (p.passes.size () != 2 || !p.passes[0].horizontal || p.passes[1].horizontal || deep_front4_variant (p.front) < 0)
    return false;
  for (int i = 0; i < 2; i++)
    if (p.passes[i].kind != SCALE_NEAREST && p.passes[i].kind != SCALE_2TAP && p.passes[i].kind != SCALE_NTAP)
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
  bool hx2 = p.passes[0].kind == SCALE_2TAP && p.passes[1].kind == SCALE_2TAP && p.front.width >= 2 * ow && ow >= 8;
  for (int x = 0; hx2 && x < ow; x++)
    hx2 = (int) p.passes[0].offset[x] == 2 * x;
  dp->hx2 = hx2;
  /* served by the block form only: the per-pixel form (at ()) is there for correctness at odd shapes, as a whole frame's path it loses to the
     multi-launch composite (P010 4K -> NV12 1080p 4-tap: 547 us against 85) */
  return hx2 && (ow % 4) == 0 && (sub.pack.kind == UNPACK_PLANAR || sub.pack.kind == UNPACK_SEMI) && !sub.pack.tail_swap;
}

}  // namespace gstamd
