// video_bilinear_fast.h - semi-planar 4:2:0 (NV12 / NV21) -> bilinear 2-tap x 2-tap scale (horizontal first) -> AYUV_ARGB
// matrix -> 4-byte RGB, one kernel, no full-resolution AYUV staging (BASELINE config 5: 8K NV12 -> 4K BGRA).
//
// Same integers as the generic chain: unpack_NV12 + chroma upsample (video-chroma.c:277-327, 687-699) of the two source
// lines a vertical 2-tap needs, ldreslinl per line (video-orc-dist.c:26162: (a * (256 - f) + b * f) >> 8 on pixels
// idx = (x * inc) >> 16 and idx + 1), video_orc_resample_v_2tap_u8_lq across the two lines, video_orc_convert_AYUV_ARGB,
// pack.  What changes is the amount of work: the chain evaluates the front for every source pixel under a tile and
// parks it in LDS as AYUV words; here the wave stages the RAW rows once (16-byte loads), luma as {line0, line1} pairs
// and each chroma row as packed {U, V} samples, and every output pixel picks the 2 x 2 source pixels it needs and
// evaluates their chroma directly from <= 3 samples per row, all in packed 16-bit lanes.
#pragma once
#include "video_scale_fast.h"

namespace gstamd {

#define BIL_MAX_SPAN 1024                 // source pixels under one tile: one 16-byte piece per lane and row
// LDS of one wave (dynamic, sized by the launcher from the plan's largest tile span - occupancy is LDS-bound):
//   y[ylen]      {Y of source line 0 | Y of source line 1 << 16} per source pixel
//   c[4][clen]   rows (line0 a, line0 b, line1 a, line1 b): {U | V << 16} per chroma sample
struct BilLds {
  uint32_t *y;
  uint32_t *c[4];
};
#ifdef __HIPCC__
#define GSTAMD_HOSTDEV __host__ __device__ inline
#else
#define GSTAMD_HOSTDEV inline
#endif
GSTAMD_HOSTDEV int bil_clen (int ylen) { return ylen / 2 + 24; }
GSTAMD_HD BilLds bil_lds (uint32_t *base, int ylen)
{
  BilLds l;
  l.y = base;
  for (int s = 0; s < 4; s++)
    l.c[s] = base + ylen + s * bil_clen (ylen);
  return l;
}
GSTAMD_HOSTDEV size_t bil_lds_words (int ylen) { return (size_t) ylen + 4 * (size_t) bil_clen (ylen); }

// the frames of one launch of k_bilinear420_rows (same VideoInfo, hence same pitches): plane pointers per frame
#define GSTAMD_BIL_MAX_BATCH 32
struct BilBatch {
  const uint8_t *p[3][GSTAMD_BIL_MAX_BATCH];
  uint8_t *dst[GSTAMD_BIL_MAX_BATCH];
  int stride[3];
};

struct BilParams {
  FastParams fp;        // width / height = SOURCE size; matrix, layout
  int out_w, out_h;
  int inc;              // ldreslinl increment of the horizontal pass
  int tile_w;           // outputs per wave (multiple of 64, <= 256)
  int ylen;             // LDS words of the luma row (largest 16-aligned tile span + 16)
  const uint32_t *voffset;   // [out_h] first source line of the vertical 2-tap
  const int16_t *vtaps;      // [out_h][2]
  const int *vpair;          // chroma line pairing of the source (planner), or NULL: rows y >> 1
  int regular_pairs;         // the pairing is the closed form of bil_rows: no table reads
  int planar;                // 1: I420 / YV12 - U in plane u_plane, V in plane v_plane (width % 16 == 0, 8-byte aligned chroma rows); 0: interleaved plane 1
  int u_plane, v_plane;
  int rows_tile_w;           // outputs per wave of k_bilinear420_rows (<= 384)
  int rows;                  // > 0: k_bilinear420_rows (video_bilinear_rows.h) with this many output rows per wave; < 0: as many
                             // waves as the chip holds at once; 0: k_bilinear420
  int strips;                // row strips per tile column of k_bilinear420_rows: strip g covers rows [g h / strips, (g + 1) h / strips)
  int half;                  // the picture shrinks by exactly two in both directions: k_bilinear420_half (video_bilinear_half.h)
};

GSTAMD_HOSTDEV void bil_rows (const BilParams &bp, int line, int *ra, int *rb, int *role)
{
  if (bp.regular_pairs) {
    // pairs (2p - 1, 2p) over chroma rows (p - 1, p), clamped into the plane: what do_upsample_lines produces when every
    // source line is consumed in order (the planner checks its simulated table against this form)
    const int p = (line + 1) >> 1;
    *ra = p - 1 > bp.fp.crow_lo ? p - 1 : bp.fp.crow_lo;
    *rb = p < bp.fp.crow_hi ? p : bp.fp.crow_hi;
    *role = (line & 1) ? 0 : 1;
    if (*ra == *rb)
      *role = 0;
    return;
  }
  if (bp.vpair) {
    const int e0 = bp.vpair[2 * line];
    *ra = vpair_row (e0);
    *role = vpair_role (e0);
    *rb = bp.vpair[2 * line + 1];
  } else {
    *ra = *rb = line >> 1;
    *role = 0;
  }
}

// source span of the outputs [t0, t1): pixels [x_lo, x_hi), chroma samples [k_lo, k_hi] (clamped into the row)
GSTAMD_HD void bil_span (const BilParams &bp, int t0, int t1, int *x_lo, int *x_hi, int *k_lo, int *k_hi)
{
  const int cw = (bp.fp.width + 1) >> 1;
  *x_lo = (t0 * bp.inc) >> 16;
  *x_hi = (((t1 - 1) * bp.inc) >> 16) + 2;
  const int kl = (*x_lo >> 1) - 1, kh = ((*x_hi - 1) >> 1) + 1;
  *k_lo = kl < 0 ? 0 : kl;
  *k_hi = kh > cw - 1 ? cw - 1 : kh;
}

// One lane's share of the raw rows of source lines r0, r0 + 1: fetched with 16-byte loads (all six issued before the first
// one is consumed), then rearranged into LDS.  A tile spans at most 1024 source pixels, so each lane owns at most one
// 16-byte piece of every row.  (Walking several output rows per wave with the next row's loads in flight was tried:
// no gain, the register cost lowers occupancy.)
struct BilRegs {
  uint32_t a[4], b[4];       // 16 luma bytes of line r0 / r0 + 1
  uint32_t m[4][4];          // 8 chroma samples of each of the four chroma rows
};

GSTAMD_HD void bil_fetch (const BilParams &bp, const Planes &pl, int t0, int t1, int r0, int lane, bool vec, BilRegs &r)
{
  const int w = bp.fp.width, cw = (w + 1) >> 1;
  int x_lo, x_hi, k_lo, k_hi;
  bil_span (bp, t0, t1, &x_lo, &x_hi, &k_lo, &k_hi);
  const int x = (x_lo & ~15) + 16 * lane, k = (k_lo & ~7) + 8 * lane;
  const uint8_t *y0 = pl.p[0] + (size_t) r0 * pl.stride[0], *y1 = y0 + pl.stride[0];
  if (vec && (w & 15) == 0) {
    // every piece of every row is whole (width % 16 == 0): all six loads in ONE straight line, lanes outside the span reading a
    // valid piece they will not commit.  With the per-piece edge fallbacks in the same control flow the compiler has to assume
    // their byte loads may still be writing these registers and drains the queue before every wide load - five serial round
    // trips per tile instead of one.
    const int xc = x < x_hi ? x : (x_lo & ~15), kc = k <= k_hi ? k : (k_lo & ~7);
    int crow[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
      int ra, rb, role;
      bil_rows (bp, r0 + (s >> 1), &ra, &rb, &role);
      crow[s] = (s & 1) ? rb : ra;
    }
    wide_load16<true> (y0 + xc, 4, true, r.a);
    wide_load16<true> (y1 + xc, 4, true, r.b);
    if (bp.planar) {
      // 8 U bytes in m[s][0..1], 8 V bytes in m[s][2..3]
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const uint2 mu = *(const uint2 *) (pl.p[bp.u_plane] + (ptrdiff_t) crow[s] * pl.stride[bp.u_plane] + kc);
        const uint2 mv = *(const uint2 *) (pl.p[bp.v_plane] + (ptrdiff_t) crow[s] * pl.stride[bp.v_plane] + kc);
        r.m[s][0] = mu.x, r.m[s][1] = mu.y, r.m[s][2] = mv.x, r.m[s][3] = mv.y;
      }
    } else {
#pragma unroll
      for (int s = 0; s < 4; s++)
        wide_load16<false> (pl.p[1] + (ptrdiff_t) crow[s] * pl.stride[1] + 2 * kc, 4, true, r.m[s]);
    }
    return;
  }
  if (x < x_hi) {
    if (((w - x) >> 2) >= 4 && vec) {
      wide_load16<true> (y0 + x, 4, true, r.a);
      wide_load16<true> (y1 + x, 4, true, r.b);
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        r.a[j] = r.b[j] = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int xx = x + 4 * j + q;
          if (xx < w) {
            r.a[j] |= (uint32_t) y0[xx] << (8 * q);
            r.b[j] |= (uint32_t) y1[xx] << (8 * q);
          }
        }
      }
    }
  }
  if (k <= k_hi) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
      int ra, rb, role;
      bil_rows (bp, r0 + (s >> 1), &ra, &rb, &role);
      const uint8_t *row = pl.p[1] + (ptrdiff_t) ((s & 1) ? rb : ra) * pl.stride[1];
      if (k + 8 <= cw && vec) {
        wide_load16<false> (row + 2 * k, 4, true, r.m[s]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          r.m[s][j] = 0;
#pragma unroll
          for (int q = 0; q < 2; q++) {
            const int kk = k + 2 * j + q;
            if (kk < cw)
              r.m[s][j] |= (uint32_t) * (const uint16_t *) (row + 2 * kk) << (16 * q);
          }
        }
      }
    }
  }
}

GSTAMD_HD void bil_commit (const BilParams &bp, int t0, int t1, int lane, const BilRegs &r, const BilLds *lds)
{
  int x_lo, x_hi, k_lo, k_hi;
  bil_span (bp, t0, t1, &x_lo, &x_hi, &k_lo, &k_hi);
  const int xa = x_lo & ~15, ka = k_lo & ~7;
  const int x = xa + 16 * lane, k = ka + 8 * lane;
  if (x < x_hi) {
    uint32_t *d = &lds->y[x - xa];
#pragma unroll
    for (int j = 0; j < 4; j++)
      *(uint4 *) (d + 4 * j) = gstamd_make_uint4 (bperm (r.b[j], r.a[j], 0x0c040c00u), bperm (r.b[j], r.a[j], 0x0c050c01u),
          bperm (r.b[j], r.a[j], 0x0c060c02u), bperm (r.b[j], r.a[j], 0x0c070c03u));
  }
  if (k <= k_hi) {
    // byte pair (b0, b1) of a sample -> U | V << 16; NV12 (u_first): U first
    const uint32_t sel_lo = bp.fp.u_first ? 0x0c010c00u : 0x0c000c01u, sel_hi = bp.fp.u_first ? 0x0c030c02u : 0x0c020c03u;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const uint32_t *m = r.m[s];
      uint32_t *d = &lds->c[s][k - ka];
      if (bp.planar) {        // sample j = U byte j | V byte j << 16
        *(uint4 *) d = gstamd_make_uint4 (bperm (m[2], m[0], 0x0c040c00u), bperm (m[2], m[0], 0x0c050c01u), bperm (m[2], m[0], 0x0c060c02u),
            bperm (m[2], m[0], 0x0c070c03u));
        *(uint4 *) (d + 4) = gstamd_make_uint4 (bperm (m[3], m[1], 0x0c040c00u), bperm (m[3], m[1], 0x0c050c01u), bperm (m[3], m[1], 0x0c060c02u),
            bperm (m[3], m[1], 0x0c070c03u));
        continue;
      }
      *(uint4 *) d = gstamd_make_uint4 (bperm (0, m[0], sel_lo), bperm (0, m[0], sel_hi), bperm (0, m[1], sel_lo), bperm (0, m[1], sel_hi));
      *(uint4 *) (d + 4) = gstamd_make_uint4 (bperm (0, m[2], sel_lo), bperm (0, m[2], sel_hi), bperm (0, m[3], sel_lo), bperm (0, m[3], sel_hi));
    }
  }
}

// upsampled chroma {U, V} of source pixels px and px + 1 of one chroma row (samples in LDS from ka on)
template <int CH>
GSTAMD_HD void bil_chroma_pair (const uint32_t *row, int ka, int cw, int px, uint32_t *c0, uint32_t *c1)
{
  const int k = px >> 1;
  const uint32_t s0 = row[k - ka];
  if (CH == CHROMA_H_NONE) {
    *c0 = s0;
    *c1 = (px & 1) ? row[(k + 1 < cw ? k + 1 : cw - 1) - ka] : s0;
    return;
  }
  const uint32_t s1 = row[(k + 1 < cw ? k + 1 : cw - 1) - ka];
  if (CH == CHROMA_H_H2_CS) {
    // even pixel: the sample; odd pixel: (a + b + 1) >> 1 with the next sample (the last sample pairs with itself)
    const uint32_t avg = pk_shr<1> (s0 + s1 + 0x00010001u);
    *c0 = (px & 1) ? avg : s0;
    *c1 = (px & 1) ? s1 : avg;
  } else {
    // even pixel: (prev + 3 cur + 2) >> 2, odd pixel: (3 cur + next + 2) >> 2, clamped neighbours
    const uint32_t sm = row[(k > 0 ? k - 1 : 0) - ka];
    const uint32_t even0 = pk_shr<2> (sm + 3u * s0 + 0x00020002u), odd0 = pk_shr<2> (3u * s0 + s1 + 0x00020002u);
    const uint32_t even1 = pk_shr<2> (s0 + 3u * s1 + 0x00020002u);
    *c0 = (px & 1) ? odd0 : even0;
    *c1 = (px & 1) ? even1 : odd0;
  }
}

GSTAMD_HD void store_px_raw (uint8_t *__restrict__ dst, int dstride, int x, int y, uint32_t v)
{
  uint32_t *p = (uint32_t *) (dst + (size_t) y * dstride + 4 * (size_t) x);
#ifdef __HIPCC__
  __builtin_nontemporal_store (v, p);
#else
  *p = v;
#endif
}

// host side: LDS words of the luma row for tiles of tile_w outputs (largest span from the 16-aligned start, rounded up
// to 16, plus 16), or 0 when a tile does not fit BIL_MAX_SPAN
inline int bil_ylen (int out_w, int inc, int tile_w)
{
  int worst = 0;
  for (int t0 = 0; t0 < out_w; t0 += tile_w) {
    const int t1 = t0 + tile_w < out_w ? t0 + tile_w : out_w;
    const int x_lo = (t0 * inc) >> 16, x_hi = (((t1 - 1) * inc) >> 16) + 2;
    const int span = (x_hi - (x_lo & ~15) + 15) & ~15;
    worst = span > worst ? span : worst;
  }
  return worst <= BIL_MAX_SPAN ? worst + 16 : 0;
}

// outputs per wave: the widest tile whose source span fits (MI355X, C5: 256 -> 37.6 us, 192 -> 39.1, 128 -> 42.2; the
// staging work per output shrinks with the tile width faster than the LDS footprint costs occupancy)
inline int bil_pick_tile (int out_w, int inc, int *ylen)
{
  for (int tw = 256; tw >= 64; tw -= 64) {
    const int yl = bil_ylen (out_w, inc, tw);
    if (yl > 0) {
      *ylen = yl;
      return tw;
    }
  }
  return 0;
}

// one pixel through video_orc_convert_AYUV_ARGB + pack (word form of video_fast.h): y8 = luma, uv = {U, V} in u16 lanes
template <int L>
GSTAMD_HD uint32_t fast_pixel1_l (const FastParams &fp, uint32_t y8, uint32_t uv, uint32_t (&q)[2])
{
  if (L & GSTAMD_LAYOUT_AYUV)
    return GSTAMD_AYUV_OUT (fp, 0xffu | ((y8 & 0xffu) << 8) | ((uv & 0xffu) << 16) | ((uv & 0x00ff0000u) << 8));
  constexpr int PR = L & 3, PG = (L >> 2) & 3, PB = (L >> 4) & 3;
  const uint32_t ys = (y8 ^ 0x80u) * 0x0101u;                         // word 0 = splatbw (y - 128)
  const uint32_t cx = uv ^ 0x00800080u, cs = bperm (cx, cx, 0x02020000u);      // words [t(U) | t(V)]
  const int wy = mul_word<0> (ys, fp.pc[0]) + 0x00800000;
  const int pgu = mul_word<0> (cs, fp.pc[3]), prv = mul_word<1> (cs, fp.pc[1]);
  const int pgv = mul_word<1> (cs, fp.pc[4]), pbu = mul_word<0> (cs, fp.pc[2]);
  const int g0 = add_hiwords (wy, pgu);
  add_hiwords_into<PR & 1> (q[PR >> 1], wy, prv);
  add_hiword_into<PG & 1> (q[PG >> 1], g0, pgv);
  add_hiwords_into<PB & 1> (q[PB >> 1], wy, pbu);
  uint32_t o = sat_pk_u8 (q[0]);
  sat_pk_u8_hi (o, q[1]);
  return o;
}

// phase 2: lane `lane` produces outputs t0 + lane + 64 i (i < 4) of output row y
template <int CH, int L>
GSTAMD_HD void bil_emit (const BilParams &bp, uint8_t *__restrict__ dst, int dstride, int t0, int t1, int y, int r0, int lane, const BilLds *lds)
{
  const int cw = (bp.fp.width + 1) >> 1;
  int x_lo, x_hi, k_lo, k_hi;
  bil_span (bp, t0, t1, &x_lo, &x_hi, &k_lo, &k_hi);
  const int xa = x_lo & ~15, ka = k_lo & ~7;
  int ra, rb, role0, role1;
  bil_rows (bp, r0, &ra, &rb, &role0);
  bil_rows (bp, r0 + 1, &ra, &rb, &role1);
  // vertical chroma blend weights of the two source lines: role 0 (3, 1), role 1 (1, 3)
  const uint32_t wa0 = role0 == 0 ? 0x00030003u : 0x00010001u, wb0 = 0x00040004u - wa0;      /* both u16 lanes */
  const uint32_t wa1 = role1 == 0 ? 0x00030003u : 0x00010001u, wb1 = 0x00040004u - wa1;
  const uint32_t p1s = ((uint32_t) (uint16_t) bp.vtaps[(size_t) y * 2 + 1]) * 0x00010001u;
  uint32_t q[4][2];
  layout_init<L> (q);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int x = t0 + lane + 64 * i;
    if (x >= t1)
      break;
    const int tmp = (int) umul24 ((uint32_t) x, (uint32_t) bp.inc);
    const int idx = tmp >> 16;
    const uint32_t fr = (uint32_t) (tmp >> 8) & 0xffu, frs = fr * 0x00010001u, nfs = 0x01000100u - frs;
    // luma of both lines: ldreslinl on {line0, line1} lanes, then the vertical 2-tap between the lanes
    const uint32_t ya = lds->y[idx - xa], yb = lds->y[idx + 1 - xa];
    const uint32_t yh = pk_shr<8> (pk_mad16 (yb, frs, pk_mad16 (ya, nfs, 0u)));
    const uint32_t yv = v2tap_pk (yh & 0xffffu, yh >> 16, p1s) & 0xffu;
    // chroma of the 2 x 2 source pixels
    uint32_t a0, a1, b0, b1, c00, c01, c10, c11;
    bil_chroma_pair<CH> (lds->c[0], ka, cw, idx, &a0, &a1);
    bil_chroma_pair<CH> (lds->c[1], ka, cw, idx, &b0, &b1);
    c00 = pk_shr<2> (pk_mad16 (a0, wa0, pk_mad16 (b0, wb0, 0x00020002u)));
    c01 = pk_shr<2> (pk_mad16 (a1, wa0, pk_mad16 (b1, wb0, 0x00020002u)));
    bil_chroma_pair<CH> (lds->c[2], ka, cw, idx, &a0, &a1);
    bil_chroma_pair<CH> (lds->c[3], ka, cw, idx, &b0, &b1);
    c10 = pk_shr<2> (pk_mad16 (a0, wa1, pk_mad16 (b0, wb1, 0x00020002u)));
    c11 = pk_shr<2> (pk_mad16 (a1, wa1, pk_mad16 (b1, wb1, 0x00020002u)));
    const uint32_t ch0 = pk_shr<8> (pk_mad16 (c01, frs, pk_mad16 (c00, nfs, 0u)));
    const uint32_t ch1 = pk_shr<8> (pk_mad16 (c11, frs, pk_mad16 (c10, nfs, 0u)));
    const uint32_t cv = v2tap_pk (ch0, ch1, p1s);
    store_px_raw (dst, dstride, x, y, fast_pixel1_l<L> (bp.fp, yv, cv, q[i]));
  }
}

}  // namespace gstamd
