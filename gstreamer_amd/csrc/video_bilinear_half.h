// video_bilinear_half.h - k_bilinear420_half: the 4:2:0 -> bilinear 2-tap x 2-tap -> matrix -> 4-byte RGB path of
// video_bilinear_rows.h for pictures that shrink by EXACTLY two in both directions (BASELINE config 5: 8K NV12 -> 4K BGRA; the
// elements' 4K -> 1080p).
//
// Same integers as k_bilinear420_rows (unpack + chroma upsample video-chroma.c:277-327, 687-699; ldreslinl
// video-orc-dist.c:26162; video_orc_resample_v_2tap_u8_lq; video_orc_convert_AYUV_ARGB).  What the ratio gives away:
//   - ldreslinl's index (x * inc) >> 16 is 2 x for every output of the row (inc = 2 + 1 / (out - 1), the fraction sweeps 0 .. 255
//     across the row) and the vertical pass reads lines 2 y, 2 y + 1 (bilh_plan_ok checks both tables): the 16 source pixels a lane
//     fetches with one 16-byte load are exactly the 2 x 8 sources of EIGHT CONSECUTIVE outputs.  No LDS, no wave barrier, no
//     per-output address: the rows kernel parked six byte planes in LDS and read them back with six 16-bit loads per output
//     (7.5 LDS and 2.2 vector-memory instructions per output and lane; here 0 and 0.9);
//   - an output's two source pixels are an EVEN and the next ODD pixel, which is how h420_filter_raw2 hands the h-filtered chroma
//     row over (even-pixel bytes, odd-pixel bytes): the interleave of bilr_filter_px is not needed, the 3:1 row blend works on the
//     split form, and v_perm_b32 with compile-time selectors puts {even | odd << 16} of an output into a register;
//   - a pass on such a pair is ONE v_dot2_i32_i16 against {256 - f | f << 16}: a (256 - f) + b f is ldreslinl before its >> 8, and
//     s1 + (((s2 - s1) p1 + 128) >> 8) of the vertical 2-tap equals (s1 (256 - p1) + s2 p1 + 128) >> 8 for taps 0 .. 256 (the
//     16-bit wrap of the ORC form cancels: the sum stays below 2^16) - the ">> 8" of a pass is the byte the next perm picks.
//     Six instructions per component and output (rows kernel: 8.5 on luma, 17 on chroma).
// A wave walks a strip of output rows down a 1024-pixel source column and carries the filtered chroma rows y - 1, y, y + 1 with it:
// one new chroma row per output row, its loads and the next two luma rows' in flight while the current row is emitted.
#pragma once
#include "video_bilinear_rows.h"

namespace gstamd {

#define BILH_SRC_PER_LANE 16
#define BILH_OUT_PER_LANE 8
#define BILH_TILE_SRC (64 * BILH_SRC_PER_LANE)

// host, once per plan: is this the exact halving the kernel is written for?  voffset / vtaps: the vertical pass's tables
inline bool bilh_plan_ok (const BilParams &bp, const uint32_t *voffset, const int16_t *vtaps)
{
  if (!bp.regular_pairs || bp.fp.px_bytes != 4 || bp.fp.lut != nullptr)
    return false;
  if (bp.fp.width != 2 * bp.out_w || bp.fp.height != 2 * bp.out_h || (bp.fp.width % BILH_SRC_PER_LANE) != 0)
    return false;
  if (bp.fp.crow_lo != 0 || bp.fp.crow_hi != bp.out_h - 1)
    return false;
  for (int x = 0; x < bp.out_w; x++)
    if ((int) (((uint32_t) x * (uint32_t) bp.inc) >> 16) != 2 * x)
      return false;
  for (int y = 0; y < bp.out_h; y++) {
    const int p1 = vtaps[2 * (size_t) y + 1];
    if ((int) voffset[y] != 2 * y || p1 < 0 || p1 > 256 || !bilr_window_matches (bp, 2 * y))
      return false;
  }
  return true;
}

GSTAMD_HD int bilh_dot2 (uint32_t pair, uint32_t w, int acc)     // lo (pair) * lo (w) + hi (pair) * hi (w) + acc on signed 16-bit halves
{
#ifdef __HIPCC__
  typedef short s2 __attribute__ ((ext_vector_type (2)));
  /* clamp = true: the sums here are nowhere near 2^31, so the saturation never acts - but it keeps the instruction in its three-operand form
   * (v_dot2_i32_i16 with an inline-constant accumulator); without it the compiler picks v_dot2c, whose accumulator is the destination, and
   * spends a v_mov per dot product on loading it */
  return __builtin_amdgcn_sdot2 (__builtin_bit_cast (s2, pair), __builtin_bit_cast (s2, w), acc, true);
#else
  return (int) (int16_t) (pair & 0xffffu) * (int) (int16_t) (w & 0xffffu) + (int) (int16_t) (pair >> 16) * (int) (int16_t) (w >> 16) + acc;
#endif
}

// what depends on the output column only: {256 - f | f << 16} of the lane's eight outputs (f = ldreslinl's fraction)
struct BilhLane {
  uint32_t w[BILH_OUT_PER_LANE];
};

GSTAMD_HD void bilh_lane_setup (const BilParams &bp, int o0, BilhLane &c)
{
#pragma unroll
  for (int i = 0; i < BILH_OUT_PER_LANE; i++) {
    const uint32_t f = (umul24 ((uint32_t) (o0 + i), (uint32_t) bp.inc) >> 8) & 0xffu;
    c.w[i] = (256u - f) | (f << 16);
  }
}

// one component of one output through both passes: pa / pb = {source 2 x | source 2 x + 1 << 16} on lines 2 y / 2 y + 1.  Byte 1 of the
// result is the value XOR 0x80 (the form the colour matrix wants: 0x8000 rides on the rounding term), the other bytes are not zero.
GSTAMD_HD uint32_t bilh_value (uint32_t pa, uint32_t pb, uint32_t w, uint32_t vw)
{
  const uint32_t h0 = (uint32_t) bilh_dot2 (pa, w, 0), h1 = (uint32_t) bilh_dot2 (pb, w, 0);
  return (uint32_t) bilh_dot2 (bperm (h1, h0, 0x0c050c01u), vw, 128 + 0x8000);
}

// fast_pixel1_x80 (video_bilinear_rows.h) with its operands as the perms above leave them: ys = word 0 splatbw (y - 128),
// cs = words [splatbw (U - 128) | splatbw (V - 128)]
template <int L>
GSTAMD_HD uint32_t bilh_pixel (const FastParams &fp, uint32_t ys, uint32_t cs, uint32_t (&q)[2])
{
  if (L & GSTAMD_LAYOUT_AYUV)
    return GSTAMD_AYUV_OUT (fp, GSTAMD_AYUV_X80 (ys, cs));
  constexpr int PR = L & 3, PG = (L >> 2) & 3, PB = (L >> 4) & 3;
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

// eight outputs of one row.  ya / yb: the 16 luma bytes of lines 2 y, 2 y + 1; c0 / c1: the upsampled chroma of those lines in the split
// form of h420_filter_raw2 ([0..1] U of the even pixels, [2..3] U of the odd pixels, [4..7] the same for V); vw = {256 - p1 | p1 << 16}
// ST: void (int half, four pixels): outputs 4 half .. 4 half + 3 of the lane
template <int L, class ST>
GSTAMD_HD void bilh_emit_row (const FastParams &fp, const BilhLane &c, const uint32_t *ya, const uint32_t *yb, const uint32_t *c0, const uint32_t *c1,
    uint32_t vw, ST st, uint32_t (&q)[4][2])
{
#pragma unroll
  for (int half = 0; half < 2; half++) {
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int j = 4 * half + k;
      const uint32_t sel_y = (j & 1) ? 0x0c030c02u : 0x0c010c00u;
      const uint32_t sel_c = 0x0c040c00u + (uint32_t) (j & 3) * 0x00010001u;
      const uint32_t ry = bilh_value (bperm (0u, ya[j >> 1], sel_y), bperm (0u, yb[j >> 1], sel_y), c.w[j], vw);
      const uint32_t ru = bilh_value (bperm (c0[2 + half], c0[half], sel_c), bperm (c1[2 + half], c1[half], sel_c), c.w[j], vw);
      const uint32_t rv = bilh_value (bperm (c0[6 + half], c0[4 + half], sel_c), bperm (c1[6 + half], c1[4 + half], sel_c), c.w[j], vw);
      o[k] = bilh_pixel<L> (fp, bperm (0u, ry, 0x0c0c0101u), bperm (rv, ru, 0x05050101u), q[k]);
    }
    st (half, o[0], o[1], o[2], o[3]);
  }
}

// a raw chroma row piece -> its h-filtered split form
template <int CH>
GSTAMD_HD void bilh_filter (const BilParams &bp, const H420Raw &raw, uint32_t *o)
{
  h420_filter_raw2<CH> (!bp.planar, bp.fp.u_first != 0, raw, o);
}

// 3:1 blend of a heavy and a light filtered row, all eight registers
GSTAMD_HD void bilh_blend (const uint32_t *h, const uint32_t *l, uint32_t *o)
{
#pragma unroll
  for (int i = 0; i < 8; i++)
    o[i] = blend31_u8 (h[i], l[i]);
}

// What is asked for ahead of time: the luma of lines 2 y, 2 y + 1 and chroma row min (y + 1, last)
struct BilhReq {
  uint32_t ya[4], yb[4];
  H420Raw raw;
};

GSTAMD_HD void bilh_request (const BilParams &bp, const Planes &pl, int y, int xc, BilhReq &rq)
{
  const uint8_t *y0 = pl.p[0] + (ptrdiff_t) (2 * y) * pl.stride[0];
  wide_load16<true> (y0 + xc, 4, true, rq.ya);
  wide_load16<true> (y0 + pl.stride[0] + xc, 4, true, rq.yb);
  const int crow = y + 1 < bp.fp.crow_hi ? y + 1 : bp.fp.crow_hi;
  bilr_load_raw (bp, pl, crow, xc >> 1, bp.fp.width >> 1, rq.raw);
}

// one lane's share of a strip: outputs [o0, o0 + 8) of rows [y0, y1).  P1: int (int row) -> the second vertical tap of a row (the
// kernel reads it out of a lane table with v_readlane, the emulator from the plan's table); ST: void (row pointer of the lane's first
// output, lane stores?, half, four pixels) - the emulator stores them where they belong, the kernel trades them between lanes first
template <int CH, int L, class P1, class ST>
GSTAMD_HD void bilh_strip (const BilParams &bp, const Planes &pl, uint8_t *__restrict__ dst, int dstride, int x0, int y0, int y1, P1 p1_of, ST st)
{
  const bool active = x0 < bp.fp.width;
  const int xc = active ? x0 : 0;             /* lanes right of the picture work on the first piece and store nothing */
  const int cw = bp.fp.width >> 1;
  BilhLane c;
  bilh_lane_setup (bp, xc >> 1, c);
  uint32_t q[4][2];
  layout_init<L> (q);
  uint32_t s[3][8];                           /* filtered chroma rows y - 1, y, y + 1 (clamped into the plane) */
  {
    H420Raw ra, rb;
    bilr_load_raw (bp, pl, y0 > 0 ? y0 - 1 : 0, xc >> 1, cw, ra);
    bilr_load_raw (bp, pl, y0, xc >> 1, cw, rb);
    bilh_filter<CH> (bp, ra, s[0]);
    bilh_filter<CH> (bp, rb, s[1]);
  }
  BilhReq rq;
  bilh_request (bp, pl, y0, xc, rq);
  uint8_t *__restrict__ d = dst + (ptrdiff_t) y0 * dstride + 2 * (ptrdiff_t) xc;       /* 4 bytes x (xc / 2) outputs */
  for (int y = y0; y < y1; y++) {
    uint32_t ya[4], yb[4], c0[8], c1[8];
#pragma unroll
    for (int i = 0; i < 4; i++)
      ya[i] = rq.ya[i], yb[i] = rq.yb[i];
    bilh_filter<CH> (bp, rq.raw, s[2]);
    if (y + 1 < y1)
      bilh_request (bp, pl, y + 1, xc, rq);                 /* in flight while this row is emitted */
    bilh_blend (s[1], s[0], c0);                            /* line 2 y: 3 B + A; line 2 y + 1: 3 B + C (bilr_window, r0 even) */
    bilh_blend (s[1], s[2], c1);
    const uint32_t p1 = (uint32_t) p1_of (y);
#if defined (GSTAMD_TUNING) && defined (__HIPCC__)
    if (bp.half >= 5) {                 /* ablations (wrong bytes): 5 = loads, chroma filter + blend, stores; 6 = loads and stores only */
      if (bp.half == 5) {
        st (d, active, 0, ya[0] ^ c0[0] ^ c1[0], ya[1] ^ c0[1] ^ c1[1], ya[2] ^ c0[2] ^ c1[2], ya[3] ^ c0[3] ^ c1[3]);
        st (d, active, 1, yb[0] ^ c0[4] ^ c1[4], yb[1] ^ c0[5] ^ c1[5], yb[2] ^ c0[6] ^ c1[6], yb[3] ^ c0[7] ^ c1[7]);
      } else {
        st (d, active, 0, ya[0] ^ s[2][0], ya[1] ^ s[2][1], ya[2] ^ s[2][2], ya[3] ^ s[2][3]);
        st (d, active, 1, yb[0] ^ s[2][4], yb[1] ^ s[2][5], yb[2] ^ s[2][6], yb[3] ^ s[2][7]);
      }
      d += dstride;
      continue;
    }
#endif
    bilh_emit_row<L> (bp.fp, c, ya, yb, c0, c1, (256u - p1) | (p1 << 16),
        [&] (int half, uint32_t a, uint32_t b, uint32_t e, uint32_t f) { st (d, active, half, a, b, e, f); }, q);
    d += dstride;
#pragma unroll
    for (int i = 0; i < 8; i++)
      s[0][i] = s[1][i], s[1][i] = s[2][i];
  }
}

// row strips per 1024-pixel column: one resident round of `slots` waves for a single frame (at least four rows a strip - a strip's first
// row fetches three chroma rows), strips of `rows` rows when given; a strip has at most 64 rows (a lane per row holds its tap)
inline int bilh_strips (int out_h, int rows, int tiles, int slots)
{
  int n = rows > 0 ? (out_h + rows - 1) / rows : slots / (tiles > 0 ? tiles : 1);
  if (rows <= 0 && n > out_h / 4)
    n = out_h / 4;
  const int least = (out_h + 63) / 64;
  n = n < least ? least : n;
  n = n < 1 ? 1 : n;
  return n > out_h ? out_h : n;
}

}  // namespace gstamd
