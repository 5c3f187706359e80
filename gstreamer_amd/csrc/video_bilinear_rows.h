// video_bilinear_rows.h - k_bilinear420_rows: the 4:2:0 -> bilinear 2-tap x 2-tap -> matrix -> 4-byte RGB path of
// video_bilinear_fast.h with the per-pixel work moved to where it is cheapest (BASELINE config 5: 8K NV12 -> 4K BGRA).
//
// Same integers as k_bilinear420 (and as the reference chain it restates: unpack + chroma upsample video-chroma.c:277-327,
// 687-699; ldreslinl video-orc-dist.c:26162; video_orc_resample_v_2tap_u8_lq; video_orc_convert_AYUV_ARGB).  k_bilinear420
// evaluated the chroma of the 2 x 2 source pixels of EVERY OUTPUT from raw samples in 16-bit lanes (~600 VALU instructions per
// 256 outputs, the kernel sat on the VALU roof at 0.30 of the HBM roofline).  Here
//   - chroma is upsampled once per SOURCE pixel, 4 pixels per instruction, in byte lanes (v_lerp_u8 identities of
//     video_hscale420.h), from the h-filtered form of the three chroma rows the two source lines blend; a wave walks `rows`
//     consecutive output rows of its tile and carries those filtered rows with it, so one new chroma row is fetched and
//     filtered per output row at 2:1 instead of four;
//   - LDS holds six BYTE planes in pixel order (Y, U, V of source lines r0 and r0 + 1: 6 bytes per source pixel, as much as the
//     raw staging took), and an output reads the byte pair [idx, idx + 1] of each plane with one unaligned 16-bit LDS load;
//   - everything that depends on the output column only (idx, the ldreslinl fraction, LDS offsets) is computed once per wave,
//     and the luma of two outputs shares the 16-bit lanes of one register (the per-lane fractions differ, the packed
//     multiply-add does not care).
#pragma once
#include "video_bilinear_fast.h"
#include "video_hscale420.h"

namespace gstamd {

#define BILR_PLANES 6                    // Y0 Y1 U0 V0 U1 V1
// bytes per LDS plane: a compile-time constant, so that the plane of a read is an instruction offset and not an address add
#define BILR_PLANE_BYTES (BIL_MAX_SPAN + 16)
#define BILR_EMPTY (-0x40000000)

// the three chroma rows around the two source lines r0, r0 + 1 (regular pairing, bil_rows): A = row c - 1, B = row c,
// C = row c + 1 with c = (r0 + 1) >> 1, clamped into the plane.  r0 even: line r0 blends 3 B + A, line r0 + 1 blends 3 B + C;
// r0 odd: line r0 blends 3 A + B, line r0 + 1 blends 3 B + A (C is not needed).
GSTAMD_HOSTDEV void bilr_window (const BilParams &bp, int r0, int *wa, int *wb, int *wc)
{
  const int c = (r0 + 1) >> 1, lo = bp.fp.crow_lo, hi = bp.fp.crow_hi;
  *wa = c - 1 > lo ? c - 1 : lo;
  *wb = c < hi ? c : hi;
  *wc = c + 1 < hi ? c + 1 : hi;
}

// does the window form give, for source lines r0 and r0 + 1, exactly the rows and weights bil_rows gives? (host, once per plan)
inline bool bilr_window_matches (const BilParams &bp, int r0)
{
  int wa, wb, wc, ra, rb, role;
  bilr_window (bp, r0, &wa, &wb, &wc);
  for (int l = 0; l < 2; l++) {
    bil_rows (bp, r0 + l, &ra, &rb, &role);
    const int heavy = role == 0 ? ra : rb, light = role == 0 ? rb : ra;
    int eh, el;
    if (!(r0 & 1))
      eh = wb, el = l == 0 ? wa : wc;
    else
      eh = l == 0 ? wa : wb, el = l == 0 ? wb : wa;
    /* equal rows blend to themselves whatever the weights */
    if (!((heavy == eh && light == el) || (heavy == light && eh == el && heavy == eh)))
      return false;
  }
  return true;
}

struct BilrState {
  uint32_t s[3][8];     // h-filtered chroma rows of the lane's 16 pixels, in pixel order: [0..3] U, [4..7] V
  int id[3];            // chroma row each holds (wave-uniform)
};

GSTAMD_HD void bilr_state_init (BilrState &st) { st.id[0] = st.id[1] = st.id[2] = BILR_EMPTY; }

// what depends on the output column only
#define BILR_MAX_PAIRS 3                 // a lane produces 2 NP outputs of a row: tiles of 256 (NP = 2) or 384 (NP = 3) outputs
struct BilrLane {
  uint32_t off[2 * BILR_MAX_PAIRS];     // LDS byte offset of source pixel idx of output t0 + lane + 64 i inside a plane
  uint32_t frs[2 * BILR_MAX_PAIRS];     // ldreslinl fraction f in both 16-bit lanes (256 - f is one subtraction away: registers are scarcer)
  uint32_t frs_pair[BILR_MAX_PAIRS];    // {f of output 2 j | f of output 2 j + 1 << 16}
  uint32_t xoff[2 * BILR_MAX_PAIRS];    // byte offset of the output in its destination row
};

template <int NP>
GSTAMD_HD void bilr_lane_setup (const BilParams &bp, int t0, int t1, int xa, int lane, BilrLane &c)
{
  uint32_t fr[2 * NP];
#pragma unroll
  for (int i = 0; i < 2 * NP; i++) {
    int x = t0 + lane + 64 * i;
    x = x < t1 ? x : t1 - 1;              /* lanes past the tile compute a valid pixel and do not store it */
    const int tmp = (int) umul24 ((uint32_t) x, (uint32_t) bp.inc);
    c.off[i] = (uint32_t) ((tmp >> 16) - xa);
    c.xoff[i] = 4u * (uint32_t) x;
    fr[i] = (uint32_t) (tmp >> 8) & 0xffu;
    c.frs[i] = fr[i] * 0x00010001u;
  }
#pragma unroll
  for (int j = 0; j < NP; j++)
    c.frs_pair[j] = fr[2 * j] | (fr[2 * j + 1] << 16);
}

GSTAMD_HD void bilr_load_raw (const BilParams &bp, const Planes &pl, int crow, int k0, int cw, H420Raw &r)
{
  const int km = k0 > 0 ? k0 - 1 : 0, kp = k0 + 8 < cw ? k0 + 8 : cw - 1;
  if (!bp.planar) {
    const uint8_t *row = pl.p[1] + (ptrdiff_t) crow * pl.stride[1];
    const uint4 m = *(const uint4 *) (row + (uint32_t) (2 * k0));
    r.u0 = m.x, r.u1 = m.y, r.v0 = m.z, r.v1 = m.w;               // still interleaved, see h420_filter_raw2
    r.um = *(const uint16_t *) (row + (uint32_t) (2 * km));
    r.up = *(const uint16_t *) (row + (uint32_t) (2 * kp));
    r.vm = r.vp = 0;
  } else {
    /* a select of the two plane pointers, not pl.p[bp.u_plane]: indexing the array loses the pointers' address space and the loads
     * come out as FLAT instructions, which wait on both counters */
    const uint8_t *ru = (bp.u_plane == 1 ? pl.p[1] : pl.p[2]) + (ptrdiff_t) crow * (bp.u_plane == 1 ? pl.stride[1] : pl.stride[2]);
    const uint8_t *rv = (bp.v_plane == 1 ? pl.p[1] : pl.p[2]) + (ptrdiff_t) crow * (bp.v_plane == 1 ? pl.stride[1] : pl.stride[2]);
    const uint2 mu = *(const uint2 *) (ru + (uint32_t) k0), mv = *(const uint2 *) (rv + (uint32_t) k0);
    r.u0 = mu.x, r.u1 = mu.y, r.v0 = mv.x, r.v1 = mv.y;
    r.um = ru[(uint32_t) km], r.up = ru[(uint32_t) kp], r.vm = rv[(uint32_t) km], r.vp = rv[(uint32_t) kp];
  }
}

// a raw chroma row piece -> its h-filtered form in pixel order (the even / odd pixel registers of h420_filter_raw2 interleaved
// here, once per row fetched, rather than after every 3:1 blend the row takes part in)
template <int CH>
GSTAMD_HD void bilr_filter_px (const BilParams &bp, const H420Raw &raw, uint32_t *o)
{
  uint32_t r[8];
  h420_filter_raw2<CH> (!bp.planar, bp.fp.u_first != 0, raw, r);
  o[0] = bperm (r[2], r[0], 0x05010400u), o[1] = bperm (r[2], r[0], 0x07030602u);
  o[2] = bperm (r[3], r[1], 0x05010400u), o[3] = bperm (r[3], r[1], 0x07030602u);
  o[4] = bperm (r[6], r[4], 0x05010400u), o[5] = bperm (r[6], r[4], 0x07030602u);
  o[6] = bperm (r[7], r[5], 0x05010400u), o[7] = bperm (r[7], r[5], 0x07030602u);
}

// 3:1 blend of a heavy and a light filtered row -> 16 bytes of U and 16 bytes of V
GSTAMD_HD void bilr_blend_line (const uint32_t *h, const uint32_t *l, uint32_t *pu, uint32_t *pv)
{
#pragma unroll
  for (int i = 0; i < 4; i++) {
    pu[i] = blend31_u8 (h[i], l[i]);
    pv[i] = blend31_u8 (h[4 + i], l[4 + i]);
  }
}

GSTAMD_HD void bilr_copy8 (uint32_t *d, const uint32_t *s)
{
#pragma unroll
  for (int i = 0; i < 8; i++)
    d[i] = s[i];
}

// What a wave asks for ahead of time: the luma of source lines r0, r0 + 1 and the ONE chroma row that is new at the usual
// ratios (the bottom row of the window); issued before the previous output row is emitted, consumed after it.
struct BilrReq {
  uint32_t ya[4], yb[4];
  H420Raw raw;
  int row;              // chroma row `raw` holds, or BILR_EMPTY (wave-uniform)
};

GSTAMD_HD int bilr_piece (const BilParams &bp, int xa, int x_hi, int lane)
{
  const int x = xa + 16 * lane;
  return (x < x_hi && x + 16 <= bp.fp.width) ? x : xa;        /* lanes outside the span read a valid piece they do not commit */
}

GSTAMD_HD void bilr_request (const BilParams &bp, const Planes &pl, const BilrState &st, int r0, int xa, int x_hi, int lane, BilrReq &rq)
{
  const int xc = bilr_piece (bp, xa, x_hi, lane);
  const uint8_t *y0 = pl.p[0] + (ptrdiff_t) r0 * pl.stride[0];
  wide_load16<true> (y0 + xc, 4, true, rq.ya);
  wide_load16<true> (y0 + pl.stride[0] + xc, 4, true, rq.yb);
  int want[3];
  bilr_window (bp, r0, &want[0], &want[1], &want[2]);
  const int t = (r0 & 1) ? want[1] : want[2];
  rq.row = (st.id[0] != t && st.id[1] != t && st.id[2] != t) ? t : BILR_EMPTY;
  if (rq.row != BILR_EMPTY)
    bilr_load_raw (bp, pl, t, xc >> 1, bp.fp.width >> 1, rq.raw);
}

// Source lines r0, r0 + 1 of the tile's span [xa, x_hi) into the six LDS planes: the lane owns the 16 pixels from
// xa + 16 * lane.  Wave-uniform control flow throughout (row numbers only).  (Turning the slot assignment instead of moving the
// rows that stay - three instances of this body - cost 27 registers and a wave of occupancy: slower.)
template <int CH>
GSTAMD_HD void bilr_install (const BilParams &bp, const Planes &pl, BilrState &st, const BilrReq &rq, int r0, int xa, int x_hi, int lane, uint8_t *lds)
{
  constexpr int P = BILR_PLANE_BYTES;
  const int x = xa + 16 * lane;
  const int xc = bilr_piece (bp, xa, x_hi, lane);
  int want[3];
  bilr_window (bp, r0, &want[0], &want[1], &want[2]);
  const bool need_c = !(r0 & 1);
  // rows already held move to their new slot (the window only ever moves down)
  if (st.id[0] != want[0]) {
    if (st.id[1] == want[0])
      bilr_copy8 (st.s[0], st.s[1]), st.id[0] = st.id[1];
    else if (st.id[2] == want[0])
      bilr_copy8 (st.s[0], st.s[2]), st.id[0] = st.id[2];
  }
  if (st.id[1] != want[1] && st.id[2] == want[1])
    bilr_copy8 (st.s[1], st.s[2]), st.id[1] = st.id[2];
  // what is still missing: every load first (one round trip, also on the first row of a wave), then the filters
  const int cw = bp.fp.width >> 1;
  const bool get_a = st.id[0] != want[0];
  const bool get_b = st.id[1] != want[1] && want[1] != want[0];
  const bool get_c = need_c && st.id[2] != want[2] && want[2] != want[1];
  H420Raw ra, rb, rc;
  if (get_a && rq.row != want[0])
    bilr_load_raw (bp, pl, want[0], xc >> 1, cw, ra);
  if (get_b && rq.row != want[1])
    bilr_load_raw (bp, pl, want[1], xc >> 1, cw, rb);
  if (get_c && rq.row != want[2])
    bilr_load_raw (bp, pl, want[2], xc >> 1, cw, rc);
  if (get_a) {
    if (rq.row == want[0])
      bilr_filter_px<CH> (bp, rq.raw, st.s[0]);
    else
      bilr_filter_px<CH> (bp, ra, st.s[0]);
    st.id[0] = want[0];
  }
  if (get_b) {
    if (rq.row == want[1])
      bilr_filter_px<CH> (bp, rq.raw, st.s[1]);
    else
      bilr_filter_px<CH> (bp, rb, st.s[1]);
    st.id[1] = want[1];
  } else if (st.id[1] != want[1]) {
    bilr_copy8 (st.s[1], st.s[0]), st.id[1] = want[1];          /* clamped at the top: B is the row A holds */
  }
  if (need_c) {
    if (get_c) {
      if (rq.row == want[2])
        bilr_filter_px<CH> (bp, rq.raw, st.s[2]);
      else
        bilr_filter_px<CH> (bp, rc, st.s[2]);
      st.id[2] = want[2];
    } else if (st.id[2] != want[2]) {
      bilr_copy8 (st.s[2], st.s[1]), st.id[2] = want[2];        /* clamped at the bottom: C is the row B holds */
    }
  }
  uint32_t u0[4], v0[4], u1[4], v1[4];
  if (need_c) {
    bilr_blend_line (st.s[1], st.s[0], u0, v0);
    bilr_blend_line (st.s[1], st.s[2], u1, v1);
  } else {
    bilr_blend_line (st.s[0], st.s[1], u0, v0);
    bilr_blend_line (st.s[1], st.s[0], u1, v1);
  }
  if (x < x_hi) {
    uint8_t *d = lds + (x - xa);
    *(uint4 *) (d + 0 * P) = gstamd_make_uint4 (rq.ya[0], rq.ya[1], rq.ya[2], rq.ya[3]);
    *(uint4 *) (d + 1 * P) = gstamd_make_uint4 (rq.yb[0], rq.yb[1], rq.yb[2], rq.yb[3]);
    *(uint4 *) (d + 2 * P) = gstamd_make_uint4 (u0[0], u0[1], u0[2], u0[3]);
    *(uint4 *) (d + 3 * P) = gstamd_make_uint4 (v0[0], v0[1], v0[2], v0[3]);
    *(uint4 *) (d + 4 * P) = gstamd_make_uint4 (u1[0], u1[1], u1[2], u1[3]);
    *(uint4 *) (d + 5 * P) = gstamd_make_uint4 (v1[0], v1[1], v1[2], v1[3]);
  }
}

GSTAMD_HD void issue_order_fence_v ()
{
#ifdef __HIPCC__
  __builtin_amdgcn_sched_barrier (0);
#endif
}

// bytes [o, o + 1] of an LDS plane (the source pixels idx and idx + 1).  EVEN (every offset of the plan is even: increments that are multiples
// of two pixels, the span starts on 16): one 16-bit load.  Otherwise two byte loads: a ds_read_u16 at an ODD address is legal and gives the right
// bytes, but the LDS serialises such lanes - at 3:2 (every other lane odd) the kernel took 47.5 us per 1440p frame against 14.5 with the
// offsets forced even and 15.7 with byte loads (profiles/r05/bilr_lds_align.log)
template <bool EVEN>
GSTAMD_HD uint32_t bilr_pair (const uint8_t *plane, uint32_t o)
{
  if (EVEN) {
    uint16_t v;
    __builtin_memcpy (&v, plane + o, 2);
    return v;
  }
  return (uint32_t) plane[o] | ((uint32_t) plane[o + 1] << 8);
}
GSTAMD_HOSTDEV bool bilr_even_offsets (const BilParams &bp) { return (bp.inc & 0x1ffff) == 0; }

// ldreslinl on two 16-bit lanes: (a * (256 - f) + b * f) >> 8
GSTAMD_HD uint32_t bilr_h (uint32_t a, uint32_t b, uint32_t frs) { return pk_shr<8> (pk_mad16 (b, frs, pk_mad16 (a, 0x01000100u - frs, 0u))); }

// the eight byte pairs two outputs (2 j, 2 j + 1) read: luma of both on the two lines, then U, V of each on the two lines
struct BilrPairs {
  uint32_t l0a, l0b, l1a, l1b;
  uint32_t u0[2], v0[2], u1[2], v1[2];
};

template <bool EVEN>
GSTAMD_HD void bilr_read_pairs (const BilrLane &c, const uint8_t *lds, int j, BilrPairs &r)
{
  constexpr int P = BILR_PLANE_BYTES;
  const uint32_t oa = c.off[2 * j], ob = c.off[2 * j + 1];
  r.l0a = bilr_pair<EVEN> (lds, oa), r.l0b = bilr_pair<EVEN> (lds, ob);
  r.l1a = bilr_pair<EVEN> (lds + P, oa), r.l1b = bilr_pair<EVEN> (lds + P, ob);
  r.u0[0] = bilr_pair<EVEN> (lds + 2 * P, oa), r.v0[0] = bilr_pair<EVEN> (lds + 3 * P, oa);
  r.u1[0] = bilr_pair<EVEN> (lds + 4 * P, oa), r.v1[0] = bilr_pair<EVEN> (lds + 5 * P, oa);
  r.u0[1] = bilr_pair<EVEN> (lds + 2 * P, ob), r.v0[1] = bilr_pair<EVEN> (lds + 3 * P, ob);
  r.u1[1] = bilr_pair<EVEN> (lds + 4 * P, ob), r.v1[1] = bilr_pair<EVEN> (lds + 5 * P, ob);
}

// v2tap_pk with 0x80 added to both results (mod 256), i.e. each XOR 0x80 - the form the AYUV -> ARGB matrix wants (value - 128 as
// a signed byte): adding 0x8000 to the rounding term adds 0x80 after the >> 8, for free
GSTAMD_HD uint32_t v2tap_pk_x80 (uint32_t s1, uint32_t s2, uint32_t p1_splat)
{
  const uint32_t m = pk_mad16 (pk_sub16 (s2, s1), p1_splat, 0x80808080u);
  return (pk_shr<8> (m) + s1) & 0x00ff00ffu;
}

// fast_pixel1_l on operands that are XOR 0x80 already: ys = word 0 splatbw (y - 128), cx = {U ^ 0x80 | V ^ 0x80 << 16}
template <int L>
GSTAMD_HD uint32_t fast_pixel1_x80 (const FastParams &fp, uint32_t ys, uint32_t cx, uint32_t (&q)[2])
{
  if (L & GSTAMD_LAYOUT_AYUV)
    return GSTAMD_AYUV_OUT (fp, GSTAMD_AYUV_X80 (ys, cx));
  constexpr int PR = L & 3, PG = (L >> 2) & 3, PB = (L >> 4) & 3;
  const uint32_t cs = bperm (cx, cx, 0x02020000u);                     // words [t(U) | t(V)]
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

GSTAMD_HD void bilr_store (uint8_t *__restrict__ row, uint32_t xoff, uint32_t v)
{
#ifdef __HIPCC__
  // uniform row pointer + 32-bit lane offset: the saddr form, no address arithmetic and no 64-bit lane pointer per output (left
  // to itself the compiler hoists dst + xoff out of the row loop into a register pair per output and adds y * stride to each)
  asm volatile ("global_store_dword %0, %1, %2 nt" : : "v" (xoff), "v" (v), "s" (row) : "memory");
#else
  *(uint32_t *) (row + xoff) = v;
#endif
}

// one output row of the tile from the staged planes: lane `lane` produces outputs t0 + lane + 64 i (i < 2 NP).  The LDS reads of
// the next pair of outputs are issued before the arithmetic of this pair starts (the compiler left to itself issues each read
// right in front of its use and the wave sits out the LDS latency once per output).  Lanes past the tile computed the tile's last
// pixel (bilr_lane_setup): they store it again, same value to the same address, which keeps the row one straight run of code.
template <int L, int NP, bool EVEN = false>
GSTAMD_HD void bilr_emit_row (const BilParams &bp, const BilrLane &c, const uint8_t *lds, uint8_t *__restrict__ dst, int dstride, int y, int p1,
    uint32_t (&q)[2])
{
  const uint32_t p1s = ((uint32_t) (uint16_t) p1) * 0x00010001u;         /* p1 = bp.vtaps[2 y + 1], handed in by the caller */
  uint8_t *__restrict__ row = dst + (ptrdiff_t) y * dstride;
  BilrPairs rp[2];
  bilr_read_pairs<EVEN> (c, lds, 0, rp[0]);
#pragma unroll
  for (int j = 0; j < NP; j++) {
    if (j + 1 < NP)
      bilr_read_pairs<EVEN> (c, lds, j + 1, rp[(j + 1) & 1]);
    issue_order_fence_v ();
    const BilrPairs &r = rp[j & 1];
    // luma of two outputs: {output 2 j | output 2 j + 1 << 16} on each source line, then the vertical 2-tap between the lines
    const uint32_t h0 = bilr_h (bperm (r.l0b, r.l0a, 0x0c040c00u), bperm (r.l0b, r.l0a, 0x0c050c01u), c.frs_pair[j]);
    const uint32_t h1 = bilr_h (bperm (r.l1b, r.l1a, 0x0c040c00u), bperm (r.l1b, r.l1a, 0x0c050c01u), c.frs_pair[j]);
    const uint32_t yv = v2tap_pk_x80 (h0, h1, p1s);
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int i = 2 * j + k;
      // chroma: {U | V << 16} of the two source pixels on each line
      const uint32_t ch0 = bilr_h (bperm (r.v0[k], r.u0[k], 0x0c040c00u), bperm (r.v0[k], r.u0[k], 0x0c050c01u), c.frs[i]);
      const uint32_t ch1 = bilr_h (bperm (r.v1[k], r.u1[k], 0x0c040c00u), bperm (r.v1[k], r.u1[k], 0x0c050c01u), c.frs[i]);
      const uint32_t cv = v2tap_pk_x80 (ch0, ch1, p1s);
      const uint32_t ys = bperm (0u, yv, k == 0 ? 0x0c0c0000u : 0x0c0c0202u);
      bilr_store (row, c.xoff[i], fast_pixel1_x80<L> (bp.fp, ys, cv, q));
    }
  }
}

// row strips per tile column.  rows > 0: strips of that many rows; rows < 0: as many strips as fill the `slots` waves the device
// holds at once, so that the whole frame is one resident round with the rows spread evenly (with a fixed strip height the last
// wave slots stay empty or a second, partial round starts: C5 at 6 rows per wave left every fourth SIMD a wave short).  A strip
// has at most 64 rows (one lane per row holds the row's table entries).
inline int bilr_strips (int out_h, int rows, int tiles, int slots)
{
  int n = rows > 0 ? (out_h + rows - 1) / rows : slots / (tiles > 0 ? tiles : 1);
  if (rows <= 0 && n > out_h / 4)
    n = out_h / 4;              /* small frames: no fewer than four rows per wave, the chroma rows a wave carries are what it is about */
  const int least = (out_h + 63) / 64;
  n = n < least ? least : n;
  n = n < 1 ? 1 : n;
  return n > out_h ? out_h : n;
}

// outputs per wave of the rows kernel: 384 (six per lane) when their source span still fits the 64 x 16 pixels a wave stages -
// the staging work is per span piece, the wider the tile the more lanes of it are busy - else what bil_pick_tile finds
inline int bilr_pick_tile (int out_w, int inc, int *ylen)
{
  const int yl = bil_ylen (out_w, inc, 384);
  if (yl > 0 && out_w > 256) {
    *ylen = yl;
    return 384;
  }
  return bil_pick_tile (out_w, inc, ylen);
}

GSTAMD_HOSTDEV size_t bilr_lds_bytes () { return (size_t) BILR_PLANES * BILR_PLANE_BYTES; }

}  // namespace gstamd
