// video_fast.h - the speed-of-light variant of the unscaled NV12/NV21 -> 4-byte RGB conversion
// (BASELINE config 2).  Same arithmetic as convert_body in video_device.h, restructured so the
// kernel is bounded by HBM and not by VALU issue:
//
//   * one lane converts an 8-pixel span of a chroma LINE PAIR (2p-1, 2p) - the pairing the reference's
//     do_upsample_lines produces (video-converter.c:2991-3021) - so each chroma row is loaded and
//     horizontally filtered once for the two luma lines that use it;
//   * chroma lives in packed 16-bit lanes {U, V} of one VGPR; the 3:1 blends are single multiply-adds
//     over both lanes (no carries: values <= 1022);
//   * mulhsw (splatbw (x - 128), p) of video_orc_convert_AYUV_ARGB is ONE v_mul_hi_i32_i24:
//     splatbw(b) << 8 == (b ^ 0x80) * 0x10100 is a sign-correct 24-bit operand, p << 8 the other;
//     their 48-bit product >> 32 is (splat * p) >> 16;
//   * bytes move with v_perm_b32; clamp (x, -128, 127) + 128 == med3 (x + 128, 0, 255).
//
// The 16-bit wrap of ORC's addw cannot trigger for the matrices this kernel accepts; the planner
// proves that (fast_matrix_ok) before selecting it, otherwise convert_body runs.
#pragma once
#include "video_device.h"

namespace gstamd {

struct FastParams {
  int width, height;
  int p8[5];            // p1..p5 << 8
  uint32_t pack_sel;    // v_perm selector: dest byte pos[A] <- 0xff, pos[R] <- lo.0, pos[G] <- lo.1, pos[B] <- hi.0
  int u_first;          // 1: NV12 (U,V), 0: NV21 (V,U)
};

#ifdef __HIPCC__
GSTAMD_HD uint32_t bperm (uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm (hi, lo, sel); }
#else
// v_perm_b32: selector byte 0-3 -> byte of lo, 4-7 -> byte of hi, 0x0c -> 0x00, >= 0x0d -> 0xff
inline uint32_t bperm (uint32_t hi, uint32_t lo, uint32_t sel)
{
  uint32_t r = 0;
  for (int i = 0; i < 4; i++) {
    const uint32_t s = (sel >> (8 * i)) & 0xff;
    uint32_t b;
    if (s < 4)
      b = (lo >> (8 * s)) & 0xff;
    else if (s < 8)
      b = (hi >> (8 * (s - 4))) & 0xff;
    else if (s == 0x0c)
      b = 0;
    else
      b = 0xff;
    r |= b << (8 * i);
  }
  return r;
}
#endif

// (a * b) >> 32 of the sign-extended low 24 bits of both operands: one v_mul_hi_i32_i24.  Written as
// asm on the device because the C form makes hipcc sign-extend the operands with extra v_bfe_i32.
GSTAMD_HD int mulhi24 (int a, int b)
{
#ifdef __HIPCC__
  int r;
  asm ("v_mul_hi_i32_i24 %0, %1, %2" : "=v" (r) : "v" (a), "s" (b));   /* b: wave-uniform coefficient */
  return r;
#else
  const int a24 = (int) ((uint32_t) a << 8) >> 8, b24 = (int) ((uint32_t) b << 8) >> 8;
  return (int) (((long long) a24 * (long long) b24) >> 32);
#endif
}

// per-16-bit-lane logical shift right (v_pk_lshrrev_b16)
template <int N>
GSTAMD_HD uint32_t pk_shr (uint32_t x)
{
#ifdef __HIPCC__
  typedef unsigned short us2 __attribute__ ((ext_vector_type (2)));
  us2 v = __builtin_bit_cast (us2, x);
  v = v >> (unsigned short) N;
  return __builtin_bit_cast (uint32_t, v);
#else
  return (x >> N) & (0x0000ffffu >> N) * 0x00010001u;
#endif
}

// write-once output: nontemporal (streaming) 16-byte store.  On MI355X a plain store costs the
// 12.4 MB-in / 33 MB-out byte mix 10.8 us per 4K frame, the nontemporal one 7.5 us (scripts/membench.hip).
GSTAMD_HD void store16_stream (uint8_t *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
#ifdef __HIPCC__
  typedef unsigned int u32x4 __attribute__ ((ext_vector_type (4)));
  u32x4 v = {a, b, c, d};
  __builtin_nontemporal_store (v, (u32x4 *) p);
#else
  *(uint4 *) p = gstamd_make_uint4 (a, b, c, d);
#endif
}

GSTAMD_HD int med3_0_255 (int v)
{
#ifdef __HIPCC__
  int r;
  asm ("v_med3_i32 %0, %1, 0, %2" : "=v" (r) : "v" (v), "s" (255));
  return r;
#else
  return v < 0 ? 0 : (v > 255 ? 255 : v);
#endif
}

// one output pixel.  yx = 4 luma bytes ^ 0x80808080, cx = packed chroma {lane0, lane1} ^ 0x00800080: the
// "- 128" of the reference (subb 128, bytewise wrap) is that XOR, done once per word instead of per byte.
GSTAMD_HD uint32_t fast_pixel (const FastParams &fp, uint32_t yx, uint32_t ysel, uint32_t cx, uint32_t usel, uint32_t vsel)
{
  const int sy = (int) bperm (yx, yx, ysel);      // ((Y ^ 0x80) * 0x10100): splatbw << 8, a signed 24-bit value
  const int su = (int) bperm (cx, cx, usel);
  const int sv = (int) bperm (cx, cx, vsel);
  const int wy = mulhi24 (sy, fp.p8[0]);
  const int r = med3_0_255 (wy + mulhi24 (sv, fp.p8[1]) + 128);
  const int b = med3_0_255 (wy + mulhi24 (su, fp.p8[2]) + 128);
  const int g = med3_0_255 (wy + mulhi24 (su, fp.p8[3]) + mulhi24 (sv, fp.p8[4]) + 128);
  // bytes: lo = {r, g, 0, 0}, hi = {b, ...}; pack_sel routes r/g/b and the constant 0xff alpha
  return bperm ((uint32_t) b, ((uint32_t) g << 8) | (uint32_t) r, fp.pack_sel);
}

// horizontally filtered chroma for the NPX pixels x0 .. x0+NPX-1 of one chroma row: out[i] packed
// {c0, c1} in u16 lanes (c0 = first byte of the interleaved pair).  x0 % NPX == 0, rows NPX-byte aligned.
template <int CH, int NPX>
GSTAMD_HD void fast_hchroma (const uint8_t *__restrict__ row, int cw, int x0, int w, uint32_t *out)
{
  constexpr int NS = NPX / 2;        // chroma samples under the span
  const int k0 = x0 >> 1;
  uint32_t raw[NPX / 4];
  if (NPX == 4) {
    raw[0] = *(const uint32_t *) (row + 2 * (size_t) k0);
  } else if (NPX == 8) {
    const uint2 m = *(const uint2 *) (row + 2 * (size_t) k0);
    raw[0] = m.x;
    raw[NPX >= 8 ? 1 : 0] = m.y;
  } else {
    const uint4 m = *(const uint4 *) (row + 2 * (size_t) k0);
    raw[0] = m.x;
    raw[NPX >= 16 ? 1 : 0] = m.y;
    raw[NPX >= 16 ? 2 : 0] = m.z;
    raw[NPX >= 16 ? 3 : 0] = m.w;
  }
  uint32_t S[NS + 2];                // samples k0-1 .. k0+NS as {c0, c1} 16-bit lanes
#pragma unroll
  for (int j = 0; j < NS; j++)
    S[j + 1] = bperm (raw[j >> 1], raw[j >> 1], (j & 1) ? 0x0c030c02u : 0x0c010c00u);
  if (CH != CHROMA_H_NONE) {
    const int kp = k0 + NS < cw ? k0 + NS : cw - 1;
    const uint32_t pp = *(const uint16_t *) (row + 2 * (size_t) kp);
    S[NS + 1] = bperm (pp, pp, 0x0c010c00u);
  } else {
    S[NS + 1] = S[NS];
  }
  if (CH == CHROMA_H_H2) {
    const int km = k0 > 0 ? k0 - 1 : 0;
    const uint32_t pm = *(const uint16_t *) (row + 2 * (size_t) km);
    S[0] = bperm (pm, pm, 0x0c010c00u);
  } else {
    S[0] = S[1];
  }
#pragma unroll
  for (int j = 0; j < NS; j++) {
    const int xe = x0 + 2 * j, xo = xe + 1;
    uint32_t e = S[j + 1], o = S[j + 1];
    if (CH == CHROMA_H_H2_CS) {
      if (xo < w - 1)
        o = pk_shr<1> (S[j + 1] + S[j + 2] + 0x00010001u);
    } else if (CH == CHROMA_H_H2) {
      if (xo < w - 1)
        o = pk_shr<2> (3u * S[j + 1] + S[j + 2] + 0x00020002u);
      if (xe >= 2)
        e = pk_shr<2> (S[j] + 3u * S[j + 1] + 0x00020002u);
    }
    out[2 * j] = e;
    out[2 * j + 1] = o;
  }
}

template <int NPX>
GSTAMD_HD void fast_load_y (const uint8_t *__restrict__ p, uint32_t *y)
{
  if (NPX == 4) {
#ifdef __HIPCC__
    y[0] = __builtin_nontemporal_load ((const uint32_t *) p);     // luma is read exactly once: streaming load
#else
    y[0] = *(const uint32_t *) p;
#endif
  } else if (NPX == 8) {
    const uint2 m = *(const uint2 *) p;
    y[0] = m.x;
    y[NPX >= 8 ? 1 : 0] = m.y;
  } else {
    const uint4 m = *(const uint4 *) p;
    y[0] = m.x;
    y[NPX >= 16 ? 1 : 0] = m.y;
    y[NPX >= 16 ? 2 : 0] = m.z;
    y[NPX >= 16 ? 3 : 0] = m.w;
  }
}

// One line of the pair: blend the two chroma rows (role 0: (3a+b+2)>>2, role 1: (a+3b+2)>>2), convert, store.
// ABL (ablation, benchmarking only): 0 = real kernel, 1 = loads + stores with trivial math.
template <int NPX, int ABL>
GSTAMD_HD void fast_line (const FastParams &fp, const Planes &pl, uint8_t *__restrict__ dst, int dstride, int line, int x0,
    const uint32_t *ca, const uint32_t *cb, bool blend, int role, uint32_t usel, uint32_t vsel)
{
  static const uint32_t ysel[4] = {0x0c00000cu, 0x0c01010cu, 0x0c02020cu, 0x0c03030cu};
  uint32_t yy[NPX / 4], o[NPX];
  fast_load_y<NPX> (pl.p[0] + (size_t) line * pl.stride[0] + x0, yy);
#pragma unroll
  for (int i = 0; i < NPX; i++) {
    uint32_t c = ca[i];
    if (blend)
      c = role == 0 ? pk_shr<2> (3u * ca[i] + cb[i] + 0x00020002u) : pk_shr<2> (ca[i] + 3u * cb[i] + 0x00020002u);
    if (ABL == 1)
      o[i] = yy[i >> 2] ^ c;
    else
      o[i] = fast_pixel (fp, yy[i >> 2] ^ 0x80808080u, ysel[i & 3], c ^ 0x00800080u, usel, vsel);
  }
  uint8_t *d = dst + (size_t) line * dstride + 4 * (size_t) x0;
#pragma unroll
  for (int q = 0; q < NPX / 4; q++)
    store16_stream (d + 16 * q, o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
}

// Pair index p (lines 2p-1 and 2p), NPX pixels starting at x0.  Requires: 4:2:0 semi-planar source,
// x0 + NPX <= width, NPX-byte aligned luma/chroma rows, 16-byte aligned destination rows.
template <int CH, int NPX, int ABL>
GSTAMD_HD void fast_pair_span (const FastParams &fp, const Planes &pl, uint8_t *__restrict__ dst, int dstride, int x0, int p)
{
  const int w = fp.width, h = fp.height;
  const int l0 = 2 * p - 1, l1 = 2 * p;
  const bool have0 = l0 >= 0, have1 = l1 < h;
  const int cw = (w + 1) >> 1;
  const int ra = have0 ? p - 1 : p;            // chroma row of line l0 (or of l1 when l0 is absent)
  const int rb = have1 ? p : p - 1;            // chroma row of line l1 (or of l0 when l1 is absent)
  uint32_t ca[NPX], cb[NPX];
  fast_hchroma<CH, NPX> (pl.p[1] + (size_t) ra * pl.stride[1], cw, x0, w, ca);
  const bool blend = rb != ra;
  if (blend)
    fast_hchroma<CH, NPX> (pl.p[1] + (size_t) rb * pl.stride[1], cw, x0, w, cb);
  // byte selectors: 16-bit lane0 / lane1 low byte -> bits 8..23
  const uint32_t sel_l0 = 0x0c00000cu, sel_l1 = 0x0c02020cu;
  const uint32_t usel = fp.u_first ? sel_l0 : sel_l1, vsel = fp.u_first ? sel_l1 : sel_l0;
  if (have0)
    fast_line<NPX, ABL> (fp, pl, dst, dstride, l0, x0, ca, cb, blend, 0, usel, vsel);
  if (have1)
    fast_line<NPX, ABL> (fp, pl, dst, dstride, l1, x0, ca, cb, blend, 1, usel, vsel);
}

// Single-line variant: one lane converts NPX pixels of ONE line (the chroma rows are fetched again by the
// partner line's lane, from L2).  Same results as fast_pair_span.
template <int CH, int NPX, int ABL>
GSTAMD_HD void fast_line_span (const FastParams &fp, const Planes &pl, uint8_t *__restrict__ dst, int dstride, int x0, int line)
{
  const int w = fp.width, h = fp.height;
  const int p = (line + 1) >> 1, l0 = 2 * p - 1, l1 = 2 * p;
  const bool have0 = l0 >= 0, have1 = l1 < h;
  const int cw = (w + 1) >> 1;
  const int ra = have0 ? p - 1 : p, rb = have1 ? p : p - 1;
  uint32_t ca[NPX], cb[NPX];
  fast_hchroma<CH, NPX> (pl.p[1] + (size_t) ra * pl.stride[1], cw, x0, w, ca);
  const bool blend = rb != ra;
  if (blend)
    fast_hchroma<CH, NPX> (pl.p[1] + (size_t) rb * pl.stride[1], cw, x0, w, cb);
  const uint32_t sel_l0 = 0x0c00000cu, sel_l1 = 0x0c02020cu;
  const uint32_t usel = fp.u_first ? sel_l0 : sel_l1, vsel = fp.u_first ? sel_l1 : sel_l0;
  fast_line<NPX, ABL> (fp, pl, dst, dstride, line, x0, ca, cb, blend, line == l0 ? 0 : 1, usel, vsel);
}

// ---- strip variant: one lane walks K consecutive line pairs of its NPX-pixel column --------------------
// Loads of pair p+1 are issued before pair p is computed (software prefetch: with ~16 waves per SIMD in the
// whole grid the hardware alone cannot overlap the load, VALU and store phases), and the horizontally
// filtered chroma row of pair p is reused as the upper row of pair p+1 (each chroma row is loaded and
// filtered exactly once).
template <int CH, int NPX>
struct ChromaRaw {
  uint32_t raw[NPX / 4];
  uint32_t nxt, prv;
};

struct __attribute__ ((aligned (4))) u32x2u { uint32_t a, b; };
struct __attribute__ ((aligned (4))) u32x3u { uint32_t a, b, c; };

template <int CH, int NPX>
GSTAMD_HD void fast_chroma_load (const uint8_t *__restrict__ row, int cw, int x0, ChromaRaw<CH, NPX> &r)
{
  const int k0 = x0 >> 1;
  constexpr int NS = NPX / 2;
  const uint8_t *p = row + 2 * (size_t) k0;
  r.nxt = r.prv = 0;
  // the sample right of the span rides in the same (dword-aligned) load unless the span ends the row
  const bool inner = CH != CHROMA_H_NONE && k0 + NS < cw;
  if (NPX == 4) {
    if (inner) {
      const u32x2u m = *(const u32x2u *) p;
      r.raw[0] = m.a;
      r.nxt = m.b;
    } else {
      r.raw[0] = *(const uint32_t *) p;
    }
  } else if (NPX == 8) {
    if (inner) {
      const u32x3u m = *(const u32x3u *) p;
      r.raw[0] = m.a;
      r.raw[NPX >= 8 ? 1 : 0] = m.b;
      r.nxt = m.c;
    } else {
      const uint2 m = *(const uint2 *) p;
      r.raw[0] = m.x;
      r.raw[NPX >= 8 ? 1 : 0] = m.y;
    }
  } else {
    const uint4 m = *(const uint4 *) p;
    r.raw[0] = m.x;
    r.raw[NPX >= 16 ? 1 : 0] = m.y;
    r.raw[NPX >= 16 ? 2 : 0] = m.z;
    r.raw[NPX >= 16 ? 3 : 0] = m.w;
    if (inner)
      r.nxt = *(const uint16_t *) (p + 2 * NS);
  }
  if (CH != CHROMA_H_NONE && !inner)
    r.nxt = *(const uint16_t *) (row + 2 * (size_t) (cw - 1));
  if (CH == CHROMA_H_H2) {
    const int km = k0 > 0 ? k0 - 1 : 0;
    r.prv = *(const uint16_t *) (row + 2 * (size_t) km);
  }
}

template <int CH, int NPX>
GSTAMD_HD void fast_chroma_filter (const ChromaRaw<CH, NPX> &r, int x0, int w, uint32_t *out)
{
  constexpr int NS = NPX / 2;
  uint32_t S[NS + 2];
#pragma unroll
  for (int j = 0; j < NS; j++)
    S[j + 1] = bperm (r.raw[j >> 1], r.raw[j >> 1], (j & 1) ? 0x0c030c02u : 0x0c010c00u);
  S[NS + 1] = CH != CHROMA_H_NONE ? bperm (r.nxt, r.nxt, 0x0c010c00u) : S[NS];
  S[0] = CH == CHROMA_H_H2 ? bperm (r.prv, r.prv, 0x0c010c00u) : S[1];
#pragma unroll
  for (int j = 0; j < NS; j++) {
    const int xe = x0 + 2 * j, xo = xe + 1;
    uint32_t e = S[j + 1], o = S[j + 1];
    if (CH == CHROMA_H_H2_CS) {
      if (xo < w - 1)
        o = pk_shr<1> (S[j + 1] + S[j + 2] + 0x00010001u);
    } else if (CH == CHROMA_H_H2) {
      if (xo < w - 1)
        o = pk_shr<2> (3u * S[j + 1] + S[j + 2] + 0x00020002u);
      if (xe >= 2)
        e = pk_shr<2> (S[j] + 3u * S[j + 1] + 0x00020002u);
    }
    out[2 * j] = e;
    out[2 * j + 1] = o;
  }
}

// convert + store one line whose luma words are already in registers
template <int NPX, int ABL>
GSTAMD_HD void fast_emit_line (const FastParams &fp, uint8_t *__restrict__ dst, int dstride, int line, int x0, const uint32_t *yy,
    const uint32_t *ca, const uint32_t *cb, bool blend, int role, uint32_t usel, uint32_t vsel)
{
  static const uint32_t ysel[4] = {0x0c00000cu, 0x0c01010cu, 0x0c02020cu, 0x0c03030cu};
  uint32_t o[NPX];
#pragma unroll
  for (int i = 0; i < NPX; i++) {
    uint32_t c = ca[i];
    if (blend)
      c = role == 0 ? pk_shr<2> (3u * ca[i] + cb[i] + 0x00020002u) : pk_shr<2> (ca[i] + 3u * cb[i] + 0x00020002u);
    if (ABL == 1)
      o[i] = yy[i >> 2] ^ c;
    else
      o[i] = fast_pixel (fp, yy[i >> 2] ^ 0x80808080u, ysel[i & 3], c ^ 0x00800080u, usel, vsel);
  }
  uint8_t *d = dst + (size_t) line * dstride + 4 * (size_t) x0;
#pragma unroll
  for (int q = 0; q < NPX / 4; q++)
    store16_stream (d + 16 * q, o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
}

// pairs [p_begin, p_end) of the column x0 .. x0+NPX-1; pair p = lines (2p-1, 2p), chroma rows (p-1, p)
template <int CH, int NPX, int ABL>
GSTAMD_HD void fast_strip (const FastParams &fp, const Planes &pl, uint8_t *__restrict__ dst, int dstride, int x0, int p_begin,
    int p_end)
{
  const int w = fp.width, h = fp.height, cw = (w + 1) >> 1;
  const int n_crows = (h + 1) >> 1;
  const uint32_t sel_l0 = 0x0c00000cu, sel_l1 = 0x0c02020cu;
  const uint32_t usel = fp.u_first ? sel_l0 : sel_l1, vsel = fp.u_first ? sel_l1 : sel_l0;
  const uint8_t *yb = pl.p[0] + x0, *cbase = pl.p[1];
  const int ys = pl.stride[0], cs = pl.stride[1];

  uint32_t cprev[NPX], ccur[NPX];
  ChromaRaw<CH, NPX> craw;
  uint32_t y0[NPX / 4], y1[NPX / 4];
  {                                       // upper chroma row of the first pair
    ChromaRaw<CH, NPX> r0;
    const int r = p_begin > 0 ? p_begin - 1 : 0;
    fast_chroma_load<CH, NPX> (cbase + (size_t) r * cs, cw, x0, r0);
    fast_chroma_filter<CH, NPX> (r0, x0, w, cprev);
  }
  // prefetch pair p_begin
  {
    const int p = p_begin, l0 = 2 * p - 1, l1 = 2 * p;
    const int cr = p < n_crows ? p : n_crows - 1;
    fast_chroma_load<CH, NPX> (cbase + (size_t) cr * cs, cw, x0, craw);
    fast_load_y<NPX> (yb + (size_t) (l0 >= 0 ? l0 : 0) * ys, y0);
    fast_load_y<NPX> (yb + (size_t) (l1 < h ? l1 : h - 1) * ys, y1);
  }
  for (int p = p_begin; p < p_end; p++) {
    const int l0 = 2 * p - 1, l1 = 2 * p;
    const bool have0 = l0 >= 0, have1 = l1 < h;
    uint32_t cy0[NPX / 4], cy1[NPX / 4];
    ChromaRaw<CH, NPX> cr_now = craw;
#pragma unroll
    for (int q = 0; q < NPX / 4; q++) {
      cy0[q] = y0[q];
      cy1[q] = y1[q];
    }
    if (p + 1 < p_end) {                  // issue the next pair's loads before this pair's math
      const int pn = p + 1, n0 = 2 * pn - 1, n1 = 2 * pn;
      const int cr = pn < n_crows ? pn : n_crows - 1;
      fast_chroma_load<CH, NPX> (cbase + (size_t) cr * cs, cw, x0, craw);
      fast_load_y<NPX> (yb + (size_t) n0 * ys, y0);
      fast_load_y<NPX> (yb + (size_t) (n1 < h ? n1 : h - 1) * ys, y1);
    }
    if (have1)
      fast_chroma_filter<CH, NPX> (cr_now, x0, w, ccur);
    const bool blend = have0 && have1;
    if (have0)
      fast_emit_line<NPX, ABL> (fp, dst, dstride, l0, x0, cy0, cprev, ccur, blend, 0, usel, vsel);
    if (have1) {
      fast_emit_line<NPX, ABL> (fp, dst, dstride, l1, x0, cy1, blend ? cprev : ccur, ccur, blend, 1, usel, vsel);
#pragma unroll
      for (int i = 0; i < NPX; i++)
        cprev[i] = ccur[i];
    }
  }
}

// the shipped shape (8 pixels per lane); kept as the emulator's and the launcher's default entry
template <int CH>
GSTAMD_HD void fast_pair_body (const FastParams &fp, const Planes &pl, uint8_t *__restrict__ dst, int dstride, int span, int p)
{
  fast_pair_span<CH, 8, 0> (fp, pl, dst, dstride, span * 8, p);
}

}  // namespace gstamd
