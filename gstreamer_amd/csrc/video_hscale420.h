// video_hscale420.h - horizontal N-tap pass straight from a 2x horizontally subsampled planar / semi-planar frame
// (I420, YV12, NV12, NV21, Y42B, NV16 ...), the front of BASELINE C3 (8K I420 -> 1080p Lanczos).
//
// A wave owns one tile of <= 256 outputs and walks down `rows_per_wave` source lines of it.  Per line every lane
// stages the 16 source pixels [xa + 16 * lane, + 16) as 16 bytes of each of the three byte planes Y / U / V (XOR 0x80)
// in LDS with ONE 16-byte luma load and, only when the line needs a chroma row the lane does not hold yet, one 8-byte
// load per chroma plane (16 bytes for interleaved UV) + the neighbour samples; then the lanes run the byte-dot-product
// filter of video_scale_fast.h (hscale_dot4_lane) on the planes with their taps kept in registers across the lines.
//
// Chroma upsampling in BYTE lanes with v_lerp_u8 (per byte (a + b + (c & 1)) >> 1), exact because
//     (a + b + 1) >> 1           = lerp (a, b, 1)                       video_chroma_up_h2_cs_u8, video-chroma.c:687-699
//     (3 a + b + 2) >> 2         = lerp (a, lerp (a, b, 0), 1)           video_chroma_up_h2_u8 / up_v2_u8, :277-327
// (second identity: with m = (a + b) >> 1, 2 a + 2 m + 2 equals 3 a + b + 2 when a + b is even and 3 a + b + 1 when it is
// odd; in the odd case 3 a + b is odd too, so adding 1 or 2 before the >> 2 gives the same quotient).
// The reference filters horizontally first, then blends the two chroma rows 3:1 (do_upsample_lines, video-converter.c:2991);
// a lane keeps the h-filtered form of two chroma rows (even-pixel bytes and odd-pixel bytes) and reuses them for the next line.
#pragma once
#include "video_scale_fast.h"

// bytes per LDS plane: 1024 staged + the aligned filter window's run past the span, a multiple of 16
#define GSTAMD_H420_PLANE_BYTES 1056

namespace gstamd {

GSTAMD_HD uint32_t lerp_u8 (uint32_t a, uint32_t b, uint32_t c)
{
#ifdef __HIPCC__
  return __builtin_amdgcn_lerp (a, b, c);
#else
  uint32_t r = 0;
  for (int i = 0; i < 4; i++) {
    const uint32_t x = (a >> (8 * i)) & 0xffu, y = (b >> (8 * i)) & 0xffu, z = (c >> (8 * i)) & 1u;
    r |= ((x + y + z) >> 1) << (8 * i);
  }
  return r;
#endif
}

// ({hi, lo} >> 8 n) as 32 bits: v_alignbyte_b32
GSTAMD_HD uint32_t align_bytes (uint32_t hi, uint32_t lo, int n)
{
#ifdef __HIPCC__
  return __builtin_amdgcn_alignbyte (hi, lo, (uint32_t) n);
#else
  return (uint32_t) ((((uint64_t) hi << 32) | lo) >> (8 * n));
#endif
}

// (3 a + b + 2) >> 2 per byte
GSTAMD_HD uint32_t blend31_u8 (uint32_t a, uint32_t b) { return lerp_u8 (a, lerp_u8 (a, b, 0u), 0x01010101u); }

// per-lane state carried down the lines of a tile: the h-filtered form of two chroma rows of the lane's 16 pixels
// ([0..1] U of the even pixels, [2..3] U of the odd pixels, [4..7] the same for V), and what has been requested for the NEXT line
// while the current one is filtered (software prefetch: luma, and the one chroma row the next line needs and the lane lacks)
struct H420Raw {
  uint32_t u0, u1, v0, v1, um, up, vm, vp;
};
struct H420State {
  uint32_t f[2][8];
  int id[2];            // chroma row each set holds (wave-uniform)
  uint4 luma;           // 16 luma bytes of the line about to be staged
  H420Raw raw;          // unfiltered chroma row raw_row, to be installed when that line is staged
  int raw_row;
};
#define GSTAMD_H420_EMPTY (-0x40000000)

// heavy (weight 3) and light chroma row of source line y, wave-uniform
GSTAMD_HD void h420_rows_of_line (const SrcFront &src, int y, int &rh, int &rl)
{
  if (src.f.chroma_v2) {
    const int e0 = src.vpair[2 * y], ra = vpair_row (e0), rb = src.vpair[2 * y + 1];
    rh = vpair_role (e0) == 0 ? ra : rb;
    rl = vpair_role (e0) == 0 ? rb : ra;
  } else {
    rh = rl = y >> src.f.h_sub;
  }
}

// chroma row `crow`, samples k0 .. k0+7 (k0 % 8 == 0) and the clamped neighbours k0-1, k0+8: loads only
GSTAMD_HD void h420_load_raw (const FrontParams &f, const Planes &pl, int crow, int k0, int cw, H420Raw &r)
{
  const int km = k0 > 0 ? k0 - 1 : 0, kp = k0 + 8 < cw ? k0 + 8 : cw - 1;
  if (f.kind == UNPACK_SEMI) {
    const uint8_t *row = pl.p[1] + (ptrdiff_t) crow * pl.stride[1];
    const uint4 m = *(const uint4 *) (row + 2 * k0);
    r.u0 = m.x, r.u1 = m.y, r.v0 = m.z, r.v1 = m.w;               // still interleaved, see h420_filter_raw
    r.um = *(const uint16_t *) (row + 2 * km);
    r.up = *(const uint16_t *) (row + 2 * kp);
    r.vm = r.vp = 0;
  } else {
    const uint8_t *ru = pl.p[f.u_plane] + (ptrdiff_t) crow * pl.stride[f.u_plane];
    const uint8_t *rv = pl.p[f.v_plane] + (ptrdiff_t) crow * pl.stride[f.v_plane];
    const uint2 mu = *(const uint2 *) (ru + k0), mv = *(const uint2 *) (rv + k0);
    r.u0 = mu.x, r.u1 = mu.y, r.v0 = mv.x, r.v1 = mv.y;
    r.um = ru[km], r.up = ru[kp], r.vm = rv[km], r.vp = rv[kp];
  }
}

template <int CH>
GSTAMD_HD void h420_filter_raw2 (bool semi, bool u_first, const H420Raw &r, uint32_t *o)
{
  uint32_t u0, u1, v0, v1, um, up, vm, vp;
  if (semi) {
    // first bytes of the pairs / second bytes of the pairs
    const uint32_t a0 = bperm (r.u1, r.u0, 0x06040200u), a1 = bperm (r.v1, r.v0, 0x06040200u);
    const uint32_t b0 = bperm (r.u1, r.u0, 0x07050301u), b1 = bperm (r.v1, r.v0, 0x07050301u);
    if (u_first) {              // NV12 / NV16: U first
      u0 = a0, u1 = a1, v0 = b0, v1 = b1;
      um = r.um & 0xffu, vm = r.um >> 8, up = r.up & 0xffu, vp = r.up >> 8;
    } else {
      v0 = a0, v1 = a1, u0 = b0, u1 = b1;
      vm = r.um & 0xffu, um = r.um >> 8, vp = r.up & 0xffu, up = r.up >> 8;
    }
  } else {
    u0 = r.u0, u1 = r.u1, v0 = r.v0, v1 = r.v1, um = r.um, up = r.up, vm = r.vm, vp = r.vp;
  }
  o[0] = u0, o[1] = u1, o[4] = v0, o[5] = v1;
  if (CH == CHROMA_H_H2_CS) {
    // odd pixel 2 j + 1: (c[j] + c[j+1] + 1) >> 1; the last pixel of the line keeps c[j] (clamped neighbour: (2 c + 1) >> 1 = c)
    o[2] = lerp_u8 (u0, align_bytes (u1, u0, 1), 0x01010101u);
    o[3] = lerp_u8 (u1, align_bytes (up, u1, 1), 0x01010101u);
    o[6] = lerp_u8 (v0, align_bytes (v1, v0, 1), 0x01010101u);
    o[7] = lerp_u8 (v1, align_bytes (vp, v1, 1), 0x01010101u);
  } else if (CH == CHROMA_H_H2) {
    // even pixel 2 j: (c[j-1] + 3 c[j] + 2) >> 2, odd pixel: (3 c[j] + c[j+1] + 2) >> 2; pixels 0 and w-1 keep c[j] (clamped)
    o[0] = blend31_u8 (u0, align_bytes (u0, um << 24, 3));
    o[1] = blend31_u8 (u1, align_bytes (u1, u0, 3));
    o[2] = blend31_u8 (u0, align_bytes (u1, u0, 1));
    o[3] = blend31_u8 (u1, align_bytes (up, u1, 1));
    o[4] = blend31_u8 (v0, align_bytes (v0, vm << 24, 3));
    o[5] = blend31_u8 (v1, align_bytes (v1, v0, 3));
    o[6] = blend31_u8 (v0, align_bytes (v1, v0, 1));
    o[7] = blend31_u8 (v1, align_bytes (vp, v1, 1));
  } else {
    o[2] = u0, o[3] = u1, o[6] = v0, o[7] = v1;
  }
}

template <int CH>
GSTAMD_HD void h420_filter_raw (const FrontParams &f, const H420Raw &r, uint32_t *o)
{
  h420_filter_raw2<CH> (f.kind == UNPACK_SEMI, f.u_plane != 0, r, o);
}

// 3:1 blend of the heavy and the light chroma row, pixel order, XOR 0x80 -> 16 bytes of the U plane and of the V plane
GSTAMD_HD void h420_blend_store (const uint32_t *h, const uint32_t *l, uint32_t *pu16, uint32_t *pv16)
{
  uint32_t r[8];
#pragma unroll
  for (int i = 0; i < 8; i++)
    r[i] = blend31_u8 (h[i], l[i]);
  // (e0 o0 e1 o1) (e2 o2 e3 o3) of each even / odd register pair
  uint4 *du = (uint4 *) pu16, *dv = (uint4 *) pv16;
  *du = gstamd_make_uint4 (bperm (r[2], r[0], 0x05010400u) ^ 0x80808080u, bperm (r[2], r[0], 0x07030602u) ^ 0x80808080u,
      bperm (r[3], r[1], 0x05010400u) ^ 0x80808080u, bperm (r[3], r[1], 0x07030602u) ^ 0x80808080u);
  *dv = gstamd_make_uint4 (bperm (r[6], r[4], 0x05010400u) ^ 0x80808080u, bperm (r[6], r[4], 0x07030602u) ^ 0x80808080u,
      bperm (r[7], r[5], 0x05010400u) ^ 0x80808080u, bperm (r[7], r[5], 0x07030602u) ^ 0x80808080u);
}

// filter a raw row into the set that does not hold row `keep`
template <int CH>
GSTAMD_HD void h420_install (const FrontParams &f, H420State &s, const H420Raw &r, int row, int keep, bool active)
{
  if (s.id[0] == keep) {
    if (active)
      h420_filter_raw<CH> (f, r, s.f[1]);
    s.id[1] = row;
  } else {
    if (active)
      h420_filter_raw<CH> (f, r, s.f[0]);
    s.id[0] = row;
  }
}

// requests for line y: its luma, and the first chroma row it needs that the lane does not hold
GSTAMD_HD void h420_request (const SrcFront &src, H420State &s, int x0, int y, bool active)
{
  int rh, rl;
  h420_rows_of_line (src, y, rh, rl);
  const int missing = (s.id[0] != rh && s.id[1] != rh) ? rh : ((s.id[0] != rl && s.id[1] != rl) ? rl : GSTAMD_H420_EMPTY);
  s.raw_row = missing;
  if (!active)
    return;
  s.luma = *(const uint4 *) (src.pl.p[0] + (ptrdiff_t) y * src.pl.stride[0] + x0);
  if (missing != GSTAMD_H420_EMPTY)
    h420_load_raw (src.f, src.pl, missing, x0 >> 1, (src.f.width + 1) >> 1, s.raw);
}

// before the first line of a wave
GSTAMD_HD void h420_begin (const SrcFront &src, H420State &s, int xa, int x_hi, int y0, int lane)
{
  const int x0 = xa + 16 * lane;
  s.id[0] = s.id[1] = GSTAMD_H420_EMPTY;
  h420_request (src, s, x0, y0, x0 < x_hi);
}

// source line y: the lane's 16 pixels [xa + 16 lane, + 16) into the byte planes (nothing for lanes past x_hi), then the
// requests for line y_next (< 0: none)
template <int CH>
GSTAMD_HD void h420_stage_line (const SrcFront &src, H420State &s, uint32_t *py, uint32_t *pu, uint32_t *pv, int xa, int x_hi, int y, int y_next,
    int lane)
{
  const FrontParams &f = src.f;
  int rh, rl;
  h420_rows_of_line (src, y, rh, rl);
  const int x0 = xa + 16 * lane;
  const bool active = x0 < x_hi;
  const int cw = (f.width + 1) >> 1, k0 = x0 >> 1;
  if (s.raw_row != GSTAMD_H420_EMPTY) {
    h420_install<CH> (f, s, s.raw, s.raw_row, s.raw_row == rh ? rl : rh, active);
    s.raw_row = GSTAMD_H420_EMPTY;
  }
  // a second missing row (first line of a wave, irregular pair tables): fetched here and now
  if (s.id[0] != rh && s.id[1] != rh) {
    H420Raw r;
    if (active)
      h420_load_raw (f, src.pl, rh, k0, cw, r);
    h420_install<CH> (f, s, r, rh, rl, active);
  }
  if (s.id[0] != rl && s.id[1] != rl) {
    H420Raw r;
    if (active)
      h420_load_raw (f, src.pl, rl, k0, cw, r);
    h420_install<CH> (f, s, r, rl, rh, active);
  }
  if (active) {
    const int w0 = (x0 - xa) >> 2;
    *(uint4 *) (py + w0) = gstamd_make_uint4 (s.luma.x ^ 0x80808080u, s.luma.y ^ 0x80808080u, s.luma.z ^ 0x80808080u, s.luma.w ^ 0x80808080u);
    const bool h0 = s.id[0] == rh, l0 = s.id[0] == rl;
    if (h0 && l0)
      h420_blend_store (s.f[0], s.f[0], pu + w0, pv + w0);
    else if (h0)
      h420_blend_store (s.f[0], s.f[1], pu + w0, pv + w0);
    else if (l0)
      h420_blend_store (s.f[1], s.f[0], pu + w0, pv + w0);
    else
      h420_blend_store (s.f[1], s.f[1], pu + w0, pv + w0);
  }
  if (y_next >= 0)
    h420_request (src, s, x0, y_next, active);
}

// the requests alone (for a loop that issues them after the filter phase's stores)
GSTAMD_HD void h420_request_line (const SrcFront &src, H420State &s, int xa, int x_hi, int y_next, int lane)
{
  const int x0 = xa + 16 * lane;
  h420_request (src, s, x0, y_next, x0 < x_hi);
}

GSTAMD_HD void h420_stage_line_any (const SrcFront &src, H420State &s, uint32_t *py, uint32_t *pu, uint32_t *pv, int xa, int x_hi, int y, int y_next,
    int lane)
{
  if (src.f.chroma_h == CHROMA_H_H2_CS)
    h420_stage_line<CHROMA_H_H2_CS> (src, s, py, pu, pv, xa, x_hi, y, y_next, lane);
  else if (src.f.chroma_h == CHROMA_H_H2)
    h420_stage_line<CHROMA_H_H2> (src, s, py, pu, pv, xa, x_hi, y, y_next, lane);
  else
    h420_stage_line<CHROMA_H_NONE> (src, s, py, pu, pv, xa, x_hi, y, y_next, lane);
}

// ------------------------------------------------------------------------------------------------
// The same pass for the regular case, lean: 4:2:0 source whose chroma line pairing is the closed form (lines 2u-1, 2u blend
// chroma rows u-1 and u, clamped into [crow_lo, crow_hi]; what do_upsample_lines produces when every line is consumed in
// order - checked by the launcher against the planner's table), AYUV intermediate as destination, filter window of NW words.
// A wave walks line PAIRS: both lines of a pair blend the same two h-filtered rows P and Q (3:1 and 1:3), the next pair
// replaces the older of the two, so with two pairs per loop turn every register has a fixed role - no table reads, no
// row bookkeeping, two lines' worth of independent loads, LDS traffic and dot products per synchronisation.
// ------------------------------------------------------------------------------------------------
struct H420RegParams {
  const uint8_t *y, *c0, *c1;     // luma; planar: U plane, V plane; semi-planar: the interleaved plane (c1 unused)
  int ystride, cstride;
  int width, height;              // source picture in pixels / lines, width % 16 == 0
  int semi, u_first;
  int crow_lo, crow_hi;           // chroma rows the upsampler may touch (frame rows around a crop)
  const uint32_t *offset, *tapw;  // ScaleDev::offset / tapw
  int nw4;
  uint8_t *dst;                   // AYUV image, out_w x (height + 1): one spare row takes the stores of lines outside the picture
  int dstride;
  int out_w, tile_w, lines_per_wave;      // even; wave b covers lines [b * lpw - 1, (b + 1) * lpw - 1)
};

struct H420Pair {                 // one line pair in flight: luma of lines 2u-1 and 2u, raw chroma row u
  uint4 la, lb;
  H420Raw raw;
};

// closed form of the pairing: heavy and light chroma row of line y
#ifdef __HIPCC__
#define GSTAMD_H420_HOSTDEV __host__ __device__ inline
#else
#define GSTAMD_H420_HOSTDEV inline
#endif
GSTAMD_H420_HOSTDEV void h420r_rows (int crow_lo, int crow_hi, int y, int *heavy, int *light)
{
  const int u = (y + 1) >> 1;
  const int ra = u - 1 < crow_lo ? crow_lo : (u - 1 > crow_hi ? crow_hi : u - 1), rb = u < crow_lo ? crow_lo : (u > crow_hi ? crow_hi : u);
  *heavy = (y & 1) ? ra : rb;
  *light = (y & 1) ? rb : ra;
}

GSTAMD_HD int h420r_crow (const H420RegParams &p, int r) { return r < p.crow_lo ? p.crow_lo : (r > p.crow_hi ? p.crow_hi : r); }

template <int SEMI>
GSTAMD_HD void h420r_load_raw (const H420RegParams &p, int crow, int k0, H420Raw &r)
{
  const int cw = p.width >> 1, km = k0 > 0 ? k0 - 1 : 0, kp = k0 + 8 < cw ? k0 + 8 : cw - 1;
  if (SEMI) {
    const uint8_t *row = p.c0 + (ptrdiff_t) crow * p.cstride;
    const uint4 m = *(const uint4 *) (row + (uint32_t) (2 * k0));
    r.u0 = m.x, r.u1 = m.y, r.v0 = m.z, r.v1 = m.w;
    r.um = *(const uint16_t *) (row + (uint32_t) (2 * km));
    r.up = *(const uint16_t *) (row + (uint32_t) (2 * kp));
    r.vm = r.vp = 0;
  } else {
    const uint8_t *ru = p.c0 + (ptrdiff_t) crow * p.cstride, *rv = p.c1 + (ptrdiff_t) crow * p.cstride;
    const uint2 mu = *(const uint2 *) (ru + (uint32_t) k0), mv = *(const uint2 *) (rv + (uint32_t) k0);
    r.u0 = mu.x, r.u1 = mu.y, r.v0 = mv.x, r.v1 = mv.y;
    r.um = ru[(uint32_t) km], r.up = ru[(uint32_t) kp], r.vm = rv[(uint32_t) km], r.vp = rv[(uint32_t) kp];
  }
}

// loads of pair u (lines 2u-1, 2u; line numbers clamped into the picture, the caller skips lines outside it)
template <int SEMI>
GSTAMD_HD void h420r_request (const H420RegParams &p, int u, int x0, H420Pair &q)
{
  const int ya = 2 * u - 1 < 0 ? 0 : 2 * u - 1, yb = 2 * u < p.height ? 2 * u : p.height - 1;
  q.la = *(const uint4 *) (p.y + (ptrdiff_t) ya * p.ystride + (uint32_t) x0);
  q.lb = *(const uint4 *) (p.y + (ptrdiff_t) yb * p.ystride + (uint32_t) x0);
  h420r_load_raw<SEMI> (p, h420r_crow (p, u), x0 >> 1, q.raw);
}

GSTAMD_HD void h420r_stage_luma (const uint4 &l, uint32_t *py16)
{
  *(uint4 *) py16 = gstamd_make_uint4 (l.x ^ 0x80808080u, l.y ^ 0x80808080u, l.z ^ 0x80808080u, l.w ^ 0x80808080u);
}

// (int16) acc >> 6 (the ORC program's 16-bit accumulator, shrsw 6), clamped to a byte; the rounding 32 is in the accumulator's start value
GSTAMD_HD uint32_t h420r_finish (int acc)
{
#ifdef __HIPCC__
  int v = __builtin_amdgcn_sbfe (acc, 6, 10);
  asm volatile ("" : "+v" (v));                  // see lq_round: keep shift and clamp apart
  return (uint32_t) (v < 0 ? 0 : (v > 255 ? 255 : v));
#else
  const int v = ((int) (int16_t) (uint16_t) acc) >> 6;
  return (uint32_t) (v < 0 ? 0 : (v > 255 ? 255 : v));
#endif
}

// LDS word read that the compiler leaves alone: one ds_read_b32 with an immediate offset from the output's base register
// (merged ds_read2_b32 pairs cannot reach across the 1056-byte planes and cost an address add each)
GSTAMD_HD uint32_t h420r_lds (const uint32_t *p)
{
#ifdef __HIPCC__
  return *(const volatile __attribute__ ((address_space (3))) uint32_t *) p;
#else
  return *p;
#endif
}

// outputs t0 + lane + 64 i of one line from its byte planes; AYUV words, plain stores (the vertical pass reads them next).
// No predicates: lanes past the tile's end repeat its last output (same taps, same bytes, same address) so that every line costs
// exactly four stores - the waits on the prefetched loads of the next pair can then count past them instead of draining them.
template <int NW>
GSTAMD_HD void h420r_filter_line (const uint32_t *line, const Dot4Taps<NW> &ft, uint8_t *drow, int t0, int t1, int lane)
{
  const int pw = GSTAMD_H420_PLANE_BYTES / 4;
  // two outputs per round: their 6 * NW LDS reads go out back to back before the first dot product needs one (the LDS reaches its
  // rate only with many reads in flight per wait)
#pragma unroll
  for (int i0 = 0; i0 < 4; i0 += 2) {
    uint32_t wy[2][NW], wu[2][NW], wv[2][NW];
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const uint32_t *b = line + ft.w0[i0 + j];
#pragma unroll
      for (int k = 0; k < NW; k++) {
        wy[j][k] = h420r_lds (b + k);
        wu[j][k] = h420r_lds (b + pw + k);
        wv[j][k] = h420r_lds (b + 2 * pw + k);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int i = i0 + j, x = t0 + lane + 64 * i, xc = x < t1 ? x : t1 - 1;
      int ay = 128 * 64 + 32, au = 128 * 64 + 32, av = 128 * 64 + 32;
#pragma unroll
      for (int k = 0; k < NW; k++) {
        const uint32_t t = ft.t[i][k];
        ay = dot4_i8 (wy[j][k], t, ay);
        au = dot4_i8 (wu[j][k], t, au);
        av = dot4_i8 (wv[j][k], t, av);
      }
      *(uint32_t *) (drow + (uint32_t) (4 * xc)) = 0xffu | (h420r_finish (ay) << 8) | (h420r_finish (au) << 16) | (h420r_finish (av) << 24);
    }
  }
}

template <int NW>
GSTAMD_HD void h420r_fetch_taps (const H420RegParams &p, int xa, int t0, int t1, int lane, Dot4Taps<NW> &ft)
{
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int x = t0 + lane + 64 * i;
    const int xc = x < t1 ? x : t1 - 1;
    ft.w0[i] = ((int) p.offset[xc] - xa) >> 2;
    const uint32_t *tw = p.tapw + (size_t) xc * p.nw4;
#pragma unroll
    for (int k = 0; k < NW; k++)
      ft.t[i][k] = tw[k];
  }
}

// LDS words of one line (three planes) in the two-line layout
#define GSTAMD_H420_LINE_WORDS (3 * GSTAMD_H420_PLANE_BYTES / 4)

// stage pair u from `q`: install its chroma row into `fresh` (the set that held the older row), then line 2u-1 = 3 * older + newer,
// line 2u = older + 3 * newer.  `older` is the other set.
template <int CH, int SEMI>
GSTAMD_HD void h420r_stage_pair (const H420RegParams &p, const H420Pair &q, const uint32_t *older, uint32_t *fresh, uint32_t *lds, int w0)
{
  const int pw = GSTAMD_H420_PLANE_BYTES / 4;
  h420_filter_raw2<CH> (SEMI != 0, p.u_first != 0, q.raw, fresh);
  h420r_stage_luma (q.la, lds + w0);
  h420_blend_store (older, fresh, lds + pw + w0, lds + 2 * pw + w0);
  h420r_stage_luma (q.lb, lds + GSTAMD_H420_LINE_WORDS + w0);
  h420_blend_store (fresh, older, lds + GSTAMD_H420_LINE_WORDS + pw + w0, lds + GSTAMD_H420_LINE_WORDS + 2 * pw + w0);
}

template <int NW>
GSTAMD_HD void h420r_filter_pair (const H420RegParams &p, const uint32_t *lds, const Dot4Taps<NW> &ft, int u, int t0, int t1, int lane)
{
  const int ya = 2 * u - 1, yb = 2 * u;
  // lines -1 and `height` (first / last pair) land in the spare row after the image
  h420r_filter_line<NW> (lds, ft, p.dst + (ptrdiff_t) (ya >= 0 ? ya : p.height) * p.dstride, t0, t1, lane);
  h420r_filter_line<NW> (lds + GSTAMD_H420_LINE_WORDS, ft, p.dst + (ptrdiff_t) (yb < p.height ? yb : p.height) * p.dstride, t0, t1, lane);
}

// source span of the outputs [t0, t1)
GSTAMD_HD void h420r_span (const H420RegParams &p, int n_taps, int t0, int t1, int *x_lo, int *x_hi)
{
  *x_lo = (int) p.offset[t0];
  *x_hi = (int) p.offset[t1 - 1] + n_taps;
}

}  // namespace gstamd
