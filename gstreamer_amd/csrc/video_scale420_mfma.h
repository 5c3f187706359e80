// video_scale420_mfma.h - the horizontal N-tap pass of the fused 4:2:0 scaler (video_scale420_fused.h) on the matrix cores.
//
// Why a matrix instruction in a byte stencil: a 16-tap horizontal filter at 4:1 is 8.6 multiply-adds per byte moved, and on the
// vector ALU its inner product (v_dot4) plus the LDS reads that feed it are what bound k_scale420_fused (VALU > 90 % busy in its
// horizontal phases, profiles/r02_c3_fused_stage_trace.log) - the pass is compute-bound there, not HBM-bound.  Written as
//     H[line][x] = sum_k  S[line][k] * T[k][x]            S: source bytes (one plane), T: the banded tap matrix
// a block of 16 lines x 16 outputs is v_mfma_i32_16x16x64_i8 over the 64-pixel chunks the 16 outputs' windows touch (three at
// 4:1 with 16 taps: 4 * 15 + 16 = 76 pixels at any alignment).  Integer, exact: the 32-bit accumulator holds the true sum, the
// reference's 16-bit wrap ((int16) acc >> 6, video-orc.orc:2388-2480) is applied to it afterwards as in the dot4 form.
//
// Operand layout (A and B use the SAME lane -> K mapping, so only the C map matters: row = 4 * (lane >> 4) + reg, col = lane & 15):
//   A  lane l: line m = l & 15 of the block, 16 consecutive source pixels 64 c + 16 (l >> 4) + j of chunk c (bytes XOR 0x80)
//        luma: the frame's bytes as they are (one 16-byte load per lane, no LDS); chroma: the lane upsamples ITS line's 16
//        pixels from the two chroma rows the line blends (h420_filter_raw2 + h420_blend_store on registers)
//   B  lane l: output n = l & 15 of the block, the int8 taps of the same 16 pixels (zero outside the output's window) - a
//        table made on the host, one 16-byte load per lane and chunk
//   C  lane l, reg r: line 4 (l >> 4) + r, output l & 15: the four lines of ONE line group of one output - exactly the ring word
//        of video_scale420_fused.h (byte r = line 4g-1+r), so a line block of 16 lines starts at line 16 kb - 1 = group 4 kb.
// The vertical pass is fused_vrow, unchanged.
//
// Applies when every block of 16 outputs finds its windows inside the three chunks bg + d0 .. bg + d0 + 2 (one new chunk per
// block: 64 source pixels per 16 outputs, i.e. 4:1); the host checks that on the offsets and builds T accordingly.  Other ratios
// take k_scale420_fused.
#pragma once
#include "video_scale420_fused.h"

#define GSTAMD_MFMA_CHUNKS 3

namespace gstamd {

struct Mfma420Params {
  Fused420Params f;             // source planes (f.h), vertical tables, chunking; f.h.tile_w is a multiple of 16
  const uint4 *btab;            // [n_blocks][GSTAMD_MFMA_CHUNKS][64] B operands
  int d0;                       // block bg reads the chunks bg + d0 .. bg + d0 + 2
  int n_blocks;                 // ceil (out_w / 16)
};

// row pointers of one lane for one line block: its line's luma row and the heavy / light chroma rows
struct Mfma420Rows {
  const uint8_t *y;
  const uint8_t *ch, *cl;       // planar: U rows (V rows at + vdelta); semi-planar: the interleaved rows
  ptrdiff_t vdelta;
};

GSTAMD_HD void mfma_rows (const H420RegParams &p, int line, Mfma420Rows &r)
{
  const int yl = line < 0 ? 0 : (line >= p.height ? p.height - 1 : line);
  int rh, rl;
  h420r_rows (p.crow_lo, p.crow_hi, line, &rh, &rl);
  r.y = p.y + (ptrdiff_t) yl * p.ystride;
  r.ch = p.c0 + (ptrdiff_t) rh * p.cstride;
  r.cl = p.c0 + (ptrdiff_t) rl * p.cstride;
  r.vdelta = p.c1 - p.c0;
}

// raw chroma samples k0 .. k0 + 7 (+ clamped neighbours) of one row
template <int SEMI>
GSTAMD_HD void mfma_load_raw (const uint8_t *row, ptrdiff_t vdelta, int k0, int cw, H420Raw &r)
{
  const int km = k0 > 0 ? k0 - 1 : 0, kp = k0 + 8 < cw ? k0 + 8 : cw - 1;
  if (SEMI) {
    const uint4 m = *(const uint4 *) (row + (uint32_t) (2 * k0));
    r.u0 = m.x, r.u1 = m.y, r.v0 = m.z, r.v1 = m.w;
    r.um = *(const uint16_t *) (row + (uint32_t) (2 * km));
    r.up = *(const uint16_t *) (row + (uint32_t) (2 * kp));
    r.vm = r.vp = 0;
  } else {
    const uint8_t *rv = row + vdelta;
    const uint2 mu = *(const uint2 *) (row + (uint32_t) k0), mv = *(const uint2 *) (rv + (uint32_t) k0);
    r.u0 = mu.x, r.u1 = mu.y, r.v0 = mv.x, r.v1 = mv.y;
    r.um = row[(uint32_t) km], r.up = row[(uint32_t) kp], r.vm = rv[(uint32_t) km], r.vp = rv[(uint32_t) kp];
  }
}

// everything a lane loads for one chunk
struct Mfma420Loads {
  uint4 luma;
  H420Raw h, l;
};

// chunk c of the lane's line: pixels 64 c + 16 kg .. + 15 (clamped into the picture: chunks that stick out meet zero taps only)
template <int SEMI>
GSTAMD_HD void mfma_request (const H420RegParams &p, const Mfma420Rows &r, int c, int kg, Mfma420Loads &q)
{
  int x0 = 64 * c + 16 * kg;
  x0 = x0 < 0 ? 0 : (x0 + 16 > p.width ? p.width - 16 : x0);
  q.luma = *(const uint4 *) (r.y + (uint32_t) x0);
  mfma_load_raw<SEMI> (r.ch, r.vdelta, x0 >> 1, p.width >> 1, q.h);
  mfma_load_raw<SEMI> (r.cl, r.vdelta, x0 >> 1, p.width >> 1, q.l);
}

// A operands of the three planes from the loaded samples
struct Mfma420A {
  uint4 y, u, v;
};

template <int CH, int SEMI>
GSTAMD_HD void mfma_make_a (const H420RegParams &p, const Mfma420Loads &q, Mfma420A &a)
{
  uint32_t fh[8], fl[8];
  struct __attribute__ ((aligned (16))) W4 { uint32_t w[4]; } pu, pv;
  h420_filter_raw2<CH> (SEMI != 0, p.u_first != 0, q.h, fh);
  h420_filter_raw2<CH> (SEMI != 0, p.u_first != 0, q.l, fl);
  h420_blend_store (fh, fl, pu.w, pv.w);                /* (3 heavy + light + 2) >> 2, pixel order, XOR 0x80 */
  a.y = gstamd_make_uint4 (q.luma.x ^ 0x80808080u, q.luma.y ^ 0x80808080u, q.luma.z ^ 0x80808080u, q.luma.w ^ 0x80808080u);
  a.u = gstamd_make_uint4 (pu.w[0], pu.w[1], pu.w[2], pu.w[3]);
  a.v = gstamd_make_uint4 (pv.w[0], pv.w[1], pv.w[2], pv.w[3]);
}

// four accumulators (the four lines of a group) -> the ring word: ((int16) acc >> 6) clamped to a byte each, XOR 0x80.  The
// accumulators start at 128 * 64 + 32 (the XOR's 128 * sum (taps) and the rounding), so this is shift, saturate, pack.
GSTAMD_HD uint32_t mfma_group_word (int a0, int a1, int a2, int a3)
{
#ifdef __HIPCC__
  typedef short s2 __attribute__ ((ext_vector_type (2)));
  const uint32_t p01 = bperm ((uint32_t) a1, (uint32_t) a0, 0x05040100u), p23 = bperm ((uint32_t) a3, (uint32_t) a2, 0x05040100u);
  const s2 s01 = __builtin_bit_cast (s2, p01) >> (short) 6, s23 = __builtin_bit_cast (s2, p23) >> (short) 6;
  const uint32_t b01 = sat_pk_u8 (__builtin_bit_cast (uint32_t, s01)), b23 = sat_pk_u8 (__builtin_bit_cast (uint32_t, s23));
  return (b01 | (b23 << 16)) ^ 0x80808080u;
#else
  return (h420r_finish (a0) | (h420r_finish (a1) << 8) | (h420r_finish (a2) << 16) | (h420r_finish (a3) << 24)) ^ 0x80808080u;
#endif
}

}  // namespace gstamd
