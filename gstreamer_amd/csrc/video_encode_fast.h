// video_encode_fast.h - the encoder-facing mirror of video_fast.h: 4-byte RGB -> 4:2:0 YUV (NV12 / NV21 / I420 / YV12), same
// size, in one kernel.  The generic path renders an AYUV image into HBM and packs it in a second kernel; here one lane
// owns a 4 x 2 pixel block: two 16-byte loads (+ the pixel left of the block for the cosited chroma filter), the colour
// matrix as byte dot products, chroma downsampling in packed 16-bit arithmetic, a 4-byte luma store per line and the
// block's two chroma samples.
//
// Reference semantics (bit-exact): unpack (video-orc.orc:334-411, byte permutations) -> video_converter_matrix8_table
// (video-converter.c:1187: the three rows summed in one int64; the planner picks it only for matrices that cannot
// clip (is_no_clip_matrix :1252), i.e. every row sum s is in [0, 65535] for every pixel, so the packed sum has no carries
// between the fields and each output is simply s >> 8) -> chroma down: video_orc_chroma_down_v2_u8 (avgub of the line
// pair), then video_orc_chroma_down_h2_u8 (avgub of the pixel pair) or video_chroma_down_h2_cs_u8 (video-chroma.c:740-762:
// 3-1 at pixel 0, 1-2-1 inside, 1-3 for the last even pixels) -> pack_NV12 / pack_NV21 / pack_planar_420.
#pragma once
#include "video_fast.h"
#include "video_pack.h"
#include <string.h>

namespace gstamd {

struct Enc420Params {
  int width, height;          // picture size, width % 4 == 0
  uint32_t cpos[3], cneg[3];  // per matrix row: |coefficient| of the positive / negative entries at the SOURCE byte of R, G, B
  int off[3];                 // im[k][3]
  int neg_mask;               // bit k: row k has negative coefficients (its second dot product is needed)
  int down_h, down_v;         // PackPlanarParams::down_h / down_v
  int u_first;                // semi-planar: 1 = U,V (NV12), 0 = V,U (NV21)
  int u_plane, v_plane;       // planar: destination plane of U and V
};

inline Enc420Params make_enc420_params (const VideoPlan &p)
{
  Enc420Params ep;
  memset (&ep, 0, sizeof (ep));
  ep.width = p.out_info.width;
  ep.height = p.out_info.height;
  for (int k = 0; k < 3; k++) {
    for (int j = 0; j < 3; j++) {
      const int cf = p.matrix.im[k][j], sh = 8 * p.fin->pos[j + 1];
      if (cf >= 0)
        ep.cpos[k] |= (uint32_t) cf << sh;
      else {
        ep.cneg[k] |= (uint32_t) (-cf) << sh;
        ep.neg_mask |= 1 << k;
      }
    }
    ep.off[k] = p.matrix.im[k][3];
  }
  ep.down_h = p.pack.down_h;
  ep.down_v = p.pack.down_v;
  ep.u_first = p.fout->u_plane;           /* semi-planar: FormatDesc::u_plane is the U-first flag */
  ep.u_plane = p.fout->u_plane;
  ep.v_plane = p.fout->v_plane;
  return ep;
}

// sum of the four byte products + c: v_dot4_u32_u8
GSTAMD_HD uint32_t dot4_u8 (uint32_t a, uint32_t b, uint32_t c)
{
#ifdef __HIPCC__
  return __builtin_amdgcn_udot4 (a, b, c, false);
#else
  uint32_t s = c;
  for (int i = 0; i < 4; i++)
    s += ((a >> (8 * i)) & 0xff) * ((b >> (8 * i)) & 0xff);
  return s;
#endif
}

// row k of the matrix on one pixel: s in [0, 65535]
template <int K>
GSTAMD_HD uint32_t enc_row (const Enc420Params &ep, uint32_t px)
{
  uint32_t s = dot4_u8 (px, ep.cpos[K], (uint32_t) ep.off[K]);
  if (ep.neg_mask & (1 << K))             // wave-uniform
    s -= dot4_u8 (px, ep.cneg[K], 0);
  return s;
}

// {U | V << 16} of one pixel
GSTAMD_HD uint32_t enc_chroma (const Enc420Params &ep, uint32_t px)
{
  return bperm (enc_row<2> (ep, px), enc_row<1> (ep, px), 0x0c050c01u);
}

// luma bytes of four pixels
GSTAMD_HD uint32_t enc_luma4 (const Enc420Params &ep, const uint4 &p)
{
  const uint32_t s0 = enc_row<0> (ep, p.x), s1 = enc_row<0> (ep, p.y), s2 = enc_row<0> (ep, p.z), s3 = enc_row<0> (ep, p.w);
  const uint32_t lo = bperm (s1, s0, 0x0c0c0501u), hi = bperm (s3, s2, 0x05010c0cu);
  return lo | hi;
}

// source pixels are read once and destination bytes written once: streaming (nontemporal) accesses, as in video_fast.h
GSTAMD_HD uint4 enc_load16 (const uint8_t *p)
{
#ifdef __HIPCC__
  typedef unsigned int u32x4 __attribute__ ((ext_vector_type (4)));
  const u32x4 v = __builtin_nontemporal_load ((const u32x4 *) p);
  return make_uint4 (v.x, v.y, v.z, v.w);
#else
  return *(const uint4 *) p;
#endif
}

GSTAMD_HD void enc_store32 (uint8_t *p, uint32_t v)
{
#ifdef __HIPCC__
  __builtin_nontemporal_store (v, (uint32_t *) p);
#else
  *(uint32_t *) p = v;
#endif
}

GSTAMD_HD uint32_t pk_avg (uint32_t a, uint32_t b) { return pk_shr<1> (a + b + 0x00010001u); }

// one lane: pixels x0 .. x0+3 of the lines 2r, 2r+1 (x0 % 4 == 0, x0 < width)
template <int SEMI>
GSTAMD_HD void enc420_block (const Enc420Params &ep, const uint8_t *__restrict__ src, int sstride, const DstPlanes &d, int x0, int r, long long dd = 0)
{
  const int w = ep.width, h = ep.height;
  const int y0 = 2 * r, y1 = y0 + 1 < h ? y0 + 1 : y0;
  if (x0 >= w || y0 >= h)
    return;
  const uint8_t *row0 = src + (size_t) y0 * sstride, *row1 = src + (size_t) y1 * sstride;
  const uint4 a = enc_load16 (row0 + 4 * (size_t) x0), b = enc_load16 (row1 + 4 * (size_t) x0);
  const int xm = x0 > 0 ? x0 - 1 : 0;
  uint32_t am = 0, bm = 0;
  if (ep.down_h == 2) {
    am = *(const uint32_t *) (row0 + 4 * (size_t) xm);
    bm = *(const uint32_t *) (row1 + 4 * (size_t) xm);
  }
  // ---- luma
  enc_store32 ((d.p[0] + dd) + (size_t) y0 * d.stride[0] + x0, enc_luma4 (ep, a));
  if (y1 != y0)
    enc_store32 ((d.p[0] + dd) + (size_t) y1 * d.stride[0] + x0, enc_luma4 (ep, b));
  // ---- chroma columns x0 .. x0+3 (and x0-1 for the cosited filter), lines averaged first
  uint32_t c[4], cm = 0;
  c[0] = enc_chroma (ep, a.x);
  c[2] = enc_chroma (ep, a.z);
  if (ep.down_h) {
    c[1] = enc_chroma (ep, a.y);
    c[3] = enc_chroma (ep, a.w);
  }
  if (ep.down_h == 2)
    cm = enc_chroma (ep, am);
  if (ep.down_v) {
    c[0] = pk_avg (c[0], enc_chroma (ep, b.x));
    c[2] = pk_avg (c[2], enc_chroma (ep, b.z));
    if (ep.down_h) {
      c[1] = pk_avg (c[1], enc_chroma (ep, b.y));
      c[3] = pk_avg (c[3], enc_chroma (ep, b.w));
    }
    if (ep.down_h == 2)
      cm = pk_avg (cm, enc_chroma (ep, bm));
  }
  uint32_t o0 = c[0], o1 = c[2];
  if (ep.down_h == 1) {
    o0 = pk_avg (c[0], c[1]);
    o1 = pk_avg (c[2], c[3]);
  } else if (ep.down_h == 2) {
    // pixel x0: 1-2-1 (x0 == 0: the clamped left neighbour is the pixel itself = the 3-1 rule); pixel x0+2: 1-2-1, or 1-3
    // when it is one of the last two pixels of the line
    o0 = pk_shr<2> (cm + 2u * c[0] + c[1] + 0x00020002u);
    o1 = x0 + 2 < w - 2 ? pk_shr<2> (c[1] + 2u * c[2] + c[3] + 0x00020002u) : pk_shr<2> (c[1] + 3u * c[2] + 0x00020002u);
  }
  if (SEMI) {
    enc_store32 ((d.p[1] + dd) + (size_t) r * d.stride[1] + x0, bperm (o1, o0, ep.u_first ? 0x06040200u : 0x04060002u));
  } else {
    *(uint16_t *) ((d.p[ep.u_plane] + dd) + (size_t) r * d.stride[ep.u_plane] + (x0 >> 1)) = (uint16_t) bperm (o1, o0, 0x0c0c0400u);
    *(uint16_t *) ((d.p[ep.v_plane] + dd) + (size_t) r * d.stride[ep.v_plane] + (x0 >> 1)) = (uint16_t) bperm (o1, o0, 0x0c0c0602u);
  }
}

}  // namespace gstamd
