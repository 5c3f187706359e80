// video_422_fast.h - unscaled packed 4:2:2 (YUY2, UYVY, YVYU, VYUY) -> 4-byte RGB, the capture-card direction.
//
// Reference path: unpack_YUY2 & co (video-format.c:155-460) -> horizontal chroma upsample (video-chroma.c:277-327, 687-699) ->
// video_orc_convert_AYUV_ARGB (video-orc.orc:1634) -> pack; no vertical chroma step, so lines are independent.  One lane owns 8
// pixels = 4 macropixels = ONE 16-byte load (+ the neighbour macropixels the upsampler looks at), sorts the bytes into a Y
// register pair and one U and one V register with v_perm_b32, upsamples U and V for its 8 pixels in byte lanes with v_lerp_u8
// (identities of video_hscale420.h), runs the matrix through fast_pixel's byte selectors and stores 32 bytes.
// Planner condition: VideoPlan::fast_422 (no scaler pass, AYUV_ARGB matrix without 16-bit wrap, opaque, width % 8 == 0).
#pragma once
#include "video_fast.h"
#include "video_hscale420.h"

namespace gstamd {

struct Fast422Params {
  FastParams fp;            // width / height, matrix, pack selector
  uint32_t sel_y;           // v_perm selector: the two luma bytes of macropixel `lo` then of macropixel `hi`
  uint32_t sel_c2;          // selector picking the byte at U's position of `lo` and of `hi` into bytes 0, 1 (V: sel_c2v)
  uint32_t sel_c2v;
  int chroma_h;
};

// host: selectors for the byte positions of Y0 / U / V inside the macropixel (FormatDesc::pos[1..3])
inline void fast422_selectors (int ypos, int upos, int vpos, Fast422Params *p)
{
  p->sel_y = (uint32_t) ypos | ((uint32_t) (ypos + 2) << 8) | ((uint32_t) (4 + ypos) << 16) | ((uint32_t) (4 + ypos + 2) << 24);
  p->sel_c2 = (uint32_t) upos | ((uint32_t) (4 + upos) << 8) | 0x0c0c0000u;
  p->sel_c2v = (uint32_t) vpos | ((uint32_t) (4 + vpos) << 8) | 0x0c0c0000u;
}

// like fast_pixel, chroma bytes from two registers
GSTAMD_HD uint32_t fast_pixel_uv (const FastParams &fp, uint32_t yx, uint32_t ysel, uint32_t ux, uint32_t vx, uint32_t csel)
{
  const int sy = (int) bperm (yx, yx, ysel);
  const int su = (int) bperm (ux, ux, csel);
  const int sv = (int) bperm (vx, vx, csel);
  const int wy = mulhi24 (sy, fp.p8[0]);
  const int r = med3_0_255 (wy + mulhi24 (sv, fp.p8[1]) + 128);
  const int b = med3_0_255 (wy + mulhi24 (su, fp.p8[2]) + 128);
  const int g = med3_0_255 (wy + mulhi24 (su, fp.p8[3]) + mulhi24 (sv, fp.p8[4]) + 128);
  return bperm ((uint32_t) b, ((uint32_t) g << 8) | (uint32_t) r, fp.pack_sel);
}

// pixels x0 .. x0+7 (x0 % 8 == 0, x0 + 8 <= width) of one line; srow / drow 16-byte aligned.  AYUV = 1: no matrix, the pixels leave as
// A = 0xff, Y, U, V bytes (the chain's AYUV image ahead of a planar pack, or an AYUV destination)
template <int CH, int AYUV = 0>
GSTAMD_HD void convert422_lane8 (const Fast422Params &p, const uint8_t *__restrict__ srow, uint8_t *__restrict__ drow, int x0)
{
  const int w = p.fp.width;
  const uint4 m = *(const uint4 *) (srow + 2 * (size_t) x0);
  // neighbour macropixels, clamped into the line (the reference repeats the edge sample)
  const uint32_t mn = x0 + 8 < w ? *(const uint32_t *) (srow + 2 * (size_t) (x0 + 8)) : m.w;
  const uint32_t mp = (CH == CHROMA_H_H2 && x0 > 0) ? *(const uint32_t *) (srow + 2 * (size_t) (x0 - 2)) : m.x;
  const uint32_t flip = AYUV ? 0u : 0x80808080u;
  const uint32_t y03 = bperm (m.y, m.x, p.sel_y) ^ flip, y47 = bperm (m.w, m.z, p.sel_y) ^ flip;
  const uint32_t u4 = bperm (bperm (m.w, m.z, p.sel_c2), bperm (m.y, m.x, p.sel_c2), 0x05040100u);       // U of macropixels 0..3
  const uint32_t v4 = bperm (bperm (m.w, m.z, p.sel_c2v), bperm (m.y, m.x, p.sel_c2v), 0x05040100u);
  const uint32_t un = bperm (0, mn, p.sel_c2) & 0xffu, vn = bperm (0, mn, p.sel_c2v) & 0xffu;
  uint32_t ue = u4, ve = v4, uo = u4, vo = v4;                  // chroma of the even / odd pixel of each macropixel
  if (CH == CHROMA_H_H2_CS) {
    uo = lerp_u8 (u4, align_bytes (un, u4, 1), 0x01010101u);
    vo = lerp_u8 (v4, align_bytes (vn, v4, 1), 0x01010101u);
  } else if (CH == CHROMA_H_H2) {
    const uint32_t up = bperm (0, mp, p.sel_c2) & 0xffu, vp = bperm (0, mp, p.sel_c2v) & 0xffu;
    ue = blend31_u8 (u4, align_bytes (u4, up << 24, 3));
    ve = blend31_u8 (v4, align_bytes (v4, vp << 24, 3));
    uo = blend31_u8 (u4, align_bytes (un, u4, 1));
    vo = blend31_u8 (v4, align_bytes (vn, v4, 1));
  }
  ue ^= flip, ve ^= flip, uo ^= flip, vo ^= flip;
  uint32_t o[8];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (AYUV) {
      const uint32_t yw = j < 2 ? y03 : y47;
      const uint32_t s0 = 0x0000000du | ((uint32_t) (2 * (j & 1)) << 8) | ((uint32_t) (4 + j) << 16) | 0x0c000000u;
      const uint32_t s1 = 0x0000000du | ((uint32_t) (2 * (j & 1) + 1) << 8) | ((uint32_t) (4 + j) << 16) | 0x0c000000u;
      const uint32_t sv = 0x00020100u | ((uint32_t) (4 + j) << 24);
      o[2 * j] = bperm (ve, bperm (ue, yw, s0), sv);
      o[2 * j + 1] = bperm (vo, bperm (uo, yw, s1), sv);
      continue;
    }
    const uint32_t csel = 0x0c00000cu | ((uint32_t) j << 8) | ((uint32_t) j << 16);
    const uint32_t yw = j < 2 ? y03 : y47;
    const uint32_t ys0 = 0x0c00000cu | ((uint32_t) (2 * (j & 1)) << 8) | ((uint32_t) (2 * (j & 1)) << 16);
    const uint32_t ys1 = 0x0c00000cu | ((uint32_t) (2 * (j & 1) + 1) << 8) | ((uint32_t) (2 * (j & 1) + 1) << 16);
    o[2 * j] = fast_pixel_uv (p.fp, yw, ys0, ue, ve, csel);
    o[2 * j + 1] = fast_pixel_uv (p.fp, yw, ys1, uo, vo, csel);
  }
  uint8_t *d = drow + 4 * (size_t) x0;
#ifdef __HIPCC__
  typedef unsigned int u32x4 __attribute__ ((ext_vector_type (4)));
  const u32x4 v0 = {o[0], o[1], o[2], o[3]}, v1 = {o[4], o[5], o[6], o[7]};
  __builtin_nontemporal_store (v0, (u32x4 *) d);
  __builtin_nontemporal_store (v1, (u32x4 *) (d + 16));
#else
  for (int i = 0; i < 8; i++)
    ((uint32_t *) d)[i] = o[i];
#endif
}

GSTAMD_HD void convert422_lane8_ayuv (const Fast422Params &p, const uint8_t *srow, uint8_t *drow, int x0)
{
  if (p.chroma_h == CHROMA_H_H2_CS)
    convert422_lane8<CHROMA_H_H2_CS, 1> (p, srow, drow, x0);
  else if (p.chroma_h == CHROMA_H_H2)
    convert422_lane8<CHROMA_H_H2, 1> (p, srow, drow, x0);
  else
    convert422_lane8<CHROMA_H_NONE, 1> (p, srow, drow, x0);
}

GSTAMD_HD void convert422_lane8_any (const Fast422Params &p, const uint8_t *srow, uint8_t *drow, int x0)
{
  if (p.chroma_h == CHROMA_H_H2_CS)
    convert422_lane8<CHROMA_H_H2_CS> (p, srow, drow, x0);
  else if (p.chroma_h == CHROMA_H_H2)
    convert422_lane8<CHROMA_H_H2> (p, srow, drow, x0);
  else
    convert422_lane8<CHROMA_H_NONE> (p, srow, drow, x0);
}

// ------------------------------------------------------------------------------------------------
// The reference's own same-size fastpaths convert_I420_BGRA / _ARGB / _pack_ARGB (video-converter.c:3409-3530, I420 / YV12 -> any
// 4-byte RGB): chroma is NOT interpolated there - sample x >> 1 of row y >> 1 (loadupdb) - and the matrix is the AYUV_ARGB arithmetic.
// Decoder output -> display is this conversion.  One lane owns 8 pixels of the two lines that share a chroma row: 8 + 8 luma bytes,
// 4 U and 4 V bytes in, 64 bytes out.  Planner condition: VideoPlan::fast_420p.
// ------------------------------------------------------------------------------------------------
struct Fast420pParams {
  FastParams fp;
  const uint8_t *y, *u, *v;
  int ystride, cstride;
};

GSTAMD_HD void store8_stream (uint8_t *d, const uint32_t *o)
{
#ifdef __HIPCC__
  typedef unsigned int u32x4 __attribute__ ((ext_vector_type (4)));
  const u32x4 v0 = {o[0], o[1], o[2], o[3]}, v1 = {o[4], o[5], o[6], o[7]};
  __builtin_nontemporal_store (v0, (u32x4 *) d);
  __builtin_nontemporal_store (v1, (u32x4 *) (d + 16));
#else
  for (int i = 0; i < 8; i++)
    ((uint32_t *) d)[i] = o[i];
#endif
}

// pixels x0 .. x0+7 (x0 % 8 == 0, x0 + 8 <= width) of lines 2r and 2r+1 (the second one only if it exists)
GSTAMD_HD void convert420p_lane8x2 (const Fast420pParams &p, uint8_t *__restrict__ dst, int dstride, int x0, int r)
{
  const int ya = 2 * r, yb = ya + 1 < p.fp.height ? ya + 1 : ya;
  const uint2 la = *(const uint2 *) (p.y + (size_t) ya * p.ystride + x0), lb = *(const uint2 *) (p.y + (size_t) yb * p.ystride + x0);
  const uint32_t u4 = *(const uint32_t *) (p.u + (size_t) r * p.cstride + (x0 >> 1)) ^ 0x80808080u;
  const uint32_t v4 = *(const uint32_t *) (p.v + (size_t) r * p.cstride + (x0 >> 1)) ^ 0x80808080u;
  const uint32_t ya0 = la.x ^ 0x80808080u, ya1 = la.y ^ 0x80808080u, yb0 = lb.x ^ 0x80808080u, yb1 = lb.y ^ 0x80808080u;
  uint32_t oa[8], ob[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t ysel = 0x0c00000cu | ((uint32_t) (i & 3) << 8) | ((uint32_t) (i & 3) << 16);
    const uint32_t csel = 0x0c00000cu | ((uint32_t) (i >> 1) << 8) | ((uint32_t) (i >> 1) << 16);
    oa[i] = fast_pixel_uv (p.fp, i < 4 ? ya0 : ya1, ysel, u4, v4, csel);
    ob[i] = fast_pixel_uv (p.fp, i < 4 ? yb0 : yb1, ysel, u4, v4, csel);
  }
  store8_stream (dst + (size_t) ya * dstride + 4 * (size_t) x0, oa);
  if (ya + 1 < p.fp.height)
    store8_stream (dst + (size_t) (ya + 1) * dstride + 4 * (size_t) x0, ob);
}

}  // namespace gstamd
