// video_deep.h - the 16-bit chain of GstVideoConverter for 10-bit sources, unscaled, into an 8-bit 4-byte destination:
//   unpack_I420_10LE / unpack_P010_10LE  -> AYUV64   (video-format.c:3836-3873, 5331-5400: 10 bits widened to 16 by bit replication)
//   video_chroma_up_h2_u16 / _h2_cs_u16 / _v2_u16    (video-chroma.c:277-327, 687-699, instantiated for guint16 at :469-471, 796:
//                                                     the same (3a+b+2)>>2 and (a+b+1)>>1 on 16-bit values, same line pairing)
//   video_converter_matrix16                         (video-converter.c:1296-1320: (im . px + offset) >> 8, clamped to 0..65535)
//   video_orc_convert_u16_to_u8                      (do_convert_lines :3133: the high byte of every component)
//   alpha / pack as in the 8-bit chain
// First version: one lane = 4 pixels of one row, per-pixel loads (correctness and coverage first, like video_planes.h).
#pragma once
#include "video_device.h"

namespace gstamd {

// a stored 16-bit little-endian word -> the unpacked 16-bit value
GSTAMD_HD int deep_widen (int hi_depth, int v)
{
  if (hi_depth == 1) {                  // I420_10LE: value in the low 10 bits: Y = v << 6 (as guint16), Y |= Y >> 10
    const int t = (v << 6) & 0xffff;
    return t | (t >> 10);
  }
  return v | (v >> 10);                 // P010_10LE: value in the high 10 bits
}

GSTAMD_HD UV deep_load_uv (const FrontParams &f, const Planes &pl, int crow, int k)
{
  UV r;
  if (f.kind == UNPACK_SEMI) {
    const uint16_t *p = (const uint16_t *) (pl.p[1] + (ptrdiff_t) crow * pl.stride[1]) + 2 * k;
    r.u = deep_widen (f.hi_depth, p[f.u_plane ? 0 : 1]);
    r.v = deep_widen (f.hi_depth, p[f.u_plane ? 1 : 0]);
  } else {
    r.u = deep_widen (f.hi_depth, ((const uint16_t *) (pl.p[f.u_plane] + (ptrdiff_t) crow * pl.stride[f.u_plane]))[k]);
    r.v = deep_widen (f.hi_depth, ((const uint16_t *) (pl.p[f.v_plane] + (ptrdiff_t) crow * pl.stride[f.v_plane]))[k]);
  }
  return r;
}

// horizontally filtered chroma of chroma row `crow` at luma position x (chroma_h_at on 16-bit samples)
GSTAMD_HD UV deep_chroma_h_at (const FrontParams &f, const Planes &pl, int crow, int x)
{
  if (f.w_sub == 0)
    return deep_load_uv (f, pl, crow, x);
  const int k = x >> 1, w = f.width;
  UV c = deep_load_uv (f, pl, crow, k);
  if (f.chroma_h == CHROMA_H_H2_CS) {
    if ((x & 1) && x < w - 1) {
      const UV n = deep_load_uv (f, pl, crow, k + 1);
      c.u = (c.u + n.u + 1) >> 1;
      c.v = (c.v + n.v + 1) >> 1;
    }
  } else if (f.chroma_h == CHROMA_H_H2) {
    if ((x & 1) && x < w - 1) {
      const UV n = deep_load_uv (f, pl, crow, k + 1);
      c.u = (3 * c.u + n.u + 2) >> 2;
      c.v = (3 * c.v + n.v + 2) >> 2;
    } else if (!(x & 1) && x >= 2) {
      const UV pv = deep_load_uv (f, pl, crow, k - 1);
      c.u = (pv.u + 3 * c.u + 2) >> 2;
      c.v = (pv.v + 3 * c.v + 2) >> 2;
    }
  }
  return c;
}

// pixel (x, y) through the whole chain: the 8-bit unpack-order word A | c1 << 8 | c2 << 16 | c3 << 24
GSTAMD_HD uint32_t deep_pixel (const FrontParams &f, const Planes &pl, const int *__restrict__ vpair, const Deep16Params &d, int x, int y)
{
  int c1 = deep_widen (f.hi_depth, ((const uint16_t *) (pl.p[0] + (ptrdiff_t) y * pl.stride[0]))[x]);
  UV c;
  if (f.chroma_v2) {
    const int e0 = vpair[2 * y], rb = vpair[2 * y + 1];
    const int ra = vpair_row (e0), role = vpair_role (e0);
    const UV a = deep_chroma_h_at (f, pl, ra, x);
    if (ra == rb) {
      c = a;
    } else {
      const UV b = deep_chroma_h_at (f, pl, rb, x);
      if (role == 0) {
        c.u = (3 * a.u + b.u + 2) >> 2;
        c.v = (3 * a.v + b.v + 2) >> 2;
      } else {
        c.u = (a.u + 3 * b.u + 2) >> 2;
        c.v = (a.v + 3 * b.v + 2) >> 2;
      }
    }
  } else {
    c = deep_chroma_h_at (f, pl, y >> f.h_sub, x);
  }
  int c2 = c.u, c3 = c.v;
  if (d.has_matrix) {
    const int r = c1, g = c2, b = c3;
    c1 = clampi ((d.im[0][0] * r + d.im[0][1] * g + d.im[0][2] * b + d.im[0][3]) >> 8, 0, 65535);
    c2 = clampi ((d.im[1][0] * r + d.im[1][1] * g + d.im[1][2] * b + d.im[1][3]) >> 8, 0, 65535);
    c3 = clampi ((d.im[2][0] * r + d.im[2][1] * g + d.im[2][2] * b + d.im[2][3]) >> 8, 0, 65535);
  }
  return 0xffu | ((uint32_t) (c1 >> 8) << 8) | ((uint32_t) (c2 >> 8) << 16) | ((uint32_t) (c3 >> 8) << 24);
}

// pixels x0 .. x0+3 of row y
GSTAMD_HD void convert16_lane4 (const FrontParams &f, const Planes &pl, const int *__restrict__ vpair, const Deep16Params &d, const PostParams &post,
    uint8_t *dst, int dstride, int x0, int y)
{
  if (x0 >= f.width || y >= f.height)
    return;
  uint32_t *out = (uint32_t *) (dst + (size_t) y * dstride + 4 * (size_t) x0);
  for (int i = 0; i < 4 && x0 + i < f.width; i++) {
    const uint32_t px = apply_alpha (post.alpha_kind, (unsigned) post.alpha_value, deep_pixel (f, pl, vpair, d, x0 + i, y));
    out[i] = pack_px (post.pack_pos, px);
  }
}

}  // namespace gstamd
