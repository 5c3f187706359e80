// video_pack.h - last stage for every destination that is not a 4-byte packed format (planar, semi-planar, RGB / BGR, packed 4:2:2): chroma downsample + pack of the final AYUV image.
//
// Reference: chain_downsample / do_downsample_lines (video-converter.c:2040, 3192) feed line pairs (2r, 2r+1) - a clamped
// copy of the last line when the height is odd - to gst_video_chroma_resample; video_orc_chroma_down_v2_u8
// (video-orc.orc:2692) averages U and V of the pair into the even line, then the horizontal resampler rewrites the
// even pixels of that line: video_orc_chroma_down_h2_u8 (:2657, pairs, an odd last pixel stays) or
// video_chroma_down_h2_cs_u8 (video-chroma.c:740-762, 3-1 / 1-2-1 / 1-3 taps); pack_planar_420 / pack_NV12 / pack_Y42B /
// pack_Y444 (video-format.c:100-151, 1644-1675, ..) then store Y of every pixel and U, V of the even pixels of the even
// lines.  All of it is per-pixel selection of a few neighbours, so one lane produces a 4-pixel-wide block directly.
#pragma once
#include "video_device.h"
#include "video_dither.h"

namespace gstamd {

// (dd, the bodies' last argument: the byte distance of the frame a workgroup of a frame-list launch writes from frame 0's planes, which
// are what p[] holds - added where a plane pointer is formed: a kernel that rebased p[] itself would hold the struct in LDS)
struct DstPlanes {
  uint8_t *p[3];
  int stride[3];
};

// chain_dither between the chroma downsampler and the packer (do_dither_lines on the AYUV line, video-converter.c:3155): component k of
// the pixel at (x, y) of the converted rectangle - for the chroma of a subsampled destination that is the even pixel of the even
// line, where the downsamplers left their result
GSTAMD_HD int pack_dither (const DitherParams &d, int k, int v, int x, int y)
{
  if (!d.on)
    return v;
  const int sh = d.shift[k];
  const int b = d.method == GSTAMD_DITHER_NONE ? 0 : dither_bayer_value (x, y + d.y0);
  int p = v + (sh < 8 ? b >> (8 - sh) : b);
  p &= ~((1 << sh) - 1) & 0xffff;
  return p > 255 ? 255 : p;
}

// U | V << 16 of AYUV word px (bytes A, Y, U, V)
GSTAMD_HD uint32_t ayuv_uv (uint32_t px) { return ((px >> 16) & 0xffu) | ((px >> 24) << 16); }

// per-lane block: pixels x0 .. x0+3 of the lines (yb << h_sub) .. ; x0 % 4 == 0
// SRC: where the AYUV pixel (x, y) of the converted picture comes from - the image a scaled / gamma / dithering chain left in HBM (SrcImage) or,
// for an unscaled 8-bit chain, the chain itself (SrcFront: unpack + chroma upsample + matrix + alpha per pixel, nothing in between in HBM)
template <class SRC>
GSTAMD_HD void pack_planar_body (const PackPlanarParams &pk, const SRC &src, const DstPlanes &d, int x0, int yb, long long dd = 0)
{
  const int w = pk.width, h = pk.height;
  const int y0 = yb << pk.h_sub;
  if (x0 >= w || y0 >= h)
    return;
  if (pk.kind == UNPACK_PACKED3) {          // pack_RGB / pack_BGR (video-format.c:1540, 1577): 4 pixels = 12 bytes
    uint8_t *q = (d.p[0] + dd) + (size_t) y0 * d.stride[0] + 3 * (size_t) x0;
    for (int i = 0; i < 4 && x0 + i < w; i++) {
      const uint32_t px = src.at (x0 + i, y0);
      q[3 * i + pk.pos[1]] = (uint8_t) pack_dither (pk.dither, 1, (int) ((px >> 8) & 0xff), x0 + i, y0);
      q[3 * i + pk.pos[2]] = (uint8_t) pack_dither (pk.dither, 2, (int) ((px >> 16) & 0xff), x0 + i, y0);
      q[3 * i + pk.pos[3]] = (uint8_t) pack_dither (pk.dither, 3, (int) (px >> 24), x0 + i, y0);
    }
    return;
  }
  if (pk.kind == UNPACK_RGB16) {           // pack_RGB16 / _BGR16 / _RGB15 / _BGR15 (video-format.c:1316-1425): the components' top bits in one word
    uint16_t *q = (uint16_t *) ((d.p[0] + dd) + (size_t) y0 * d.stride[0]) + x0;
    for (int i = 0; i < 4 && x0 + i < w; i++) {
      const uint32_t px = src.at (x0 + i, y0);
      q[i] = (uint16_t) rgb16_pack (pk.pos, pack_dither (pk.dither, 1, (int) ((px >> 8) & 0xff), x0 + i, y0),
          pack_dither (pk.dither, 2, (int) ((px >> 16) & 0xff), x0 + i, y0), pack_dither (pk.dither, 3, (int) (px >> 24), x0 + i, y0));
    }
    return;
  }
  if (pk.kind == UNPACK_GRAY) {            // pack_GRAY8 (video-format.c:1221): the luma byte of every pixel
    uint8_t *q = (d.p[0] + dd) + (size_t) y0 * d.stride[0] + x0;
    for (int i = 0; i < 4 && x0 + i < w; i++)
      q[i] = (uint8_t) pack_dither (pk.dither, 1, (int) ((src.at (x0 + i, y0) >> 8) & 0xff), x0 + i, y0);
    return;
  }
  const int nlines = 1 << pk.h_sub;
  if (pk.kind == UNPACK_PACKED422) {       // luma into the macropixels (pack_YUY2 & co, video-format.c:201-460)
    uint8_t *q = (d.p[0] + dd) + (size_t) y0 * d.stride[0] + 2 * (size_t) x0;
    for (int i = 0; i < 4 && x0 + i < w; i++)
      q[4 * (i >> 1) + pk.pos[1] + 2 * (i & 1)] = (uint8_t) pack_dither (pk.dither, 1, (int) ((src.at (x0 + i, y0) >> 8) & 0xff), x0 + i, y0);
  }
  if (pk.kind == UNPACK_PACKED411) {       // pack_IYU1 (video-format.c:2438-2470): the lumas of the group's pixels that exist at bytes 1, 2, 4, 5
    uint8_t *q = (d.p[0] + dd) + (size_t) y0 * d.stride[0] + 6 * (size_t) (x0 >> 2);
    for (int i = 0; i < 4 && x0 + i < w; i++)
      q[1 + i + (i >> 1)] = (uint8_t) pack_dither (pk.dither, 1, (int) ((src.at (x0 + i, y0) >> 8) & 0xff), x0 + i, y0);
  }
  if (pk.kind == UNPACK_SEMI_TILED) {      // pack_TILED -> pack_NV12 on each tile (video-format.c:5135-5183): the block's bytes at their tile addresses (blocks never straddle tiles: widths >= 4)
    for (int r = 0; r < nlines && y0 + r < h; r++)
      for (int i = 0; i < 4 && x0 + i < w; i++)
        (d.p[0] + dd)[tiled_luma_offset (pk.pos, d.stride[0], x0 + i, y0 + r)] =
            (uint8_t) pack_dither (pk.dither, 1, (int) ((src.at (x0 + i, y0 + r) >> 8) & 0xff), x0 + i, y0 + r);
  }
  // ---- luma of every line of the block
  for (int r = 0; r < nlines && pk.kind != UNPACK_PACKED422 && pk.kind != UNPACK_PACKED411 && pk.kind != UNPACK_SEMI_TILED; r++) {
    const int y = y0 + r;
    if (y >= h)
      break;
    uint8_t *dy = (d.p[0] + dd) + (size_t) y * d.stride[0] + x0;
    for (int i = 0; i < 4 && x0 + i < w; i++)
      dy[i] = (uint8_t) pack_dither (pk.dither, 1, (int) ((src.at (x0 + i, y) >> 8) & 0xff), x0 + i, y);
  }
  // ---- chroma of the block's first line (the "chroma line"), vertically averaged with the next one
  const int yb1 = y0 + 1 < h ? y0 + 1 : (pk.virtual_line ? h : h - 1);
  uint32_t v[6];                            // packed {U, V} of pixels x0-1 .. x0+4, clamped into the row
  for (int i = 0; i < 6; i++) {
    int x = x0 - 1 + i;
    x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
    uint32_t c = ayuv_uv (src.at (x, y0));
    if (pk.down_v)
      c = ((c + ayuv_uv (src.at (x, yb1)) + 0x00010001u) >> 1) & 0x00ff00ffu;      // avgub on both components
    v[i] = c;
  }
  if (pk.w_sub == 2) {        // pack_Y41B (video-format.c:976-1006): the chroma of pixel x0 = 4 k after the 4:1:1 downsampler, which works in place on that pixel
    const int x = x0, k = x0 >> 2;
    uint32_t c = v[1];
    auto at = [&] (int xx) {          // chroma pair of pixel xx of the chroma line (vertical average applied), xx inside the row
      if (xx >= x0 - 1 && xx <= x0 + 4)
        return v[xx - x0 + 1];
      uint32_t cc = ayuv_uv (src.at (xx, y0));
      if (pk.down_v)
        cc = ((cc + ayuv_uv (src.at (xx, yb1)) + 0x00010001u) >> 1) & 0x00ff00ffu;
      return cc;
    };
    if (pk.down_h == 3) {     // video_chroma_down_h4_u8: for (i = 0; i < width - 4; i += 4) PR (i) = FILT_1_3_3_1 (PR (i) .. PR (i + 3))
      if (x < w - 4)
        c = ((v[1] + 3u * (v[2] + v[3]) + v[4] + 0x00040004u) >> 3) & 0x00ff00ffu;
    } else if (pk.down_h == 4 && w >= 4) {      // video_chroma_down_h4_cs_u8
      if (x == 0)
        c = ((10u * v[1] + 3u * v[2] + 2u * v[3] + v[4] + 0x00080008u) >> 4) & 0x00ff00ffu;
      else if (x < w - 4)
        c = ((at (x - 3) + 2u * (at (x - 2) + at (x + 2)) + 3u * (at (x - 1) + at (x + 1)) + 4u * at (x) + at (x + 3) + 0x00080008u) >> 4) & 0x00ff00ffu;
      else
        c = ((at (x - 3) + 2u * at (x - 2) + 3u * at (x - 1) + 10u * at (x) + 0x00080008u) >> 4) & 0x00ff00ffu;
    }
    if (pk.kind == UNPACK_PACKED411) {      // pack_IYU1: U and V of the group's first pixel at bytes 0 and 3
      uint8_t *q = (d.p[0] + dd) + (size_t) y0 * d.stride[0] + 6 * (size_t) k;
      q[0] = (uint8_t) pack_dither (pk.dither, 2, (int) (c & 0xff), x, y0);
      q[3] = (uint8_t) pack_dither (pk.dither, 3, (int) ((c >> 16) & 0xff), x, y0);
      return;
    }
    (d.p[pk.u_plane] + dd)[(size_t) yb * d.stride[pk.u_plane] + k] = (uint8_t) pack_dither (pk.dither, 2, (int) (c & 0xff), x, y0);
    (d.p[pk.v_plane] + dd)[(size_t) yb * d.stride[pk.v_plane] + k] = (uint8_t) pack_dither (pk.dither, 3, (int) ((c >> 16) & 0xff), x, y0);
    return;
  }
  const int step = 1 << pk.w_sub;
  for (int i = 0; i < 4; i += step) {
    const int x = x0 + i;                   // an even pixel when w_sub == 1
    if (x >= w)
      break;
    const uint32_t cm = v[i], c0 = v[i + 1], cp = v[i + 2];      // pixels x-1, x, x+1
    uint32_t c = c0;
    if (pk.w_sub == 1) {
      if (pk.down_h == 1) {
        if (x + 1 < w)
          c = ((c0 + cp + 0x00010001u) >> 1) & 0x00ff00ffu;
      } else if (pk.down_h == 2 && w >= 2) {
        if (x == 0)
          c = ((3u * c0 + cp + 0x00020002u) >> 2) & 0x00ff00ffu;
        else if (x < w - 2)
          c = ((cm + 2u * c0 + cp + 0x00020002u) >> 2) & 0x00ff00ffu;
        else
          c = ((cm + 3u * c0 + 0x00020002u) >> 2) & 0x00ff00ffu;
      }
    }
    uint8_t cu = (uint8_t) pack_dither (pk.dither, 2, (int) (c & 0xff), x, y0), cv = (uint8_t) pack_dither (pk.dither, 3, (int) ((c >> 16) & 0xff), x, y0);
    if (pk.tail_swap && x == w - 1) {
      const uint8_t t = cu;
      cu = cv;
      cv = t;
    }
    const int k = x >> pk.w_sub;
    if (pk.kind == UNPACK_PACKED422) {
      uint8_t *q = (d.p[0] + dd) + (size_t) y0 * d.stride[0] + 4 * (size_t) k;
      q[pk.pos[2]] = cu;
      q[pk.pos[3]] = cv;
    } else if (pk.kind == UNPACK_SEMI_TILED) {
      uint8_t *duv = (d.p[1] + dd) + tiled_uv_offset (pk.pos, d.stride[1], k, yb);
      duv[0] = cu;
      duv[1] = cv;
    } else if (GSTAMD_KIND_SEMI (pk.kind)) {
      uint8_t *duv = (d.p[1] + dd) + (size_t) yb * d.stride[1] + 2 * k;
      duv[0] = pk.u_plane ? cu : cv;
      duv[1] = pk.u_plane ? cv : cu;
    } else {
      (d.p[pk.u_plane] + dd)[(size_t) yb * d.stride[pk.u_plane] + k] = cu;
      (d.p[pk.v_plane] + dd)[(size_t) yb * d.stride[pk.v_plane] + k] = cv;
    }
  }
}

// the fourth plane of an A420 destination (pack_A420 video-format.c:2148-2185: the alpha byte of every pixel of every line) from the chain's AYUV image:
// pixels x0 .. x0 + 3 of line y
GSTAMD_HD void pack_alpha_plane_body (const PackPlanarParams &pk, const uint8_t *img, int istride, uint8_t *da, int dstride, int x0, int y)
{
  if (x0 >= pk.width || y >= pk.height)
    return;
  const uint32_t *row = (const uint32_t *) (img + (size_t) y * istride);
  for (int i = 0; i < 4 && x0 + i < pk.width; i++)
    da[(size_t) y * dstride + x0 + i] = (uint8_t) pack_dither (pk.dither, 0, (int) (row[x0 + i] & 0xffu), x0 + i, y);
}

// pixel source of k_convert_pack for a packed 4:2:2 frame whose chroma is duplicated sideways and taken line by line (the reference's
// YUY2 / UYVY -> planar fastpaths, video_orc_convert_YUY2_I420 & co; chroma_h NONE, no vertical pairing), no matrix, no alpha stage: the
// macropixel word, three byte picks
struct Src422Dup {
  const uint8_t *p;
  int stride;
  int ysh, ush, vsh;    // bit offsets of Y0, U, V inside the macropixel word (Y1 = Y0 + 16)
  int swap_k;           // FrontParams::swap_k
  GSTAMD_HD uint32_t at (int x, int y) const
  {
    const int k = x >> 1;
    const uint32_t m = *(const uint32_t *) (p + (size_t) y * stride + 4 * (size_t) k);
    const uint32_t Y = (m >> (ysh + 16 * (x & 1))) & 0xffu;
    const bool sw = k == swap_k;
    const uint32_t U = (m >> (sw ? vsh : ush)) & 0xffu, V = (m >> (sw ? ush : vsh)) & 0xffu;
    return 0xffu | (Y << 8) | (U << 16) | (V << 24);
  }
};

GSTAMD_HD uint32_t avgub4 (uint32_t a, uint32_t b) { return (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu); }          /* (a + b + 1) >> 1 on four bytes */

// Src422Dup -> planar / semi-planar YUV in wide accesses: 8 pixels (four macropixels, one 16-byte load) of one or two lines per lane - the
// shape of video_orc_convert_YUY2_I420 / _Y42B / _Y444 and their UYVY twins (video-orc.orc:1452-1620): luma copied, chroma copied
// (4:2:2), averaged over the line pair (4:2:0) or doubled (4:4:4).  False for what it leaves to pack_planar_body (picture edge, the swapped
// tail macropixel, a cosited horizontal filter, dither, the line past the picture).
GSTAMD_HD bool pack_422dup_block8 (const PackPlanarParams &pk, const Src422Dup &s, const DstPlanes &d, int x0, int yb, long long dd = 0)
{
  const int w = pk.width, h = pk.height, y0 = yb << pk.h_sub;
  if (y0 >= h)
    return true;
  if (x0 + 8 > w || pk.down_h > 1 || pk.dither.on || pk.virtual_line || (s.swap_k >= 0 && s.swap_k >= (x0 >> 1) && s.swap_k < (x0 >> 1) + 4))
    return false;
  const int y1 = y0 + 1 < h ? y0 + 1 : h - 1;
  const uint4 a = *(const uint4 *) (s.p + (size_t) y0 * s.stride + 2 * (size_t) x0);
  uint4 b = a;
  if (pk.h_sub || pk.down_v)
    b = *(const uint4 *) (s.p + (size_t) y1 * s.stride + 2 * (size_t) x0);
#define GSTAMD_LUM2(m) ((((m) >> s.ysh) & 0xffu) | ((((m) >> (s.ysh + 16)) & 0xffu) << 8))
#define GSTAMD_PICK4(q, sh) ((((q).x >> (sh)) & 0xffu) | ((((q).y >> (sh)) & 0xffu) << 8) | ((((q).z >> (sh)) & 0xffu) << 16) | ((((q).w >> (sh)) & 0xffu) << 24))
  uint2 ya;
  ya.x = GSTAMD_LUM2 (a.x) | (GSTAMD_LUM2 (a.y) << 16), ya.y = GSTAMD_LUM2 (a.z) | (GSTAMD_LUM2 (a.w) << 16);
  *(uint2 *) ((d.p[0] + dd) + (size_t) y0 * d.stride[0] + x0) = ya;
  if (pk.h_sub && y0 + 1 < h) {
    uint2 yb2;
    yb2.x = GSTAMD_LUM2 (b.x) | (GSTAMD_LUM2 (b.y) << 16), yb2.y = GSTAMD_LUM2 (b.z) | (GSTAMD_LUM2 (b.w) << 16);
    *(uint2 *) ((d.p[0] + dd) + (size_t) (y0 + 1) * d.stride[0] + x0) = yb2;
  }
  uint32_t u = GSTAMD_PICK4 (a, s.ush), v = GSTAMD_PICK4 (a, s.vsh);
  if (pk.down_v) {
    u = avgub4 (u, GSTAMD_PICK4 (b, s.ush));
    v = avgub4 (v, GSTAMD_PICK4 (b, s.vsh));
  }
#undef GSTAMD_LUM2
#undef GSTAMD_PICK4
  if (pk.w_sub == 1) {
    const int k = x0 >> 1;
    if (pk.kind == UNPACK_SEMI) {
      const uint32_t f = pk.u_plane ? u : v, g = pk.u_plane ? v : u;
      uint2 o;
      o.x = (f & 0xffu) | ((g & 0xffu) << 8) | ((f & 0xff00u) << 8) | ((g & 0xff00u) << 16);
      o.y = ((f >> 16) & 0xffu) | (((g >> 16) & 0xffu) << 8) | ((f >> 24) << 16) | ((g >> 24) << 24);
      *(uint2 *) ((d.p[1] + dd) + (size_t) yb * d.stride[1] + 2 * k) = o;
    } else {
      *(uint32_t *) ((d.p[pk.u_plane] + dd) + (size_t) yb * d.stride[pk.u_plane] + k) = u;
      *(uint32_t *) ((d.p[pk.v_plane] + dd) + (size_t) yb * d.stride[pk.v_plane] + k) = v;
    }
    return true;
  }
  /* 4:4:4: every pixel its macropixel's chroma */
  uint2 uu, vv;
  uu.x = (u & 0xffu) * 0x0101u | (((u >> 8) & 0xffu) * 0x0101u) << 16, uu.y = ((u >> 16) & 0xffu) * 0x0101u | ((u >> 24) * 0x0101u) << 16;
  vv.x = (v & 0xffu) * 0x0101u | (((v >> 8) & 0xffu) * 0x0101u) << 16, vv.y = ((v >> 16) & 0xffu) * 0x0101u | ((v >> 24) * 0x0101u) << 16;
  if (pk.kind == UNPACK_SEMI) {
    uint8_t *q = (d.p[1] + dd) + (size_t) yb * d.stride[1] + 2 * (size_t) x0;
    for (int i = 0; i < 8; i++) {
      const uint32_t cu = ((i < 4 ? uu.x : uu.y) >> (8 * (i & 3))) & 0xffu, cv = ((i < 4 ? vv.x : vv.y) >> (8 * (i & 3))) & 0xffu;
      q[2 * i] = (uint8_t) (pk.u_plane ? cu : cv);
      q[2 * i + 1] = (uint8_t) (pk.u_plane ? cv : cu);
    }
  } else {
    *(uint2 *) ((d.p[pk.u_plane] + dd) + (size_t) yb * d.stride[pk.u_plane] + x0) = uu;
    *(uint2 *) ((d.p[pk.v_plane] + dd) + (size_t) yb * d.stride[pk.v_plane] + x0) = vv;
  }
  return true;
}

// The common shape of the pack in wide accesses: AYUV image -> planar / semi-planar YUV, a block of 4 pixels wholly inside the picture, no
// dither stage: 16-byte loads of the block's pixels (the image rows are 16-byte aligned: the caller checks), the luma of a line as one
// 32-bit store, the chroma as 16-bit (4:2:x planar), 32-bit (semi-planar, 4:4:4) stores; the arithmetic is pack_planar_body's on the same
// packed {U, V} words.  Returns false for the blocks it leaves to the general body (picture edge, tail_swap pixel).
// ROWS: where four / one AYUV pixels of a row come from - the AYUV image (ImgRows), or a 4-byte source frame through its unpack permutation and
// colour stage (SrcPacked4: k_convert_pack's block form)
struct ImgRows {
  const uint8_t *img;
  int sstride;
  GSTAMD_HD bool ok4 (int, int) const { return true; }
  GSTAMD_HD uint4 row4 (int x0, int y) const { return *(const uint4 *) (img + (size_t) y * sstride + 4 * (size_t) x0); }
  GSTAMD_HD uint32_t px (int x, int y) const { return *(const uint32_t *) (img + (size_t) y * sstride + 4 * (size_t) x); }
  GSTAMD_HD uint4 row4n (int x0, int y, bool edges, int xm, int xp, uint32_t &em, uint32_t &ep) const
  {
    if (edges)
      em = px (xm, y), ep = px (xp, y);
    return row4 (x0, y);
  }
};

// (store false: the lane takes part in the row source's calls - a source may trade values between the lanes of a wave - and writes nothing)
template <class ROWS>
GSTAMD_HD bool pack_planar_block4 (const PackPlanarParams &pk, const ROWS &rows, const DstPlanes &d, int x0, int yb, long long dd = 0, bool store = true)
{
  const int w = pk.width, h = pk.height, y0 = yb << pk.h_sub;
  if (x0 + 4 > w || y0 >= h || (pk.tail_swap && x0 + 4 == w && (w & 1)))
    return false;
  const int yb1 = y0 + 1 < h ? y0 + 1 : (pk.virtual_line ? h : h - 1);
  if (!rows.ok4 (x0, y0) || !rows.ok4 (x0, yb1))
    return false;
  /* the cosited horizontal downsampler also reads the pixels left and right of the block (their chroma): handed over by the row source together
   * with the block where that is cheaper than two more pixel fetches per line (Src422Up has them in the macropixels it loaded anyway) */
  const bool edges = pk.w_sub == 1 && pk.down_h == 2;
  uint32_t em = 0, ep = 0, fm = 0, fp = 0;
  const int xm = x0 > 0 ? x0 - 1 : 0, xp = x0 + 4 < w ? x0 + 4 : w - 1;
  const uint4 a = rows.row4n (x0, y0, edges, xm, xp, em, ep);
  uint4 b = a;
  if (pk.h_sub || pk.down_v)
    b = rows.row4n (x0, yb1, edges && pk.down_v, xm, xp, fm, fp);
  if (!store)
    return true;
  // luma
  *(uint32_t *) ((d.p[0] + dd) + (size_t) y0 * d.stride[0] + x0) = ((a.x >> 8) & 0xffu) | (a.y & 0xff00u) | ((a.z << 8) & 0xff0000u) | ((a.w << 16) & 0xff000000u);
  if (pk.h_sub && y0 + 1 < h)
    *(uint32_t *) ((d.p[0] + dd) + (size_t) (y0 + 1) * d.stride[0] + x0) = ((b.x >> 8) & 0xffu) | (b.y & 0xff00u) | ((b.z << 8) & 0xff0000u) | ((b.w << 16) & 0xff000000u);
  // chroma of pixels x0 - 1 .. x0 + 4 of the chroma line
  uint32_t v[6];
  v[1] = ayuv_uv (a.x), v[2] = ayuv_uv (a.y), v[3] = ayuv_uv (a.z), v[4] = ayuv_uv (a.w);
  v[0] = ayuv_uv (em), v[5] = ayuv_uv (ep);
  if (pk.down_v) {
    v[1] = ((v[1] + ayuv_uv (b.x) + 0x00010001u) >> 1) & 0x00ff00ffu;
    v[2] = ((v[2] + ayuv_uv (b.y) + 0x00010001u) >> 1) & 0x00ff00ffu;
    v[3] = ((v[3] + ayuv_uv (b.z) + 0x00010001u) >> 1) & 0x00ff00ffu;
    v[4] = ((v[4] + ayuv_uv (b.w) + 0x00010001u) >> 1) & 0x00ff00ffu;
    v[0] = ((v[0] + ayuv_uv (fm) + 0x00010001u) >> 1) & 0x00ff00ffu;
    v[5] = ((v[5] + ayuv_uv (fp) + 0x00010001u) >> 1) & 0x00ff00ffu;
  }
  uint32_t c[4];                            // what is stored: all four (w_sub == 0) or c[0], c[2]
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int x = x0 + i;
    const uint32_t cm = v[i], c0 = v[i + 1], cp = v[i + 2];
    uint32_t r = c0;
    if (pk.w_sub == 1 && !(i & 1)) {
      if (pk.down_h == 1) {
        r = ((c0 + cp + 0x00010001u) >> 1) & 0x00ff00ffu;              /* x + 1 < w: the block is inside the picture */
      } else if (pk.down_h == 2 && w >= 2) {
        if (x == 0)
          r = ((3u * c0 + cp + 0x00020002u) >> 2) & 0x00ff00ffu;
        else if (x < w - 2)
          r = ((cm + 2u * c0 + cp + 0x00020002u) >> 2) & 0x00ff00ffu;
        else
          r = ((cm + 3u * c0 + 0x00020002u) >> 2) & 0x00ff00ffu;
      }
    }
    c[i] = r;
  }
  if (pk.w_sub == 1) {
    const uint32_t u2 = (c[0] & 0xffu) | ((c[2] & 0xffu) << 8), v2 = (c[0] >> 16) | ((c[2] >> 16) << 8);
    const int k = x0 >> 1;
    if (pk.kind == UNPACK_SEMI) {
      const uint32_t first = pk.u_plane ? u2 : v2, second = pk.u_plane ? v2 : u2;
      *(uint32_t *) ((d.p[1] + dd) + (size_t) yb * d.stride[1] + 2 * k) = (first & 0xffu) | ((second & 0xffu) << 8) | ((first & 0xff00u) << 8) | ((second & 0xff00u) << 16);
    } else {
      *(uint16_t *) ((d.p[pk.u_plane] + dd) + (size_t) yb * d.stride[pk.u_plane] + k) = (uint16_t) u2;
      *(uint16_t *) ((d.p[pk.v_plane] + dd) + (size_t) yb * d.stride[pk.v_plane] + k) = (uint16_t) v2;
    }
  } else {
    const uint32_t u4 = (c[0] & 0xffu) | ((c[1] & 0xffu) << 8) | ((c[2] & 0xffu) << 16) | ((c[3] & 0xffu) << 24);
    const uint32_t v4 = (c[0] >> 16) | ((c[1] >> 16) << 8) | ((c[2] >> 16) << 16) | ((c[3] >> 16) << 24);
    if (pk.kind == UNPACK_SEMI) {           /* NV24 family: 8 interleaved bytes */
      uint8_t *q = (d.p[1] + dd) + (size_t) yb * d.stride[1] + 2 * (size_t) x0;
      for (int i = 0; i < 4; i++) {
        q[2 * i] = (uint8_t) (pk.u_plane ? c[i] : c[i] >> 16);
        q[2 * i + 1] = (uint8_t) (pk.u_plane ? c[i] >> 16 : c[i]);
      }
    } else {
      *(uint32_t *) ((d.p[pk.u_plane] + dd) + (size_t) yb * d.stride[pk.u_plane] + x0) = u4;
      *(uint32_t *) ((d.p[pk.v_plane] + dd) + (size_t) yb * d.stride[pk.v_plane] + x0) = v4;
    }
  }
  return true;
}

// pixel source of k_convert_pack_422up: a packed 4:2:2 frame through unpack and the HORIZONTAL chroma upsampler of the generic chain
// (chroma_h_at of video_device.h: video_chroma_up_h2 / _h2_cs on the macropixels' chroma), no vertical pairing, no matrix, no alpha stage -
// what YUY2 / UYVY -> NV12 / NV21 and the 4:2:2 -> 4:4:4 packs run through (the reference has fastpaths for the PLANAR destinations only:
// those are Src422Dup's).  at (): any pixel, with the edge rules; row4 (): four pixels of a block whose neighbours exist (ok4), from one
// 16-byte load of the macropixels k0 - 1 .. k0 + 2 and packed {U | V << 16} arithmetic.
struct Src422Up {
  const uint8_t *p;
  int stride;
  int ysh, ush, vsh;    // bit offsets of Y0, U, V inside the macropixel word (Y1 = Y0 + 16)
  int swap_k;           // FrontParams::swap_k
  int chroma_h;         // CHROMA_H_H2 or CHROMA_H_H2_CS
  int width, luma_last;
  GSTAMD_HD uint32_t uv_of (uint32_t m, bool sw) const
  {
    const uint32_t u = (m >> (sw ? vsh : ush)) & 0xffu, v = (m >> (sw ? ush : vsh)) & 0xffu;
    return u | (v << 16);
  }
  GSTAMD_HD uint32_t at (int x, int y) const
  {
    const int k = x >> 1;
    const uint8_t *row = p + (size_t) y * stride;
    const uint8_t *lrow = p + (size_t) (y < luma_last ? y : luma_last) * stride;
    const uint32_t Y = (*(const uint32_t *) (lrow + 4 * (size_t) k) >> (ysh + 16 * (x & 1))) & 0xffu;
    uint32_t c = uv_of (*(const uint32_t *) (row + 4 * (size_t) k), k == swap_k);
    if ((x & 1) && x < width - 1) {
      const uint32_t n = uv_of (*(const uint32_t *) (row + 4 * (size_t) (k + 1)), k + 1 == swap_k);
      c = chroma_h == CHROMA_H_H2_CS ? ((c + n + 0x00010001u) >> 1) & 0x00ff00ffu : ((3u * c + n + 0x00020002u) >> 2) & 0x00ff00ffu;
    } else if (!(x & 1) && x >= 2 && chroma_h == CHROMA_H_H2) {
      const uint32_t pv = uv_of (*(const uint32_t *) (row + 4 * (size_t) (k - 1)), false);
      c = ((pv + 3u * c + 0x00020002u) >> 2) & 0x00ff00ffu;
    }
    return 0xffu | (Y << 8) | ((c & 0xffu) << 16) | ((c >> 16) << 24);
  }
  GSTAMD_HD uint32_t px (int x, int y) const { return at (x, y); }
  /* whole blocks of a picture at least eight pixels wide, none of their macropixels the swapped tail.  (The picture's first and last block come out of
   * the same formulas on a clamped neighbour - (c + c + 1) >> 1 == (3 c + c + 2) >> 2 == c are the edge rules of chroma_h_at - so only a partial
   * last block and the tail of an odd width go through the general body: with a lane of every row in it, its waves set the kernel's time.) */
  GSTAMD_HD bool ok4 (int x0, int y) const { return x0 + 4 <= width && width >= 8 && (swap_k < 0 || swap_k > (x0 >> 1) + 2) && y <= luma_last; }
  /* + the pixels x0 - 1 and x0 + 4 (clamped into the line: xm, xp): the odd pixel of macropixel k0 - 1, the even one of k0 + 2 */
  GSTAMD_HD uint4 row4n (int x0, int y, bool edges, int, int, uint32_t &em, uint32_t &ep) const
  {
    /* macropixels k0 - 1 .. k0 + 2; at the picture's left edge k0 .. k0 + 3 with the first one standing in for its missing neighbour, at the right
     * edge k0 - 2 .. k0 + 1 with the last one doing so */
    const bool left = x0 == 0, right = x0 + 4 >= width;
    const uint8_t *q = p + (size_t) y * stride + 2 * (size_t) x0 - (left ? 0 : (right ? 8 : 4));
#ifdef __HIPCC__
    typedef unsigned int u32x4a __attribute__ ((ext_vector_type (4), aligned (4)));
    const u32x4a m = *(const u32x4a *) q;
    const uint32_t l0 = m.x, l1 = m.y, l2 = m.z, l3 = m.w;
#else
    uint32_t mm[4];
    __builtin_memcpy (mm, q, 16);
    const uint32_t l0 = mm[0], l1 = mm[1], l2 = mm[2], l3 = mm[3];
#endif
    uint32_t m0 = l0, m1 = l1, m2 = l2, m3 = l3;
    if (left)
      m1 = l0, m2 = l1, m3 = l2;
    else if (right)
      m0 = l1, m1 = l2, m2 = l3;
    const uint32_t c0 = uv_of (m0, false), c1 = uv_of (m1, false), c2 = uv_of (m2, false), c3 = uv_of (m3, false);
    uint32_t e1, o1, e2, o2;          /* pixels x0 (even, macropixel 1), x0 + 1, x0 + 2 (macropixel 2), x0 + 3 */
    if (chroma_h == CHROMA_H_H2_CS) {
      e1 = c1, e2 = c2;
      o1 = ((c1 + c2 + 0x00010001u) >> 1) & 0x00ff00ffu;
      o2 = ((c2 + c3 + 0x00010001u) >> 1) & 0x00ff00ffu;
    } else {
      e1 = ((c0 + 3u * c1 + 0x00020002u) >> 2) & 0x00ff00ffu;
      o1 = ((3u * c1 + c2 + 0x00020002u) >> 2) & 0x00ff00ffu;
      e2 = ((c1 + 3u * c2 + 0x00020002u) >> 2) & 0x00ff00ffu;
      o2 = ((3u * c2 + c3 + 0x00020002u) >> 2) & 0x00ff00ffu;
    }
#define GSTAMD_AYUV(m, odd, c) (0xffu | ((((m) >> (ysh + 16 * (odd))) & 0xffu) << 8) | (((c) & 0xffu) << 16) | (((c) >> 16) << 24))
    if (edges) {
      const uint32_t om = chroma_h == CHROMA_H_H2_CS ? ((c0 + c1 + 0x00010001u) >> 1) & 0x00ff00ffu : ((3u * c0 + c1 + 0x00020002u) >> 2) & 0x00ff00ffu;
      const uint32_t e3 = chroma_h == CHROMA_H_H2_CS ? c3 : ((c2 + 3u * c3 + 0x00020002u) >> 2) & 0x00ff00ffu;
      em = GSTAMD_AYUV (m0, 1, om), ep = GSTAMD_AYUV (m3, 0, e3);
    }
    const uint4 r = gstamd_make_uint4 (GSTAMD_AYUV (m1, 0, e1), GSTAMD_AYUV (m1, 1, o1), GSTAMD_AYUV (m2, 0, e2), GSTAMD_AYUV (m2, 1, o2));
#undef GSTAMD_AYUV
    return r;
  }
};

GSTAMD_HD bool pack_planar_block4 (const PackPlanarParams &pk, const uint8_t *__restrict__ img, int sstride, const DstPlanes &d, int x0, int yb, long long dd = 0)
{
  const ImgRows rows = {img, sstride};
  return pack_planar_block4 (pk, rows, d, x0, yb, dd);
}

// The chroma downsamplers alone, on the AYUV image in place, exactly as the reference's line caches leave the lines (the error-diffusion
// dither methods read every pixel of every line afterwards, also the chroma the packers never store): first the vertical pass - U, V of
// every pixel of the pair's first line become the average - then, on that line (every line without vertical subsampling), the
// horizontal pass rewrites the even pixels from their unchanged odd neighbours.  Two launches: each is free of read-after-write overlap.
GSTAMD_HD void pack_down_v_px (const PackPlanarParams &pk, uint8_t *img, int stride, int x, int yb)
{
  const int w = pk.width, h = pk.height, y0 = yb << pk.h_sub;
  if (x >= w || y0 >= h || !pk.down_v)
    return;
  uint32_t *ra = (uint32_t *) (img + (size_t) y0 * stride);
  const uint32_t *rb = (const uint32_t *) (img + (size_t) (y0 + 1 < h ? y0 + 1 : (pk.virtual_line ? h : h - 1)) * stride);
  const uint32_t c = ((ayuv_uv (ra[x]) + ayuv_uv (rb[x]) + 0x00010001u) >> 1) & 0x00ff00ffu;
  ra[x] = (ra[x] & 0xffffu) | ((c & 0xffu) << 16) | ((c >> 16) << 24);
}

GSTAMD_HD void pack_down_h_px (const PackPlanarParams &pk, uint8_t *img, int stride, int x, int yb)
{
  const int w = pk.width, h = pk.height, y0 = yb << pk.h_sub;
  if (pk.w_sub == 2) {          /* video_chroma_down_h4_u8 / _h4_cs_u8 in place: only the pixels 4 k are rewritten, from neighbours that are not */
    if (x >= w || y0 >= h || !pk.down_h || (x & 3))
      return;
    uint32_t *r4 = (uint32_t *) (img + (size_t) y0 * stride);
    auto at = [&] (int xx) { return ayuv_uv (r4[xx]); };
    uint32_t c4 = at (x);
    if (pk.down_h == 3) {
      if (x < w - 4)
        c4 = ((at (x) + 3u * (at (x + 1) + at (x + 2)) + at (x + 3) + 0x00040004u) >> 3) & 0x00ff00ffu;
    } else if (pk.down_h == 4 && w >= 4) {
      if (x == 0)
        c4 = ((10u * at (0) + 3u * at (1) + 2u * at (2) + at (3) + 0x00080008u) >> 4) & 0x00ff00ffu;
      else if (x < w - 4)
        c4 = ((at (x - 3) + 2u * (at (x - 2) + at (x + 2)) + 3u * (at (x - 1) + at (x + 1)) + 4u * at (x) + at (x + 3) + 0x00080008u) >> 4) & 0x00ff00ffu;
      else
        c4 = ((at (x - 3) + 2u * at (x - 2) + 3u * at (x - 1) + 10u * at (x) + 0x00080008u) >> 4) & 0x00ff00ffu;
    }
    r4[x] = (r4[x] & 0xffffu) | ((c4 & 0xffu) << 16) | ((c4 >> 16) << 24);
    return;
  }
  if (x >= w || y0 >= h || pk.w_sub != 1 || !pk.down_h || (x & 1))
    return;
  uint32_t *ra = (uint32_t *) (img + (size_t) y0 * stride);
  const uint32_t c0 = ayuv_uv (ra[x]), cm = ayuv_uv (ra[x > 0 ? x - 1 : 0]), cp = ayuv_uv (ra[x + 1 < w ? x + 1 : w - 1]);
  uint32_t c = c0;
  if (pk.down_h == 1) {
    if (x + 1 < w)
      c = ((c0 + cp + 0x00010001u) >> 1) & 0x00ff00ffu;
  } else if (pk.down_h == 2 && w >= 2) {
    if (x == 0)
      c = ((3u * c0 + cp + 0x00020002u) >> 2) & 0x00ff00ffu;
    else if (x < w - 2)
      c = ((cm + 2u * c0 + cp + 0x00020002u) >> 2) & 0x00ff00ffu;
    else
      c = ((cm + 3u * c0 + 0x00020002u) >> 2) & 0x00ff00ffu;
  }
  ra[x] = (ra[x] & 0xffffu) | ((c & 0xffu) << 16) | ((c >> 16) << 24);
}

// the packer's view of the image once the downsamplers and the dither stage have run over it: selection only
inline PackPlanarParams pack_select_only (PackPlanarParams pk)     /* host */
{
  pk.down_h = pk.down_v = 0;
  pk.dither.on = 0;
  return pk;
}

}  // namespace gstamd
