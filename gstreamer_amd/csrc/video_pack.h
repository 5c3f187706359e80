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
GSTAMD_HD void pack_planar_body (const PackPlanarParams &pk, const uint8_t *__restrict__ src, int sstride, const DstPlanes &d, int x0, int yb)
{
  const int w = pk.width, h = pk.height;
  const int y0 = yb << pk.h_sub;
  if (x0 >= w || y0 >= h)
    return;
  if (pk.kind == UNPACK_PACKED3) {          // pack_RGB / pack_BGR (video-format.c:1540, 1577): 4 pixels = 12 bytes
    const uint32_t *row = (const uint32_t *) (src + (size_t) y0 * sstride);
    uint8_t *q = d.p[0] + (size_t) y0 * d.stride[0] + 3 * (size_t) x0;
    for (int i = 0; i < 4 && x0 + i < w; i++) {
      const uint32_t px = row[x0 + i];
      q[3 * i + pk.pos[1]] = (uint8_t) pack_dither (pk.dither, 1, (int) ((px >> 8) & 0xff), x0 + i, y0);
      q[3 * i + pk.pos[2]] = (uint8_t) pack_dither (pk.dither, 2, (int) ((px >> 16) & 0xff), x0 + i, y0);
      q[3 * i + pk.pos[3]] = (uint8_t) pack_dither (pk.dither, 3, (int) (px >> 24), x0 + i, y0);
    }
    return;
  }
  const int nlines = 1 << pk.h_sub;
  if (pk.kind == UNPACK_PACKED422) {       // luma into the macropixels (pack_YUY2 & co, video-format.c:201-460)
    const uint32_t *row = (const uint32_t *) (src + (size_t) y0 * sstride);
    uint8_t *q = d.p[0] + (size_t) y0 * d.stride[0] + 2 * (size_t) x0;
    for (int i = 0; i < 4 && x0 + i < w; i++)
      q[4 * (i >> 1) + pk.pos[1] + 2 * (i & 1)] = (uint8_t) pack_dither (pk.dither, 1, (int) ((row[x0 + i] >> 8) & 0xff), x0 + i, y0);
  }
  // ---- luma of every line of the block
  for (int r = 0; r < nlines && pk.kind != UNPACK_PACKED422; r++) {
    const int y = y0 + r;
    if (y >= h)
      break;
    const uint32_t *row = (const uint32_t *) (src + (size_t) y * sstride);
    uint8_t *dy = d.p[0] + (size_t) y * d.stride[0] + x0;
    for (int i = 0; i < 4 && x0 + i < w; i++)
      dy[i] = (uint8_t) pack_dither (pk.dither, 1, (int) ((row[x0 + i] >> 8) & 0xff), x0 + i, y);
  }
  // ---- chroma of the block's first line (the "chroma line"), vertically averaged with the next one
  const uint32_t *ra = (const uint32_t *) (src + (size_t) y0 * sstride);
  const uint32_t *rb = (const uint32_t *) (src + (size_t) (y0 + 1 < h ? y0 + 1 : (pk.virtual_line ? h : h - 1)) * sstride);
  uint32_t v[6];                            // packed {U, V} of pixels x0-1 .. x0+4, clamped into the row
  for (int i = 0; i < 6; i++) {
    int x = x0 - 1 + i;
    x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
    uint32_t c = ayuv_uv (ra[x]);
    if (pk.down_v)
      c = ((c + ayuv_uv (rb[x]) + 0x00010001u) >> 1) & 0x00ff00ffu;      // avgub on both components
    v[i] = c;
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
      uint8_t *q = d.p[0] + (size_t) y0 * d.stride[0] + 4 * (size_t) k;
      q[pk.pos[2]] = cu;
      q[pk.pos[3]] = cv;
    } else if (pk.kind == UNPACK_SEMI) {
      uint8_t *duv = d.p[1] + (size_t) yb * d.stride[1] + 2 * k;
      duv[0] = pk.u_plane ? cu : cv;
      duv[1] = pk.u_plane ? cv : cu;
    } else {
      d.p[pk.u_plane][(size_t) yb * d.stride[pk.u_plane] + k] = cu;
      d.p[pk.v_plane][(size_t) yb * d.stride[pk.v_plane] + k] = cv;
    }
  }
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
