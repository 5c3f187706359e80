// compositor_wide.h - the compositor's blend loops for ARGB64 / AYUV64 canvases (16 bits per component, alpha first, little endian), per
// destination pixel like k_aggregate: background -> pad 0 -> pad 1 ... in registers, one store.
//
// Reference (subprojects/gst-plugins-base/gst/compositor/blend.c):
//   compositor_blend_argb64            :701-754   dst = (src * a + dst * (65535 - a)) / 65535, alpha forced to 0xffff
//   compositor_source_argb64           :756-785   source pixels with alpha a_src * a / 65535 (a plain copy when a == 65535, :1245-1252)
//   compositor_overlay_argb64          :787-858   associative over: normalised by the final alpha
//   compositor_overlay_argb64_addition :860-937   the same with the destination alpha accumulated
//   BLEND_A64 (clipping, s_alpha = CLAMP ((gint) (alpha * 65535)))  :1179-1234;  fill_checker_argb64_c / _ayuv64_c :1310-1352;  fill_color_argb64 :1354-1384
// The arithmetic is the reference's 64-bit integer arithmetic as it stands (products of two 16-bit values and sums of two of them).
#pragma once
#include <stdint.h>

#include "compositor_device.h"

namespace gstamd {

struct Wide64Params {
  int overlay;                  // 1: the overlay_* functions (transparent background), 0: blend_*
  int bg_kind;                  // 0 checker, 1 colour, 2 keep the canvas
  int checker_yuv;
  uint64_t bg_px;               // colour background: the whole pixel
  int n_pads;
  PadDev pads[GSTAMD_MAX_FUSED_PADS];   // s_alpha on the 16-bit scale
};

GSTAMD_CD uint64_t wide64_blend (uint64_t s, uint64_t d, uint64_t a)
{
  uint64_t sa = ((s & 0xffffull) * a) / 65535ull;
  sa = sa < 65535ull ? sa : 65535ull;
  const uint64_t inv = 65535ull - sa;
  uint64_t r = 0xffffull;
#pragma unroll
  for (int k = 1; k < 4; k++) {
    uint64_t c = (((s >> (16 * k)) & 0xffffull) * sa + ((d >> (16 * k)) & 0xffffull) * inv) / 65535ull;
    c = c < 65535ull ? c : 65535ull;
    r |= c << (16 * k);
  }
  return r;
}

GSTAMD_CD uint64_t wide64_source (uint64_t s, uint64_t a)
{
  if (a == 65535ull)
    return s;                   /* memcpy */
  uint64_t sa = ((s & 0xffffull) * a) / 65535ull;
  sa = sa < 65535ull ? sa : 65535ull;
  return (s & 0xffffffffffff0000ull) | sa;
}

GSTAMD_CD uint64_t wide64_overlay (uint64_t s, uint64_t d, uint64_t a, bool addition)
{
  uint64_t sa = ((s & 0xffffull) * a) / 65535ull;
  sa = sa < 65535ull ? sa : 65535ull;
  const uint64_t inv = 65535ull - sa;
  uint64_t da = ((d & 0xffffull) * inv) / 65535ull;     /* dst_alpha / alpha_factor */
  uint64_t fa = da + sa;
  fa = fa < 65535ull ? fa : 65535ull;
  uint64_t r = 0;
#pragma unroll
  for (int k = 1; k < 4; k++) {
    uint64_t c = ((d >> (16 * k)) & 0xffffull) * da + ((s >> (16 * k)) & 0xffffull) * sa;
    if (fa > 0)
      c /= fa;
    c = c < 65535ull ? c : 65535ull;
    r |= c << (16 * k);
  }
  uint64_t out_a = fa;
  if (addition) {
    out_a = (d & 0xffffull) + sa;
    out_a = out_a < 65535ull ? out_a : 65535ull;
  }
  return r | out_a;
}

// destination pixel (x, y) of the rectangle being rendered
GSTAMD_CD void wide64_px (const Wide64Params &p, uint8_t *dst, int dstride, int x, int y)
{
  uint64_t *q = (uint64_t *) (dst + (size_t) y * dstride) + x;
  uint64_t px;
  if (p.bg_kind == 0) {
    const uint64_t v = ((((y & 8) >> 3) + ((x & 8) >> 3)) & 1) ? 40960ull : 20480ull;
    px = p.checker_yuv ? (0xffffull | (v << 16) | (32768ull << 32) | (32768ull << 48)) : (0xffffull | (v << 16) | (v << 32) | (v << 48));
  } else if (p.bg_kind == 1) {
    px = p.bg_px;
  } else {
    px = *q;
  }
  for (int i = 0; i < p.n_pads; i++) {
    const PadDev &pd = p.pads[i];
    const int sx = x - pd.xpos, sy = y - pd.ypos;
    if (sx < 0 || sy < 0 || sx >= pd.width || sy >= pd.height)
      continue;
    const uint64_t s = *((const uint64_t *) (pd.data + (size_t) sy * pd.stride) + sx);
    const uint64_t a = (uint64_t) pd.s_alpha;
    if (pd.mode == 0)
      px = wide64_source (s, a);
    else if (!p.overlay)
      px = wide64_blend (s, px, a);
    else
      px = wide64_overlay (s, px, a, pd.mode == 2);
  }
  *q = px;
}

}  // namespace gstamd
