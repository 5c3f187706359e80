// compositor_device.h - per-pixel device code of the compositor blend kernels (bodies only; see
// video_device.h for the host-emulation arrangement).
//
// Reference semantics reproduced bit-exactly (paths under
// /root/reference/subprojects/gst-plugins-base/gst/compositor/):
//   BLEND_A32 clipping                      blend.c:41-99
//   _blend_loop_* / _overlay_loop_*         blend.c:101-158
//   compositor_orc_blend_argb/bgra          compositororc.orc:158-195, 225-265  (C: compositororc-dist.c:1900-2160)
//   compositor_orc_source_argb/bgra         compositororc.orc:196-224, 266-294
//   compositor_orc_overlay_* (+_addition)   compositororc.orc:295-559 (divluw: compositororc-dist.c:3345)
//   fill_checker_*_c / fill_color_*         blend.c:177-240
//   _draw_background + blend_pads           compositor.c:1619-1697
//
// "argb" family = alpha in memory byte 0 (ARGB, ABGR, AYUV); "bgra" family = alpha in byte 3
// (BGRA, RGBA): blend.h:56-61.  A pixel is the little-endian uint32 of its 4 memory bytes.
#pragma once
#include <stdint.h>
#include <stddef.h>

#include "../../include/gstamd_video.h"

#ifdef __HIPCC__
#define GSTAMD_CD __device__ __forceinline__
#else
#define GSTAMD_CD inline
#endif

namespace gstamd {

#define GSTAMD_MAX_FUSED_PADS 32

struct PadDev {
  const uint8_t *data;
  int width, height, stride;
  int xpos, ypos;
  int s_alpha;      // CLAMP ((gint) (src_alpha * 255), 0, 255)
  int mode;         // GstCompositorBlendMode
};

struct AggregateParams {
  int ashift;       // 0 ("argb" family) or 24 ("bgra" family): bit position of alpha in the LE word
  int overlay;      // 1: transparent background -> overlay functions
  int bg_kind;      // 0 checker, 1 solid colour word, 2 keep destination (continuation chunk)
  int checker_yuv;  // checker for AYUV: Y = tab, U = V = 128
  uint32_t bg_word;
  int n_pads;
  PadDev pads[GSTAMD_MAX_FUSED_PADS];
};

GSTAMD_CD uint32_t div255w (uint32_t x) { return ((x & 0xffffu) * 0x8081u) >> 23; }

// compositor_orc_blend_*: opaque destination
GSTAMD_CD uint32_t px_blend (uint32_t s, uint32_t d, uint32_t alpha, int ashift)
{
  const uint32_t a = div255w (((s >> ashift) & 0xff) * alpha);
  const uint32_t ia = (0xffu - a) & 0xffffu;
  uint32_t r = 0;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const uint32_t sw = (((s >> (8 * c)) & 0xff) * a) & 0xffffu;
    const uint32_t dw = (((d >> (8 * c)) & 0xff) * ia) & 0xffffu;
    r |= (div255w ((dw + sw) & 0xffffu) & 0xff) << (8 * c);
  }
  return r | (0xffu << ashift);
}

// compositor_orc_source_*: copy colour, scale alpha
GSTAMD_CD uint32_t px_source (uint32_t s, uint32_t alpha, int ashift)
{
  const uint32_t a = div255w (((s >> ashift) & 0xff) * alpha) & 0xff;
  return (s & ~(0xffu << ashift)) | (a << ashift);
}

// compositor_orc_overlay_* and _addition: destination may be transparent
GSTAMD_CD uint32_t px_overlay (uint32_t s, uint32_t d, uint32_t alpha, int ashift, bool addition)
{
  const uint32_t alpha_s = div255w (((s >> ashift) & 0xff) * alpha);
  const uint32_t alpha_s_inv = (0xffu - alpha_s) & 0xffffu;
  const uint32_t da = (d >> ashift) & 0xff;
  uint32_t alpha_d = div255w ((da * alpha_s_inv) & 0xffffu);
  const uint32_t norm = (alpha_d + alpha_s) & 0xffffu;          // final alpha (over) / alpha factor (add)
  uint32_t r = 0;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const uint32_t sw = (((s >> (8 * c)) & 0xff) * alpha_s) & 0xffffu;
    const uint32_t dw = (((d >> (8 * c)) & 0xff) * alpha_d) & 0xffffu;
    const uint32_t sum = (dw + sw) & 0xffffu;
    uint32_t q;
    if ((norm & 0xff) == 0)
      q = 255;
    else {
      q = sum / (norm & 0xff);                                  // divluw
      q = q > 255 ? 255 : q;
    }
    r |= (q & 0xff) << (8 * c);
  }
  const uint32_t out_a = addition ? ((da + alpha_s) & 0xff) : (norm & 0xff);
  return (r & ~(0xffu << ashift)) | (out_a << ashift);
}

// one pad applied to one destination pixel value (BlendFunction semantics, any background)
GSTAMD_CD uint32_t apply_pad (uint32_t d, uint32_t s, int s_alpha, int mode, int ashift, int overlay)
{
  if (mode == GSTAMD_COMPOSITOR_BLEND_MODE_SOURCE)
    return s_alpha == 255 ? s : px_source (s, (uint32_t) s_alpha, ashift);
  if (!overlay)
    return px_blend (s, d, (uint32_t) s_alpha, ashift);
  return px_overlay (s, d, (uint32_t) s_alpha, ashift, mode == GSTAMD_COMPOSITOR_BLEND_MODE_ADD);
}

GSTAMD_CD uint32_t checker_px (int x, int y, int ashift, int yuv)
{
  const uint32_t val = (((y & 0x8) >> 3) + ((x & 0x8) >> 3)) == 1 ? 160u : 80u;   // tab = {80,160,80,160}
  if (!yuv)
    return (ashift == 0) ? (0xffu | (val << 8) | (val << 16) | (val << 24)) : (val | (val << 8) | (val << 16) | (0xffu << 24));
  return (ashift == 0) ? (0xffu | (val << 8) | (128u << 16) | (128u << 24)) : (128u | (128u << 8) | (val << 16) | (0xffu << 24));
}

// fused aggregate: value of destination pixel (x, y) after background + all pads of this chunk
GSTAMD_CD uint32_t aggregate_px (const AggregateParams &p, uint32_t dest_in, int x, int y)
{
  uint32_t d = p.bg_kind == 0 ? checker_px (x, y, p.ashift, p.checker_yuv) : (p.bg_kind == 1 ? p.bg_word : dest_in);
  for (int i = 0; i < p.n_pads; i++) {
    const PadDev &pad = p.pads[i];
    const int sx = x - pad.xpos, sy = y - pad.ypos;
    if (sx >= 0 && sy >= 0 && sx < pad.width && sy < pad.height) {
      const uint32_t s = *(const uint32_t *) (pad.data + (size_t) sy * pad.stride + 4 * (size_t) sx);
      d = apply_pad (d, s, pad.s_alpha, pad.mode, p.ashift, p.overlay);
    }
  }
  return d;
}

}  // namespace gstamd
