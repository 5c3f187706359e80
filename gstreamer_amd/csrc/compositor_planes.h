// compositor_planes.h - compositor output in the formats that have no per-pixel alpha: planar YUV (I420, YV12, Y42B, Y444 and their
// 10 / 12 / 16-bit forms), semi-planar (NV12, NV21), 24-bit RGB / BGR, the 32-bit xRGB / xBGR / RGBx / BGRx (RGB_BLEND with bpp 4,
// blend.c:1768-1783) and packed 4:2:2 YUY2 / UYVY / YVYU (PACKED_422_BLEND :1785-1925).  The reference converts every pad to the output format first and then
// blends PLANE BY PLANE with the pad alpha only (gst/compositor/blend.c: PLANAR_YUV_BLEND :247-405, NV_YUV_BLEND :1387-1500,
// RGB_BLEND :1610-1684): a plane rectangle is copied (alpha 1.0 or operator `source`), left alone (alpha 0.0) or run
// through compositor_orc_blend_u8 (compositororc.orc:20-36: d = (d * 256 + (s - d) * alpha) >> 8 in 16 bits, alpha =
// (int) (pad_alpha * 255)), byte by byte - chroma bytes included.  Background: fill_checker_* (:405-450, 1500-1540,
// 1686-1711: 80 / 160 in 8 x 8 pixel squares on luma / on R, G and B; chroma 0x80), fill_color_*, or memset 0
// (compositor.c:1641-1668).
//
// GPU shape: ONE pass per destination plane; a lane owns 4 consecutive bytes of a row, starts from the background (or the
// canvas for follow-up chunks) and applies the pads' rectangles in z-order in registers - every destination byte is
// written once, whatever the number of pads.  The host computes the rectangles with the reference's own clipping rules.
#pragma once
#include <stdint.h>
#include <string.h>

#include "planner.h"

#ifdef __HIPCC__
#define GSTAMD_CP __device__ __forceinline__
#else
#define GSTAMD_CP inline
#endif

namespace gstamd {

#define GSTAMD_PLANE_MAX_PADS 24

struct PlaneRect {
  const uint8_t *src;     // first source byte of the rectangle
  int sstride;
  int x, y, w, h;         // destination rectangle in BYTES x ROWS of the plane
  int mode;               // 1: copy, 2: blend with `alpha`
  int alpha;              // 0 .. 255
};

struct PlaneJob {
  uint8_t *dst;
  int dstride;
  int wbytes, rows;       // plane size
  int bg_kind;            // 0: checker, 1: constant bytes, 2: keep the canvas (follow-up chunk)
  int px_bytes;           // bytes per pixel of the plane (checker squares are 8 PIXELS wide)
  int bg_period;          // constants repeat every bg_period bytes (= px_bytes; 4 for packed 4:2:2, whose U and V differ)
  int keep_mask;          // checker only: bit k = byte k of a pixel is NOT written (the x byte of xRGB ...: RGB_FILL_CHECKER_C writes r, g, b)
  int const_mask;         // checker only: bit k = byte k of a pixel is bg[k] (the chroma bytes of packed 4:2:2: 128)
  int bits;               // 8, or 10 / 12 / 16: the plane holds 16-bit little-endian samples of that depth (compositor_orc_blend_u10 / u12 / u16,
                          // compositororc.orc:38-88: d = (d << bits + (s - d) * alpha) >> bits in 32 bits, saturated to 16; alpha = (int) (a * (2^bits - 1)))
  uint8_t bg[4];          // constant background: byte k of every pixel
  int n;
  PlaneRect r[GSTAMD_PLANE_MAX_PADS];
};

// one destination byte
GSTAMD_CP uint32_t plane_byte (const PlaneJob &j, int x, int y, uint32_t canvas)
{
  uint32_t d;
  if (j.bg_kind == 2)
    d = canvas;
  else if (j.bg_kind == 0) {
    const int k = x % j.px_bytes;
    if ((j.keep_mask >> k) & 1)
      d = canvas;
    else if ((j.const_mask >> k) & 1)
      d = j.bg[k];
    else
      d = ((((unsigned) y & 8u) >> 3) + ((((unsigned) (x / j.px_bytes)) & 8u) >> 3)) & 1u ? 160u : 80u;
  } else
    d = j.bg[x % j.bg_period];
  for (int k = 0; k < j.n; k++) {
    const PlaneRect &r = j.r[k];
    if (x < r.x || x >= r.x + r.w || y < r.y || y >= r.y + r.h)
      continue;
    const uint32_t s = r.src[(size_t) (y - r.y) * r.sstride + (x - r.x)];
    d = r.mode == 1 ? s : ((d << 8) + (uint32_t) (((int) s - (int) d) * r.alpha)) >> 8 & 0xffu;
  }
  return d;
}

// one 16-bit sample (x = its first byte, even)
GSTAMD_CP uint32_t plane_sample16 (const PlaneJob &j, int x, int y, uint32_t canvas)
{
  uint32_t d;
  if (j.bg_kind == 2)
    d = canvas;
  else if (j.bg_kind == 0)          /* PLANAR_YUV_HIGH_FILL_CHECKER (blend.c:503-551): 80 / 160 << (bits - 8) in 8 x 8 pixel squares */
    d = (((((unsigned) y & 8u) >> 3) + ((((unsigned) (x / j.px_bytes)) & 8u) >> 3)) & 1u ? 160u : 80u) << (j.bits - 8);
  else
    d = (uint32_t) j.bg[0] | ((uint32_t) j.bg[1] << 8);
  for (int k = 0; k < j.n; k++) {
    const PlaneRect &r = j.r[k];
    if (x < r.x || x >= r.x + r.w || y < r.y || y >= r.y + r.h)
      continue;
    const uint8_t *sp = r.src + (size_t) (y - r.y) * r.sstride + (x - r.x);
    const uint32_t s = (uint32_t) sp[0] | ((uint32_t) sp[1] << 8);
    if (r.mode == 1) {
      d = s;
    } else {
      const uint32_t t = ((d << j.bits) + (uint32_t) (((int) s - (int) d) * r.alpha)) >> j.bits;       /* subl, mulll, shll, addl, shrul: 32-bit wrap */
      d = (int32_t) t < 0 ? 0u : (t > 65535u ? 65535u : t);                                             /* convsuslw */
    }
  }
  return d;
}

// one lane: bytes x0 .. x0+3 of row y
GSTAMD_CP void plane_word_body (const PlaneJob &j, int x0, int y)
{
  if (x0 >= j.wbytes || y >= j.rows)
    return;
  uint8_t *q = j.dst + (size_t) y * j.dstride + x0;
  if (j.bits > 8) {                     /* two 16-bit samples (plane rows are whole samples: wbytes is even) */
    const uint16_t *q16 = (const uint16_t *) q;
    const uint32_t a = plane_sample16 (j, x0, y, j.bg_kind == 2 ? q16[0] : 0);
    if (x0 + 2 < j.wbytes) {
      const uint32_t b = plane_sample16 (j, x0 + 2, y, j.bg_kind == 2 ? q16[1] : 0);
      if ((((uintptr_t) q) & 3) == 0)
        *(uint32_t *) q = a | (b << 16);
      else
        ((uint16_t *) q)[0] = (uint16_t) a, ((uint16_t *) q)[1] = (uint16_t) b;
    } else {
      *(uint16_t *) q = (uint16_t) a;
    }
    return;
  }
  const int n = j.wbytes - x0 < 4 ? j.wbytes - x0 : 4;
  uint32_t v[4];
  for (int i = 0; i < n; i++)
    v[i] = plane_byte (j, x0 + i, y, (j.bg_kind == 2 || j.keep_mask) ? q[i] : 0);
  if (n == 4 && (((uintptr_t) q) & 3) == 0)
    *(uint32_t *) q = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
  else
    for (int i = 0; i < n; i++)
      q[i] = (uint8_t) v[i];
}

// ---- host side: the reference's rectangle arithmetic ---------------------------------------------------------------------
struct FramePad {           // one pad, already in the output format
  const uint8_t *data[3];
  int stride[3];
  int width, height, xpos, ypos;
  double alpha;
  int mode;                 // GstCompositorBlendMode
};

inline int sub_scale (int v, int sub) { return -((-v) >> sub); }     /* GST_VIDEO_SUB_SCALE */

struct PlaneGeom {          // plane i of a format: which bytes it holds
  int w_sub, h_sub;         // subsampling of the plane's component(s)
  int px_bytes;             // bytes per (subsampled) pixel
};

// planes of the supported output formats; returns the number of planes or 0
inline int compositor_plane_geometry (const FormatDesc *f, PlaneGeom g[3])
{
  if (!f)
    return 0;
  if (f->kind == UNPACK_PLANAR && (f->hi_depth == 0 || f->hi_depth == 1 || f->hi_depth == 4 || f->hi_depth == 6)) {
    /* 8-bit planes, or 16-bit little-endian samples with the value in the low bits (I420_10LE ... Y444_16LE: blend.c:609-681) */
    const int sb = f->hi_depth ? 2 : 1;
    g[0] = {0, 0, sb};
    g[1] = g[2] = {f->w_sub, f->h_sub, sb};
    return 3;
  }
  if (f->hi_depth)
    return 0;
  if (f->kind == UNPACK_SEMI && f->w_sub == 1 && f->h_sub == 1) {
    g[0] = {0, 0, 1};
    g[1] = {1, 1, 2};
    return 2;
  }
  if (f->kind == UNPACK_PACKED3) {
    g[0] = {0, 0, 3};
    return 1;
  }
  if (f->kind == UNPACK_PACKED4 && !f->alpha) {         /* xRGB, xBGR, RGBx, BGRx: blend_xrgb on 4 bytes per pixel, the x byte included */
    g[0] = {0, 0, 4};
    return 1;
  }
  if (f->kind == UNPACK_PACKED422 && f->format != GSTAMD_VIDEO_FORMAT_VYUY) {
    g[0] = {0, 0, 2};                                   /* YUY2, UYVY, YVYU: rows of 2 bytes per pixel, xpos on even pixels */
    return 1;
  }
  return 0;
}

// bytes of a row of plane `pl` the background writes (PACKED_422_FILL_*: whole macropixels; the transparent memset: 2 * width, compositor.c:1657)
inline int compositor_plane_row_bytes (const FormatDesc *f, const PlaneGeom &g, int dw, int background)
{
  if (f->kind == UNPACK_PACKED422)
    return (background == 3 ? dw : ((dw + 1) & ~1)) * 2;
  return sub_scale (dw, g.w_sub) * g.px_bytes;
}


// the rectangle of plane `pl` one pad contributes (blend_<format> of blend.c); false: nothing to do
inline bool compositor_pad_rect (const FormatDesc *f, const PlaneGeom &g, int pl, const FramePad &pad, int dest_w, int dest_h, PlaneRect *out)
{
  double alpha = pad.mode == 0 /* SOURCE */ ? 1.0 : pad.alpha;
  if (alpha == 0.0 || !pad.data[pl])
    return false;
  int xpos = pad.xpos, ypos = pad.ypos;
  if (f->kind != UNPACK_PACKED3) {
    /* x_round / y_round: GST_ROUND_UP_2 where the format subsamples (i420, nv12: both; y42b: x only) */
    if (f->w_sub)
      xpos = (xpos + 1) & ~1;
    if (f->h_sub)
      ypos = (ypos + 1) & ~1;
  }
  int bw = pad.width, bh = pad.height, xoff = 0, yoff = 0;
  if (xpos < 0) {
    xoff = -xpos;
    bw -= -xpos;
    xpos = 0;
  }
  if (ypos < 0) {
    yoff = -ypos;
    bh -= -ypos;
    ypos = 0;
  }
  if (xoff >= pad.width || yoff >= pad.height)
    return false;
  if (xpos + bw > dest_w)
    bw = dest_w - xpos;
  if (ypos + bh > dest_h)
    bh = dest_h - ypos;
  if (bw <= 0 || bh <= 0)
    return false;
  /* component geometry: sizes round up, positions and offsets of the chroma planes: x rounds up, y rounds DOWN (:358-367) */
  const int cw = sub_scale (bw, g.w_sub), ch = sub_scale (bh, g.h_sub);
  const int cx = xpos == 0 ? 0 : sub_scale (xpos, g.w_sub), cxo = xoff == 0 ? 0 : sub_scale (xoff, g.w_sub);
  const int cy = pl == 0 ? ypos : ypos >> g.h_sub, cyo = pl == 0 ? yoff : yoff >> g.h_sub;
  out->src = pad.data[pl] + (size_t) cyo * pad.stride[pl] + (size_t) cxo * g.px_bytes;
  out->sstride = pad.stride[pl];
  out->x = cx * g.px_bytes;
  out->y = cy;
  out->w = cw * g.px_bytes;
  out->h = ch;
  const int range = (1 << (f->hi_depth ? hi_depth_bits (f->hi_depth) : 8)) - 1;          /* PLANAR_YUV_BLEND: range = (1 << n_bits) - 1 (blend.c:280-281) */
  if (alpha == 1.0) {
    out->mode = 1;
    out->alpha = range;
  } else {
    int a = (int) (alpha * range);
    out->mode = 2;
    out->alpha = a < 0 ? 0 : (a > range ? range : a);
  }
  return true;
}

// background of plane `pl`: kind (GstCompositorBackground) + the black / white colours in component order (Y,U,V or R,G,B)
inline void compositor_plane_background (const FormatDesc *f, const PlaneGeom &g, int pl, int background, const int black[3], const int white[3],
    PlaneJob *job)
{
  job->px_bytes = g.px_bytes;
  job->bg_period = g.px_bytes;
  job->keep_mask = job->const_mask = 0;
  job->bits = f->hi_depth ? hi_depth_bits (f->hi_depth) : 8;
  memset (job->bg, 0, sizeof (job->bg));
  if (job->bits > 8) {
    /* PLANAR_YUV_HIGH_FILL_CHECKER / _FILL_COLOR (blend.c:503-607): luma checker, chroma 1 << (bits - 1); colours as 16-bit values */
    const bool luma = pl == 0;
    int v = 0;
    if (background == 3) {
      job->bg_kind = 1;
    } else if (background == 0) {
      job->bg_kind = luma ? 0 : 1;
      v = 1 << (job->bits - 1);
    } else {
      const int *c = background == 1 ? black : white;
      job->bg_kind = 1;
      v = luma ? c[0] : (pl == f->u_plane ? c[1] : c[2]);
    }
    job->bg[0] = (uint8_t) (v & 0xff), job->bg[1] = (uint8_t) ((v >> 8) & 0xff);
    return;
  }
  if (background == 3) {                  /* transparent: memset 0 */
    job->bg_kind = 1;
    return;
  }
  const bool luma_like = pl == 0;         /* the Y plane, or the single RGB plane */
  if (f->kind == UNPACK_PACKED422) {
    /* PACKED_422_FILL_CHECKER_C (blend.c:1847-1876: Y = 80 / 160 by pixel, U = V = 128; YVYU takes YUY2's) and _FILL_COLOR (:1878-1917) */
    const int ybyte = f->pos[1] & 1;
    if (background == 0) {
      job->bg_kind = 0;
      job->const_mask = 1 << (ybyte ^ 1);
      job->bg[ybyte ^ 1] = 128;
    } else {
      const int *c = background == 1 ? black : white;
      job->bg_kind = 1;
      job->bg_period = 4;
      job->bg[f->pos[1]] = job->bg[f->pos[1] + 2] = (uint8_t) c[0];
      job->bg[f->pos[2]] = (uint8_t) c[1];
      job->bg[f->pos[3]] = (uint8_t) c[2];
    }
    return;
  }
  if (background == 0) {                  /* checker */
    if (luma_like) {
      job->bg_kind = 0;
      if (f->kind == UNPACK_PACKED4)      /* RGB_FILL_CHECKER_C (xrgb / rgbx, :1693-1717) leaves the x byte alone */
        job->keep_mask = 1 << f->pos[0];
    } else {
      job->bg_kind = 1;
      job->bg[0] = job->bg[1] = 0x80;
    }
    return;
  }
  const int *c = background == 1 ? black : white;
  job->bg_kind = 1;
  if (f->kind == UNPACK_PACKED4 && f->pos[0] == 0) {
    /* MEMSET_XRGB (xrgb, 24, 16, 0) / (xbgr, 0, 16, 24) (blend.c:1770, 1773): the word's shifts put the three colours into bytes 0, 1 and 3 -
     * R G 0 B for xRGB, B G 0 R for xBGR - not into the format's own byte order; followed as is */
    const bool rgb_order = f->pos[1] == 1;
    job->bg[0] = (uint8_t) c[rgb_order ? 0 : 2];
    job->bg[1] = (uint8_t) c[1];
    job->bg[3] = (uint8_t) c[rgb_order ? 2 : 0];
  } else if (f->kind == UNPACK_PACKED3 || f->kind == UNPACK_PACKED4) {       /* MEMSET_XRGB (rgbx / bgrx): the x byte is 0 */
    for (int k = 1; k < 4; k++)
      job->bg[f->pos[k]] = (uint8_t) c[k - 1];
  } else if (pl == 0) {
    job->bg[0] = (uint8_t) c[0];
  } else if (f->kind == UNPACK_SEMI) {
    job->bg[0] = (uint8_t) (f->u_plane ? c[1] : c[2]);
    job->bg[1] = (uint8_t) (f->u_plane ? c[2] : c[1]);
  } else {
    job->bg[0] = (uint8_t) (pl == f->u_plane ? c[1] : c[2]);
  }
}

}  // namespace gstamd
