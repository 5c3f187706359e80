// compositor_scaled.h - pads that are SCALED on their way into the canvas, sampled inside the blend kernel.
//
// The reference gives every pad whose frames differ from the canvas a converter (GstVideoAggregatorConvertPad,
// gstvideoaggregator.c:479-513: gst_video_converter_new (pad info -> canvas format at the pad's width x height, the pad's
// converter-config) and gst_video_converter_frame per buffer) and blends the converted frame (compositor.c:1678-1697).  When only
// the SIZE differs and the format is a 4-byte 8-bit packed one, that converter is two scaler passes on the raw pixels
// (convert_scale_planes -> gst_video_scaler_2d, video-converter.c:7757, video-scaler.c:1342-1530) - planner.cpp: plan_is_pad_scaler.
// Here the scaled pixel is evaluated where it is needed: V (H (source)) or H (V (source)) in the plan's pass order with the passes'
// own tables and the 4 x u8 arithmetic of video_device.h (hscale_px / vscale_px; the first pass's result is the clamped u8 the
// reference stores in its temporary line), so the scaled frame never exists in HBM and the frame is one launch.
#pragma once
#include "compositor_device.h"
#include "video_scale_fast.h"

namespace gstamd {

#define GSTAMD_MAX_SCALED_PADS 16

struct ScaledPadDev {
  PadDev pad;           // data / stride: the frame as it arrived; width / height: the pad's size ON THE CANVAS
  int n_pass;           // 0: the frame is blended as it is
  int h_first;          // both passes: the horizontal one runs first
  int src_w;            // pixels per row of the frame in pad.data
  ScaleDev sh, sv;      // kind SCALE_NONE: no pass in that direction
};

struct ScaledAggParams {
  int ashift, overlay, bg_kind, checker_yuv;
  uint32_t bg_word;
  int n_pads;
  ScaledPadDev pads[GSTAMD_MAX_SCALED_PADS];
};

struct PadImage {
  const uint8_t *p;
  int stride;
  GSTAMD_HD uint32_t at (int x, int y) const { return *(const uint32_t *) (p + (size_t) y * stride + 4 * (size_t) x); }
};

struct PadHRows {       // the horizontally scaled frame, a pixel at a time
  PadImage img;
  const ScaleDev *sh;
  GSTAMD_HD uint32_t at (int x, int y) const
  {
    const RowOfSrc<PadImage> row = {img, y};
    return hscale_px (row, *sh, x);
  }
};

struct PadVRow {        // row sy of the vertically scaled frame
  PadImage img;
  const ScaleDev *sv;
  int sy;
  GSTAMD_HD uint32_t at (int x) const { return vscale_px (img, *sv, x, sy); }
};

GSTAMD_HD uint32_t scaled_pad_px (const ScaledPadDev &sp, int sx, int sy)
{
  const PadImage img = {sp.pad.data, sp.pad.stride};
  if (sp.n_pass == 0)
    return img.at (sx, sy);
  const bool has_h = sp.sh.kind != SCALE_NONE, has_v = sp.sv.kind != SCALE_NONE;
  if (has_h && has_v) {
    if (sp.h_first) {
      const PadHRows rows = {img, &sp.sh};
      return vscale_px (rows, sp.sv, sx, sy);
    }
    const PadVRow row = {img, &sp.sv, sy};
    return hscale_px (row, sp.sh, sx);
  }
  if (has_h) {
    const RowOfSrc<PadImage> row = {img, sy};
    return hscale_px (row, sp.sh, sx);
  }
  return vscale_px (img, sp.sv, sx, sy);
}

// destination pixel (x, y): background -> pads in order, as aggregate_px
GSTAMD_HD uint32_t aggregate_scaled_px (const ScaledAggParams &p, uint32_t dest_in, int x, int y)
{
  uint32_t d = p.bg_kind == 0 ? checker_px (x, y, p.ashift, p.checker_yuv) : (p.bg_kind == 1 ? p.bg_word : dest_in);
  for (int i = 0; i < p.n_pads; i++) {
    const ScaledPadDev &sp = p.pads[i];
    const int sx = x - sp.pad.xpos, sy = y - sp.pad.ypos;
    if (sx >= 0 && sy >= 0 && sx < sp.pad.width && sy < sp.pad.height)
      d = apply_pad (d, scaled_pad_px (sp, sx, sy), sp.pad.s_alpha, sp.pad.mode, p.ashift, p.overlay);
  }
  return d;
}

// ---- workgroup form: a tile of the canvas per workgroup, a scaled pad's FIRST pass evaluated once per tile into LDS -------------------
// Evaluating V (H (source)) per destination pixel repeats the first pass n_taps times (64 loads per pixel for an 8 x 8 tap
// downscale).  A workgroup owns a SCALED_TILE_W x SCALED_TILE_H tile; for every scaled pad that touches the tile, in pad order:
// all lanes fill the first pass's results for the part of the pad under the tile (vertical first: the tile's rows over the
// horizontal source span; horizontal first: the vertical source span over the tile's columns), barrier, every lane runs the second
// pass for its pixels from LDS and blends, barrier.  Pads whose intermediate does not fit SCALED_LDS_PX (extreme ratios), one-pass
// and unscaled pads take the per-pixel form - the values are the same either way.
#define SCALED_TILE_W 64
#define SCALED_TILE_H 16
#define SCALED_LDS_PX 3072

struct ScaledTileGeom {
  int mode;             // 0: the pad misses the tile, 1: per pixel, 2: vertical pass in LDS, 3: horizontal pass in LDS
  int sx0, sx1, sy0, sy1;       // the part of the pad (in its own scaled coordinates) under the tile
  int lo, w, rows;      // LDS image: mode 2 - `rows` tile rows x source columns [lo, lo + w); mode 3 - source rows [lo, lo + rows) x w tile columns
  int pitch;            // LDS pixels per row (mode 2: w rounded up to whole quads)
};

// packed 16-bit accumulators of the *_u8_lq scaler programs (mullw / addw wrap, then (acc + 32) >> 6 and saturation; video_scale_fast.h):
// bytes 0, 2 of the pixel in `e`, bytes 1, 3 in `o`
struct PkAcc {
  uint32_t e, o;
  GSTAMD_HD void mad (uint32_t px, uint32_t tap_splat)
  {
    e = pk_mad16 (px & 0x00ff00ffu, tap_splat, e);
    o = pk_mad16 (pk_shr<8> (px), tap_splat, o);
  }
  GSTAMD_HD uint32_t finish () const { return pk_lq_finish (e) | (pk_lq_finish (o) << 8); }
};

GSTAMD_HD uint32_t tap_splat (int16_t t) { return (uint32_t) (uint16_t) t * 0x00010001u; }

// the eight taps of an output as four words with ONE load (a row of the tap table: 16 bytes on a 16-byte boundary), tap l in both halves of a
// register with one v_perm - tap by tap the kernel issued a 2-byte load per tap and lane, ~10 per canvas pixel, and sat on the vector-memory
// issue rate (C4-A: 103 us)
struct Taps8 { uint32_t w[4]; };
GSTAMD_HD Taps8 taps8_load (const int16_t *t)
{
  Taps8 r;
  const uint4 v = *(const uint4 *) t;
  r.w[0] = v.x, r.w[1] = v.y, r.w[2] = v.z, r.w[3] = v.w;
  return r;
}
template <int L>
GSTAMD_HD uint32_t taps8_splat (const Taps8 &t) { return bperm (0u, t.w[L >> 1], (L & 1) ? 0x03020302u : 0x01000100u); }

GSTAMD_HD void vscale_span (const ScaleDev &sd, int t0, int t1, int *y_lo, int *y_hi)
{
  const int n = sd.kind == SCALE_NEAREST ? 1 : (sd.kind == SCALE_2TAP ? 2 : sd.n_taps);
  *y_lo = (int) sd.offset[t0];
  *y_hi = (int) sd.offset[t1 - 1] + n;
}

GSTAMD_HD ScaledTileGeom scaled_tile_geom (const ScaledPadDev &sp, int tx0, int ty0, int tx1, int ty1)
{
  ScaledTileGeom g;
  g.mode = 0;
  g.lo = g.w = g.rows = g.pitch = 0;
  g.sx0 = (tx0 > sp.pad.xpos ? tx0 : sp.pad.xpos) - sp.pad.xpos;
  g.sy0 = (ty0 > sp.pad.ypos ? ty0 : sp.pad.ypos) - sp.pad.ypos;
  g.sx1 = (tx1 < sp.pad.xpos + sp.pad.width ? tx1 : sp.pad.xpos + sp.pad.width) - sp.pad.xpos;
  g.sy1 = (ty1 < sp.pad.ypos + sp.pad.height ? ty1 : sp.pad.ypos + sp.pad.height) - sp.pad.ypos;
  if (g.sx1 <= g.sx0 || g.sy1 <= g.sy0)
    return g;
  g.mode = 1;
  if (sp.n_pass < 2)
    return g;
  int lo, hi;
  if (sp.h_first) {
    vscale_span (sp.sv, g.sy0, g.sy1, &lo, &hi);
    g.w = g.sx1 - g.sx0;
    g.rows = hi - lo;
  } else {
    hscale_span (sp.sh, g.sx0, g.sx1, &lo, &hi);
    g.w = hi - lo;
    g.rows = g.sy1 - g.sy0;
  }
  g.lo = lo;
  g.pitch = sp.h_first ? g.w : (g.w + 3) & ~3;
  if (g.pitch * g.rows <= SCALED_LDS_PX)
    g.mode = sp.h_first ? 3 : 2;
  return g;
}

// first pass of a scaled pad under the tile -> LDS (lane `tid` of `nthreads`)
GSTAMD_HD void scaled_tile_stage (const ScaledPadDev &sp, const ScaledTileGeom &g, uint32_t *lds, int tid, int nthreads, int src_w)
{
  const PadImage img = {sp.pad.data, sp.pad.stride};
  if (g.mode == 2) {
    /* vertical pass: four neighbouring source columns per item (one 16-byte load per tap), items = rows x quads over all lanes */
    struct __attribute__ ((aligned (4))) W4 { uint32_t v[4]; };
    const int nq = g.pitch >> 2, n = nq * g.rows;
    for (int i = tid; i < n; i += nthreads) {
      const int r = i / nq, c = (i - r * nq) * 4, sy = g.sy0 + r;
      uint32_t *out = lds + r * g.pitch + c;
      if (sp.sv.kind == SCALE_NTAP && sp.sv.n_taps == 8 && g.lo + c + 4 <= src_w) {
        const Taps8 t8 = taps8_load (sp.sv.taps + (size_t) sy * 8);
        const uint8_t *q = img.p + (size_t) sp.sv.offset[sy] * img.stride + 4 * (size_t) (g.lo + c);
        PkAcc a[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
        W4 w[8];
#pragma unroll
        for (int l = 0; l < 8; l++)
          w[l] = *(const W4 *) (q + (size_t) l * img.stride);
#define GSTAMD_V8(L) { const uint32_t ts = taps8_splat<L> (t8); for (int k = 0; k < 4; k++) a[k].mad (w[L].v[k], ts); }
        GSTAMD_V8 (0) GSTAMD_V8 (1) GSTAMD_V8 (2) GSTAMD_V8 (3) GSTAMD_V8 (4) GSTAMD_V8 (5) GSTAMD_V8 (6) GSTAMD_V8 (7)
#undef GSTAMD_V8
#pragma unroll
        for (int k = 0; k < 4; k++)
          out[k] = a[k].finish ();
      } else if (sp.sv.kind == SCALE_NTAP && g.lo + c + 4 <= src_w) {
        const int16_t *t = sp.sv.taps + (size_t) sy * sp.sv.n_taps;
        const uint8_t *q = img.p + (size_t) sp.sv.offset[sy] * img.stride + 4 * (size_t) (g.lo + c);
        PkAcc a[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
#pragma unroll 4
        for (int l = 0; l < sp.sv.n_taps; l++) {
          const W4 w = *(const W4 *) (q + (size_t) l * img.stride);
          const uint32_t ts = tap_splat (t[l]);
#pragma unroll
          for (int k = 0; k < 4; k++)
            a[k].mad (w.v[k], ts);
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
          out[k] = a[k].finish ();
      } else {
        for (int k = 0; k < 4 && c + k < g.w; k++)
          out[k] = vscale_px (img, sp.sv, g.lo + c + k, sy);
      }
    }
    return;
  }
  const int n = g.w * g.rows;
  for (int i = tid; i < n; i += nthreads) {
    const int r = i / g.w, c = i - r * g.w, sx = g.sx0 + c;
    if (sp.sh.kind == SCALE_NTAP && sp.sh.n_taps == 8) {
      struct __attribute__ ((aligned (4))) W4 { uint32_t v[4]; };
      const Taps8 t8 = taps8_load (sp.sh.taps + (size_t) sx * 8);
      const uint32_t *q = (const uint32_t *) (img.p + (size_t) (g.lo + r) * img.stride) + sp.sh.offset[sx];
      const W4 qa = *(const W4 *) q, qb = *(const W4 *) (q + 4);
      PkAcc a = {0, 0};
      a.mad (qa.v[0], taps8_splat<0> (t8)), a.mad (qa.v[1], taps8_splat<1> (t8)), a.mad (qa.v[2], taps8_splat<2> (t8)), a.mad (qa.v[3], taps8_splat<3> (t8));
      a.mad (qb.v[0], taps8_splat<4> (t8)), a.mad (qb.v[1], taps8_splat<5> (t8)), a.mad (qb.v[2], taps8_splat<6> (t8)), a.mad (qb.v[3], taps8_splat<7> (t8));
      lds[i] = a.finish ();
    } else if (sp.sh.kind == SCALE_NTAP) {
      const int16_t *t = sp.sh.taps + (size_t) sx * sp.sh.n_taps;
      const uint32_t *q = (const uint32_t *) (img.p + (size_t) (g.lo + r) * img.stride) + sp.sh.offset[sx];
      PkAcc a = {0, 0};
#pragma unroll 8
      for (int l = 0; l < sp.sh.n_taps; l++)
        a.mad (q[l], tap_splat (t[l]));
      lds[i] = a.finish ();
    } else {
      const RowOfSrc<PadImage> row = {img, g.lo + r};
      lds[i] = hscale_px (row, sp.sh, sx);
    }
  }
}

struct TileCols {       // mode 3: the horizontally scaled rows [lo, lo + rows) of the tile's columns
  const uint32_t *lds;
  int lo, w, sx0;
  GSTAMD_HD uint32_t at (int x, int y) const { return lds[(y - lo) * w + (x - sx0)]; }
};

// second pass for pad pixel (sx, sy) from LDS
GSTAMD_HD uint32_t scaled_tile_px (const ScaledPadDev &sp, const ScaledTileGeom &g, const uint32_t *lds, int sx, int sy)
{
  if (g.mode == 2) {
    const uint32_t *rowp = lds + (sy - g.sy0) * g.pitch;
    if (sp.sh.kind == SCALE_NTAP && sp.sh.n_taps == 8) {
      const Taps8 t8 = taps8_load (sp.sh.taps + (size_t) sx * 8);
      const uint32_t *q = rowp + ((int) sp.sh.offset[sx] - g.lo);
      PkAcc a = {0, 0};
      a.mad (q[0], taps8_splat<0> (t8)), a.mad (q[1], taps8_splat<1> (t8)), a.mad (q[2], taps8_splat<2> (t8)), a.mad (q[3], taps8_splat<3> (t8));
      a.mad (q[4], taps8_splat<4> (t8)), a.mad (q[5], taps8_splat<5> (t8)), a.mad (q[6], taps8_splat<6> (t8)), a.mad (q[7], taps8_splat<7> (t8));
      return a.finish ();
    }
    if (sp.sh.kind == SCALE_NTAP) {
      const int16_t *t = sp.sh.taps + (size_t) sx * sp.sh.n_taps;
      const uint32_t *q = rowp + ((int) sp.sh.offset[sx] - g.lo);
      PkAcc a = {0, 0};
#pragma unroll 8
      for (int l = 0; l < sp.sh.n_taps; l++)
        a.mad (q[l], tap_splat (t[l]));
      return a.finish ();
    }
    const RowOfLds row = {rowp, g.lo};
    return hscale_px (row, sp.sh, sx);
  }
  if (sp.sv.kind == SCALE_NTAP && sp.sv.n_taps == 8) {
    const Taps8 t8 = taps8_load (sp.sv.taps + (size_t) sy * 8);
    const uint32_t *q = lds + ((int) sp.sv.offset[sy] - g.lo) * g.w + (sx - g.sx0);
    PkAcc a = {0, 0};
    a.mad (q[0], taps8_splat<0> (t8)), a.mad (q[g.w], taps8_splat<1> (t8)), a.mad (q[2 * g.w], taps8_splat<2> (t8)), a.mad (q[3 * g.w], taps8_splat<3> (t8));
    a.mad (q[4 * g.w], taps8_splat<4> (t8)), a.mad (q[5 * g.w], taps8_splat<5> (t8)), a.mad (q[6 * g.w], taps8_splat<6> (t8)), a.mad (q[7 * g.w], taps8_splat<7> (t8));
    return a.finish ();
  }
  if (sp.sv.kind == SCALE_NTAP) {
    const int16_t *t = sp.sv.taps + (size_t) sy * sp.sv.n_taps;
    const uint32_t *q = lds + ((int) sp.sv.offset[sy] - g.lo) * g.w + (sx - g.sx0);
    PkAcc a = {0, 0};
#pragma unroll 8
    for (int l = 0; l < sp.sv.n_taps; l++)
      a.mad (q[l * g.w], tap_splat (t[l]));
    return a.finish ();
  }
  const TileCols cols = {lds, g.lo, g.w, g.sx0};
  return vscale_px (cols, sp.sv, sx, sy);
}

}  // namespace gstamd
