// video_dither_ed.h - the error-diffusion methods of the dither stage on 8-bit lines: GST_VIDEO_DITHER_VERTERR, _FLOYD_STEINBERG and
// _SIERRA_LITE (video-dither.c dither_verterr_u8 :75-87, dither_floyd_steinberg_u8 :117-151, dither_sierra_lite_u8 :182-206; ORC programs
// video_orc_dither_verterr_4u8_mask video-orc.orc:2859-2871 and video_orc_dither_fs_muladd_u8 :2873-2883).
//
// The reference keeps ONE line of 16-bit errors, slot k (4 components each) holding the error of pixel k - 1 of the line dithered
// last (verterr: of pixel k), cleared when the frame's line 0 comes by.  Restated per pixel x of a line, with P(j) the error the
// previous line left at its pixel j (0 for the first line and for j >= width) and L the error of this line's pixel x - 1:
//
//   verterr      v = p + P(x)                                      (16-bit add)
//   sierra-lite  v = p + ((2 L + P(x+1) + P(x+2)) >> 2)            L = 0 at x = 0
//   floyd-st.    v = p + ((7 L + X) >> 4)                          X = P(x) + 5 P(x+1) + 3 P(x+2) (16-bit, the muladd pass), but the bare
//                                                                  P(x) for the line's last pixel (the pass covers slots 0 .. width-1 only);
//                L at x = 0 is slot 0, which no pixel ever writes and the muladd pass keeps accumulating:
//                A(y) = A(y-1) + 5 P(0) + 3 P(1), A = 0 before line 0
//   all          err = v & mask;  v &= ~mask;  p = min (v, 255)    (v a guint16)
//
// Every component runs its own chain (slot index i & 3), so - as for the ordered method - the pass may run over the packed pixel with
// the shift of the component stored in each byte.
//
// Lines depend on each other through P with a reach of two pixels ahead: line r can run three pixels behind line r - 1.  The kernel
// (video_kernels.hip k_dither_ed) puts one line on each lane of a 1024-lane workgroup and steps them as a skewed wavefront, errors in an
// 4-slot LDS ring per line; frames taller than 1024 lines go band by band with the last line's errors carried through HBM.  Vertical
// error carry needs no wavefront (a lane per column, k_dither_verterr).
#pragma once
#include <stdint.h>

#include "video_device.h"

namespace gstamd {

struct Err4 {
  uint16_t c[4];
};

GSTAMD_HD Err4 err4_zero ()
{
  Err4 e;
  e.c[0] = e.c[1] = e.c[2] = e.c[3] = 0;
  return e;
}

// quantise component k: v is the pixel plus the diffused error as a guint16
GSTAMD_HD uint32_t ed_quantise (const DitherParams &d, int k, uint32_t v, uint16_t *err)
{
  const uint32_t mask = (1u << d.shift[k]) - 1u;
  v &= 0xffffu;
  *err = (uint16_t) (v & mask);
  v &= ~mask;
  return v > 255u ? 255u : v;
}

// dither_verterr_u8: px with the error of the pixel above; err = in: that error, out: this pixel's
GSTAMD_HD uint32_t ed_verterr_px (const DitherParams &d, uint32_t px, Err4 &err)
{
  uint32_t r = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t v = ((px >> (8 * k)) & 0xffu) + err.c[k];                 /* addw */
    /* andnw + convsuswb: the sum is below 0x8000 (a byte plus an error smaller than the mask), so the signed saturation is a min */
    r |= ed_quantise (d, k, v, &err.c[k]) << (8 * k);
  }
  return r;
}

// dither_sierra_lite_u8: left = error of this line's previous pixel (0 at x = 0), p1 / p2 = P(x+1) / P(x+2); left becomes this pixel's error
GSTAMD_HD uint32_t ed_sierra_px (const DitherParams &d, uint32_t px, Err4 &left, const Err4 &p1, const Err4 &p2)
{
  uint32_t r = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t v = ((px >> (8 * k)) & 0xffu) + ((2u * left.c[k] + p1.c[k] + p2.c[k]) >> 2);
    r |= ed_quantise (d, k, v, &left.c[k]) << (8 * k);
  }
  return r;
}

// dither_floyd_steinberg_u8: p0 / p1 / p2 = P(x) / P(x+1) / P(x+2); at x = 0 `left` comes in as the previous line's slot-0 accumulator A(y-1)
// and *a0 receives A(y); left becomes this pixel's error
GSTAMD_HD uint32_t ed_floyd_px (const DitherParams &d, uint32_t px, Err4 &left, const Err4 &p0, const Err4 &p1, const Err4 &p2, bool first, bool last, Err4 *a0)
{
  uint32_t r = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint16_t m = (uint16_t) (p0.c[k] + (uint16_t) (5u * p1.c[k]) + (uint16_t) (3u * p2.c[k]));          /* mullw / addw: 16-bit */
    uint32_t l = left.c[k];
    if (first) {
      l = (uint16_t) (l + (uint16_t) (5u * p0.c[k]) + (uint16_t) (3u * p1.c[k]));          /* slot 0 after the muladd pass */
      a0->c[k] = (uint16_t) l;
    }
    const uint32_t x = last ? p0.c[k] : m;
    const uint32_t v = ((px >> (8 * k)) & 0xffu) + ((7u * l + x) >> 4);
    r |= ed_quantise (d, k, v, &left.c[k]) << (8 * k);
  }
  return r;
}

// ---- the same methods on 16-bit lines (round 5): dither_verterr_u16 (video-dither.c:89-115), dither_floyd_steinberg_u16 (:153-180),
// dither_sierra_lite_u16 (:208-232) ahead of the 10 / 12 / 16-bit packers and of pack_ARGB64 / pack_AYUV64.  Plain C there, 32-bit sums:
//   verterr      v = p + P(x)
//   sierra-lite  v = p + ((2 L + P(x+1) + P(x+2)) >> 2)
//   floyd-st.    v = p + ((7 L + P(x) + 5 P(x+1) + 3 P(x+2)) >> 4)     (one expression: no muladd pass, slot 0 stays 0 - L = 0 at x = 0)
//   all          err = v & mask;  v &= ~mask;  p = min (v, 65535)
// A pixel is 4 x u16 (A, Y, U, V / A, R, G, B in unpack order), DitherParams::shift[] per component.
struct Px16 {
  uint32_t lo, hi;              // components 0, 1 / 2, 3 (16 bits each, little-endian)
};

GSTAMD_HD uint32_t ed16_quantise (const DitherParams &d, int k, uint32_t v, uint16_t *err)
{
  const uint32_t mask = (1u << d.shift[k]) - 1u;
  *err = (uint16_t) (v & mask);
  v &= ~mask;
  return v > 65535u ? 65535u : v;
}

GSTAMD_HD uint32_t ed16_comp (const Px16 &p, int k) { return k < 2 ? (p.lo >> (16 * k)) & 0xffffu : (p.hi >> (16 * (k - 2))) & 0xffffu; }

GSTAMD_HD Px16 ed16_pack (const uint32_t *c)
{
  Px16 r;
  r.lo = c[0] | (c[1] << 16);
  r.hi = c[2] | (c[3] << 16);
  return r;
}

GSTAMD_HD Px16 ed16_verterr_px (const DitherParams &d, const Px16 &px, Err4 &err)
{
  uint32_t c[4];
#pragma unroll
  for (int k = 0; k < 4; k++)
    c[k] = ed16_quantise (d, k, ed16_comp (px, k) + err.c[k], &err.c[k]);
  return ed16_pack (c);
}

GSTAMD_HD Px16 ed16_sierra_px (const DitherParams &d, const Px16 &px, Err4 &left, const Err4 &p1, const Err4 &p2)
{
  uint32_t c[4];
#pragma unroll
  for (int k = 0; k < 4; k++)
    c[k] = ed16_quantise (d, k, ed16_comp (px, k) + ((2u * left.c[k] + p1.c[k] + p2.c[k]) >> 2), &left.c[k]);
  return ed16_pack (c);
}

GSTAMD_HD Px16 ed16_floyd_px (const DitherParams &d, const Px16 &px, Err4 &left, const Err4 &p0, const Err4 &p1, const Err4 &p2)
{
  uint32_t c[4];
#pragma unroll
  for (int k = 0; k < 4; k++)
    c[k] = ed16_quantise (d, k, ed16_comp (px, k) + ((7u * left.c[k] + p0.c[k] + 5u * p1.c[k] + 3u * p2.c[k]) >> 4), &left.c[k]);
  return ed16_pack (c);
}

#ifndef __HIPCC__
// the whole 16-bit rectangle on the host (emulator and tests): prev / cur = errors of the previous / this line per pixel
inline void ed16_image_host (const DitherParams &d, uint8_t *img, int stride, int w, int h)
{
  Err4 *a = new Err4[2 * (size_t) (w + 2)] (), *b = a + (w + 2);
  Err4 *base = a;
  for (int y = 0; y < h; y++) {
    Px16 *row = (Px16 *) (img + (size_t) y * stride);
    Err4 left = err4_zero ();
    for (int x = 0; x < w; x++) {
      if (d.method == GSTAMD_DITHER_VERTERR) {
        left = a[x];
        row[x] = ed16_verterr_px (d, row[x], left);
      } else if (d.method == GSTAMD_DITHER_SIERRA_LITE) {
        row[x] = ed16_sierra_px (d, row[x], left, a[x + 1], a[x + 2]);
      } else {
        row[x] = ed16_floyd_px (d, row[x], left, a[x], a[x + 1], a[x + 2]);
      }
      b[x] = left;
    }
    b[w] = b[w + 1] = err4_zero ();
    Err4 *t = a;
    a = b;
    b = t;
  }
  delete[] base;
}
#endif

GSTAMD_VP bool dither_is_diffusion (const DitherParams &d)
{
  return d.on && (d.method == GSTAMD_DITHER_VERTERR || d.method == GSTAMD_DITHER_FLOYD_STEINBERG || d.method == GSTAMD_DITHER_SIERRA_LITE);
}

#ifndef __HIPCC__
// One line on the host (emulator and tests): prev / cur = errors of the previous / this line per pixel (w + 2 entries, the last two 0),
// a0 = the Floyd-Steinberg slot-0 accumulator (in: previous line's, out: this line's)
inline void ed_line_host (const DitherParams &d, uint32_t *row, int w, const Err4 *prev, Err4 *cur, Err4 *a0)
{
  Err4 left = d.method == GSTAMD_DITHER_FLOYD_STEINBERG ? *a0 : err4_zero ();
  for (int x = 0; x < w; x++) {
    if (d.method == GSTAMD_DITHER_VERTERR) {
      left = prev[x];
      row[x] = ed_verterr_px (d, row[x], left);
    } else if (d.method == GSTAMD_DITHER_SIERRA_LITE) {
      row[x] = ed_sierra_px (d, row[x], left, prev[x + 1], prev[x + 2]);
    } else {
      row[x] = ed_floyd_px (d, row[x], left, prev[x], prev[x + 1], prev[x + 2], x == 0, x == w - 1, a0);
    }
    cur[x] = left;
  }
  cur[w] = cur[w + 1] = err4_zero ();
}

// the whole rectangle on the host
inline void ed_image_host (const DitherParams &d, uint8_t *img, int stride, int w, int h)
{
  Err4 *a = new Err4[2 * (size_t) (w + 2)] (), *b = a + (w + 2);
  Err4 a0 = err4_zero ();
  for (int y = 0; y < h; y++) {
    ed_line_host (d, (uint32_t *) (img + (size_t) y * stride), w, a, b, &a0);
    Err4 *t = a;
    a = b;
    b = t;
  }
  delete[] (a < b ? a : b);
}
#endif

}  // namespace gstamd
