// video_testsrc.h - GstVideoTestSrc's frame painters (gst/videotestsrc/videotestsrc.c) as a function of the pixel (round 6; SURVEY 8 f4: frames that are
// born in HBM, so that a pipeline's source is not a PCIe upload).
//
// The reference paints one line of A, c1, c2, c3 bytes at a time (paint_tmpline_ARGB / _AYUV :1590-1624 for RGB / every other format), runs the line
// through the caps' chroma downsampler and packs it with the format's own pack function (convert_hline_generic :1626-1683).  Here the painted lines of a
// whole frame are one image made by k_test_pattern (a lane per pixel; every painter below is a closed form of x, y and the frame number), and the rest is
// the generic chain of this library from AYUV / ARGB into the caps' format (chroma downsampler of the caps' chroma-site + pack, no matrix, no dither:
// video_testsrc.hip) - the same two library calls the reference makes (gst_video_chroma_resample, finfo->pack_func).
//
// Painters built: smpte, snow, black, white, red, green, blue, checkers-1 / -2 / -4 / -8, blink, smpte75, smpte100, solid-color, bar, gradient, colors, ball
// (motion wavy, animation-mode frames: the element's defaults).  Not built (refused): circular, zone-plate, chroma-zone-plate, gamut, pinwheel, spokes,
// smpte-rp-219, horizontal-speed, the Bayer formats.
#pragma once
#include <math.h>
#include <stdint.h>

#include "planner.h"

namespace gstamd {

// GstVideoTestSrcPattern (gstvideotestsrc.h:84-112): the values of the element's `pattern` property
enum TestPattern : int {
  TESTSRC_SMPTE = 0, TESTSRC_SNOW = 1, TESTSRC_BLACK = 2, TESTSRC_WHITE = 3, TESTSRC_RED = 4, TESTSRC_GREEN = 5, TESTSRC_BLUE = 6, TESTSRC_CHECKERS1 = 7,
  TESTSRC_CHECKERS2 = 8, TESTSRC_CHECKERS4 = 9, TESTSRC_CHECKERS8 = 10, TESTSRC_CIRCULAR = 11, TESTSRC_BLINK = 12, TESTSRC_SMPTE75 = 13, TESTSRC_ZONE_PLATE = 14,
  TESTSRC_GAMUT = 15, TESTSRC_CHROMA_ZONE_PLATE = 16, TESTSRC_SOLID = 17, TESTSRC_BALL = 18, TESTSRC_SMPTE100 = 19, TESTSRC_BAR = 20, TESTSRC_PINWHEEL = 21,
  TESTSRC_SPOKES = 22, TESTSRC_GRADIENT = 23, TESTSRC_COLORS = 24, TESTSRC_SMPTE_RP_219 = 25
};

struct TestPatternParams {
  int pattern, w, h;
  uint32_t colors[12];          // vts_colors_*_100 (:59-155) as painted: A | c1 << 8 | c2 << 16 | c3 << 24 (c = Y, U, V or R, G, B)
  uint32_t colors75[8];         // vts_colors_*_75
  uint32_t fg, bg;              // foreground-color / background-color the same way (videotestsrc_setup_paintinfo :205-287)
  uint32_t rand_state;          // v->random_state when the frame begins (random_char :38-43: an LCG that runs on from frame to frame)
  int odd_frame;                // n_frames & 1 (blink)
  double ball_x, ball_y;        // the ball's centre (gst_video_test_src_ball :1467-1588; sin / sqrt on the host, like the reference)
  int ball_radius;
};

// n steps of state = state * 1103515245 + 12345 at once: the affine map composed with itself by squaring
GSTAMD_VP uint32_t testsrc_lcg_skip (uint32_t state, uint32_t n)
{
  uint32_t a = 1103515245u, c = 12345u, ra = 1u, rc = 0u;
  while (n) {
    if (n & 1u) {
      ra = ra * a;
      rc = rc * a + c;
    }
    c = c * a + c;
    a = a * a;
    n >>= 1;
  }
  return state * ra + rc;
}
// the k-th random_char () after `state` (k = 0: the first)
GSTAMD_VP int testsrc_random (uint32_t state, uint32_t k) { return (int) ((testsrc_lcg_skip (state, k + 1u) >> 16) & 0xffu); }

// BLEND (a, b, x) = DIV255 (a x + b (255 - x)) on the four bytes (videotestsrc_blend_line :357-378)
GSTAMD_VP uint32_t testsrc_blend (uint32_t a, uint32_t b, int x)
{
  uint32_t r = 0;
  for (int k = 0; k < 4; k++) {
    const int va = (int) ((a >> (8 * k)) & 0xffu), vb = (int) ((b >> (8 * k)) & 0xffu);
    const int v = va * x + vb * (255 - x);
    r |= (uint32_t) (((v + ((v + 128) >> 8) + 128) >> 8) & 0xff) << (8 * k);
  }
  return r;
}

// the seven bars of a line (gst_video_test_src_smpte :395-405 & co: bar i covers [i w / 7, (i + 1) w / 7))
GSTAMD_VP int testsrc_bar7 (int x, int w)
{
  int i = 0;
  while (i < 6 && x >= (i + 1) * w / 7)
    i++;
  return i;
}

GSTAMD_VP uint32_t testsrc_smpte_px (const TestPatternParams &p, int x, int y)
{
  const int w = p.w, h = p.h, y1 = 2 * h / 3, y2 = 3 * h / 4;
  const int i7 = testsrc_bar7 (x, w);
  const uint32_t band1 = p.colors[i7], band2 = p.colors[(i7 & 1) ? 7 : 6 - i7];
  if (y < y1)
    return band1;
  if (y < y2)
    return band2;
  /* the lowest band paints [0, w / 2) in three runs, [w / 2, w / 2 + 3 (w / 12)) in three more and noise from 3 w / 4 on: a pixel none of them covers
     keeps what the line buffer held - the band above it */
  uint32_t v = y2 > y1 ? band2 : (y1 > 0 ? band1 : 0u);
  for (int i = 0; i < 3; i++)
    if (x >= i * w / 6 && x < (i + 1) * w / 6)
      v = p.colors[i == 0 ? 8 : (i == 1 ? 0 : 9)];          /* -I, white, +Q */
  for (int i = 0; i < 3; i++)
    if (x >= w / 2 + i * w / 12 && x < w / 2 + (i + 1) * w / 12)
      v = p.colors[i == 0 ? 10 : (i == 1 ? 7 : 11)];        /* super black, black, dark grey */
  const int x1 = w * 3 / 4;
  if (x >= x1)
    v = testsrc_blend (p.fg, p.bg, testsrc_random (p.rand_state, (uint32_t) (y - y2) * (uint32_t) (w - x1) + (uint32_t) (x - x1)));
  return v;
}

// random_char () calls of one frame (the state the next frame starts from)
GSTAMD_VP uint32_t testsrc_draws_per_frame (int pattern, int w, int h)
{
  if (pattern == TESTSRC_SNOW)
    return (uint32_t) w * (uint32_t) h;
  if (pattern == TESTSRC_SMPTE)
    return (uint32_t) (h - 3 * h / 4) * (uint32_t) (w - w * 3 / 4);
  return 0u;
}

#if defined(__HIPCC__)
#pragma clang fp contract(off)
#endif
// alpha of the ball at (x, y) (gst_video_test_src_ball :1524-1560: doubles throughout; the int conversions truncate like the C assignments)
GSTAMD_VP int testsrc_ball_alpha (const TestPatternParams &p, int x, int y)
{
  const double bx = p.ball_x, by = p.ball_y;
  const int radius = p.ball_radius;
  if ((double) y < by - radius || (double) y > by + radius)
    return 0;
  const double dy = (double) y - by;
  double o = (double) (radius * radius) - dy * dy;
  if (o < 0)
    o = 0;
  const int r = (int) rint (sqrt (o));
  const double lo = bx - r > 0 ? bx - r : 0, hi = (double) p.w < bx + r + 1 ? (double) p.w : bx + r + 1;
  const int x1 = (int) lo, x2 = (int) hi;
  if (x < x1 || x >= x2)
    return 0;
  const double dx = (double) x - bx;
  double rr = (double) radius - sqrt (dx * dx + dy * dy);
  rr *= 0.5;
  const int a = (int) floor (256 * rr);
  return a < 0 ? 0 : (a > 255 ? 255 : a);
}

GSTAMD_VP uint32_t test_pattern_px (const TestPatternParams &p, int x, int y)
{
  switch (p.pattern) {
    case TESTSRC_SMPTE:
      return testsrc_smpte_px (p, x, y);
    case TESTSRC_SNOW:
      return testsrc_blend (p.fg, p.bg, testsrc_random (p.rand_state, (uint32_t) y * (uint32_t) p.w + (uint32_t) x));
    case TESTSRC_BLACK:
      return p.bg;              /* gst_video_test_src_unicolor :935-940: black is the background colour, white the foreground colour */
    case TESTSRC_WHITE:
      return p.fg;
    case TESTSRC_RED:
      return p.colors[5];
    case TESTSRC_GREEN:
      return p.colors[3];
    case TESTSRC_BLUE:
      return p.colors[6];
    case TESTSRC_CHECKERS1:
      return p.colors[((x ^ y) & 1) ? 3 : 5];
    case TESTSRC_CHECKERS2:
      return p.colors[((x ^ y) & 2) ? 3 : 5];
    case TESTSRC_CHECKERS4:
      return p.colors[((x ^ y) & 4) ? 3 : 5];
    case TESTSRC_CHECKERS8:
      return p.colors[((x ^ y) & 8) ? 3 : 5];
    case TESTSRC_BLINK:
      return p.odd_frame ? p.fg : p.bg;
    case TESTSRC_SMPTE75:
      return p.colors75[testsrc_bar7 (x, p.w)];
    case TESTSRC_SMPTE100:
      return p.colors[testsrc_bar7 (x, p.w)];
    case TESTSRC_SOLID:
      return p.fg;
    case TESTSRC_BAR:
      return x < p.w / 7 ? p.fg : p.bg;
    case TESTSRC_GRADIENT:
      return testsrc_blend (p.fg, p.bg, (int) ((double) y * 255.0 / (double) p.h));
    case TESTSRC_COLORS:        /* (:1880-1885: the four bytes as they are, RGB or YUV) */
      return 0xffu | ((uint32_t) (((x * 4096) / p.w) % 256) << 8) | ((uint32_t) ((((y * 16) / p.h) << 4) | ((x * 16) / p.w)) << 16) |
          ((uint32_t) (((y * 4096) / p.h) % 256) << 24);
    case TESTSRC_BALL:
      return testsrc_blend (p.fg, p.bg, testsrc_ball_alpha (p, x, y));
    default:
      return 0;
  }
}

// does this library paint the pattern?
GSTAMD_VP bool test_pattern_built (int pattern)
{
  switch (pattern) {
    case TESTSRC_CIRCULAR: case TESTSRC_ZONE_PLATE: case TESTSRC_GAMUT: case TESTSRC_CHROMA_ZONE_PLATE: case TESTSRC_PINWHEEL: case TESTSRC_SPOKES:
    case TESTSRC_SMPTE_RP_219:
      return false;
    default:
      return pattern >= 0 && pattern <= TESTSRC_COLORS;
  }
}

// host side: the colours as painted and the per-frame values (video_testsrc.hip, tests/emu)
void test_pattern_setup (TestPatternParams *p, const GstAmdVideoInfo *info, int pattern, uint32_t foreground_argb, uint32_t background_argb);
void test_pattern_frame (TestPatternParams *p, uint64_t n_frames);
// the info of the painted image (AYUV / ARGB, the caps' size and colorimetry) and the options of its conversion into the caps' format
void test_pattern_conversion (const GstAmdVideoInfo *info, GstAmdVideoInfo *painted, GstAmdVideoConverterConfig *cfg);

}  // namespace gstamd
