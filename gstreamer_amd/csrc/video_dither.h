// video_dither.h - the dither stage of GstVideoConverter (chain_dither video-converter.c:2035-2100, do_dither_lines :3155-3175) for
// 8-bit chains: GST_VIDEO_DITHER_BAYER with GST_VIDEO_DITHER_FLAG_QUANTIZE (video-dither.c dither_ordered_u8_mask :262-270,
// video_orc_dither_ordered_4u8_mask video-orc.orc:2913-2924): per component  p = c + (bayer[y & 15][x & 15] >> (8 - shift));
// p &= ~((1 << shift) - 1);  c = min (p, 255)  - x counted from the left edge of the converted rectangle, y the line of the destination FRAME
// (DitherParams::y0: do_dither_lines passes out_line = i + out_y).
// Run as a pass over the packed 4-byte destination (planner.h DitherParams: why that is the same thing).
#pragma once
#include <stdint.h>

#include "video_device.h"

namespace gstamd {

// the ordered-dither matrix of the reference, value for value (video-dither.c:234-251; it is not the textbook 16 x 16 Bayer matrix:
// rows 4, 7, 12 and 15 carry entries of their own)
GSTAMD_HD int dither_bayer_value (int x, int y)
{
  const uint8_t m[16][16] = {
    {0, 128, 32, 160, 8, 136, 40, 168, 2, 130, 34, 162, 10, 138, 42, 170},
    {192, 64, 224, 96, 200, 72, 232, 104, 194, 66, 226, 98, 202, 74, 234, 106},
    {48, 176, 16, 144, 56, 184, 24, 152, 50, 178, 18, 146, 58, 186, 26, 154},
    {240, 112, 208, 80, 248, 120, 216, 88, 242, 114, 210, 82, 250, 122, 218, 90},
    {12, 240, 44, 172, 4, 132, 36, 164, 14, 242, 46, 174, 6, 134, 38, 166},
    {204, 76, 236, 108, 196, 68, 228, 100, 206, 78, 238, 110, 198, 70, 230, 102},
    {60, 188, 28, 156, 52, 180, 20, 148, 62, 190, 30, 158, 54, 182, 22, 150},
    {252, 142, 220, 92, 244, 116, 212, 84, 254, 144, 222, 94, 246, 118, 214, 86},
    {3, 131, 35, 163, 11, 139, 43, 171, 1, 129, 33, 161, 9, 137, 41, 169},
    {195, 67, 227, 99, 203, 75, 235, 107, 193, 65, 225, 97, 201, 73, 233, 105},
    {51, 179, 19, 147, 59, 187, 27, 155, 49, 177, 17, 145, 57, 185, 25, 153},
    {243, 115, 211, 83, 251, 123, 219, 91, 241, 113, 209, 81, 249, 121, 217, 89},
    {15, 243, 47, 175, 7, 135, 39, 167, 13, 241, 45, 173, 5, 133, 37, 165},
    {207, 79, 239, 111, 199, 71, 231, 103, 205, 77, 237, 109, 197, 69, 229, 101},
    {63, 191, 31, 159, 55, 183, 23, 151, 61, 189, 29, 157, 53, 181, 21, 149},
    {255, 145, 223, 95, 247, 119, 215, 87, 253, 143, 221, 93, 245, 117, 213, 85},
  };
  return m[y & 15][x & 15];
}

// the eight matrix values of row y from column x8 (a multiple of 8) on, one byte each: one 8-byte read instead of eight lookups
GSTAMD_HD uint2 dither_bayer_row8 (int x8, int y)
{
  const uint32_t m[16][4] __attribute__ ((aligned (16))) = {
#define B4(a, b, c, d) ((uint32_t) (a) | ((uint32_t) (b) << 8) | ((uint32_t) (c) << 16) | ((uint32_t) (d) << 24))
    {B4 (0, 128, 32, 160), B4 (8, 136, 40, 168), B4 (2, 130, 34, 162), B4 (10, 138, 42, 170)},
    {B4 (192, 64, 224, 96), B4 (200, 72, 232, 104), B4 (194, 66, 226, 98), B4 (202, 74, 234, 106)},
    {B4 (48, 176, 16, 144), B4 (56, 184, 24, 152), B4 (50, 178, 18, 146), B4 (58, 186, 26, 154)},
    {B4 (240, 112, 208, 80), B4 (248, 120, 216, 88), B4 (242, 114, 210, 82), B4 (250, 122, 218, 90)},
    {B4 (12, 240, 44, 172), B4 (4, 132, 36, 164), B4 (14, 242, 46, 174), B4 (6, 134, 38, 166)},
    {B4 (204, 76, 236, 108), B4 (196, 68, 228, 100), B4 (206, 78, 238, 110), B4 (198, 70, 230, 102)},
    {B4 (60, 188, 28, 156), B4 (52, 180, 20, 148), B4 (62, 190, 30, 158), B4 (54, 182, 22, 150)},
    {B4 (252, 142, 220, 92), B4 (244, 116, 212, 84), B4 (254, 144, 222, 94), B4 (246, 118, 214, 86)},
    {B4 (3, 131, 35, 163), B4 (11, 139, 43, 171), B4 (1, 129, 33, 161), B4 (9, 137, 41, 169)},
    {B4 (195, 67, 227, 99), B4 (203, 75, 235, 107), B4 (193, 65, 225, 97), B4 (201, 73, 233, 105)},
    {B4 (51, 179, 19, 147), B4 (59, 187, 27, 155), B4 (49, 177, 17, 145), B4 (57, 185, 25, 153)},
    {B4 (243, 115, 211, 83), B4 (251, 123, 219, 91), B4 (241, 113, 209, 81), B4 (249, 121, 217, 89)},
    {B4 (15, 243, 47, 175), B4 (7, 135, 39, 167), B4 (13, 241, 45, 173), B4 (5, 133, 37, 165)},
    {B4 (207, 79, 239, 111), B4 (199, 71, 231, 103), B4 (205, 77, 237, 109), B4 (197, 69, 229, 101)},
    {B4 (63, 191, 31, 159), B4 (55, 183, 23, 151), B4 (61, 189, 29, 157), B4 (53, 181, 21, 149)},
    {B4 (255, 145, 223, 95), B4 (247, 119, 215, 87), B4 (253, 143, 221, 93), B4 (245, 117, 213, 85)},
#undef B4
  };
  const int c = (x8 >> 2) & 2;
  uint2 r;
  r.x = m[y & 15][c];
  r.y = m[y & 15][c + 1];
  return r;
}

GSTAMD_HD uint32_t dither_px (const DitherParams &d, uint32_t px, int x, int y)
{
  const int b = dither_bayer_value (x, y + d.y0);
  uint32_t r = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int sh = d.shift[k];
    const int v = sh < 8 ? b >> (8 - sh) : b;
    int p = (int) ((px >> (8 * k)) & 0xffu) + v;                /* addw */
    p &= ~((1 << sh) - 1) & 0xffff;                             /* andnw with the 16-bit mask */
    r |= (uint32_t) (p > 255 ? 255 : p) << (8 * k);             /* convsuswb */
  }
  return r;
}

// pixels x0 .. x0+3 of row y of the rectangle, in place
GSTAMD_HD void dither_lane4 (const DitherParams &d, uint8_t *img, int stride, int w, int h, int x0, int y)
{
  if (x0 >= w || y >= h)
    return;
  uint32_t *row = (uint32_t *) (img + (size_t) y * stride);
  for (int i = 0; i < 4 && x0 + i < w; i++)
    row[x0 + i] = dither_px (d, row[x0 + i], x0 + i, y);
}

}  // namespace gstamd
