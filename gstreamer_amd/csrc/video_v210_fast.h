// video_v210_fast.h - the reference's OWN v210 fastpaths between v210 and the 8-bit 4:2:0 / 4:2:2 formats (video-converter.c):
//   convert_I420_v210 :3645-3800 (I420, YV12)   convert_Y42B_v210 :6115-6230   convert_YUY2_v210 :4431-4545   convert_UYVY_v210 :5045-5160
//   convert_v210_I420 :4031-4199                convert_v210_Y42B :4665-4787   convert_v210_YUY2 :5317-5428   convert_v210_UYVY :5204-5315
// Their arithmetic is not the chain's: an 8-bit sample becomes the 10-bit one shifted left by two (no bit replication), a 10-bit sample the 8-bit one
// shifted right by two; 4:2:0 chroma is the pair's row for both lines on the way in and (u1 + u2) / 2 of the two SHIFTED values, truncated, on the
// way out; samples of the last group past the line's end are 0 (to v210) or left alone (from v210).  The odd last line of a 4:2:0 frame goes
// through unpack + pack of that one line in the reference (:3789-3799, 4187-4198), which is the same formulas with the line's own chroma row.
// Whole frames only (the planner refuses crops and rectangles: the 4:2:2 functions offset v210 rows by ROUND_UP_2 (x) * 2 bytes, :4503-4506).
// One lane = one group of six pixels of one line (4:2:2) or of a line pair (4:2:0).
#pragma once
#include "video_device.h"

namespace gstamd {

struct V210FastParams {
  int to_v210;          // 1: 8-bit -> v210, 0: v210 -> 8-bit
  int kind;             // the 8-bit side: UNPACK_PLANAR or UNPACK_PACKED422
  int h_sub;            // 1: its chroma rows serve two lines (I420 / YV12)
  int pos[4];           // packed 4:2:2: byte of Y0, U, V in the macropixel (FormatDesc::pos[1..3])
  int u_plane, v_plane;
  int width, height;
  const uint8_t *s[3];  // the 8-bit side's planes / the v210 plane in s[0] (from v210)
  int sstride[3];
  uint8_t *d[3];
  int dstride[3];
};

GSTAMD_VP int v210_fast_groups (const V210FastParams &p) { return (p.width + 5) / 6; }
GSTAMD_VP int v210_fast_rows (const V210FastParams &p) { return p.h_sub ? (p.height + 1) / 2 : p.height; }

GSTAMD_HD void v210_fast_body (const V210FastParams &p, int g, int r)
{
  if (g >= v210_fast_groups (p) || r >= v210_fast_rows (p))
    return;
  const int j = 6 * g, w = p.width;
  const int l0 = p.h_sub ? 2 * r : r, nl = p.h_sub && l0 + 1 < p.height ? 2 : 1;
  const bool packed = p.kind == UNPACK_PACKED422;
  if (p.to_v210) {
    for (int t = 0; t < nl; t++) {
      const int l = l0 + t, crow = p.h_sub ? r : l;
      const uint8_t *sy = p.s[0] + (size_t) l * p.sstride[0];
      const uint8_t *su = packed ? sy : p.s[p.u_plane] + (size_t) crow * p.sstride[p.u_plane];
      const uint8_t *sv = packed ? sy : p.s[p.v_plane] + (size_t) crow * p.sstride[p.v_plane];
      uint32_t y[6], u[3], v[3];
      for (int k = 0; k < 6; k++)
        y[k] = (k == 0 || j < w - k) ? (packed ? sy[4 * ((j + k) >> 1) + p.pos[1] + 2 * ((j + k) & 1)] : sy[j + k]) : 0u;
      for (int k = 0; k < 3; k++) {
        const bool in = k == 0 || j < w - 2 * k;
        u[k] = in ? (packed ? su[4 * (j / 2 + k) + p.pos[2]] : su[j / 2 + k]) : 0u;
        v[k] = in ? (packed ? sv[4 * (j / 2 + k) + p.pos[3]] : sv[j / 2 + k]) : 0u;
      }
      uint32_t *d = (uint32_t *) (p.d[0] + (size_t) l * p.dstride[0]) + 4 * g;
      d[0] = (u[0] << 2) | (y[0] << 12) | (v[0] << 22);
      d[1] = (y[1] << 2) | (u[1] << 12) | (y[2] << 22);
      d[2] = (v[1] << 2) | (y[3] << 12) | (u[2] << 22);
      d[3] = (y[4] << 2) | (v[2] << 12) | (y[5] << 22);
    }
    return;
  }
  uint32_t y[2][6], u[2][3], v[2][3];
  for (int t = 0; t < nl; t++) {
    const uint32_t *a = (const uint32_t *) (p.s[0] + (size_t) (l0 + t) * p.sstride[0]) + 4 * g;
    const uint32_t a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
    u[t][0] = ((a0 >> 0) & 0x3ffu) >> 2, y[t][0] = ((a0 >> 10) & 0x3ffu) >> 2, v[t][0] = ((a0 >> 20) & 0x3ffu) >> 2;
    y[t][1] = ((a1 >> 0) & 0x3ffu) >> 2, u[t][1] = ((a1 >> 10) & 0x3ffu) >> 2, y[t][2] = ((a1 >> 20) & 0x3ffu) >> 2;
    v[t][1] = ((a2 >> 0) & 0x3ffu) >> 2, y[t][3] = ((a2 >> 10) & 0x3ffu) >> 2, u[t][2] = ((a2 >> 20) & 0x3ffu) >> 2;
    y[t][4] = ((a3 >> 0) & 0x3ffu) >> 2, v[t][2] = ((a3 >> 10) & 0x3ffu) >> 2, y[t][5] = ((a3 >> 20) & 0x3ffu) >> 2;
  }
  for (int t = 0; t < nl; t++) {
    uint8_t *dy = p.d[0] + (size_t) (l0 + t) * p.dstride[0];
    for (int k = 0; k < 6; k++)
      if (k == 0 || j < w - k) {
        if (packed)
          dy[4 * ((j + k) >> 1) + p.pos[1] + 2 * ((j + k) & 1)] = (uint8_t) y[t][k];
        else
          dy[j + k] = (uint8_t) y[t][k];
      }
  }
  for (int t = 0; t < (p.h_sub ? 1 : nl); t++) {
    const int l = l0 + t, crow = p.h_sub ? r : l;
    uint8_t *du = packed ? p.d[0] + (size_t) l * p.dstride[0] : p.d[p.u_plane] + (size_t) crow * p.dstride[p.u_plane];
    uint8_t *dv = packed ? p.d[0] + (size_t) l * p.dstride[0] : p.d[p.v_plane] + (size_t) crow * p.dstride[p.v_plane];
    for (int k = 0; k < 3; k++) {
      if (!(k == 0 || j < w - 2 * k))
        continue;
      const uint32_t cu = p.h_sub && nl == 2 ? (u[0][k] + u[1][k]) / 2 : u[t][k], cv = p.h_sub && nl == 2 ? (v[0][k] + v[1][k]) / 2 : v[t][k];
      if (packed)
        du[4 * (j / 2 + k) + p.pos[2]] = (uint8_t) cu, dv[4 * (j / 2 + k) + p.pos[3]] = (uint8_t) cv;
      else
        du[j / 2 + k] = (uint8_t) cu, dv[j / 2 + k] = (uint8_t) cv;
    }
  }
}

}  // namespace gstamd
