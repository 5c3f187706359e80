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
  int bps;              // bytes per sample of the other side: 1, or 2 for I420_10LE / I422_10LE (convert_I420_10_v210 :3803, _v210_I420_10 :4201,
                        // _I422_10_v210 :6232, _v210_I422_10 :4789): the ten bits as they are - words carrying more than ten bits run into their neighbours'
                        // fields exactly as `u0 | (y0 << 10) | (v0 << 20)` does, except on the odd last line of a 4:2:0 frame, which unpack + pack mask
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
      const bool wide = p.bps == 2;
      /* (the odd last line of a 4:2:0 frame goes through unpack_I420_10LE + pack_v210: ten bits of every word) */
      const uint32_t keep = wide && p.h_sub && nl == 1 ? 0x3ffu : 0xffffu;
      for (int k = 0; k < 6; k++)
        y[k] = (k == 0 || j < w - k) ? (packed ? sy[4 * ((j + k) >> 1) + p.pos[1] + 2 * ((j + k) & 1)] : wide ? ((const uint16_t *) sy)[j + k] & keep : sy[j + k]) : 0u;
      for (int k = 0; k < 3; k++) {
        const bool in = k == 0 || j < w - 2 * k;
        u[k] = in ? (packed ? su[4 * (j / 2 + k) + p.pos[2]] : wide ? ((const uint16_t *) su)[j / 2 + k] & keep : su[j / 2 + k]) : 0u;
        v[k] = in ? (packed ? sv[4 * (j / 2 + k) + p.pos[3]] : wide ? ((const uint16_t *) sv)[j / 2 + k] & keep : sv[j / 2 + k]) : 0u;
      }
      uint32_t *d = (uint32_t *) (p.d[0] + (size_t) l * p.dstride[0]) + 4 * g;
      const int s0 = wide ? 0 : 2, s1 = s0 + 10, s2 = s0 + 20;
      d[0] = (u[0] << s0) | (y[0] << s1) | (v[0] << s2);
      d[1] = (y[1] << s0) | (u[1] << s1) | (y[2] << s2);
      d[2] = (v[1] << s0) | (y[3] << s1) | (u[2] << s2);
      d[3] = (y[4] << s0) | (v[2] << s1) | (y[5] << s2);
    }
    return;
  }
  uint32_t y[2][6], u[2][3], v[2][3];
  for (int t = 0; t < nl; t++) {
    const uint32_t *a = (const uint32_t *) (p.s[0] + (size_t) (l0 + t) * p.sstride[0]) + 4 * g;
    const uint32_t a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
    const int dn = p.bps == 2 ? 0 : 2;
    u[t][0] = ((a0 >> 0) & 0x3ffu) >> dn, y[t][0] = ((a0 >> 10) & 0x3ffu) >> dn, v[t][0] = ((a0 >> 20) & 0x3ffu) >> dn;
    y[t][1] = ((a1 >> 0) & 0x3ffu) >> dn, u[t][1] = ((a1 >> 10) & 0x3ffu) >> dn, y[t][2] = ((a1 >> 20) & 0x3ffu) >> dn;
    v[t][1] = ((a2 >> 0) & 0x3ffu) >> dn, y[t][3] = ((a2 >> 10) & 0x3ffu) >> dn, u[t][2] = ((a2 >> 20) & 0x3ffu) >> dn;
    y[t][4] = ((a3 >> 0) & 0x3ffu) >> dn, v[t][2] = ((a3 >> 10) & 0x3ffu) >> dn, y[t][5] = ((a3 >> 20) & 0x3ffu) >> dn;
  }
  for (int t = 0; t < nl; t++) {
    uint8_t *dy = p.d[0] + (size_t) (l0 + t) * p.dstride[0];
    for (int k = 0; k < 6; k++)
      if (k == 0 || j < w - k) {
        if (packed)
          dy[4 * ((j + k) >> 1) + p.pos[1] + 2 * ((j + k) & 1)] = (uint8_t) y[t][k];
        else if (p.bps == 2)
          ((uint16_t *) dy)[j + k] = (uint16_t) y[t][k];
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
      else if (p.bps == 2)
        ((uint16_t *) du)[j / 2 + k] = (uint16_t) cu, ((uint16_t *) dv)[j / 2 + k] = (uint16_t) cv;
      else
        du[j / 2 + k] = (uint8_t) cu, dv[j / 2 + k] = (uint8_t) cv;
    }
  }
}

// ---- the same four groups (24 pixels) at a time with vector accesses: 24 luma bytes as three 8-byte loads, 12 bytes of each chroma plane as three words
// (or 48 bytes of a packed 4:2:2 line as three 16-byte loads), 64 bytes of v210 as four 16-byte accesses.  Whole blocks only (24 (b + 1) <= width) on rows
// aligned for those accesses (v210_fast_vec_ok); the blocks that hold the line's end take v210_fast_body group by group.
GSTAMD_VP bool v210_fast_vec_ok (const V210FastParams &p)
{
  if (p.bps != 1)
    return false;               /* the 10-bit forms: group by group */
  const bool packed = p.kind == UNPACK_PACKED422;
  const uint8_t *const *p8 = p.to_v210 ? p.s : (const uint8_t *const *) p.d;
  const int *s8 = p.to_v210 ? p.sstride : p.dstride;
  const uint8_t *pv = p.to_v210 ? p.d[0] : p.s[0];
  const int sv = p.to_v210 ? p.dstride[0] : p.sstride[0];
  if (((size_t) pv & 15) || (sv & 15))
    return false;
  if (packed)
    return !((size_t) p8[0] & 15) && !(s8[0] & 15);
  return !((size_t) p8[0] & 7) && !(s8[0] & 7) && !((size_t) p8[p.u_plane] & 3) && !(s8[p.u_plane] & 3) && !((size_t) p8[p.v_plane] & 3) && !(s8[p.v_plane] & 3);
}
GSTAMD_VP int v210_fast_blocks (const V210FastParams &p) { return (p.width + 23) / 24; }

GSTAMD_HD uint32_t v210_byte (const uint32_t *w, int i) { return (w[i >> 2] >> (8 * (i & 3))) & 0xffu; }

GSTAMD_HD void v210_fast_block (const V210FastParams &p, int b, int r)
{
  if (b >= v210_fast_blocks (p) || r >= v210_fast_rows (p))
    return;
  if (24 * (b + 1) > p.width) {           /* the line's end: group by group */
    for (int g = 4 * b; g < 4 * b + 4; g++)
      v210_fast_body (p, g, r);
    return;
  }
  const int l0 = p.h_sub ? 2 * r : r, nl = p.h_sub && l0 + 1 < p.height ? 2 : 1;
  const bool packed = p.kind == UNPACK_PACKED422;
  if (p.to_v210) {
    uint32_t cu[3], cv[3];
    for (int t = 0; t < nl; t++) {
      const int l = l0 + t, crow = p.h_sub ? r : l;
      uint32_t yw[6], q[12];
      if (packed) {
        const uint4 *sp = (const uint4 *) (p.s[0] + (size_t) l * p.sstride[0] + 48 * (size_t) b);
        const uint4 m0 = sp[0], m1 = sp[1], m2 = sp[2];
        q[0] = m0.x, q[1] = m0.y, q[2] = m0.z, q[3] = m0.w, q[4] = m1.x, q[5] = m1.y, q[6] = m1.z, q[7] = m1.w, q[8] = m2.x, q[9] = m2.y, q[10] = m2.z, q[11] = m2.w;
      } else {
        const uint2 *sy = (const uint2 *) (p.s[0] + (size_t) l * p.sstride[0] + 24 * (size_t) b);
        const uint2 a = sy[0], c = sy[1], e = sy[2];
        yw[0] = a.x, yw[1] = a.y, yw[2] = c.x, yw[3] = c.y, yw[4] = e.x, yw[5] = e.y;
        if (t == 0 || !p.h_sub) {
          const uint32_t *su = (const uint32_t *) (p.s[p.u_plane] + (size_t) crow * p.sstride[p.u_plane] + 12 * (size_t) b);
          const uint32_t *sv = (const uint32_t *) (p.s[p.v_plane] + (size_t) crow * p.sstride[p.v_plane] + 12 * (size_t) b);
          cu[0] = su[0], cu[1] = su[1], cu[2] = su[2], cv[0] = sv[0], cv[1] = sv[1], cv[2] = sv[2];
        }
      }
      uint4 *d = (uint4 *) (p.d[0] + (size_t) l * p.dstride[0] + 64 * (size_t) b);
#pragma unroll
      for (int g = 0; g < 4; g++) {
        uint32_t y[6], u[3], v[3];
#pragma unroll
        for (int k = 0; k < 6; k++)
          y[k] = packed ? (q[(6 * g + k) >> 1] >> (8 * (p.pos[1] + 2 * (k & 1)))) & 0xffu : v210_byte (yw, 6 * g + k);
#pragma unroll
        for (int k = 0; k < 3; k++) {
          u[k] = packed ? (q[3 * g + k] >> (8 * p.pos[2])) & 0xffu : v210_byte (cu, 3 * g + k);
          v[k] = packed ? (q[3 * g + k] >> (8 * p.pos[3])) & 0xffu : v210_byte (cv, 3 * g + k);
        }
        d[g] = gstamd_make_uint4 ((u[0] << 2) | (y[0] << 12) | (v[0] << 22), (y[1] << 2) | (u[1] << 12) | (y[2] << 22),
            (v[1] << 2) | (y[3] << 12) | (u[2] << 22), (y[4] << 2) | (v[2] << 12) | (y[5] << 22));
      }
    }
    return;
  }
  uint32_t y[2][24], u[2][12], v[2][12];
  for (int t = 0; t < nl; t++) {
    const uint4 *a = (const uint4 *) (p.s[0] + (size_t) (l0 + t) * p.sstride[0] + 64 * (size_t) b);
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const uint4 m = a[g];
      u[t][3 * g + 0] = ((m.x >> 0) & 0x3ffu) >> 2, y[t][6 * g + 0] = ((m.x >> 10) & 0x3ffu) >> 2, v[t][3 * g + 0] = ((m.x >> 20) & 0x3ffu) >> 2;
      y[t][6 * g + 1] = ((m.y >> 0) & 0x3ffu) >> 2, u[t][3 * g + 1] = ((m.y >> 10) & 0x3ffu) >> 2, y[t][6 * g + 2] = ((m.y >> 20) & 0x3ffu) >> 2;
      v[t][3 * g + 1] = ((m.z >> 0) & 0x3ffu) >> 2, y[t][6 * g + 3] = ((m.z >> 10) & 0x3ffu) >> 2, u[t][3 * g + 2] = ((m.z >> 20) & 0x3ffu) >> 2;
      y[t][6 * g + 4] = ((m.w >> 0) & 0x3ffu) >> 2, v[t][3 * g + 2] = ((m.w >> 10) & 0x3ffu) >> 2, y[t][6 * g + 5] = ((m.w >> 20) & 0x3ffu) >> 2;
    }
  }
  if (p.h_sub && nl == 2)
#pragma unroll
    for (int k = 0; k < 12; k++)
      u[0][k] = (u[0][k] + u[1][k]) / 2, v[0][k] = (v[0][k] + v[1][k]) / 2;
  for (int t = 0; t < nl; t++) {
    if (packed) {
      uint32_t q[12];
#pragma unroll
      for (int k = 0; k < 12; k++)
        q[k] = (y[t][2 * k] << (8 * p.pos[1])) | (y[t][2 * k + 1] << (8 * (p.pos[1] + 2))) | (u[t][k] << (8 * p.pos[2])) | (v[t][k] << (8 * p.pos[3]));
      uint4 *d = (uint4 *) (p.d[0] + (size_t) (l0 + t) * p.dstride[0] + 48 * (size_t) b);
      d[0] = gstamd_make_uint4 (q[0], q[1], q[2], q[3]), d[1] = gstamd_make_uint4 (q[4], q[5], q[6], q[7]), d[2] = gstamd_make_uint4 (q[8], q[9], q[10], q[11]);
    } else {
      uint32_t w[6];
#pragma unroll
      for (int k = 0; k < 6; k++)
        w[k] = y[t][4 * k] | (y[t][4 * k + 1] << 8) | (y[t][4 * k + 2] << 16) | (y[t][4 * k + 3] << 24);
      uint2 *d = (uint2 *) (p.d[0] + (size_t) (l0 + t) * p.dstride[0] + 24 * (size_t) b);
      d[0] = gstamd_make_uint2 (w[0], w[1]), d[1] = gstamd_make_uint2 (w[2], w[3]), d[2] = gstamd_make_uint2 (w[4], w[5]);
    }
  }
  if (!packed)
    for (int t = 0; t < (p.h_sub ? 1 : nl); t++) {
      const int crow = p.h_sub ? r : l0 + t;
      uint32_t *du = (uint32_t *) (p.d[p.u_plane] + (size_t) crow * p.dstride[p.u_plane] + 12 * (size_t) b);
      uint32_t *dv = (uint32_t *) (p.d[p.v_plane] + (size_t) crow * p.dstride[p.v_plane] + 12 * (size_t) b);
#pragma unroll
      for (int k = 0; k < 3; k++) {
        du[k] = u[t][4 * k] | (u[t][4 * k + 1] << 8) | (u[t][4 * k + 2] << 16) | (u[t][4 * k + 3] << 24);
        dv[k] = v[t][4 * k] | (v[t][4 * k + 1] << 8) | (v[t][4 * k + 2] << 16) | (v[t][4 * k + 3] << 24);
      }
    }
}

}  // namespace gstamd
