// video_relayout.h - planar / semi-planar 8-bit YUV to another such format with the same subsampling when the chain neither filters nor
// mixes (VideoPlan::relayout): the luma plane is copied, the chroma samples change their arrangement - two planes into one interleaved
// plane (I420 / YV12 -> NV12 / NV21, Y42B -> NV16, Y444 -> NV24), one into two, or the byte order of the pairs (NV12 <-> NV21).  What the
// reference does for these pairs is its generic chain (unpack to AYUV with the chroma sample duplicated over its pixels, no resampler,
// pack taking the even pixel's chroma or the average of equal values: video-format.c unpack_I420 :1009, pack_NV12 :1644 ...) - every
// output sample IS an input sample.  16 output bytes per lane with 16-byte accesses; the caller checks the alignment and otherwise keeps
// the chain's kernels.
#pragma once
#include "video_device.h"

namespace gstamd {

struct RelayoutParams {
  int width, height;            // luma
  int cw, ch;                   // chroma samples per row, chroma rows
  int in_semi, out_semi;
  int in_u, in_v, out_u, out_v; // planar: plane of U / V; semi: in_u / out_u = 1 when U is the first byte of a pair
  const uint8_t *in[3];
  int in_stride[3];
  uint8_t *out[3];
  int out_stride[3];
};

GSTAMD_HD uint32_t rl_even (uint32_t lo, uint32_t hi) { return (lo & 0xffu) | ((lo >> 8) & 0xff00u) | ((hi & 0xffu) << 16) | ((hi << 8) & 0xff000000u); }
GSTAMD_HD uint32_t rl_odd (uint32_t lo, uint32_t hi) { return ((lo >> 8) & 0xffu) | ((lo >> 16) & 0xff00u) | ((hi << 8) & 0xff0000u) | (hi & 0xff000000u); }
// bytes a0 b0 a1 b1 of the low halves (k = 0) or the high halves (k = 1) of a and b
GSTAMD_HD uint32_t rl_zip (uint32_t a, uint32_t b, int k)
{
  const uint32_t x = k ? a >> 16 : a & 0xffffu, y = k ? b >> 16 : b & 0xffffu;
  return (x & 0xffu) | ((y & 0xffu) << 8) | ((x & 0xff00u) << 8) | ((y & 0xff00u) << 16);
}

GSTAMD_HD void rl_copy_bytes (uint8_t *d, const uint8_t *s, int n)
{
  for (int i = 0; i < n; i++)
    d[i] = s[i];
}

// row < height: 16 luma bytes of row `row` from byte 16 * lane.  Then the chroma rows: for a semi-planar destination `ch` rows of 16
// interleaved bytes (8 samples of each component); for a planar one `ch` rows of U and `ch` rows of V, 16 samples each.
GSTAMD_HD void relayout_body (const RelayoutParams &p, int lane, int row, long long ds = 0, long long dd = 0)
{
  const int b0 = 16 * lane;
  if (row < p.height) {
    if (b0 >= p.width)
      return;
    const uint8_t *s = (p.in[0] + ds) + (size_t) row * p.in_stride[0] + b0;
    uint8_t *d = (p.out[0] + dd) + (size_t) row * p.out_stride[0] + b0;
    if (b0 + 16 <= p.width)
      *(uint4 *) d = *(const uint4 *) s;
    else
      rl_copy_bytes (d, s, p.width - b0);
    return;
  }
  row -= p.height;
  if (p.out_semi) {
    if (row >= p.ch || b0 >= 2 * p.cw)
      return;
    const int k0 = b0 / 2;              /* first chroma sample of the lane, 8 of them */
    uint8_t *d = (p.out[1] + dd) + (size_t) row * p.out_stride[1] + b0;
    if (b0 + 16 <= 2 * p.cw) {
      uint4 o;
      if (p.in_semi) {
        const uint4 a = *(const uint4 *) ((p.in[1] + ds) + (size_t) row * p.in_stride[1] + b0);
        if (p.in_u == p.out_u) {
          o = a;
        } else {                        /* NV12 <-> NV21: the bytes of every pair trade places */
          o.x = ((a.x >> 8) & 0x00ff00ffu) | ((a.x << 8) & 0xff00ff00u), o.y = ((a.y >> 8) & 0x00ff00ffu) | ((a.y << 8) & 0xff00ff00u);
          o.z = ((a.z >> 8) & 0x00ff00ffu) | ((a.z << 8) & 0xff00ff00u), o.w = ((a.w >> 8) & 0x00ff00ffu) | ((a.w << 8) & 0xff00ff00u);
        }
      } else {
        const uint2 u = *(const uint2 *) ((p.in[p.in_u] + ds) + (size_t) row * p.in_stride[p.in_u] + k0);
        const uint2 v = *(const uint2 *) ((p.in[p.in_v] + ds) + (size_t) row * p.in_stride[p.in_v] + k0);
        const uint2 f = p.out_u ? u : v, g = p.out_u ? v : u;
        o.x = rl_zip (f.x, g.x, 0), o.y = rl_zip (f.x, g.x, 1), o.z = rl_zip (f.y, g.y, 0), o.w = rl_zip (f.y, g.y, 1);
      }
      *(uint4 *) d = o;
      return;
    }
    for (int k = k0; k < p.cw; k++) {
      uint8_t cu, cv;
      if (p.in_semi) {
        const uint8_t *s = (p.in[1] + ds) + (size_t) row * p.in_stride[1] + 2 * k;
        cu = p.in_u ? s[0] : s[1], cv = p.in_u ? s[1] : s[0];
      } else {
        cu = (p.in[p.in_u] + ds)[(size_t) row * p.in_stride[p.in_u] + k], cv = (p.in[p.in_v] + ds)[(size_t) row * p.in_stride[p.in_v] + k];
      }
      d[2 * (k - k0)] = p.out_u ? cu : cv;
      d[2 * (k - k0) + 1] = p.out_u ? cv : cu;
    }
    return;
  }
  // planar destination: U rows, then V rows
  const int second = row >= p.ch;
  if (second)
    row -= p.ch;
  if (row >= p.ch || b0 >= p.cw)
    return;
  const int want_u = !second;         /* first the U plane's rows */
  uint8_t *d = (p.out[want_u ? p.out_u : p.out_v] + dd) + (size_t) row * p.out_stride[want_u ? p.out_u : p.out_v] + b0;
  if (!p.in_semi) {
    const int sp = want_u ? p.in_u : p.in_v;
    const uint8_t *s = (p.in[sp] + ds) + (size_t) row * p.in_stride[sp] + b0;
    if (b0 + 16 <= p.cw)
      *(uint4 *) d = *(const uint4 *) s;
    else
      rl_copy_bytes (d, s, p.cw - b0);
    return;
  }
  const int first_byte = (want_u ? 1 : 0) == (p.in_u ? 1 : 0);          /* is the wanted component the first byte of a pair? */
  const uint8_t *s = (p.in[1] + ds) + (size_t) row * p.in_stride[1] + 2 * b0;
  if (b0 + 16 <= p.cw) {
    const uint4 a = *(const uint4 *) s, b = *(const uint4 *) (s + 16);
    uint4 o;
    if (first_byte)
      o.x = rl_even (a.x, a.y), o.y = rl_even (a.z, a.w), o.z = rl_even (b.x, b.y), o.w = rl_even (b.z, b.w);
    else
      o.x = rl_odd (a.x, a.y), o.y = rl_odd (a.z, a.w), o.z = rl_odd (b.x, b.y), o.w = rl_odd (b.z, b.w);
    *(uint4 *) d = o;
    return;
  }
  for (int k = b0; k < p.cw; k++)
    d[k - b0] = s[2 * (k - b0) + (first_byte ? 0 : 1)];
}

// lanes per row a launch needs: the widest row in bytes (an interleaved 4:4:4 chroma row is twice the luma's) / 16
inline int relayout_lanes (const RelayoutParams &p)       /* host */
{
  const int wide = p.out_semi && 2 * p.cw > p.width ? 2 * p.cw : p.width;
  return (wide + 15) / 16;
}

// grid rows of a launch
inline int relayout_rows (const RelayoutParams &p)      /* host */ { return p.height + (p.out_semi ? p.ch : 2 * p.ch); }

}  // namespace gstamd
