// video_device.h - per-pixel device code of the GstVideoConverter kernels (bodies only).
//
// Included by video_kernels.hip (compiled for gfx950, GSTAMD_HD = __device__ __forceinline__) and by
// the test-only host emulator tests/emu/emu_video.cpp (GSTAMD_HD = inline), which runs the very same
// bodies over the launch grid on the CPU so kernel logic can be debugged without a GPU.  The
// emulator is test infrastructure; nothing in the product calls it.
//
// Reference semantics being reproduced bit-exactly (paths under
// /root/reference/subprojects/gst-plugins-base/gst-libs/gst/video/):
//   unpack      video-format.c:92-151 (planar 4:2:0), :1593-1640 (NV12/NV21), video-orc.orc:334-411
//   chroma up   video-chroma.c:277-327 (h2, v2), :687-699 (h2 cosited), pairing video-converter.c:2991-3021
//   matrix      video-orc.orc:1634-1693 (AYUV_ARGB), video-converter.c:1139-1207 (matrix8, table)
//   alpha       video-converter.c:1871-1920
//   scalers     video-orc-dist.c:26162-26195 (ldreslinl), video-orc.orc:2208-2224 (v_2tap_lq),
//               :2388-2480 / :2557-2655 (N-tap LQ H/V), video-scaler.c:546-760, 829-1072
//   pack        video-orc.orc:340-411 (byte permutations)
//
// A pixel in flight is a uint32 holding the reference's intermediate "AYUV"/"ARGB" pixel:
// byte0 = A, byte1 = Y|R, byte2 = U|G, byte3 = V|B.
#pragma once
#include <stdint.h>
#include <stddef.h>

#include "planner.h"
#include "video_types.h"

#ifdef __HIPCC__
#define GSTAMD_HD __device__ __forceinline__
#define gstamd_make_uint4 make_uint4
#define gstamd_make_uint2 make_uint2
#else
#define GSTAMD_HD inline
#ifndef __restrict__
#define __restrict__
#endif
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
static inline uint4 gstamd_make_uint4 (uint32_t x, uint32_t y, uint32_t z, uint32_t w) { uint4 r = {x, y, z, w}; return r; }
static inline uint2 gstamd_make_uint2 (uint32_t x, uint32_t y) { uint2 r = {x, y}; return r; }
#endif

namespace gstamd {

// ------------------------------------------------------------------------------------------------
// per-pixel stages
// ------------------------------------------------------------------------------------------------
GSTAMD_HD int clampi (int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// video_orc_convert_AYUV_ARGB, C semantics of video-orc-dist.c:22162-22318
GSTAMD_HD uint32_t matrix_ayuv_argb (uint32_t px, const int *p)
{
  const int a = px & 0xff;
  const int y = (int) ((px >> 8) & 0xff) - 128, u = (int) ((px >> 16) & 0xff) - 128, v = (int) ((px >> 24) & 0xff) - 128;
  // splatbw: ((b & 0xff) << 8) | (b & 0xff) reinterpreted as int16
  const int yb = y & 0xff, ub = u & 0xff, vb = v & 0xff;
  const int wy0 = (int) (int16_t) ((yb << 8) | yb), wu = (int) (int16_t) ((ub << 8) | ub), wv = (int) (int16_t) ((vb << 8) | vb);
  const int wy = (int) (int16_t) ((wy0 * (int) (int16_t) p[0]) >> 16);                 // mulhsw
  const int r16 = (int) (int16_t) (wy + (int) (int16_t) ((wv * (int) (int16_t) p[1]) >> 16));   // addw wraps
  const int b16 = (int) (int16_t) (wy + (int) (int16_t) ((wu * (int) (int16_t) p[2]) >> 16));
  int g16 = (int) (int16_t) (wy + (int) (int16_t) ((wu * (int) (int16_t) p[3]) >> 16));
  g16 = (int) (int16_t) (g16 + (int) (int16_t) ((wv * (int) (int16_t) p[4]) >> 16));
  const int r = clampi (r16, -128, 127) + 128, g = clampi (g16, -128, 127) + 128, b = clampi (b16, -128, 127) + 128;
  return (uint32_t) a | ((uint32_t) r << 8) | ((uint32_t) g << 16) | ((uint32_t) b << 24);
}

GSTAMD_HD uint32_t apply_matrix (const MatrixParams &m, uint32_t px)
{
  if (m.kind == MATRIX_NONE)
    return px;
  if (m.kind == MATRIX_AYUV_ARGB)
    return matrix_ayuv_argb (px, m.p);
  const int r = (px >> 8) & 0xff, g = (px >> 16) & 0xff, b = (px >> 24) & 0xff;
  if (m.kind == MATRIX_TABLE) {
    // video_converter_matrix8_table: three 16-bit lanes in one int64, borrows included
    const long long s0 = (long long) (m.im[0][0] * r + m.im[0][1] * g + m.im[0][2] * b + m.im[0][3]);
    const long long s1 = (long long) (m.im[1][0] * r + m.im[1][1] * g + m.im[1][2] * b + m.im[1][3]);
    const long long s2 = (long long) (m.im[2][0] * r + m.im[2][1] * g + m.im[2][2] * b + m.im[2][3]);
    const long long x = (s0 << 32) + (s1 << 16) + s2;
    const uint32_t o1 = (uint32_t) (x >> 40) & 0xff, o2 = (uint32_t) (x >> 24) & 0xff, o3 = (uint32_t) (x >> 8) & 0xff;
    return (px & 0xff) | (o1 << 8) | (o2 << 16) | (o3 << 24);
  }
  // _custom_video_orc_matrix8
  const int y = clampi (((m.im[0][0] * r + m.im[0][1] * g + m.im[0][2] * b) >> 8) + m.im[0][3], 0, 255);
  const int u = clampi (((m.im[1][0] * r + m.im[1][1] * g + m.im[1][2] * b) >> 8) + m.im[1][3], 0, 255);
  const int v = clampi (((m.im[2][0] * r + m.im[2][1] * g + m.im[2][2] * b) >> 8) + m.im[2][3], 0, 255);
  return (px & 0xff) | ((uint32_t) y << 8) | ((uint32_t) u << 16) | ((uint32_t) v << 24);
}

GSTAMD_HD uint32_t apply_alpha (int kind, unsigned value, uint32_t px)
{
  if (kind == ALPHA_NONE)
    return px;
  if (kind == ALPHA_SET)
    return (px & 0xffffff00u) | (value < 255u ? value : 255u);
  const int a = (int) (((px & 0xff) * value) / 255u);
  return (px & 0xffffff00u) | (uint32_t) clampi (a, 0, 255);
}

GSTAMD_HD uint32_t apply_color (const ColorParams &c, uint32_t px)
{
  return apply_alpha (c.alpha_kind, (unsigned) c.alpha_value, apply_matrix (c.matrix, px));
}

// pack: destination byte pos[i] receives component i
GSTAMD_HD uint32_t pack_px (const int *pos, uint32_t px)
{
  return ((px & 0xff) << (8 * pos[0])) | (((px >> 8) & 0xff) << (8 * pos[1])) |
      (((px >> 16) & 0xff) << (8 * pos[2])) | (((px >> 24) & 0xff) << (8 * pos[3]));
}

// ------------------------------------------------------------------------------------------------
// front: unpack + chroma upsample, functional (one pixel)
// ------------------------------------------------------------------------------------------------
struct UV { int u, v; };

// the four bytes of a packed 4:2:2 macropixel as one load (rows are 4-byte multiples; a frame pointer of any alignment is fine for a global dword load)
GSTAMD_HD uint32_t load_macropixel (const uint8_t *p)
{
  uint32_t m;
  __builtin_memcpy (&m, p, 4);
  return m;
}

GSTAMD_HD UV load_uv (const FrontParams &f, const Planes &pl, int crow, int k)
{
  UV r;
  if (f.kind == UNPACK_PACKED422) {      // macropixel k of line crow: U and V at their bytes (unpack_YUY2 & co, video-format.c:155-460)
    const uint32_t m = load_macropixel (pl.p[0] + (ptrdiff_t) crow * pl.stride[0] + 4 * k);          // one request, not two
    const bool swap = k == f.swap_k;
    r.u = (int) ((m >> (8 * f.pos[swap ? 3 : 2])) & 0xffu);
    r.v = (int) ((m >> (8 * f.pos[swap ? 2 : 3])) & 0xffu);
  } else if (f.kind == UNPACK_SEMI_TILED) {  // tiled NV12: the pair's bytes inside its UV tile (planner.h tiled_uv_offset)
    const uint8_t *p = pl.p[1] + tiled_uv_offset (f.pos, pl.stride[1], k, crow);
    r.u = p[0];
    r.v = p[1];
  } else if (f.kind == UNPACK_PACKED411) {   // group k of the line: U Y0 Y1 V Y2 Y3 (unpack_IYU1 video-format.c:2370-2436)
    const uint8_t *p = pl.p[0] + (ptrdiff_t) crow * pl.stride[0] + 6 * k;
    r.u = p[0];
    r.v = p[3];
  } else if (GSTAMD_KIND_SEMI (f.kind)) {
    const uint8_t *p = pl.p[1] + (ptrdiff_t) crow * pl.stride[1] + 2 * k;
    const int c0 = p[0], c1 = p[1];
    r.u = f.u_plane ? c0 : c1;           // NV12: U first; NV21: V first
    r.v = f.u_plane ? c1 : c0;
  } else {
    r.u = pl.p[f.u_plane][(ptrdiff_t) crow * pl.stride[f.u_plane] + k];
    r.v = pl.p[f.v_plane][(ptrdiff_t) crow * pl.stride[f.v_plane] + k];
  }
  return r;
}

// horizontally filtered chroma of chroma row `crow` at luma position x
GSTAMD_HD UV chroma_h_at (const FrontParams &f, const Planes &pl, int crow, int x)
{
  if (f.w_sub == 0)
    return load_uv (f, pl, crow, x);
  if (f.w_sub == 2) {                     // 4:1:1: the unpacker gave pixel x the sample x >> 2 (unpack_Y41B video-format.c:923-974)
    const int w = f.width;
    UV c = load_uv (f, pl, crow, x >> 2);
    if (f.chroma_h == CHROMA_H_H4) {      // video_chroma_up_h4_u8: for (i = 2; i < width - 3; i += 4) pixels i .. i + 3 from samples PR (i - 2), PR (i + 2)
      const int i = x >= 2 ? 2 + ((x - 2) & ~3) : -1;
      if (i >= 0 && i < w - 3) {
        const int g = (i - 2) >> 2, j = x - i;
        const UV a = load_uv (f, pl, crow, g), b = load_uv (f, pl, crow, g + 1);
        const int wa = 7 - 2 * j, wb = 1 + 2 * j;         // FILT_7_1, _5_3, _3_5, _1_7
        c.u = (wa * a.u + wb * b.u + 4) >> 3;
        c.v = (wa * a.v + wb * b.v + 4) >> 3;
      }
    } else if (f.chroma_h == CHROMA_H_H4_CS) {      // video_chroma_up_h4_cs_u8: for (i = 0; i < width - 4; i += 4) pixels i + 1 .. i + 3 from PR (i), PR (i + 4)
      const int i = x & ~3, j = x & 3;
      if (j && i < w - 4) {
        const UV b = load_uv (f, pl, crow, (i >> 2) + 1);
        if (j == 2) {
          c.u = (c.u + b.u + 1) >> 1;
          c.v = (c.v + b.v + 1) >> 1;
        } else {
          const int wa = j == 1 ? 3 : 1, wb = 4 - wa;
          c.u = (wa * c.u + wb * b.u + 2) >> 2;
          c.v = (wa * c.v + wb * b.v + 2) >> 2;
        }
      }
    }
    return c;
  }
  const int k = x >> 1, w = f.width;
  UV c = load_uv (f, pl, crow, k);
  if (f.chroma_h == CHROMA_H_H2_CS) {
    if ((x & 1) && x < w - 1) {           // PR(i) = FILT_1_1 (PR(i-1), PR(i+1)), odd i < width-1
      const UV n = load_uv (f, pl, crow, k + 1);
      c.u = (c.u + n.u + 1) >> 1;
      c.v = (c.v + n.v + 1) >> 1;
    }
  } else if (f.chroma_h == CHROMA_H_H2) {
    if ((x & 1) && x < w - 1) {           // FILT_3_1 (tr0, tr1)
      const UV n = load_uv (f, pl, crow, k + 1);
      c.u = (3 * c.u + n.u + 2) >> 2;
      c.v = (3 * c.v + n.v + 2) >> 2;
    } else if (!(x & 1) && x >= 2) {      // FILT_1_3 (tr0, tr1) written to PR(i+1)
      const UV pv = load_uv (f, pl, crow, k - 1);
      c.u = (pv.u + 3 * c.u + 2) >> 2;
      c.v = (pv.v + 3 * c.v + 2) >> 2;
    }
  }
  return c;
}

GSTAMD_HD uint32_t fetch_front (const FrontParams &f, const Planes &pl, const int *__restrict__ vpair,
    int x, int y)
{
  if (f.kind == UNPACK_PACKED4) {
    const uint32_t raw = *(const uint32_t *) (pl.p[0] + (size_t) y * pl.stride[0] + 4 * (size_t) x);
    return ((raw >> (8 * f.pos[0])) & 0xff) | (((raw >> (8 * f.pos[1])) & 0xff) << 8) |
        (((raw >> (8 * f.pos[2])) & 0xff) << 16) | (((raw >> (8 * f.pos[3])) & 0xff) << 24);
  }
  if (f.kind == UNPACK_PACKED3) {          // unpack_RGB / unpack_BGR (video-format.c:1521, 1558)
    const uint8_t *q = pl.p[0] + (size_t) y * pl.stride[0] + 3 * (size_t) x;
    return 0xffu | ((uint32_t) q[f.pos[1]] << 8) | ((uint32_t) q[f.pos[2]] << 16) | ((uint32_t) q[f.pos[3]] << 24);
  }
  if (f.kind == UNPACK_RGB16) {          // unpack_RGB16 / _BGR16 / _RGB15 / _BGR15 (video-format.c:1301-1425)
    const int wd = *(const uint16_t *) (pl.p[0] + (size_t) y * pl.stride[0] + 2 * (size_t) x);
    return 0xffu | ((uint32_t) rgb16_field (wd, f.pos[1], 5) << 8) | ((uint32_t) rgb16_field (wd, f.pos[2], f.pos[0]) << 16) |
        ((uint32_t) rgb16_field (wd, f.pos[3], 5) << 24);
  }
  if (f.kind == UNPACK_GRAY)             // unpack_GRAY8 (video-format.c:1209, video_orc_unpack_GRAY8): A = 0xff, Y, U = V = 0x80
    return 0x808000ffu | ((uint32_t) pl.p[0][(size_t) y * pl.stride[0] + x] << 8);
  const int yl = y < f.luma_last ? y : f.luma_last;          /* a no-op for the picture's own lines; the line past it clamps like do_unpack_lines */
  const int Y = f.kind == UNPACK_PACKED422 ? (int) ((load_macropixel (pl.p[0] + (ptrdiff_t) yl * pl.stride[0] + 4 * (x >> 1)) >> (8 * (f.pos[1] + 2 * (x & 1)))) & 0xffu)
      : f.kind == UNPACK_SEMI_TILED ? pl.p[0][tiled_luma_offset (f.pos, pl.stride[0], x, yl)]
      : f.kind == UNPACK_PACKED411 ? pl.p[0][(size_t) yl * pl.stride[0] + 6 * (size_t) (x >> 2) + 1 + (x & 3) + ((x & 3) >> 1)]          /* bytes 1, 2, 4, 5 of the group */
      : pl.p[0][(size_t) yl * pl.stride[0] + x];
  UV c;
  if (f.chroma_v2) {
    const VPairW t = vpair_get (vpair, y, f.chroma_v2);
    const int ra = t.ra, rb = t.rb;
    const UV a = chroma_h_at (f, pl, ra, x);
    if (ra == rb) {
      c = a;
    } else {
      const UV b = chroma_h_at (f, pl, rb, x);
      // video_chroma_up_v2: d1 = (3*s1 + s2 + 2) >> 2, d2 = (s1 + 3*s2 + 2) >> 2 - weights 6 / 2 over 8; a field's table: video_chroma_up_vi2's (vpair_get)
      c.u = (t.wa * a.u + (8 - t.wa) * b.u + 4) >> 3;
      c.v = (t.wa * a.v + (8 - t.wa) * b.v + 4) >> 3;
    }
  } else {
    c = chroma_h_at (f, pl, y >> f.h_sub, x);
  }
  /* unpack_A420 (video-format.c:2118-2146): the alpha plane's sample of the same line (clamped like the luma: do_unpack_lines) */
  const int ap = GSTAMD_KIND_ALPHA_PLANE (f.kind);       /* ... unpack_AV12 :1655 */
  const uint32_t A = ap == 3 ? pl.p[3][(size_t) yl * pl.stride[3] + x] : (ap == 2 ? pl.p[2][(size_t) yl * pl.stride[2] + x] : 0xffu);
  return A | ((uint32_t) Y << 8) | ((uint32_t) c.u << 16) | ((uint32_t) c.v << 24);
}

// ------------------------------------------------------------------------------------------------
// sources for the scaler kernels
// ------------------------------------------------------------------------------------------------
GSTAMD_HD void front_span8_any (const FrontParams &f, const Planes &pl, const int *__restrict__ vpair, int x0, int y, uint32_t *out);

struct SrcFront {
  FrontParams f;
  Planes pl;
  const int *vpair;
  ColorParams pre;      // matrix+alpha applied before scaling (upscale case), kind NONE otherwise
  int vec_ok;           // planes aligned for front_span8
  GSTAMD_HD uint32_t at (int x, int y) const { return apply_color (pre, fetch_front (f, pl, vpair, x, y)); }
  // stage pixels [x_lo, x_hi) of line y into lds[0 ..): 8-pixel groups with word loads where possible
  GSTAMD_HD void stage (uint32_t *lds, int x_lo, int x_hi, int y, int tid, int nthreads) const
  {
    if (vec_ok && f.w_sub == 1 && kind_has_planes (f.kind)) {
      const int xa = x_lo & ~7;
      const int groups = (x_hi - xa + 7) >> 3;
      for (int g = tid; g < groups; g += nthreads) {
        const int x0 = xa + 8 * g;
        if (x0 + 8 <= f.width) {
          uint32_t px[8];
          front_span8_any (f, pl, vpair, x0, y, px);
#pragma unroll
          for (int i = 0; i < 8; i++)
            if (x0 + i >= x_lo && x0 + i < x_hi)
              lds[x0 + i - x_lo] = apply_color (pre, px[i]);
        } else {
          for (int x = x0 > x_lo ? x0 : x_lo; x < x_hi && x < x0 + 8; x++)
            lds[x - x_lo] = at (x, y);
        }
      }
    } else {
      for (int i = tid; i < x_hi - x_lo; i += nthreads)
        lds[i] = at (x_lo + i, y);
    }
  }
};

// a 4-byte packed source through unpack (a byte permutation) and the colour stage - the pixel source of k_convert_pack.  SrcFront does the same
// through fetch_front, whose switch over every source layout a packer body inlines twenty times: 25 000 instructions of wave-uniform branches
struct SrcPacked4 {
  const uint8_t *p;
  int stride;
  int pos[4];
  uint32_t sel;         // v_perm_b32 selector: unpacked byte c <- memory byte pos[c]
  ColorParams pre;
  GSTAMD_HD uint32_t at (int x, int y) const
  {
    const uint32_t raw = *(const uint32_t *) (p + (size_t) y * stride + 4 * (size_t) x);
#ifdef __HIPCC__
    const uint32_t px = __builtin_amdgcn_perm (0u, raw, sel);
#else
    const uint32_t px = ((raw >> (8 * pos[0])) & 0xff) | (((raw >> (8 * pos[1])) & 0xff) << 8) | (((raw >> (8 * pos[2])) & 0xff) << 16) |
        (((raw >> (8 * pos[3])) & 0xff) << 24);
#endif
    return apply_color (pre, px);
  }
  /* pack_planar_block4's row source: four pixels with one 16-byte load (frame rows on 16 bytes: the kernel's `wide` switch) */
  GSTAMD_HD uint32_t conv (uint32_t raw) const
  {
#ifdef __HIPCC__
    const uint32_t px = __builtin_amdgcn_perm (0u, raw, sel);
#else
    const uint32_t px = ((raw >> (8 * pos[0])) & 0xff) | (((raw >> (8 * pos[1])) & 0xff) << 8) | (((raw >> (8 * pos[2])) & 0xff) << 16) |
        (((raw >> (8 * pos[3])) & 0xff) << 24);
#endif
    return apply_color (pre, px);
  }
  GSTAMD_HD bool ok4 (int, int) const { return true; }
  GSTAMD_HD uint4 row4 (int x0, int y) const
  {
    const uint4 r = *(const uint4 *) (p + (size_t) y * stride + 4 * (size_t) x0);
    return gstamd_make_uint4 (conv (r.x), conv (r.y), conv (r.z), conv (r.w));
  }
  GSTAMD_HD uint4 row4n (int x0, int y, bool edges, int xm, int xp, uint32_t &em, uint32_t &ep) const
  {
    if (edges)
      em = px (xm, y), ep = px (xp, y);
    return row4 (x0, y);
  }
  GSTAMD_HD uint32_t px (int x, int y) const { return at (x, y); }
};

inline SrcPacked4 make_src_packed4 (const FrontParams &f, const Planes &pl, const ColorParams &color)
{
  SrcPacked4 s;
  s.p = pl.p[0];
  s.stride = pl.stride[0];
  s.sel = 0;
  for (int c = 0; c < 4; c++) {
    s.pos[c] = f.pos[c];
    s.sel |= (uint32_t) f.pos[c] << (8 * c);
  }
  s.pre = color;
  return s;
}

struct SrcImage {
  const uint8_t *p;
  int stride;
  int width;            // pixels per row (wave-tile staging stops there)
  GSTAMD_HD uint32_t at (int x, int y) const
  {
    return *(const uint32_t *) (p + (size_t) y * stride + 4 * (size_t) x);
  }
  GSTAMD_HD void stage (uint32_t *lds, int x_lo, int x_hi, int y, int tid, int nthreads) const
  {
    for (int i = tid; i < x_hi - x_lo; i += nthreads)
      lds[i] = at (x_lo + i, y);
  }
};

struct Dst {
  uint8_t *p;
  int stride;
  int final;            // 1: apply post colour + pack (last kernel), 0: raw intermediate
  ColorParams post;
  int pack_pos[4];
  GSTAMD_HD void put (int x, int y, uint32_t px) const
  {
    if (final)
      px = pack_px (pack_pos, apply_color (post, px));
    *(uint32_t *) (p + (size_t) y * stride + 4 * (size_t) x) = px;
  }
};

// ------------------------------------------------------------------------------------------------
// K1: fused unscaled convert.  One lane = 8 consecutive pixels of one row.
// Fast path (4:2:x sources, full span inside the row): Y as one 8-byte load, the 4 chroma samples
// under the span as one vector load per chroma row (+2 scalar neighbours), two 16-byte stores.
// ------------------------------------------------------------------------------------------------
#define K1_PX 8

struct Chroma6 { int u[6], v[6]; };     // samples k0-1 .. k0+4

GSTAMD_HD void load_chroma6 (const FrontParams &f, const Planes &pl, int crow, int k0, int cw, Chroma6 &c)
{
  if (f.kind == UNPACK_SEMI) {
    const uint8_t *row = pl.p[1] + (ptrdiff_t) crow * pl.stride[1];
    const uint2 mid = *(const uint2 *) (row + 2 * k0);           // samples k0..k0+3 (8-byte aligned: k0 % 4 == 0)
    const uint32_t w[2] = {mid.x, mid.y};
    const int su = f.u_plane ? 0 : 8, sv = f.u_plane ? 8 : 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t pair = (w[i >> 1] >> (16 * (i & 1))) & 0xffff;
      c.u[i + 1] = (pair >> su) & 0xff;
      c.v[i + 1] = (pair >> sv) & 0xff;
    }
    const int km = k0 > 0 ? k0 - 1 : 0, kp = k0 + 4 < cw ? k0 + 4 : cw - 1;
    const uint32_t pm = *(const uint16_t *) (row + 2 * km), pp = *(const uint16_t *) (row + 2 * kp);
    c.u[0] = (pm >> su) & 0xff;
    c.v[0] = (pm >> sv) & 0xff;
    c.u[5] = (pp >> su) & 0xff;
    c.v[5] = (pp >> sv) & 0xff;
  } else {
    const uint8_t *ru = pl.p[f.u_plane] + (ptrdiff_t) crow * pl.stride[f.u_plane];
    const uint8_t *rv = pl.p[f.v_plane] + (ptrdiff_t) crow * pl.stride[f.v_plane];
    const uint32_t mu = *(const uint32_t *) (ru + k0), mv = *(const uint32_t *) (rv + k0);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      c.u[i + 1] = (mu >> (8 * i)) & 0xff;
      c.v[i + 1] = (mv >> (8 * i)) & 0xff;
    }
    const int km = k0 > 0 ? k0 - 1 : 0, kp = k0 + 4 < cw ? k0 + 4 : cw - 1;
    c.u[0] = ru[km];
    c.v[0] = rv[km];
    c.u[5] = ru[kp];
    c.v[5] = rv[kp];
  }
}

// h-filtered chroma for the 8 pixels x0..x0+7 out of the 6 samples
template <int CH>
GSTAMD_HD void hfilter8 (const Chroma6 &c, int x0, int w, int *u, int *v)
{
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int x = x0 + i, j = (i >> 1) + 1;     // j: index of sample x>>1
    int uu = c.u[j], vv = c.v[j];
    if (CH == CHROMA_H_H2_CS) {
      if ((i & 1) && x < w - 1) {
        uu = (uu + c.u[j + 1] + 1) >> 1;
        vv = (vv + c.v[j + 1] + 1) >> 1;
      }
    } else if (CH == CHROMA_H_H2) {
      if ((i & 1) && x < w - 1) {
        uu = (3 * uu + c.u[j + 1] + 2) >> 2;
        vv = (3 * vv + c.v[j + 1] + 2) >> 2;
      } else if (!(i & 1) && x >= 2) {
        uu = (c.u[j - 1] + 3 * uu + 2) >> 2;
        vv = (c.v[j - 1] + 3 * vv + 2) >> 2;
      }
    }
    u[i] = uu;
    v[i] = vv;
  }
}

// 8 unpacked + chroma-upsampled pixels x0 .. x0+7 of line y (x0 % 8 == 0, x0 + 8 <= width, w_sub == 1, vector
// alignment as checked by the launcher): the word-load version of fetch_front
template <int CH>
GSTAMD_HD void front_span8 (const FrontParams &f, const Planes &pl, const int *__restrict__ vpair, int x0, int y, uint32_t *out)
{
  const int w = f.width;
  const uint2 yy = *(const uint2 *) (pl.p[0] + (size_t) (y < f.luma_last ? y : f.luma_last) * pl.stride[0] + x0);
  const int cw = (w + 1) >> 1, k0 = x0 >> 1;
  int ra, rb, wa = 6;
  if (f.chroma_v2) {
    const VPairW t = vpair_get (vpair, y, f.chroma_v2);
    ra = t.ra, rb = t.rb, wa = t.wa;
  } else {
    ra = rb = y >> f.h_sub;
  }
  int u[8], v[8];
  {
    Chroma6 c;
    load_chroma6 (f, pl, ra, k0, cw, c);
    hfilter8<CH> (c, x0, w, u, v);
  }
  if (ra != rb) {
    Chroma6 c;
    int u2[8], v2[8];
    load_chroma6 (f, pl, rb, k0, cw, c);
    hfilter8<CH> (c, x0, w, u2, v2);
    const int wb = 8 - wa;          // over 8: vpair_get
#pragma unroll
    for (int i = 0; i < 8; i++) {
      u[i] = (wa * u[i] + wb * u2[i] + 4) >> 3;
      v[i] = (wa * v[i] + wb * v2[i] + 4) >> 3;
    }
  }
  const uint32_t yw[2] = {yy.x, yy.y};
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t Y = (yw[i >> 2] >> (8 * (i & 3))) & 0xff;
    out[i] = 0xffu | (Y << 8) | ((uint32_t) u[i] << 16) | ((uint32_t) v[i] << 24);
  }
}

GSTAMD_HD void front_span8_any (const FrontParams &f, const Planes &pl, const int *__restrict__ vpair, int x0, int y, uint32_t *out)
{
  if (f.chroma_h == CHROMA_H_H2_CS)
    front_span8<CHROMA_H_H2_CS> (f, pl, vpair, x0, y, out);
  else if (f.chroma_h == CHROMA_H_H2)
    front_span8<CHROMA_H_H2> (f, pl, vpair, x0, y, out);
  else
    front_span8<CHROMA_H_NONE> (f, pl, vpair, x0, y, out);
}

// a per-pixel step between the colour stage and the packer (default: none; the fused gamma kernel passes video_gamma.h's chain)
struct PxIdentity {
  GSTAMD_HD uint32_t operator() (uint32_t px) const { return px; }
};

template <int CH, class PXF = PxIdentity>
GSTAMD_HD void convert_body (const FrontParams &f, const Planes &pl, const int *__restrict__ vpair, const ColorParams &color,
    int pack0, int pack1, int pack2, int pack3, uint8_t *__restrict__ dst, int dstride, int spans_per_row, int vec_ok,
    int span, int y, const PXF &pxf = PXF ())
{
  if (span >= spans_per_row)
    return;
  const int x0 = span * K1_PX, w = f.width;
  const int pos[4] = {pack0, pack1, pack2, pack3};
  uint8_t *drow = dst + (size_t) y * dstride;
  if (vec_ok && f.w_sub == 1 && kind_has_planes (f.kind) && x0 + K1_PX <= w) {
    uint32_t px[8], out[8];
    front_span8<CH> (f, pl, vpair, x0, y, px);
#pragma unroll
    for (int i = 0; i < 8; i++)
      out[i] = pack_px (pos, pxf (apply_color (color, px[i])));
    uint4 *d = (uint4 *) (drow + 4 * (size_t) x0);
    d[0] = gstamd_make_uint4 (out[0], out[1], out[2], out[3]);
    d[1] = gstamd_make_uint4 (out[4], out[5], out[6], out[7]);
    return;
  }
  if (vec_ok == 2 && f.kind == UNPACK_PACKED4 && x0 + K1_PX <= w) {
    /* 4-byte packed source (swizzles, RGB <-> AYUV): eight pixels through two 16-byte loads and two 16-byte stores per lane */
    const uint4 *sp = (const uint4 *) (pl.p[0] + (size_t) y * pl.stride[0] + 4 * (size_t) x0);
    const uint4 a = sp[0], b = sp[1];
    const uint32_t raw[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t out[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const uint32_t r = raw[i];
      const uint32_t px = ((r >> (8 * f.pos[0])) & 0xff) | (((r >> (8 * f.pos[1])) & 0xff) << 8) | (((r >> (8 * f.pos[2])) & 0xff) << 16) |
          (((r >> (8 * f.pos[3])) & 0xff) << 24);
      out[i] = pack_px (pos, pxf (apply_color (color, px)));
    }
    uint4 *d = (uint4 *) (drow + 4 * (size_t) x0);
    d[0] = gstamd_make_uint4 (out[0], out[1], out[2], out[3]);
    d[1] = gstamd_make_uint4 (out[4], out[5], out[6], out[7]);
    return;
  }
  const int x1 = x0 + K1_PX < w ? x0 + K1_PX : w;
  for (int x = x0; x < x1; x++) {
    const uint32_t px = fetch_front (f, pl, vpair, x, y);
    *(uint32_t *) (drow + 4 * (size_t) x) = pack_px (pos, pxf (apply_color (color, px)));
  }
}

// ------------------------------------------------------------------------------------------------
// 4-byte packed -> 4-byte packed with neither a matrix nor an alpha operation (BGRA -> RGBA, ARGB -> BGRx ...: the unpack and the pack are
// both byte permutations, video-format.c pack / unpack_<4-byte formats>): ONE selector word, v_perm_b32 per pixel.  Destination byte
// pack_pos[i] takes component i, which the source keeps at byte pos[i].
inline uint32_t swizzle4_selector (const int *src_pos, const int *pack_pos)     /* host */
{
  uint32_t sel = 0;
  for (int i = 0; i < 4; i++)
    sel |= (uint32_t) src_pos[i] << (8 * pack_pos[i]);
  return sel;
}

GSTAMD_HD uint32_t swizzle4_px (uint32_t raw, uint32_t sel)
{
#ifdef __HIPCC__
  return __builtin_amdgcn_perm (0u, raw, sel);
#else
  uint32_t r = 0;
  for (int j = 0; j < 4; j++)
    r |= ((raw >> (8 * ((sel >> (8 * j)) & 3))) & 0xffu) << (8 * j);
  return r;
#endif
}

// ------------------------------------------------------------------------------------------------
// scaler kernels: one lane per output pixel, x fastest
// ------------------------------------------------------------------------------------------------
// (acc + 32) as int16, arithmetic >> 6, unsigned saturate to a byte.
// NB gfx950: hipcc (ROCm 7.2) fuses "clamp (x >> n, 0, 255)" pairs into v_ashr_pk_u8_i32, whose second
// lane came back wrong on MI355X for this pattern (first GPU run, 2 of 4 channels off).  The opaque
// register barrier keeps shift and saturate as separate VALU ops.
GSTAMD_HD int lq_round (int acc)
{
  int v = ((int) (int16_t) (uint16_t) (acc + 32)) >> 6;
#ifdef __HIPCC__
  asm volatile ("" : "+v" (v));
#endif
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

GSTAMD_HD uint32_t lq_finish (int a0, int a1, int a2, int a3)
{
  // addw 32 (wrap), shrsw 6, convsuswb
  return (uint32_t) lq_round (a0) | ((uint32_t) lq_round (a1) << 8) | ((uint32_t) lq_round (a2) << 16) |
      ((uint32_t) lq_round (a3) << 24);
}

// one horizontally scaled pixel; ROW::at (xs) yields source pixel xs of the current row
template <class ROW>
GSTAMD_HD uint32_t hscale_px (const ROW &row, const ScaleDev &sd, int x)
{
  if (sd.kind == SCALE_NEAREST)
    return row.at ((int) sd.offset[x]);
  if (sd.kind == SCALE_2TAP) {
    const int tmp = x * sd.inc;                      // ldreslinl, p1 = 0
    const int idx = tmp >> 16, fr = (tmp >> 8) & 0xff;
    const uint32_t a = row.at (idx), b = row.at (idx + 1);
    uint32_t r = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int av = (a >> (8 * c)) & 0xff, bv = (b >> (8 * c)) & 0xff;
      r |= (uint32_t) (((av * (256 - fr) + bv * fr) >> 8) & 0xff) << (8 * c);
    }
    return r;
  }
  const int off = (int) sd.offset[x];
  const int16_t *t = sd.taps + (size_t) x * sd.n_taps;
  const int st = sd.merged == 0 ? 1 : (((x & 1) == sd.merged - 1) ? 2 : 4);         /* realize_taps (video-scaler.c:431-436) */
  int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (int l = 0; l < sd.n_taps; l++) {
    const uint32_t p = row.at (off + l * st);
    const int tp = t[l];
    a0 += (int) (p & 0xff) * tp;                     // mullw/addw: only the low 16 bits survive
    a1 += (int) ((p >> 8) & 0xff) * tp;
    a2 += (int) ((p >> 16) & 0xff) * tp;
    a3 += (int) (p >> 24) * tp;
  }
  return lq_finish (a0, a1, a2, a3);
}

// video_orc_resample_v_2tap_u8_lq on two pixels
GSTAMD_HD uint32_t v2tap_px (uint32_t a, uint32_t b, int p1)
{
  uint32_t r = 0;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int s1 = (a >> (8 * c)) & 0xff, s2 = (b >> (8 * c)) & 0xff;
    int w2 = (int) (int16_t) (s2 - s1);
    w2 = (int) (int16_t) (w2 * p1);               // mullw
    w2 = (int) (int16_t) (w2 + 128);              // addw
    const int t = (w2 >> 8) & 0xff;               // convhwb
    r |= (uint32_t) ((t + s1) & 0xff) << (8 * c); // addb
  }
  return r;
}

template <class SRC>
struct RowOfSrc {
  const SRC &src;
  int y;
  GSTAMD_HD uint32_t at (int x) const { return src.at (x, y); }
};

template <class SRC>
struct RowVFiltered {
  const SRC &src;
  int ya, p1;
  bool v2;
  GSTAMD_HD uint32_t at (int xs) const
  {
    const uint32_t a = src.at (xs, ya);
    return v2 ? v2tap_px (a, src.at (xs, ya + 1), p1) : a;
  }
};

struct RowVLds {
  const uint32_t *a, *b;
  int x_lo, p1;
  bool v2;
  GSTAMD_HD uint32_t at (int x) const { return v2 ? v2tap_px (a[x - x_lo], b[x - x_lo], p1) : a[x - x_lo]; }
};

struct RowOfLds {
  const uint32_t *lds;
  int x_lo;
  GSTAMD_HD uint32_t at (int x) const { return lds[x - x_lo]; }
};

template <class SRC>
GSTAMD_HD void hscale_body (const SRC &src, const ScaleDev &sd, const Dst &dst, int out_w, int rows, int x, int y)
{
  if (x >= out_w || y >= rows)
    return;
  const RowOfSrc<SRC> row = {src, y};
  dst.put (x, y, hscale_px (row, sd, x));
}

// source span [x_lo, x_hi) of row pixels needed by outputs [t0, t1)
GSTAMD_HD void hscale_span (const ScaleDev &sd, int t0, int t1, int *x_lo, int *x_hi)
{
  if (sd.kind == SCALE_2TAP) {
    *x_lo = (t0 * sd.inc) >> 16;
    *x_hi = (((t1 - 1) * sd.inc) >> 16) + 2;
  } else {
    const int n = sd.kind == SCALE_NEAREST ? 1 : sd.n_taps;
    *x_lo = (int) sd.offset[t0];
    *x_hi = (int) sd.offset[t1 - 1] + n;
  }
}

// LDS-staged horizontal pass, phase 1: lane `tid` of `nthreads` evaluates its share of the span ONCE
// (the front functor - unpack + chroma upsample - is no longer re-evaluated per tap)
template <class SRC>
GSTAMD_HD void hscale_stage (const SRC &src, uint32_t *lds, int x_lo, int x_hi, int y, int tid, int nthreads)
{
  src.stage (lds, x_lo, x_hi, y, tid, nthreads);
}

// fused 2x2 scaler from staged rows: lds_a / lds_b hold pixels [x_lo, x_hi) of source lines ya / ya+1
GSTAMD_HD uint32_t scale2x2_from_lds (const uint32_t *lds_a, const uint32_t *lds_b, int x_lo, const ScaleDev &sh, const ScaleDev &sv,
    int h_first, int x, int y)
{
  const bool v2 = sv.kind == SCALE_2TAP;
  const int p1 = v2 ? (int) sv.taps[(size_t) y * 2 + 1] : 0;
  if (h_first) {
    const RowOfLds ra = {lds_a, x_lo};
    const uint32_t ha = hscale_px (ra, sh, x);
    if (!v2)
      return ha;
    const RowOfLds rb = {lds_b, x_lo};
    return v2tap_px (ha, hscale_px (rb, sh, x), p1);
  }
  const RowVLds vr = {lds_a, lds_b, x_lo, p1, v2};
  return hscale_px (vr, sh, x);
}

// phase 2 (after the barrier): one output pixel from the staged row
GSTAMD_HD void hscale_from_lds (const uint32_t *lds, int x_lo, const ScaleDev &sd, const Dst &dst, int x, int y)
{
  const RowOfLds row = {lds, x_lo};
  dst.put (x, y, hscale_px (row, sd, x));
}

// Fused 2x2 scaler: both passes are nearest or 2-tap ("bilinear", the element default): one lane = one output
// pixel computed straight from <= 4 source pixels, pass order as planned (h_first), no intermediate image.
template <class SRC>
GSTAMD_HD uint32_t scale2x2_px (const SRC &src, const ScaleDev &sh, const ScaleDev &sv, int h_first, int x, int y)
{
  const int ya = (int) sv.offset[y];
  const bool v2 = sv.kind == SCALE_2TAP;
  const int p1 = v2 ? (int) sv.taps[(size_t) y * 2 + 1] : 0;
  if (h_first) {
    const RowOfSrc<SRC> ra = {src, ya};
    const uint32_t ha = hscale_px (ra, sh, x);
    if (!v2)
      return ha;
    const RowOfSrc<SRC> rb = {src, ya + 1};
    return v2tap_px (ha, hscale_px (rb, sh, x), p1);
  }
  const RowVFiltered<SRC> vr = {src, ya, p1, v2};
  return hscale_px (vr, sh, x);
}

template <class SRC>
GSTAMD_HD void scale2x2_body (const SRC &src, const ScaleDev &sh, const ScaleDev &sv, int h_first, const Dst &dst, int out_w,
    int out_h, int x, int y)
{
  if (x >= out_w || y >= out_h)
    return;
  const int ya = (int) sv.offset[y];
  const bool v2 = sv.kind == SCALE_2TAP;
  const int p1 = v2 ? (int) sv.taps[(size_t) y * 2 + 1] : 0;
  uint32_t r;
  if (h_first) {
    const RowOfSrc<SRC> ra = {src, ya};
    const uint32_t ha = hscale_px (ra, sh, x);
    if (v2) {
      const RowOfSrc<SRC> rb = {src, ya + 1};
      r = v2tap_px (ha, hscale_px (rb, sh, x), p1);
    } else {
      r = ha;
    }
  } else {
    // vertical first: the horizontal filter then reads vertically filtered pixels
    const RowVFiltered<SRC> vr = {src, ya, p1, v2};
    r = hscale_px (vr, sh, x);
  }
  dst.put (x, y, r);
}

// vertical pass, one output pixel (4 x u8 word): video_scale_v_near_u8 / v_2tap_u8 / v_4tap_u8 / v_ntap_u8
template <class SRC>
GSTAMD_HD uint32_t vscale_px (const SRC &src, const ScaleDev &sd, int x, int y)
{
  const int off = (int) sd.offset[y];
  if (sd.kind == SCALE_NEAREST)
    return src.at (x, off);
  if (sd.kind == SCALE_2TAP) {
    const int p1 = sd.taps[(size_t) y * 2 + 1];
    return v2tap_px (src.at (x, off), src.at (x, off + 1), p1);
  }
  const int16_t *t = sd.taps + (size_t) y * sd.n_taps;
  int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (int l = 0; l < sd.n_taps; l++) {
    const uint32_t p = src.at (x, off + l);
    const int tp = t[l];
    a0 += (int) (p & 0xff) * tp;
    a1 += (int) ((p >> 8) & 0xff) * tp;
    a2 += (int) ((p >> 16) & 0xff) * tp;
    a3 += (int) (p >> 24) * tp;
  }
  return lq_finish (a0, a1, a2, a3);
}

template <class SRC>
GSTAMD_HD void vscale_body (const SRC &src, const ScaleDev &sd, const Dst &dst, int width, int out_h, int x, int y)
{
  if (x >= width || y >= out_h)
    return;
  dst.put (x, y, vscale_px (src, sd, x, y));
}


}  // namespace gstamd
