// planner.h - host-side "bit-exactness brain" of the GPU GstVideoConverter.
//
// Reproduces the DECISIONS gst_video_converter_new() takes (reference:
// subprojects/gst-plugins-base/gst-libs/gst/video/video-converter.c:2421-2563 and the chain_*
// builders :851-2112) and turns them into a launch plan of fused HIP kernels.  No pixel is touched
// here; everything is integer/double set-up math that must agree with the reference bit for bit
// (matrix coefficients, resampler taps, chroma line pairing).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/gstamd_video.h"

namespace gstamd {

// ---- per-format description (own table; facts from video-format.c:8190-8235) ------------------
enum UnpackKind : int {
  UNPACK_PACKED4 = 0,   // one plane, 4 bytes / pixel, pure byte permutation to A,c1,c2,c3
  UNPACK_PLANAR = 1,    // separate Y, U, V planes (I420, YV12, Y42B, Y444)
  UNPACK_SEMI = 2,      // Y plane + interleaved UV plane (NV12, NV21, NV16, NV61, NV24)
  UNPACK_PACKED3 = 3,   // one plane, 3 bytes / pixel (RGB, BGR)
  UNPACK_PACKED422 = 4, // one plane, 4-byte macropixels of two pixels (YUY2, UYVY, YVYU, VYUY)
  UNPACK_GRAY = 5,      // one plane of luma (GRAY8): a YUV format as far as the chain goes (unpack format AYUV), U = V = 0x80
  UNPACK_P422_16 = 6,   // one plane, macropixels of four 16-bit little-endian words Y0 U Y1 V (Y210, Y212_LE): a source / destination of the 16-bit chain
  UNPACK_Y410 = 7,      // one plane, a little-endian 32-bit word per pixel: three 10-bit fields and 2 bits of alpha on top (Y410: U, Y, V from the low
                        // bits; RGB10A2_LE: R, G, B; BGR10A2_LE: B, G, R); hi_depth code 7; FormatDesc::pos[1..3] = first bit of component c1, c2, c3
  UNPACK_V210 = 8,      // one plane, groups of six pixels in 16 bytes: three 10-bit samples per little-endian 32-bit word (v210); hi_depth code 8
  UNPACK_PACKED64 = 9,  // one plane, four 16-bit words per pixel in an order of their own (RGBA64_LE / _BE, BGRA64, ABGR64, ARGB64_BE; unpack format
                        // ARGB64): FormatDesc::pos[c] = word of component c (A, R, G, B); hi_depth code 9 (little endian) / 10 (big endian)
  UNPACK_GRAY16 = 10,   // one plane of 16-bit luma (GRAY16_LE / _BE; unpack format AYUV64, U = V = 0x8000); hi_depth code 9 / 10
  UNPACK_PLANAR_A = 12, // I420's planes plus a full-size alpha plane (A420: plane 3); generic per-pixel kernels only (kind_has_planes is false for it)
  UNPACK_SEMI_A = 13,   // NV12's planes plus a full-size alpha plane (AV12: plane 2); generic per-pixel kernels only
  UNPACK_PLANAR_H4 = 14, // Y, U, V planes with one chroma sample per FOUR pixels of a line (Y41B); generic per-pixel kernels only (kind_has_planes is false)
  UNPACK_PACKED411 = 15, // one plane, groups of four pixels in six bytes U Y0 Y1 V Y2 Y3 (IYU1; unpack format AYUV, 8-bit chain, w_sub 2 like Y41B); generic per-pixel
                        // kernels only; whole frames only (unpack_IYU1 steps a horizontal offset by x * 4 BYTES, "FIXME" video-format.c:2382: crops are refused)
  UNPACK_SEMI_LE32 = 16, // NV12 / NV16 with three 10-bit samples per little-endian 32-bit word in both planes (NV12_10LE32, NV16_10LE32; hi_depth code 13; the UV
                        // plane's samples run U0 V0 U1 | V1 U2 V2 ...); 16-bit chain, a lane per pixel in, a lane per six pixels out; whole frames only
  UNPACK_GRAY_LE32 = 17, // the luma plane of those alone (GRAY10_LE32; U = V = 0x8000)
  UNPACK_SEMI_LE40 = 18, // NV12 / NV16 whose planes are little-endian streams of 10-bit samples, four in five bytes (NV12_10LE40, NV16_10LE40; hi_depth code 14; the UV
                        // plane's samples run U0 V0 U1 V1 ...); 16-bit chain, a lane per pixel in, the packer's four-pixel block = five bytes out; whole frames only
  UNPACK_P422_UYVP = 19, // one plane, macropixels of four 10-bit samples U Y0 V Y1 as a big-endian bit stream in five bytes (UYVP; hi_depth code 15): Y210's chain and
                        // Y210's rules (an odd line's last macropixel repeats its luma), a lane per macropixel out; whole frames only
  UNPACK_SEMI_TILED = 20, // NV12 stored in tiles (NV12_64Z32, NV12_4L4, NV12_32L32, NV12_16L32S, NV12_8L128; unpack_TILED / pack_TILED video-format.c:5083-5183 hand each
                        // tile to unpack_NV12 / pack_NV12): FormatDesc::pos = {tile mode (0 linear, 1 ZFLIPZ_2X2), log2 tile width, log2 tile height, sub-tiled UV plane};
                        // stride[] of such a frame is GST_VIDEO_TILE_MAKE_STRIDE (x tiles, y tiles).  8-bit chain, generic per-pixel kernels, whole frames only
  UNPACK_SEMI_LE40_TILED = 21, // NV12_10LE40_4L4: UNPACK_SEMI_LE40's sample stream restarting in every row of a 4 x 4 tile (five bytes; TILE_10bit_4x4, video-format.c:8134), tiles
                        // addressed like UNPACK_SEMI_TILED's (pos = {mode, ws, hs, sub-tiles})
  UNPACK_RGB16 = 11,    // one plane, a little-endian 16-bit word per pixel with 5-6-5 or 5-5-5 bit fields (RGB16, BGR16, RGB15, BGR15; unpack format ARGB,
                        // 8-bit chain): FormatDesc::pos = {bits of G, first bit of R, of G, of B}
};
// one plane of whole pixels whose samples reach the 16-bit chain through deep_front_px and leave it through pack16_packed_body, a lane per pixel
// the full-size alpha plane of a kind that has one (A420 & co: 3, AV12: 2), else -1; interleaved U / V samples in plane 1
#define GSTAMD_KIND_ALPHA_PLANE(k) ((k) == UNPACK_PLANAR_A ? 3 : ((k) == UNPACK_SEMI_A ? 2 : -1))
#define GSTAMD_KIND_SEMI(k) ((k) == UNPACK_SEMI || (k) == UNPACK_SEMI_A)
#define GSTAMD_KIND_LE32(k) ((k) == UNPACK_SEMI_LE32 || (k) == UNPACK_GRAY_LE32 || (k) == UNPACK_SEMI_LE40 || (k) == UNPACK_P422_UYVP || (k) == UNPACK_SEMI_LE40_TILED)          /* (the sample-stream kinds: three per word, four per five bytes) */
#define GSTAMD_KIND_PX16(k) ((k) == UNPACK_Y410 || (k) == UNPACK_PACKED64 || (k) == UNPACK_GRAY16)

struct FormatDesc {
  int format;
  const char *name;
  bool yuv;             // GST_VIDEO_FORMAT_FLAG_YUV (else RGB)
  bool alpha;           // GST_VIDEO_FORMAT_FLAG_ALPHA
  int n_planes;
  int kind;             // UnpackKind
  int w_sub, h_sub;     // log2 chroma subsampling (finfo->w_sub[1], h_sub[1])
  int u_plane, v_plane; // planar: plane index of U and V; semi: u_first (1) / v_first (0) in u_plane
  int pos[4];           // packed4 / packed3: memory byte index of unpacked component 0..3 (A,R,G,B / A,Y,U,V; packed3 has no A);
                        // packed422: byte of Y0, U, V inside the macropixel in pos[1..3] (Y1 is at pos[1] + 2)
  int hi_depth;         // 0: 8-bit samples.  3: 4 x 16-bit components per pixel (ARGB64 / AYUV64; kind UNPACK_PACKED4 with 8-byte pixels).
                        // 1 / 2: 10-bit samples in little-endian 16-bit words, in the low bits (I420_10LE) / the
                        // high bits (P010_10LE); such formats unpack to AYUV64 in the reference (video-format.c:3836, 5331)
                        // 4 / 5: the same with 12 bits (I420_12LE ... / P012_LE), 6: all 16 bits (P016_LE, Y444_16LE)
                        // 7: Y410's 10 + 10 + 10 + 2 bits in a 32-bit word, 8: v210's three 10-bit samples per 32-bit word
};
// GBR keeps its planes in the order G, B, R; inside a plan (VideoPlan::in_info / out_info, the copies every launcher takes plane pointers from)
// they are R, G, B - component order, what the planar 4:4:4 code expects (unpack_GBR is video_orc_unpack_Y444 on the R, G, B lines,
// video-format.c:1121-1147).  Callers keep the frame's own layout: only the plan's copies are permuted.
// perm[i]: the FRAME plane that is plane i of a plan (component order R, G, B, A).  false: the frame's own order is the plan's.
// (the planar RGB family of video-format.c: GBR & its 10 / 12 / 16-bit forms and GBRA keep G, B, R (A); BGRP B, G, R; RGBP R, G, B)
inline bool format_plane_perm (int format, int perm[4])
{
  perm[0] = 0, perm[1] = 1, perm[2] = 2, perm[3] = 3;
  switch (format) {
    case GSTAMD_VIDEO_FORMAT_GBR:
    case GSTAMD_VIDEO_FORMAT_GBRA:
    case GSTAMD_VIDEO_FORMAT_GBR_10LE:
    case GSTAMD_VIDEO_FORMAT_GBR_12LE:
    case GSTAMD_VIDEO_FORMAT_GBR_16LE:
    case GSTAMD_VIDEO_FORMAT_GBRA_10LE:
    case GSTAMD_VIDEO_FORMAT_GBRA_12LE:
    case GSTAMD_VIDEO_FORMAT_GBR_10BE:
    case GSTAMD_VIDEO_FORMAT_GBR_12BE:
    case GSTAMD_VIDEO_FORMAT_GBR_16BE:
    case GSTAMD_VIDEO_FORMAT_GBRA_10BE:
    case GSTAMD_VIDEO_FORMAT_GBRA_12BE:
      perm[0] = 2, perm[1] = 0, perm[2] = 1;
      return true;
    case GSTAMD_VIDEO_FORMAT_BGRP:
      perm[0] = 2, perm[1] = 1, perm[2] = 0;
      return true;
    default:
      return false;
  }
}
inline void format_plan_planes (const FormatDesc *f, GstAmdVideoInfo *info)
{
  int perm[4];
  if (f && format_plane_perm (f->format, perm)) {
    const GstAmdVideoInfo o = *info;
    for (int i = 0; i < 4; i++)
      info->offset[i] = o.offset[perm[i]], info->stride[i] = o.stride[perm[i]];
  }
}

#if defined(__HIPCC__)
#define GSTAMD_VP __host__ __device__ inline
#else
#define GSTAMD_VP inline
#endif
// significant bits of a sample for a FormatDesc::hi_depth code, and whether the format keeps its samples in 16-bit words in planes
// big-endian forms of the word-plane codes: code + 20 (21 / 24: 10 / 12 bits in the low bits, 22 / 25: in the high bits, 26: 16 bits); the 16-byte vector
// kernels of the little-endian forms do not serve them (deep_front4_variant, planes_fast, k_encode16), the per-sample ones swap on the way
GSTAMD_VP bool hi_depth_be (int hi) { return hi >= 20; }
GSTAMD_VP int hi_depth_le (int hi) { return hi >= 20 ? hi - 20 : hi; }
GSTAMD_VP int bswap16i (int v) { return ((v >> 8) | (v << 8)) & 0xffff; }
GSTAMD_VP int hi_depth_bits_le (int hi);
GSTAMD_VP int hi_depth_bits (int hi) { return hi_depth_bits_le (hi_depth_le (hi)); }
GSTAMD_VP int hi_depth_bits_le (int hi) { return hi == 1 || hi == 2 || hi == 7 || hi == 8 || hi == 13 || hi == 14 || hi == 15 ? 10 : (hi == 4 || hi == 5 || hi == 11 || hi == 12 ? 12 : (hi == 3 || hi == 6 || hi == 9 || hi == 10 || hi == 16 ? 16 : 8)); }
// ---- tiled NV12 (UNPACK_SEMI_TILED): byte offsets inside the two planes
// gst_video_tile_get_index (video-tile.c:48-118)
GSTAMD_VP int tile_index (int mode, int x, int y, int x_tiles, int y_tiles)
{
  if (mode == 0)
    return y * x_tiles + x;
  int off = (y & ~1) * x_tiles + x;
  if (y & 1)
    off += 2 + (x & ~3);
  else if ((y_tiles & 1) == 0 || y != y_tiles - 1)
    off += (x + 2) & ~3;
  return off;
}
// luma sample (x, y): tile (x >> ws, y >> hs) of get_tile_NV12 (:5054-5081), then unpack_NV12's addressing inside the tile (tile stride = tile width)
GSTAMD_VP size_t tiled_luma_offset (const int *tp, int stride, int x, int y)
{
  const int ws = tp[1], hs = tp[2], tw = 1 << ws, th = 1 << hs;
  const int idx = tile_index (tp[0], x >> ws, y >> hs, stride & 0xffff, stride >> 16);
  return (size_t) idx * (size_t) (tw * th) + (size_t) ((y & (th - 1)) * tw + (x & (tw - 1)));
}
// the U byte of chroma pair k of chroma row crow (V follows it): the UV tile under luma tile row ty - its own tile for sub-tiled formats, else the tile
// of row ty / 2, odd rows in its second half -, row (y in tile) >> 1, byte (x in tile) & ~1 (unpack_NV12 :1597-1630 on the tile)
// the same two for tiles whose rows are `rowb` bytes of a packed sample stream (NV12_10LE40_4L4: 5): the first byte of the tile row that holds the sample
GSTAMD_VP size_t tiled_luma_row (const int *tp, int stride, int rowb, int x, int y)
{
  const int ws = tp[1], hs = tp[2], th = 1 << hs;
  const int idx = tile_index (tp[0], x >> ws, y >> hs, stride & 0xffff, stride >> 16);
  return (size_t) idx * (size_t) (rowb * th) + (size_t) ((y & (th - 1)) * rowb);
}
GSTAMD_VP size_t tiled_uv_row (const int *tp, int stride, int rowb, int k, int crow)
{
  const int ws = tp[1], hs = tp[2], th = 1 << hs, sub = tp[3];
  const int x = 2 * k, y = 2 * crow, ty = y >> hs;
  const int size1 = sub ? rowb * (th >> 1) : rowb * th;
  const int idx = tile_index (tp[0], x >> ws, sub ? ty : ty >> 1, stride & 0xffff, stride >> 16);
  size_t base = (size_t) idx * (size_t) size1;
  if (!sub && (ty & 1))
    base += (size_t) (size1 >> 1);
  return base + (size_t) (((y & (th - 1)) >> 1) * rowb);
}
GSTAMD_VP size_t tiled_uv_offset (const int *tp, int stride, int k, int crow)
{
  const int ws = tp[1], hs = tp[2], tw = 1 << ws, th = 1 << hs, sub = tp[3];
  const int x = 2 * k, y = 2 * crow, ty = y >> hs;
  const int size1 = sub ? tw * (th >> 1) : tw * th;
  const int idx = tile_index (tp[0], x >> ws, sub ? ty : ty >> 1, stride & 0xffff, stride >> 16);
  size_t base = (size_t) idx * (size_t) size1;
  if (!sub && (ty & 1))
    base += (size_t) (size1 >> 1);
  return base + (size_t) (((y & (th - 1)) >> 1) * tw + (x & (tw - 1)));
}

// video_orc_unpack_RGB16 & co (video-orc.orc: mulhsw by 0x4200 / 0x2080 / 0x0210 = field * 8.25 or * 4.0625, floored): the field's bits replicated
GSTAMD_VP int rgb16_field (int word, int shift, int bits) { const int f = (word >> shift) & ((1 << bits) - 1); return bits == 6 ? (f << 2) | (f >> 4) : (f << 3) | (f >> 2); }
// video_orc_pack_RGB16_le & co: the top bits of every component at its field
GSTAMD_VP int rgb16_pack (const int *pos, int r, int g, int b) { return ((r >> 3) << pos[1]) | ((g >> (8 - pos[0])) << pos[2]) | ((b >> 3) << pos[3]); }
// a stored 16-bit word of a UNPACK_PACKED64 / UNPACK_GRAY16 format -> its value (GST_READ_UINT16_LE / _BE; the same function stores)
GSTAMD_VP int px16_word (int hi, int v) { return hi == 10 ? ((v >> 8) | (v << 8)) & 0xffff : v; }
// hi_depth code 11 (Y412_LE): 12 bits in the high bits of a little-endian word - read: masked and widened (v | v >> 12), stored: masked
// (code 12: the same big endian, Y412_BE)
// RGBA_F16LE / _BE (hi_depth codes 16 / 36): unpack_RGBA_F16LE (video-format.c:2830-2851) = float_to_u16 (gst_half_to_float (word)), pack = gst_float_to_half
// (value * (1.0f / 65535.0f)); the two conversions restated from gstvideoutilsprivate.h:36-118 (binary16 <-> binary32 on the bit patterns, round to nearest
// even, subnormals both ways), float_to_u16 from video-format.c:2818-2826 (NaN and negatives 0, >= 1 65535, else v * 65535.0f + 0.5f truncated)
GSTAMD_VP float half_bits_to_float (int h)
{
  const uint32_t sign = ((uint32_t) h & 0x8000u) << 16;
  uint32_t exponent = ((uint32_t) h >> 10) & 0x1fu, mantissa = (uint32_t) h & 0x3ffu, bits;
  if (exponent == 0) {
    if (mantissa == 0) {
      bits = sign;
    } else {
      exponent = 127 - 15 + 1;
      while ((mantissa & 0x400u) == 0) {
        mantissa <<= 1;
        exponent--;
      }
      mantissa &= 0x3ffu;
      bits = sign | (exponent << 23) | (mantissa << 13);
    }
  } else if (exponent == 31) {
    bits = sign | 0x7f800000u | (mantissa << 13);
  } else {
    bits = sign | ((exponent - 15 + 127) << 23) | (mantissa << 13);
  }
  float f;
  __builtin_memcpy (&f, &bits, 4);
  return f;
}
GSTAMD_VP int float_to_half_bits (float f)
{
  uint32_t bits;
  __builtin_memcpy (&bits, &f, 4);
  const uint32_t sign = (bits >> 16) & 0x8000u;
  const int exponent = (int) ((bits >> 23) & 0xffu) - 127 + 15;
  uint32_t mantissa = bits & 0x7fffffu;
  if (((bits >> 23) & 0xffu) == 0xffu) {
    if (mantissa == 0)
      return (int) (sign | 0x7c00u);
    mantissa >>= 13;
    return (int) (sign | 0x7c00u | mantissa | (mantissa == 0 ? 1u : 0u));
  }
  if (exponent >= 31)
    return (int) (sign | 0x7c00u);
  if (exponent <= 0) {
    if (exponent < -10)
      return (int) sign;
    mantissa |= 0x800000u;
    const uint32_t shift = (uint32_t) (14 - exponent);
    uint32_t h = mantissa >> shift;
    const uint32_t rest = mantissa & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rest > halfway || (rest == halfway && (h & 1u)))
      h++;
    return (int) ((sign | h) & 0xffffu);
  }
  uint32_t h = ((uint32_t) exponent << 10) | (mantissa >> 13);
  const uint32_t rest = mantissa & 0x1fffu;
  if (rest > 0x1000u || (rest == 0x1000u && (h & 1u)))
    h++;
  return (int) ((sign | h) & 0xffffu);
}
GSTAMD_VP int half_word_to_u16 (int h)
{
  const float v = half_bits_to_float (h);
  if (!(v > 0.0f))
    return 0;
  if (v >= 1.0f)
    return 65535;
  const float m = v * 65535.0f;          /* (two roundings, as the reference compiles it: no fused multiply-add - the library is built with -ffp-contract=off) */
  return (int) (uint16_t) (m + 0.5f);
}
GSTAMD_VP int u16_to_half_word (int v) { return float_to_half_bits ((float) v * (1.0f / 65535.0f)); }
GSTAMD_VP int px16_load (int hi, int v) {
  if (hi == 16 || hi == 36)
    return half_word_to_u16 (hi == 36 ? bswap16i (v) : v); if (hi == 12) v = bswap16i (v); return hi == 11 || hi == 12 ? (v & 0xfff0) | ((v & 0xfff0) >> 12) : px16_word (hi, v); }
GSTAMD_VP int px16_store (int hi, int v) {
  if (hi == 16)
    return u16_to_half_word (v);
  if (hi == 36)
    return bswap16i (u16_to_half_word (v)); return hi == 11 ? v & 0xfff0 : hi == 12 ? bswap16i (v & 0xfff0) : px16_word (hi, v); }
GSTAMD_VP bool hi_depth_words (int hi) { hi = hi_depth_le (hi); return hi == 1 || hi == 2 || (hi >= 4 && hi <= 6); }
const FormatDesc *format_desc (int format);

// ---- device-consumable plan pieces (POD, passed to kernels by value) --------------------------
enum ChromaH : int { CHROMA_H_NONE = 0, CHROMA_H_H2 = 1, CHROMA_H_H2_CS = 2,
  CHROMA_H_H4 = 3, CHROMA_H_H4_CS = 4 };        // 4:1:1 sources (video_chroma_up_h4_u8 / _h4_cs_u8 video-chroma.c:493-517, 819-838): chroma_h_at only
enum MatrixKind : int { MATRIX_NONE = 0, MATRIX_AYUV_ARGB = 1, MATRIX_TABLE = 2, MATRIX_8 = 3 };
enum AlphaKind : int { ALPHA_NONE = 0, ALPHA_SET = 1, ALPHA_MULT = 2 };

struct MatrixParams {
  int kind;
  int p[5];        // MATRIX_AYUV_ARGB: p1..p5 of video_orc_convert_AYUV_ARGB
  int im[3][4];    // MATRIX_TABLE / MATRIX_8: integer matrix rows (im[k][3] = offset)
};

struct PostParams {     // stages after the (optional) scaler, fused into the last kernel
  MatrixParams matrix;
  int alpha_kind;
  int alpha_value;      // convert->alpha_value = 255 * alpha (video-converter.c:2368)
  int pack_pos[4];      // destination byte index of component 0..3
};

struct FrontParams {    // unpack + chroma upsample of the source frame (functional form)
  int kind;             // UnpackKind
  int width, height;    // in_width / in_height (no crop support yet: == maxwidth/maxheight)
  int w_sub, h_sub;
  int u_plane, v_plane; // see FormatDesc
  int pos[4];
  int chroma_h;         // ChromaH
  int chroma_v2;        // 1: vertical 2x upsample through the pair table (weights 3:1 / 1:3 by the entry's role); 2: one FIELD of an interlaced frame -
                        // every line's two chroma rows come from the table, in rows of the FRAME's chroma planes (which the field conversion is
                        // handed unsplit), with the weight of the first over 8 in the entry (vpair_get)
  int swap_k;           // chroma sample whose U and V trade places, -1: none.  unpack_VYUY (video-format.c:346-352) reads the last
                        // pixel of an odd-width line in UYVY order
  int hi_depth;         // FormatDesc::hi_depth: samples are 16-bit words widened to 16 significant bits (v << 6 | v >> 4, or v | v >> 10)
  int luma_last;        // last luma row the unpacker can deliver, relative to the crop origin (do_unpack_lines clamps to the FRAME, :2966): only a
                        // line past the picture (PackPlanarParams::virtual_line) ever reaches it
};

// the 16-bit chain (unpack to AYUV64 -> chroma upsample on u16 -> video_converter_matrix16 -> video_orc_convert_u16_to_u8):
// what a 10-bit source takes on its way to an 8-bit destination (video-converter.c:1719-1868, 3096-3145)
// The dither stage (chain_dither video-converter.c:2035-2100, video-dither.c) as a pass over the packed 4-byte destination: it works per
// component and the pack stage after it only permutes bytes, so dithering byte b of the packed pixel with the shift of the component
// stored there is the same thing.  method: GSTAMD_DITHER_NONE (quantise only) or GSTAMD_DITHER_BAYER.
struct DitherParams {
  int on;
  int method;
  int shift[4];         // per BYTE of the packed pixel: log2 of the quantiser of the component stored there (0: untouched)
  int y0;               // row of the destination FRAME the converted rectangle starts at: do_dither_lines hands gst_video_dither_line the
                        // frame's line number (out_line = i + out_y, video-converter.c:3236, 3390), x counts from the rectangle's left edge
};

// plane-to-plane form of the 16-bit chain when nothing mixes samples (video_deep.h deep_planes_body)
struct DeepPlanesParams {
  int width, height;
  int w_sub, h_sub;
  int in_kind, out_kind;        // UNPACK_PLANAR / UNPACK_SEMI
  int in_hi, out_hi;            // FormatDesc::hi_depth of the two formats
  int in_u, in_v, out_u, out_v; // planar: plane index of U and V; semi: u_first (1) / v_first (0) in *_u
  DitherParams dither;          // 16-bit dither of a 10-bit destination (shift[1..3] = Y, U, V)
};

struct Deep16Params {
  int has_matrix;       // 0: same matrix on both sides, the stage only narrows 16 -> 8 bits
  int im[3][4];         // video_converter_matrix16's integer matrix (8 fractional bits; im[k][3] the offset)
};

// planar / semi-planar destination: chroma downsample (video-chroma.c) + pack (video-format.c pack_planar_420 / pack_NV12 /
// pack_Y42B / pack_Y444) of the final AYUV image
struct PackPlanarParams {
  int width, height;    // output size
  int kind;             // UnpackKind of the destination (anything but UNPACK_PACKED4)
  int pos[4];           // FormatDesc::pos of the destination (packed3 / packed422)
  int w_sub, h_sub;
  int u_plane, v_plane; // planar: plane index of U and V; semi: u_first (1) / v_first (0) in u_plane
  int down_h;           // 0: the even pixel's chroma as it is, 1: video_orc_chroma_down_h2_u8, 2: video_chroma_down_h2_cs_u8; 4:1:1 destinations: 3
                        // video_chroma_down_h4_u8, 4 video_chroma_down_h4_cs_u8 (video-chroma.c:595-612, 880-905)
  int down_v;           // 0: the even line's chroma as it is, 1: video_orc_chroma_down_v2_u8 over lines (2r, 2r+1)
  int tail_swap;        // 1: the last pixel of an odd-width line stores U and V the other way round - pack_VYUY writes it in UYVY
                        // order (video-format.c:374-380), pack_NV61 in NV16 order (:2005-2011); 2 (Y210 / Y212): the second luma of an odd-width line's
                        // last macropixel is not the packer's to write (it is the border's luma: border_picture_positions)
  int virtual_line;     // 1: row `height` of the AYUV image holds the line past an odd-height picture as the chain delivers it (unpack clamped to the
                        // last line, chroma upsampler pairing it anew) - the last 4:2:0 chroma row averages the last line with THAT line
  DitherParams dither;  // chain_dither ahead of the pack (between chroma downsampling and packing): shift[] in unpack order (A, Y, U, V)
  // v210 destinations with a rectangle: pack_v210 packs the FRAME line in groups of six pixels (video_converter_generic packs out_maxwidth pixels a
  // line, :3274), so a group can hold border and picture pixels.  frame_on 1: the rectangle's rows only (fill-border off: groups that touch the
  // picture), 2: every row of the frame, borders included (no k_fill_border for this format); the picture's first pixel sits at (frame_x, frame_y)
  // of a frame_w x frame_h frame; border10: the border's Y, U, V as pack_v210 stores them ((v * 257) >> 6)
  int frame_on, frame_x, frame_y, frame_w, frame_h;
  uint32_t border10[3];
};

enum ScaleKind : int { SCALE_NONE = 0, SCALE_NEAREST = 1, SCALE_2TAP = 2, SCALE_NTAP = 3 };

struct ScalePass {
  int kind;             // ScaleKind
  bool horizontal;
  int in_size, out_size;
  int n_taps;
  int inc;              // scale->inc for the ldreslinl horizontal 2-tap (video-scaler.c:254-257)
  std::vector<uint32_t> offset;   // [out_size] first source index
  std::vector<int16_t> taps;      // [out_size][n_taps] quantised taps (precision depends on kind)
  int precision;
  // horizontal N-tap passes whose taps fit int8 and sum to 1.0 in every phase: byte-dot-product form (video_scale_fast.h)
  bool dot4_ok;
  int nw, nw4;                    // tap words per output that are used / allocated (row stride of tapw)
  std::vector<uint32_t> tapw;     // [out_size][nw4]: 4 int8 taps per word, shifted by (offset & 3), zero padded
  int max_span;        // horizontal passes: largest source span under any 256-output tile (LDS staging)
  int merged = 0;      // ScaleDev::merged
};

// wave-tile geometry of a horizontal pass (video_scale_fast.h): outputs per wave and LDS words per staged row
struct TileGeom {
  int tile_w;           // multiple of 4, <= 256
  int lds_px;           // multiple of 8: covers [x_lo & ~7, x_hi) of every tile
  int tile16_w;         // video_hscale420.h: tile width whose spans [x_lo & ~15, x_hi) all fit 1024 pixels (64 lanes x 16), 0: none
};
TileGeom pass_tile_geom (const ScalePass &pass);

// tables of the fused 4:2:0 scaler (video_scale420_fused.h): the vertical N-tap pass as byte dot products over GROUPS of four
// source lines (lines 4g-1 .. 4g+2), built from the pass's offsets and 6-bit taps
struct Fused420Tables {
  int ngv;                        // tap words per output row
  int n_groups;                   // line groups of a picture of `height` lines
  std::vector<int32_t> vgroup;    // [out_h] first group of the row's window
  std::vector<uint32_t> vtapw;    // [out_h][ngv]
};
// false: the pass cannot take the byte form (a tap outside int8, a phase whose taps do not sum to 64, offsets not ascending)
bool make_fused420_tables (const ScalePass &vpass, int height, Fused420Tables *t);
// most groups any round of any chunk needs in the ring at once (rounds of `nwaves` rows inside chunks of rows_per_chunk rows)
int fused420_ring_groups (const Fused420Tables &t, int rows_per_chunk, int nwaves, int first_rows = 0);
// rows for a chunk's first round: the most (<= nwaves) whose windows span no more than nwaves groups
int fused420_ring_groups2 (const Fused420Tables &t, int rows_per_chunk, int nwaves, int first_rows);
int fused420_first_rows (const Fused420Tables &t, int nwaves);

// tables of the column-walk scaler (video_scale_col.h): both N-tap passes (horizontal first) of a 2x horizontally subsampled planar /
// semi-planar source in one kernel, one WAVE per column tile walking down its rows.
//   tiles[t] = {o0, n, s0, p0}: outputs [o0, o0 + n) of every row; the staged source span is the pixels [p0, p0 + 256 opl) and source
//              pixel q sits at byte q - s0 of a staged byte plane (p0 = s0 except for the first tile when its s0 < 0: p0 = 0 then; loads past
//              the picture's right edge read row padding or, out of the plane, zeros - such bytes only ever meet zero taps)
//   hout[x]  = {wbase, init, tw[0 .. 5]}: byte offset (word aligned) of output x's first window word inside a staged plane, the accumulator's
//              start value 128 * sum (taps) + 32, the int8 x 4 tap words (zero padded); with `wstep` >= 0 (opl = 2) the two outputs of a lane
//              share ONE window of nw + wstep words from the even output's wbase, the odd output's words start wstep words in
//   vrow[j]  = {gfirst, glast, init, tw[0 .. 4]}: the line groups (lines 4g-1 .. 4g+2, byte = line) of row j's window with their tap words;
//              tw[k] belongs to group glast - ngv_aligned + 1 + k (col_align_rows: the kernel keeps the LAST ngv groups, oldest first)
struct ColTables {
  int opl;                        // outputs per lane: 1 (4 staged pixels per lane) or 2 (8)
  int nw, ngv;                    // tap words per output (horizontal, <= 6) and per row (vertical, <= 4)
  int wstep;                      // -1: every output reads its own nw words
  int a8;                         // the shared window starts on an 8-byte boundary everywhere (two ds_read_b64 + ...)
  int n_groups;
  int pubn;                       // most groups the last row of a wave shares with the first row of the wave below
  int ngv_aligned;                // window groups per row the tap words of vrow are laid out for (>= ngv: the kernel form's)
  int min_rows_per_wave;          // fewest rows a wave may own for the hand-over of those groups to work (a wave publishes only groups it makes)
  std::vector<int32_t> tiles;     // [n_tiles][4]
  std::vector<uint32_t> hout;     // [out_w][8]
  std::vector<uint32_t> vrow;     // [out_h][8]
};
// false: no byte form (a tap outside int8, a phase whose taps sum to less than 64 - the alpha byte would not stay 0xff -, windows too long)
// share: let the two outputs of a lane read one window where that saves LDS traffic (opl = 2)
bool make_col_tables (const ScalePass &hpass, const ScalePass &vpass, int width, int height, int opl, bool share, ColTables *t);
// lay the rows' tap words out for a kernel that keeps ngv_form >= t->ngv groups per row (leading zero words for shorter windows)
void col_align_rows (ColTables *t, int ngv_form);

// does the column-walk scaler apply to this plan?  Two N-tap passes, horizontal first, ahead of the convert stage, from an 8-bit planar /
// semi-planar 4:2:0 source whose chroma line pairing is the closed form (lines 2u-1, 2u blend rows u-1 and u, clamped into
// [*crow_lo, *crow_hi] = the frame's rows around a crop): what do_upsample_lines produces when every line is consumed in order
struct VideoPlan;
bool col_plan_regular (const VideoPlan &plan, int *crow_lo, int *crow_hi);

// one destination plane of convert_scale_planes on a planar / semi-planar format
enum PlaneKind : int { PLANE_COPY = 0, PLANE_H_HALVE, PLANE_H_DOUBLE, PLANE_V_HALVE, PLANE_V_DOUBLE, PLANE_HV_HALVE, PLANE_HV_DOUBLE, PLANE_SCALE,
  PLANE_FILL /* a destination plane whose component the source does not have: convert_plane_fill (video-converter.c:7310), 0x80 for chroma */ };
struct PlanePlan {
  int kind;                       // PlaneKind
  int src_plane, dst_plane;
  int n_elems;                    // bytes per pixel of the plane (1; 2 for the UV plane of NV12 / NV21)
  int iw, ih, ow, oh;             // plane sizes in pixels
  std::vector<ScalePass> passes;  // PLANE_SCALE: 1 or 2 passes in execution order
};

// source crop / destination rectangle / borders (gst_video_converter_new :2307-2366, convert_fill_border :7190)
struct RectPlan {
  int in_x, in_y;               // crop origin in the source frame
  int out_x, out_y;             // origin of the converted picture in the destination frame
  int out_maxw, out_maxh;       // destination frame size
  int in_maxw, in_maxh;         // source frame size (lines above / below the crop exist and are read by the chroma upsampler)
  bool fill;                    // borders exist and are to be filled
  uint8_t border[4];            // border pixel in unpack order: A, R, G, B or A, Y, U, V
};

// gamma-mode = remap (video-converter.c chain_convert_to_RGB :1566-1610, chain_convert :1719-1868, chain_alpha, chain_convert_to_YUV
// :1955-2015): the chain runs on linear-light ARGB64 lines between the chroma upsampler and the chroma downsampler.  The plan is a
// composite: sub-conversion `in -> mid_in` (unpack + upsample into the 8-bit unpack format AYUV / ARGB, the source crop), the stages
// of video_gamma.h and the 16-bit scalers on images in HBM, sub-conversion `mid_out -> out` (downsample, dither, pack, destination
// rectangle and borders).
// The same composite serves 10-bit DESTINATIONS (I420_10LE / P010_10LE, pack16): there the first stage only widens (table i * 257 =
// video_orc_convert_u8_to_u16) or is the 16-bit front of a 10-bit source (src16), the middle stage is the convert matrix on 16-bit
// values, and the last stage is chroma downsample + dither + pack on 16-bit lines (video_deep.h pack16_body) instead of the encode table.
struct GammaPlan {
  bool on = false;
  bool src16 = false;           // the source is 10-bit: k_front16 of this plan's front / vpair makes the first ARGB64 / AYUV64 image
  bool src64 = false;           // the source is ARGB64 / AYUV64: the frame itself is the first 16-bit image
  bool store64 = false;         // the destination is ARGB64 / AYUV64: the last 16-bit image is the frame
  bool pack16 = false;          // the destination is 10-bit: pack16_body finishes
  PackPlanarParams pack;        // pack16: geometry and chroma downsampler of the destination
  int pack_hi_depth;            // FormatDesc::hi_depth of the destination
  DitherParams dither16;        // pack16: shift[] in unpack order (A, Y, U, V) on 16-bit values
  bool fused = false;           // gamma remap, unscaled, 4-byte destination: ONE kernel (k_convert_gamma) - cfg_in describes the direct conversion
                                // in -> out without a matrix, whose per-pixel step is the whole gamma chain with the tables in LDS
  bool planes_fast = false;     // same size, same chroma grid, no resampler, no matrix: every destination sample comes from ONE source
                                // sample (widen / narrow, dither, pack) - k_deep_planes goes from the source planes to the destination planes
  DeepPlanesParams planes;
  GstAmdVideoInfo sub_in_info, mid_in, mid_out, sub_out_info;
  GstAmdVideoConverterConfig cfg_in, cfg_out;
  MatrixParams to_rgb, to_yuv;  // 8-bit matrices around the tables (kind NONE: RGB on that side)
  Deep16Params prim;            // primaries matrix on 16-bit values (video_converter_matrix16)
  int alpha_kind;               // ALPHA_* on 16-bit alpha (convert_set_alpha_u16 / convert_mult_alpha_u16)
  unsigned alpha_value;
  bool shrink;                  // the scalers run before the primaries / alpha stages (chain_scale's first call)
  std::vector<uint16_t> dec;    // [256] gamma_convert_u8_u16's table
  std::vector<uint8_t> enc;     // [65536] gamma_convert_u16_u8's table
  bool lut_direct = false;      // the fused form collapses further: to_rgb IS the direct conversion's matrix and `comp` is all the gamma chain does, so the
                                // frame is the DIRECT conversion (cfg_in then carries the matrix: every fast kernel applies) followed by `comp` on the
                                // colour bytes of the destination rectangle (k_lut3)
  std::vector<uint8_t> comp;    // [256] enc[dec[v]]: the fused kernel's whole gamma part when nothing sits between the two tables (no primaries
                                // matrix, no alpha operation) - both lookups are per component and exact, so is their composition
  /* gamma-mode = remap with a 16-bit unpack / pack format (setup_gamma_decode / _encode's 16 -> 16 tables, video-converter.c:1497-1564,
   * and the to-RGB / to-YUV matrices of chain_convert_to_RGB :1567 / _to_YUV :1956 prepared for 16 bits = video_converter_matrix16) */
  bool in16 = false, out16 = false;
  Deep16Params to_rgb16, to_yuv16;
  std::vector<uint16_t> dec16;  // [65536] gamma_convert_u16_u16 on the decode side
  std::vector<uint16_t> enc16;  // [65536] ... on the encode side
};

struct VideoPlan {
  GstAmdVideoInfo in_info, out_info;
  GstAmdVideoInfo orig_in, orig_out;   // the frames as given (in_info / out_info are the crop / the destination rectangle)
  GammaPlan gamma;
  GstAmdVideoConverterConfig config;
  const FormatDesc *fin, *fout;
  FrontParams front;
  bool matrix_before_scale;   // false: scale first (downscale), matrix in the post stage
  MatrixParams matrix;        // the convert_matrix stage
  PostParams post;
  std::vector<ScalePass> passes;  // 0, 1 or 2 passes in execution order
  // chroma vertical pairing, one entry per source line: chroma rows of the pair's first and second
  // line and which of the two this line is (0 first / 1 second); rows equal => plain copy
  std::vector<int32_t> vpair;     // [in_height][2]: (row_a | role << 30), row_b
  RectPlan rect;
  bool ref_same_size;         // video_converter_lookup_fastpath's same_size: FULL input size == destination rectangle (:8942)
  // An interleaved frame (GstAmdVideoInfo::interlace_mode) is converted as two FIELD conversions (plan_field_infos): `interlaced` marks the frame's
  // plan, whose own kernels never run - it carries the description, the divergence notes and the algorithmic bytes of the two; `field` (1 top, 2 bottom)
  // marks a field's plan: a progressive plan over the field's lines (pitches doubled) whose fastpath lookup, chroma line pairing, vertical taps and pass
  // order are the FRAME's (video-converter.c:3303-3312, 1651-1660, 7977; video-scaler.c:229-249; video-chroma.c:347).
  bool interlaced = false;
  int field = 0;
  bool field_src_chroma_frame = false;   // a field plan whose SOURCE chroma planes are addressed as the frame's (rows from the pair table), not as the field's
  bool plane_mode;            // convert_scale_planes on a planar / semi-planar format: `planes` is the whole plan
  std::vector<PlanePlan> planes;
  bool out_planar;            // destination is planar / semi-planar: the chain renders AYUV, pack_planar finishes
  PackPlanarParams pack;
  bool fast_pair;             // BASELINE C2 shape: line-pair kernel of video_fast.h is applicable
  bool v210_fast = false;     // the reference's own v210 <-> 8-bit 4:2:0 / 4:2:2 fastpaths (video_v210_fast.h); nothing else of the plan is used
  bool relayout;              // the chain changes nothing but where the samples sit (I420 <-> NV12 <-> NV21 <-> YV12, Y42B <-> NV16, Y444 <-> NV24): video_relayout.h
  bool fast_enc420;           // unscaled 4-byte RGB -> 4:2:0 YUV through the table matrix: the block kernel of video_encode_fast.h applies
  bool fast_420p;             // unscaled planar 4:2:0 -> 4-byte RGB with nearest chroma (the reference's convert_I420_BGRA family): video_422_fast.h
  bool fast_422_ayuv;         // unscaled packed 4:2:2 with neither matrix nor alpha stage: the same kernel leaving A, Y, U, V bytes (AYUV destinations, the image ahead of a planar pack)
  bool fast_422;              // unscaled packed 4:2:2 -> 4-byte RGB through the no-wrap AYUV_ARGB matrix: video_422_fast.h applies
  bool fast_post;             // scaled plans: the post stage may run fast_pixel (matrix provably wrap-free, alpha stays 0xff)
  bool fast_pre = false;      // enlarging plans (the colour stage runs on the SOURCE's pixels, chain_convert ahead of chain_scale): the source-size A, R, G, B image
                              // ahead of the scaler is the unscaled NV12 / NV21 -> ARGB conversion - the line-pair kernel of video_fast.h makes it
  bool deep_out;              // 10-bit destination: the composite of GammaPlan with pack16
  bool deep16;                // 10-bit source, unscaled, 8-bit 4-byte destination: k_convert16 (video_deep.h)
  Deep16Params deep;
  DitherParams dither;
  int im_raw[3][4];           // the rint()ed 8-bit matrix before the per-kind adjustments
  std::string ref_fastpath;   // name of the reference fastpath this plan reproduces (empty: generic chain)
  std::string description;
  std::string divergence;     // where the reference's own output is undefined (uninitialised lines, line aliasing) and this plan computes the chain's intended result instead: why
  uint64_t algorithmic_bytes;
};

// byte offset of pixel (x, y) inside plane `plane` of a frame of format f (x, y multiples of the subsampling)
inline size_t plane_origin (const FormatDesc *f, int plane, int x, int y, int stride)
{
  if (f->kind == UNPACK_PACKED4)
    return (size_t) y * stride + (size_t) x * (f->hi_depth == 3 ? 8 : 4);
  if (f->kind == UNPACK_PACKED3)
    return (size_t) y * stride + (size_t) x * 3;
  if (f->kind == UNPACK_PACKED422)
    return (size_t) y * stride + (size_t) x * 2;
  if (f->kind == UNPACK_PACKED411)        /* whole groups (x is 0: the planner refuses rectangles on these frames) */
    return (size_t) y * stride + (size_t) (x >> 2) * 6;
  if (f->kind == UNPACK_P422_16 || f->kind == UNPACK_Y410)          /* 8 bytes per pair of pixels / 4 bytes per pixel */
    return (size_t) y * stride + (size_t) x * 4;
  if (f->kind == UNPACK_PACKED64)
    return (size_t) y * stride + (size_t) x * 8;
  if (f->kind == UNPACK_GRAY16 || f->kind == UNPACK_RGB16)
    return (size_t) y * stride + (size_t) x * 2;
  if (f->kind == UNPACK_V210)            /* rows only: a horizontal offset inside the 6-pixel groups is refused by the planner */
    return (size_t) y * stride;
  if (f->kind == UNPACK_SEMI_TILED)          /* whole frames only: the frame's first byte */
    return 0;
  if (f->kind == UNPACK_P422_UYVP)
    return (size_t) y * stride + (size_t) (x >> 1) * 5;
  if (GSTAMD_KIND_LE32 (f->kind))          /* whole frames only (the planner refuses rectangles) */
    return (size_t) (plane == 0 ? y : y >> f->h_sub) * stride;
  const size_t bps = f->hi_depth ? 2 : 1;          /* planes of 10 / 12 / 16-bit formats hold 16-bit samples */
  if (plane == 0 || plane == GSTAMD_KIND_ALPHA_PLANE (f->kind))
    return (size_t) y * stride + (size_t) x * bps;
  const size_t row = (size_t) (y >> f->h_sub) * stride;
  return GSTAMD_KIND_SEMI (f->kind) ? row + (size_t) (x >> f->w_sub) * 2 * bps : row + (size_t) (x >> f->w_sub) * bps;
}

// Width, in plane positions, of the part of a destination plane's rows the border fill leaves to the picture.  Planes of whole pixels and
// subsampled chroma planes: every position the picture touches.  Packed 4:2:2 (positions = macropixels; out_x is even, gst_video_converter_new
// clears the subsampling bits :2330): a picture of odd width that ends left of the frame's right edge shares its last macropixel with the border -
// the generic chain packs the frame line pair by pair (pack_YUY2 & co over out_maxwidth pixels, video-converter.c:3274), so that macropixel is
// {picture luma, picture chroma of the last pixel, BORDER luma}: the fill lays the border pair there first and the packer's odd tail (luma,
// U, V of pixel width - 1; it never writes the second luma) goes over it on the same stream.
inline int border_picture_positions (const FormatDesc *f, const RectPlan &rc, int width, int ws)
{
  const bool pairs = f->kind == UNPACK_PACKED422 || f->kind == UNPACK_P422_16;
  if (pairs && (width & 1) && rc.out_x + width < rc.out_maxw)
    return width >> 1;
  return -((-width) >> ws);
}

// The border sample(s) of plane `plane` of the destination (setup_borderline :2189-2262 packs one border pixel with the format's own pack
// function): *es = bytes a plane position takes, lo / hi = its value (hi: the upper half of an 8-byte pixel).  16-bit formats pack the
// border widened by video_orc_splat2_u64 (v * 257).  swap: the position is the odd-width tail of an NV61 line (see fill_borders).
inline void border_plane_value (const FormatDesc *f, const uint8_t border[4], int plane, int *es, uint32_t *lo, uint32_t *hi)
{
  *hi = 0;
  auto s16 = [&](int c) -> uint32_t {
    const uint32_t v = (uint32_t) border[c] * 257u;
    if (f->hi_depth == 3 || f->hi_depth == 6 || f->hi_depth == 9 || f->hi_depth == 10 || f->hi_depth == 11 || f->hi_depth == 12 || f->hi_depth == 16 || f->hi_depth == 36)
      return (uint32_t) px16_store (f->hi_depth, (int) v);
    const int le = hi_depth_le (f->hi_depth), drop = 16 - hi_depth_bits (f->hi_depth);
    const uint32_t w = le == 6 ? v : (le == 1 || le == 4 ? v >> drop : v & ~((1u << drop) - 1u));
    return hi_depth_be (f->hi_depth) ? (uint32_t) bswap16i ((int) w) : w;
  };
  if (f->kind == UNPACK_PACKED422) {
    /* the unit is the macropixel (the planner admits borders on these formats only where no macropixel holds border and picture: even offsets, widths
       and frame width - every path of the reference, the generic chain's packed border line and convert_fill_border's group 42 alike, then lays the
       border pair from an even pixel on): two border pixels packed, 4 bytes */
    uint8_t m[4];
    m[f->pos[1]] = m[f->pos[1] + 2] = border[1], m[f->pos[2]] = border[2], m[f->pos[3]] = border[3];
    *es = 4, *lo = (uint32_t) m[0] | ((uint32_t) m[1] << 8) | ((uint32_t) m[2] << 16) | ((uint32_t) m[3] << 24);
  } else if (f->kind == UNPACK_P422_16) {
    uint32_t w[4] = {0, 0, 0, 0};               /* pack_Y210: words Y0 U Y1 V; pack_v216: U Y0 V Y1 */
    w[f->pos[1]] = w[f->pos[1] + 2] = s16 (1), w[f->pos[2]] = s16 (2), w[f->pos[3]] = s16 (3);
    *es = 8, *lo = w[0] | (w[1] << 16), *hi = w[2] | (w[3] << 16);
  } else if (f->kind == UNPACK_Y410) {            /* pack_Y410 (video-format.c:898-921) of the widened border */
    const uint32_t a = (uint32_t) border[0] * 257u, y = (uint32_t) border[1] * 257u, u = (uint32_t) border[2] * 257u, v = (uint32_t) border[3] * 257u;
    *es = 4, *lo = ((y >> 6) << f->pos[1]) | ((u >> 6) << f->pos[2]) | ((v >> 6) << f->pos[3]) | (f->hi_depth == 27 ? 0u : (a & 0xc000u) << 16);
    if (f->hi_depth == 27)        /* r210: the word big endian */
      *lo = (*lo >> 24) | ((*lo >> 8) & 0xff00u) | ((*lo << 8) & 0xff0000u) | (*lo << 24);
  } else if (f->kind == UNPACK_GRAY16) {
    *es = 2, *lo = s16 (1);
  } else if (f->kind == UNPACK_RGB16) {
    *es = 2, *lo = (uint32_t) rgb16_pack (f->pos, border[1], border[2], border[3]);
  } else if ((f->kind == UNPACK_PACKED4 && f->hi_depth == 3) || f->kind == UNPACK_PACKED64) {
    uint32_t w[4] = {0, 0, 0, 0};
    for (int c = 0; c < 4; c++)
      w[f->pos[c]] = s16 (c);
    *es = 8, *lo = w[0] | (w[1] << 16), *hi = w[2] | (w[3] << 16);
  } else if (f->kind == UNPACK_PACKED4 || f->kind == UNPACK_PACKED3) {
    uint32_t v = 0;
    for (int c = f->kind == UNPACK_PACKED4 ? 0 : 1; c < 4; c++)
      v |= (uint32_t) border[c] << (8 * f->pos[c]);
    *es = f->kind == UNPACK_PACKED4 ? 4 : 3, *lo = v;
  } else if (f->hi_depth) {
    if (plane == GSTAMD_KIND_ALPHA_PLANE (f->kind))
      *es = 2, *lo = s16 (0);
    else if (plane == 0)
      *es = 2, *lo = s16 (1);
    else if (f->kind == UNPACK_SEMI)
      *es = 4, *lo = f->u_plane ? s16 (2) | (s16 (3) << 16) : s16 (3) | (s16 (2) << 16);
    else
      *es = 2, *lo = s16 (plane == f->u_plane ? 2 : 3);
  } else {
    if (plane == GSTAMD_KIND_ALPHA_PLANE (f->kind))
      *es = 1, *lo = border[0];
    else if (plane == 0)
      *es = 1, *lo = border[1];
    else if (GSTAMD_KIND_SEMI (f->kind))
      *es = 2, *lo = f->u_plane ? (uint32_t) border[2] | ((uint32_t) border[3] << 8) : (uint32_t) border[3] | ((uint32_t) border[2] << 8);
    else
      *es = 1, *lo = border[plane == f->u_plane ? 2 : 3];
  }
}

// vpair table entry 0: chroma row of the pair's first line (signed 30 bits: with a source crop the row above the crop
// origin is -1) | role << 30; entry 1: row of the second line
GSTAMD_VP bool kind_has_planes (int kind) { return kind == UNPACK_PLANAR || kind == UNPACK_SEMI; }
GSTAMD_VP bool kind_is_tiled (int kind) { return kind == UNPACK_SEMI_TILED || kind == UNPACK_SEMI_LE40_TILED; }          /* NV12 in tiles: 4:2:0 like NV12, generic per-pixel kernels */
GSTAMD_VP int vpair_row (int e0) { return (int) ((uint32_t) e0 << 2) >> 2; }
GSTAMD_VP int vpair_role (int e0) { return (e0 >> 30) & 1; }
GSTAMD_VP int vpair_pack (int row, int role) { return (int) (((uint32_t) row & 0x3fffffffu) | ((uint32_t) role << 30)); }
// One line's vertical chroma blend: c = (wa * C[ra] + (8 - wa) * C[rb] + 4) >> 3 (ra == rb: a copy).  Progressive tables (mode 1): video_chroma_up_v2's
// (3 a + b + 2) >> 2 and (a + 3 b + 2) >> 2 = weights 6 / 2 over 8.  Field tables (mode 2): the second entry carries the weight in bits 28 .. 30 over a
// 28-bit signed row - video_chroma_up_vi2's FILT_5_3, _7_1, _1_7, _3_5 (video-chroma.c:347-384).
struct VPairW { int ra, rb, wa; };
GSTAMD_VP int vpair_pack_w8 (int row, int wa) { return (int) (((uint32_t) row & 0x0fffffffu) | ((uint32_t) wa << 28)); }
GSTAMD_VP VPairW vpair_get (const int *vpair, int y, int mode)
{
  VPairW r;
  const int e0 = vpair[2 * y], e1 = vpair[2 * y + 1];
  r.ra = vpair_row (e0);
  if (mode == 2) {
    r.rb = (int) ((uint32_t) e1 << 4) >> 4;
    r.wa = (e1 >> 28) & 7;
  } else {
    r.rb = e1;
    r.wa = vpair_role (e0) ? 2 : 6;
  }
  return r;
}

// Returns GSTAMD_OK and fills `plan`, or an error code (GSTAMD_ERR_UNSUPPORTED for conversions the
// reference would run through a path this library has no kernel for yet).
// a scaled plan without a colour stage between the scaler and the pack: the chain's lines stay A Y U V from the unpacker to the packer (no matrix, no alpha
// operation, unpack order = pack order) - the bilinear 4:2:0 kernels serve it with the layout that stores A Y U V (video_fast.h GSTAMD_LAYOUT_AYUV)
inline bool bilinear420_ayuv_plan (const VideoPlan &p)
{
  /* (or with the 8-bit convert stage of two YUV colorimetries behind the scaler: FastParams::m8) */
  if (p.passes.empty () || p.matrix_before_scale || (p.matrix.kind != MATRIX_NONE && p.matrix.kind != MATRIX_8 && p.matrix.kind != MATRIX_TABLE) || p.post.alpha_kind != ALPHA_NONE || p.front.hi_depth != 0 || p.gamma.on || p.deep16 ||
      p.plane_mode || p.interlaced || p.field)
    return false;
  for (int i = 0; i < 4; i++)
    if (p.post.pack_pos[i] != i)
      return false;
  return true;
}

int plan_video_converter (const GstAmdVideoInfo *in, const GstAmdVideoInfo *out,
    const GstAmdVideoConverterConfig *config, VideoPlan *plan, std::string *error);

// A plan that is nothing but the scaler passes on the raw 4-byte pixels of one format (same 8-bit 4-byte packed format on both sides,
// whole frames, no colour step, no dither): what a compositor pad's converter is when only the pad's size differs from its frames'
// (GstVideoAggregatorConvertPad, gstvideoaggregator.c:479-513).  Such a plan can be sampled per destination pixel inside the blend
// kernel (compositor_scaled.h).  *h / *v: index into plan.passes of the horizontal / vertical pass, -1: none.
void plan_set_matrix_override (const MatrixParams *m);
void plan_set_border_override (const uint8_t *border);
bool plan_is_pad_scaler (const VideoPlan &plan, int *h, int *v);
// rows per 64-column canvas tile (12..16) of k_aggregate_scaled for pads scaled by this plan: the first pass under a tile is
// rows x quads items (vertical first) or rows x 64 (horizontal first) spread over 256 lanes; the height that leaves the fewest idle
// lanes in the last round wins (2:1 with 8 taps: 34 quads x 15 rows = 510 items = two full rounds, 16 rows would need a third)
int scaled_tile_rows_for (const VideoPlan &plan);

int video_info_set_format (GstAmdVideoInfo *info, int format, int width, int height);
// The two infos of field `f` (0 top, 1 bottom) of an interleaved frame pair: lines f, f + 2, ... of every plane (offset + f * pitch, pitch * 2, the field's
// height), marked GSTAMD_INTERLACE_FIELD_* with the frame's height beside it.  Plane pointers for a field conversion follow the same rule, except the
// source chroma planes of a plan with field_src_chroma_frame, which stay the frame's.
void plan_field_infos (const GstAmdVideoInfo *in, const GstAmdVideoInfo *out, int f, GstAmdVideoInfo *fin, GstAmdVideoInfo *fout);
// The options of the field conversions of an interleaved frame: the frame's, with rectangle options that name the whole frames (what the elements
// always pass: dest-x 0, dest-width = the frame's ...) dropped.  false: an actual source crop or destination rectangle (not built for interlaced frames).
bool plan_field_config (const GstAmdVideoInfo *in, const GstAmdVideoInfo *out, const GstAmdVideoConverterConfig *cfg, GstAmdVideoConverterConfig *fcfg);
inline bool info_is_field (const GstAmdVideoInfo &i) { return i.interlace_mode == GSTAMD_INTERLACE_FIELD_TOP || i.interlace_mode == GSTAMD_INTERLACE_FIELD_BOTTOM; }
inline bool info_is_interleaved (const GstAmdVideoInfo &i) { return i.interlace_mode == GSTAMD_INTERLACE_MODE_INTERLEAVED || i.interlace_mode == GSTAMD_INTERLACE_MODE_MIXED; }
void converter_config_init (GstAmdVideoConverterConfig *config);

// building blocks exposed for tests
void compute_convert_matrix (const VideoPlan &plan_inputs, int in_range, int in_matrix, int out_range,
    int out_matrix, bool in_yuv, bool out_yuv, int matrix_mode, double dm[4][4]);
bool make_scale_pass (int method, unsigned n_taps_opt, const GstAmdVideoConverterConfig &cfg, int in_size,
    int out_size, bool horizontal, ScalePass *pass, bool h2_as_ntap = false, bool deep16 = false);

}  // namespace gstamd
