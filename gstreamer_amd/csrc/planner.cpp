// planner.cpp - see planner.h.  Host-only set-up math; compiled with -ffp-contract=off so that the
// double computations round exactly like the reference's C code.
#include "planner.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace gstamd {

// capi_video.cpp sets this around the planning of a fused gamma plan's direct conversion (config.internal_flags & 2)
static thread_local const MatrixParams *g_matrix_override = nullptr;
void plan_set_matrix_override (const MatrixParams *m) { g_matrix_override = m; }
// the border pixel of a composite plan's sub-conversion is the composite's (computed from the caller's matrix-mode and source colorimetry, which
// the sub-conversion's own config no longer has): set around the planning of the sub-converters (config.internal_flags & 1)
static thread_local const uint8_t *g_border_override = nullptr;
void plan_set_border_override (const uint8_t *border) { g_border_override = border; }

// ------------------------------------------------------------------------------------------------
// format table (facts: video-format.c:8190-8235; byte orders: video-orc.orc:334-411)
// ------------------------------------------------------------------------------------------------
static const FormatDesc g_formats[] = {
  // format, name, yuv, alpha, planes, kind, w_sub, h_sub, u_plane, v_plane, pos{A,c1,c2,c3}
  {GSTAMD_VIDEO_FORMAT_I420, "I420", true, false, 3, UNPACK_PLANAR, 1, 1, 1, 2, {0, 0, 0, 0}},
  {GSTAMD_VIDEO_FORMAT_YV12, "YV12", true, false, 3, UNPACK_PLANAR, 1, 1, 2, 1, {0, 0, 0, 0}},
  {GSTAMD_VIDEO_FORMAT_A420, "A420", true, true, 4, UNPACK_PLANAR_A, 1, 1, 1, 2, {0, 0, 0, 0}},          /* unpack_A420 / pack_A420 video-format.c:2118-2185 */
  {GSTAMD_VIDEO_FORMAT_Y42B, "Y42B", true, false, 3, UNPACK_PLANAR, 1, 0, 1, 2, {0, 0, 0, 0}},
  {GSTAMD_VIDEO_FORMAT_Y444, "Y444", true, false, 3, UNPACK_PLANAR, 0, 0, 1, 2, {0, 0, 0, 0}},
  // planar RGB (video-format.c:1119-1147: the Y444 unpack / pack on the R, G, B lines); planes R, G, B inside a plan (format_plan_planes)
  {GSTAMD_VIDEO_FORMAT_GBR, "GBR", false, false, 3, UNPACK_PLANAR, 0, 0, 1, 2, {0, 0, 0, 0}},
  /* the rest of the planar RGB family (format_plane_perm says which frame plane is which component): unpack_RGBP / _BGRP video-format.c:7480-7540,
     unpack_GBRA :1151 (the fourth plane is alpha), unpack_GBR_10LE / _12LE / _16LE :3117, 3370, 3486 (Y444_10LE's arithmetic) */
  {GSTAMD_VIDEO_FORMAT_RGBP, "RGBP", false, false, 3, UNPACK_PLANAR, 0, 0, 1, 2, {0, 0, 0, 0}},
  {GSTAMD_VIDEO_FORMAT_BGRP, "BGRP", false, false, 3, UNPACK_PLANAR, 0, 0, 1, 2, {0, 0, 0, 0}},
  {GSTAMD_VIDEO_FORMAT_GBRA, "GBRA", false, true, 4, UNPACK_PLANAR_A, 0, 0, 1, 2, {0, 0, 0, 0}},
  {GSTAMD_VIDEO_FORMAT_GBR_10LE, "GBR_10LE", false, false, 3, UNPACK_PLANAR, 0, 0, 1, 2, {0, 0, 0, 0}, 1},
  {GSTAMD_VIDEO_FORMAT_GBR_12LE, "GBR_12LE", false, false, 3, UNPACK_PLANAR, 0, 0, 1, 2, {0, 0, 0, 0}, 4},
  {GSTAMD_VIDEO_FORMAT_GBR_16LE, "GBR_16LE", false, false, 3, UNPACK_PLANAR, 0, 0, 1, 2, {0, 0, 0, 0}, 6},
  /* the 10 / 12 / 16-bit forms with an alpha plane (unpack_A420_16 / _A422_16 / _A444_16 :4645-5040: samples in the low bits, every plane widened the same
     way; unpack_GBRA_10LE :3240, _12LE :3590) */
  {GSTAMD_VIDEO_FORMAT_A420_10LE, "A420_10LE", true, true, 4, UNPACK_PLANAR_A, 1, 1, 1, 2, {0, 0, 0, 0}, 1},
  {GSTAMD_VIDEO_FORMAT_A422_10LE, "A422_10LE", true, true, 4, UNPACK_PLANAR_A, 1, 0, 1, 2, {0, 0, 0, 0}, 1},
  {GSTAMD_VIDEO_FORMAT_A444_10LE, "A444_10LE", true, true, 4, UNPACK_PLANAR_A, 0, 0, 1, 2, {0, 0, 0, 0}, 1},
  {GSTAMD_VIDEO_FORMAT_A420_12LE, "A420_12LE", true, true, 4, UNPACK_PLANAR_A, 1, 1, 1, 2, {0, 0, 0, 0}, 4},
  {GSTAMD_VIDEO_FORMAT_A422_12LE, "A422_12LE", true, true, 4, UNPACK_PLANAR_A, 1, 0, 1, 2, {0, 0, 0, 0}, 4},
  {GSTAMD_VIDEO_FORMAT_A444_12LE, "A444_12LE", true, true, 4, UNPACK_PLANAR_A, 0, 0, 1, 2, {0, 0, 0, 0}, 4},
  {GSTAMD_VIDEO_FORMAT_A420_16LE, "A420_16LE", true, true, 4, UNPACK_PLANAR_A, 1, 1, 1, 2, {0, 0, 0, 0}, 6},
  {GSTAMD_VIDEO_FORMAT_A422_16LE, "A422_16LE", true, true, 4, UNPACK_PLANAR_A, 1, 0, 1, 2, {0, 0, 0, 0}, 6},
  {GSTAMD_VIDEO_FORMAT_A444_16LE, "A444_16LE", true, true, 4, UNPACK_PLANAR_A, 0, 0, 1, 2, {0, 0, 0, 0}, 6},
  {GSTAMD_VIDEO_FORMAT_GBRA_10LE, "GBRA_10LE", false, true, 4, UNPACK_PLANAR_A, 0, 0, 1, 2, {0, 0, 0, 0}, 1},
  {GSTAMD_VIDEO_FORMAT_GBRA_12LE, "GBRA_12LE", false, true, 4, UNPACK_PLANAR_A, 0, 0, 1, 2, {0, 0, 0, 0}, 4},
  /* A422 / A444 (unpack_A422 :4859, unpack_A444 :4588): Y42B / Y444 plus the alpha plane */
  {GSTAMD_VIDEO_FORMAT_A422, "A422", true, true, 4, UNPACK_PLANAR_A, 1, 0, 1, 2, {0, 0, 0, 0}},
  {GSTAMD_VIDEO_FORMAT_A444, "A444", true, true, 4, UNPACK_PLANAR_A, 0, 0, 1, 2, {0, 0, 0, 0}},
  {GSTAMD_VIDEO_FORMAT_NV12, "NV12", true, false, 2, UNPACK_SEMI, 1, 1, 1, 0, {0, 0, 0, 0}},
  {GSTAMD_VIDEO_FORMAT_NV21, "NV21", true, false, 2, UNPACK_SEMI, 1, 1, 0, 1, {0, 0, 0, 0}},
  {GSTAMD_VIDEO_FORMAT_Y41B, "Y41B", true, false, 3, UNPACK_PLANAR_H4, 2, 0, 1, 2, {0, 0, 0, 0}},          /* unpack_Y41B / pack_Y41B video-format.c:923-1006 */
  {GSTAMD_VIDEO_FORMAT_IYU1, "IYU1", true, false, 1, UNPACK_PACKED411, 2, 0, 0, 0, {0, 0, 0, 0}},          /* unpack_IYU1 / pack_IYU1 video-format.c:2368-2470 */
  /* unpack / pack_GRAY10_LE32, _NV12_10LE32, _NV16_10LE32 (video-format.c:5490-5868): three 10-bit samples per little-endian word */
  {GSTAMD_VIDEO_FORMAT_GRAY10_LE32, "GRAY10_LE32", true, false, 1, UNPACK_GRAY_LE32, 0, 0, 0, 0, {0, 0, 0, 0}, 13},
  {GSTAMD_VIDEO_FORMAT_NV12_10LE32, "NV12_10LE32", true, false, 2, UNPACK_SEMI_LE32, 1, 1, 1, 0, {0, 0, 0, 0}, 13},
  {GSTAMD_VIDEO_FORMAT_NV16_10LE32, "NV16_10LE32", true, false, 2, UNPACK_SEMI_LE32, 1, 0, 1, 0, {0, 0, 0, 0}, 13},
  /* unpack / pack_NV12_10LE40, _NV16_10LE40 (video-format.c:5868-6200): sample n at bit 10 n of the row's little-endian byte stream */
  {GSTAMD_VIDEO_FORMAT_NV12_10LE40, "NV12_10LE40", true, false, 2, UNPACK_SEMI_LE40, 1, 1, 1, 0, {0, 0, 0, 0}, 14},
  {GSTAMD_VIDEO_FORMAT_NV16_10LE40, "NV16_10LE40", true, false, 2, UNPACK_SEMI_LE40, 1, 0, 1, 0, {0, 0, 0, 0}, 14},
  {GSTAMD_VIDEO_FORMAT_UYVP, "UYVP", true, false, 1, UNPACK_P422_UYVP, 1, 0, 0, 0, {0, 0, 0, 0}, 15},          /* unpack_UYVP / pack_UYVP video-format.c:2042-2118 */
  /* the tiled NV12 family (format table video-format.c:8316-8442, TILE_64x32 / _4x4 / _32x32 / _16x32s / _8x128 :8127-8131): pos = {mode, ws, hs, sub-tiles} */
  {GSTAMD_VIDEO_FORMAT_NV12_64Z32, "NV12_64Z32", true, false, 2, UNPACK_SEMI_TILED, 1, 1, 1, 0, {1, 6, 5, 0}},
  {GSTAMD_VIDEO_FORMAT_NV12_4L4, "NV12_4L4", true, false, 2, UNPACK_SEMI_TILED, 1, 1, 1, 0, {0, 2, 2, 0}},
  {GSTAMD_VIDEO_FORMAT_NV12_32L32, "NV12_32L32", true, false, 2, UNPACK_SEMI_TILED, 1, 1, 1, 0, {0, 5, 5, 0}},
  {GSTAMD_VIDEO_FORMAT_NV12_16L32S, "NV12_16L32S", true, false, 2, UNPACK_SEMI_TILED, 1, 1, 1, 0, {0, 4, 5, 1}},
  {GSTAMD_VIDEO_FORMAT_NV12_8L128, "NV12_8L128", true, false, 2, UNPACK_SEMI_TILED, 1, 1, 1, 0, {0, 3, 7, 0}},
  {GSTAMD_VIDEO_FORMAT_NV12_10LE40_4L4, "NV12_10LE40_4L4", true, false, 2, UNPACK_SEMI_LE40_TILED, 1, 1, 1, 0, {0, 2, 2, 0}, 14},          /* :8446, unpack_TILED on unpack_NV12_10LE40 */
  {GSTAMD_VIDEO_FORMAT_AV12, "AV12", true, true, 3, UNPACK_SEMI_A, 1, 1, 1, 0, {0, 0, 0, 0}},          /* unpack_AV12 / pack_AV12 video-format.c: NV12 + an alpha plane */
  {GSTAMD_VIDEO_FORMAT_NV16, "NV16", true, false, 2, UNPACK_SEMI, 1, 0, 1, 0, {0, 0, 0, 0}},
  {GSTAMD_VIDEO_FORMAT_NV61, "NV61", true, false, 2, UNPACK_SEMI, 1, 0, 0, 1, {0, 0, 0, 0}},
  {GSTAMD_VIDEO_FORMAT_NV24, "NV24", true, false, 2, UNPACK_SEMI, 0, 0, 1, 0, {0, 0, 0, 0}},
  // packed 4:2:2 (video-format.c:153-460): pos[1..3] = byte of Y0, U, V inside the 4-byte macropixel
  {GSTAMD_VIDEO_FORMAT_YUY2, "YUY2", true, false, 1, UNPACK_PACKED422, 1, 0, 0, 0, {0, 0, 1, 3}},
  {GSTAMD_VIDEO_FORMAT_UYVY, "UYVY", true, false, 1, UNPACK_PACKED422, 1, 0, 0, 0, {0, 1, 0, 2}},
  {GSTAMD_VIDEO_FORMAT_YVYU, "YVYU", true, false, 1, UNPACK_PACKED422, 1, 0, 0, 0, {0, 0, 3, 1}},
  {GSTAMD_VIDEO_FORMAT_VYUY, "VYUY", true, false, 1, UNPACK_PACKED422, 1, 0, 0, 0, {0, 1, 2, 0}},
  // packed 10 / 12-bit formats of the 16-bit chain (video-format.c:760-861 Y210, :863-921 Y410, :6995-7092 Y212_LE): pos[1..3] = WORD of Y0, U, V.
  // Y410 stores two bits of alpha but is declared without GST_VIDEO_FORMAT_FLAG_ALPHA (MAKE_YUV_FORMAT :8380): the alpha options pass it by
  {GSTAMD_VIDEO_FORMAT_Y210, "Y210", true, false, 1, UNPACK_P422_16, 1, 0, 0, 0, {0, 0, 1, 3}, 2},
  {GSTAMD_VIDEO_FORMAT_Y212_LE, "Y212_LE", true, false, 1, UNPACK_P422_16, 1, 0, 0, 0, {0, 0, 1, 3}, 5},
  {GSTAMD_VIDEO_FORMAT_Y216_LE, "Y216_LE", true, false, 1, UNPACK_P422_16, 1, 0, 0, 0, {0, 0, 1, 3}, 6},
  {GSTAMD_VIDEO_FORMAT_v216, "v216", true, false, 1, UNPACK_P422_16, 1, 0, 0, 0, {0, 1, 0, 2}, 6},          /* unpack_v216 video-format.c: words U Y0 V Y1 */
  /* unpack_r210: one BIG-endian 32-bit word per pixel, x 2, R 10, G 10, B 10 - Y410's kind with hi_depth code 27 (word swapped, no alpha bits) */
  {GSTAMD_VIDEO_FORMAT_r210, "r210", false, false, 1, UNPACK_Y410, 0, 0, 0, 0, {0, 20, 10, 0}, 27},
  /* unpack_GRAY10_LE16: 10 bits in the low bits of little-endian words */
  {GSTAMD_VIDEO_FORMAT_GRAY10_LE16, "GRAY10_LE16", true, false, 1, UNPACK_GRAY16, 0, 0, 0, 0, {0, 0, 0, 0}, 1},          /* unpack_Y216_LE video-format.c:7181 */
  /* unpack_Y412_LE :7323 (12 bits in the high bits, widened by v | v >> 12), unpack_Y416_LE :7430: words U Y V A; like Y410 their alpha word is a
     component the format flags do not declare (MAKE_YUV_LE_FORMAT :8407, 8495) */
  {GSTAMD_VIDEO_FORMAT_Y412_LE, "Y412_LE", true, false, 1, UNPACK_PACKED64, 0, 0, 0, 0, {3, 1, 0, 2}, 11},
  {GSTAMD_VIDEO_FORMAT_Y416_LE, "Y416_LE", true, false, 1, UNPACK_PACKED64, 0, 0, 0, 0, {3, 1, 0, 2}, 9},
  {GSTAMD_VIDEO_FORMAT_Y410, "Y410", true, false, 1, UNPACK_Y410, 0, 0, 0, 0, {0, 10, 0, 20}, 7},
  /* unpack_rgb10a2_le / unpack_bgr10a2_le (video-format.c:6210-6330; format table :8384-8388): Y410's word with R, G, B fields, unpack format ARGB64 */
  {GSTAMD_VIDEO_FORMAT_RGB10A2_LE, "RGB10A2_LE", false, true, 1, UNPACK_Y410, 0, 0, 0, 0, {0, 0, 10, 20}, 7},
  {GSTAMD_VIDEO_FORMAT_BGR10A2_LE, "BGR10A2_LE", false, true, 1, UNPACK_Y410, 0, 0, 0, 0, {0, 20, 10, 0}, 7},
  /* BGR10x2_LE / RGB10x2_LE (format table :8504-8507): PACK_BGR10A2_LE / PACK_RGB10A2_LE again - the two top bits are read as alpha and written from it -
     under three declared components: no alpha flag, depth[3] = 0 (format_alpha_bits: no alpha quantiser in chain_dither) */
  {GSTAMD_VIDEO_FORMAT_BGR10x2_LE, "BGR10x2_LE", false, false, 1, UNPACK_Y410, 0, 0, 0, 0, {0, 20, 10, 0}, 7},
  {GSTAMD_VIDEO_FORMAT_RGB10x2_LE, "RGB10x2_LE", false, false, 1, UNPACK_Y410, 0, 0, 0, 0, {0, 0, 10, 20}, 7},
  {GSTAMD_VIDEO_FORMAT_v210, "v210", true, false, 1, UNPACK_V210, 1, 0, 0, 0, {0, 0, 0, 0}, 8},          /* video-format.c:558-757 */
  // luma only (video-format.c:1207-1229)
  {GSTAMD_VIDEO_FORMAT_GRAY8, "GRAY8", true, false, 1, UNPACK_GRAY, 0, 0, 0, 0, {0, 0, 0, 0}},
  // 3 bytes per pixel (video-format.c:1519-1593)
  {GSTAMD_VIDEO_FORMAT_RGB, "RGB", false, false, 1, UNPACK_PACKED3, 0, 0, 0, 0, {0, 0, 1, 2}},
  {GSTAMD_VIDEO_FORMAT_BGR, "BGR", false, false, 1, UNPACK_PACKED3, 0, 0, 0, 0, {0, 2, 1, 0}},
  // packed 4:4:4 YUV in 3 bytes (video-format.c:461-495 v308, :498-532 IYU2) and VUYA (:6186-6215)
  /* unpack_RGB16 / _BGR16 / _RGB15 / _BGR15 (video-format.c:1301-1425; format table :8249-8256) */
  {GSTAMD_VIDEO_FORMAT_RGB16, "RGB16", false, false, 1, UNPACK_RGB16, 0, 0, 0, 0, {6, 11, 5, 0}},
  {GSTAMD_VIDEO_FORMAT_BGR16, "BGR16", false, false, 1, UNPACK_RGB16, 0, 0, 0, 0, {6, 0, 5, 11}},
  {GSTAMD_VIDEO_FORMAT_RGB15, "RGB15", false, false, 1, UNPACK_RGB16, 0, 0, 0, 0, {5, 10, 5, 0}},
  {GSTAMD_VIDEO_FORMAT_BGR15, "BGR15", false, false, 1, UNPACK_RGB16, 0, 0, 0, 0, {5, 0, 5, 10}},
  {GSTAMD_VIDEO_FORMAT_v308, "v308", true, false, 1, UNPACK_PACKED3, 0, 0, 0, 0, {0, 0, 1, 2}},
  {GSTAMD_VIDEO_FORMAT_IYU2, "IYU2", true, false, 1, UNPACK_PACKED3, 0, 0, 0, 0, {0, 1, 0, 2}},
  {GSTAMD_VIDEO_FORMAT_VUYA, "VUYA", true, true, 1, UNPACK_PACKED4, 0, 0, 0, 0, {3, 2, 1, 0}},
  {GSTAMD_VIDEO_FORMAT_AYUV, "AYUV", true, true, 1, UNPACK_PACKED4, 0, 0, 0, 0, {0, 1, 2, 3}},
  {GSTAMD_VIDEO_FORMAT_ARGB, "ARGB", false, true, 1, UNPACK_PACKED4, 0, 0, 0, 0, {0, 1, 2, 3}},
  {GSTAMD_VIDEO_FORMAT_xRGB, "xRGB", false, false, 1, UNPACK_PACKED4, 0, 0, 0, 0, {0, 1, 2, 3}},
  {GSTAMD_VIDEO_FORMAT_BGRA, "BGRA", false, true, 1, UNPACK_PACKED4, 0, 0, 0, 0, {3, 2, 1, 0}},
  {GSTAMD_VIDEO_FORMAT_BGRx, "BGRx", false, false, 1, UNPACK_PACKED4, 0, 0, 0, 0, {3, 2, 1, 0}},
  {GSTAMD_VIDEO_FORMAT_RGBA, "RGBA", false, true, 1, UNPACK_PACKED4, 0, 0, 0, 0, {3, 0, 1, 2}},
  {GSTAMD_VIDEO_FORMAT_RGBx, "RGBx", false, false, 1, UNPACK_PACKED4, 0, 0, 0, 0, {3, 0, 1, 2}},
  {GSTAMD_VIDEO_FORMAT_ABGR, "ABGR", false, true, 1, UNPACK_PACKED4, 0, 0, 0, 0, {0, 3, 2, 1}},
  {GSTAMD_VIDEO_FORMAT_RBGA, "RBGA", false, true, 1, UNPACK_PACKED4, 0, 0, 0, 0, {3, 0, 2, 1}},          /* packed R B G A (unpack_RBGA video-format.c:7999) */
  {GSTAMD_VIDEO_FORMAT_xBGR, "xBGR", false, false, 1, UNPACK_PACKED4, 0, 0, 0, 0, {0, 3, 2, 1}},
  // 10 bits per sample in 16-bit little-endian words (video-format.c:3834-3873, 5329-5400); sources of the 16-bit chain
  {GSTAMD_VIDEO_FORMAT_I420_10LE, "I420_10LE", true, false, 3, UNPACK_PLANAR, 1, 1, 1, 2, {0, 0, 0, 0}, 1},
  {GSTAMD_VIDEO_FORMAT_P010_10LE, "P010_10LE", true, false, 2, UNPACK_SEMI, 1, 1, 1, 0, {0, 0, 0, 0}, 2},
  // 12 bits and 16 bits in the same layouts (video-format.c: unpack_I420_12LE ... pack_P016_LE, pack_Y444_16LE)
  {GSTAMD_VIDEO_FORMAT_I420_12LE, "I420_12LE", true, false, 3, UNPACK_PLANAR, 1, 1, 1, 2, {0, 0, 0, 0}, 4},
  {GSTAMD_VIDEO_FORMAT_I422_12LE, "I422_12LE", true, false, 3, UNPACK_PLANAR, 1, 0, 1, 2, {0, 0, 0, 0}, 4},
  {GSTAMD_VIDEO_FORMAT_Y444_12LE, "Y444_12LE", true, false, 3, UNPACK_PLANAR, 0, 0, 1, 2, {0, 0, 0, 0}, 4},
  {GSTAMD_VIDEO_FORMAT_P012_LE, "P012_LE", true, false, 2, UNPACK_SEMI, 1, 1, 1, 0, {0, 0, 0, 0}, 5},
  {GSTAMD_VIDEO_FORMAT_P016_LE, "P016_LE", true, false, 2, UNPACK_SEMI, 1, 1, 1, 0, {0, 0, 0, 0}, 6},
  {GSTAMD_VIDEO_FORMAT_Y444_16LE, "Y444_16LE", true, false, 3, UNPACK_PLANAR, 0, 0, 1, 2, {0, 0, 0, 0}, 6},
  {GSTAMD_VIDEO_FORMAT_I422_10LE, "I422_10LE", true, false, 3, UNPACK_PLANAR, 1, 0, 1, 2, {0, 0, 0, 0}, 1},
  {GSTAMD_VIDEO_FORMAT_Y444_10LE, "Y444_10LE", true, false, 3, UNPACK_PLANAR, 0, 0, 1, 2, {0, 0, 0, 0}, 1},
  // 16 bits per component, packed, native endianness (video-format.c:2426-2473, 2523-2570): the unpack formats of the 16-bit chain themselves
  /* the big-endian forms of the word-plane formats (GST_READ_UINT16_BE / GST_WRITE_UINT16_BE around the same arithmetic): hi_depth code + 20 */
  {GSTAMD_VIDEO_FORMAT_I420_10BE, "I420_10BE", true, false, 3, UNPACK_PLANAR, 1, 1, 1, 2, {0, 0, 0, 0}, 21},
  {GSTAMD_VIDEO_FORMAT_I422_10BE, "I422_10BE", true, false, 3, UNPACK_PLANAR, 1, 0, 1, 2, {0, 0, 0, 0}, 21},
  {GSTAMD_VIDEO_FORMAT_Y444_10BE, "Y444_10BE", true, false, 3, UNPACK_PLANAR, 0, 0, 1, 2, {0, 0, 0, 0}, 21},
  {GSTAMD_VIDEO_FORMAT_I420_12BE, "I420_12BE", true, false, 3, UNPACK_PLANAR, 1, 1, 1, 2, {0, 0, 0, 0}, 24},
  {GSTAMD_VIDEO_FORMAT_I422_12BE, "I422_12BE", true, false, 3, UNPACK_PLANAR, 1, 0, 1, 2, {0, 0, 0, 0}, 24},
  {GSTAMD_VIDEO_FORMAT_Y444_12BE, "Y444_12BE", true, false, 3, UNPACK_PLANAR, 0, 0, 1, 2, {0, 0, 0, 0}, 24},
  {GSTAMD_VIDEO_FORMAT_Y444_16BE, "Y444_16BE", true, false, 3, UNPACK_PLANAR, 0, 0, 1, 2, {0, 0, 0, 0}, 26},
  {GSTAMD_VIDEO_FORMAT_P010_10BE, "P010_10BE", true, false, 2, UNPACK_SEMI, 1, 1, 1, 0, {0, 0, 0, 0}, 22},
  {GSTAMD_VIDEO_FORMAT_P012_BE, "P012_BE", true, false, 2, UNPACK_SEMI, 1, 1, 1, 0, {0, 0, 0, 0}, 25},
  {GSTAMD_VIDEO_FORMAT_P016_BE, "P016_BE", true, false, 2, UNPACK_SEMI, 1, 1, 1, 0, {0, 0, 0, 0}, 26},
  {GSTAMD_VIDEO_FORMAT_GBR_10BE, "GBR_10BE", false, false, 3, UNPACK_PLANAR, 0, 0, 1, 2, {0, 0, 0, 0}, 21},
  {GSTAMD_VIDEO_FORMAT_GBR_12BE, "GBR_12BE", false, false, 3, UNPACK_PLANAR, 0, 0, 1, 2, {0, 0, 0, 0}, 24},
  {GSTAMD_VIDEO_FORMAT_GBR_16BE, "GBR_16BE", false, false, 3, UNPACK_PLANAR, 0, 0, 1, 2, {0, 0, 0, 0}, 26},
  {GSTAMD_VIDEO_FORMAT_GBRA_10BE, "GBRA_10BE", false, true, 4, UNPACK_PLANAR_A, 0, 0, 1, 2, {0, 0, 0, 0}, 21},
  {GSTAMD_VIDEO_FORMAT_GBRA_12BE, "GBRA_12BE", false, true, 4, UNPACK_PLANAR_A, 0, 0, 1, 2, {0, 0, 0, 0}, 24},
  {GSTAMD_VIDEO_FORMAT_A420_10BE, "A420_10BE", true, true, 4, UNPACK_PLANAR_A, 1, 1, 1, 2, {0, 0, 0, 0}, 21},
  {GSTAMD_VIDEO_FORMAT_A422_10BE, "A422_10BE", true, true, 4, UNPACK_PLANAR_A, 1, 0, 1, 2, {0, 0, 0, 0}, 21},
  {GSTAMD_VIDEO_FORMAT_A444_10BE, "A444_10BE", true, true, 4, UNPACK_PLANAR_A, 0, 0, 1, 2, {0, 0, 0, 0}, 21},
  {GSTAMD_VIDEO_FORMAT_A420_12BE, "A420_12BE", true, true, 4, UNPACK_PLANAR_A, 1, 1, 1, 2, {0, 0, 0, 0}, 24},
  {GSTAMD_VIDEO_FORMAT_A422_12BE, "A422_12BE", true, true, 4, UNPACK_PLANAR_A, 1, 0, 1, 2, {0, 0, 0, 0}, 24},
  {GSTAMD_VIDEO_FORMAT_A444_12BE, "A444_12BE", true, true, 4, UNPACK_PLANAR_A, 0, 0, 1, 2, {0, 0, 0, 0}, 24},
  {GSTAMD_VIDEO_FORMAT_A420_16BE, "A420_16BE", true, true, 4, UNPACK_PLANAR_A, 1, 1, 1, 2, {0, 0, 0, 0}, 26},
  {GSTAMD_VIDEO_FORMAT_A422_16BE, "A422_16BE", true, true, 4, UNPACK_PLANAR_A, 1, 0, 1, 2, {0, 0, 0, 0}, 26},
  {GSTAMD_VIDEO_FORMAT_A444_16BE, "A444_16BE", true, true, 4, UNPACK_PLANAR_A, 0, 0, 1, 2, {0, 0, 0, 0}, 26},
  {GSTAMD_VIDEO_FORMAT_Y212_BE, "Y212_BE", true, false, 1, UNPACK_P422_16, 1, 0, 0, 0, {0, 0, 1, 3}, 25},
  {GSTAMD_VIDEO_FORMAT_Y216_BE, "Y216_BE", true, false, 1, UNPACK_P422_16, 1, 0, 0, 0, {0, 0, 1, 3}, 26},
  {GSTAMD_VIDEO_FORMAT_Y412_BE, "Y412_BE", true, false, 1, UNPACK_PACKED64, 0, 0, 0, 0, {3, 1, 0, 2}, 12},
  {GSTAMD_VIDEO_FORMAT_Y416_BE, "Y416_BE", true, false, 1, UNPACK_PACKED64, 0, 0, 0, 0, {3, 1, 0, 2}, 10},
  {GSTAMD_VIDEO_FORMAT_ARGB64, "ARGB64", false, true, 1, UNPACK_PACKED4, 0, 0, 0, 0, {0, 1, 2, 3}, 3},
  /* the endian-specific 64-bit formats (video-format.c:2467-2815, table :8421-8436): ARGB64_LE is ARGB64 on this host (unpack_copy8), the others
     reorder and / or byte-swap the four words */
  {GSTAMD_VIDEO_FORMAT_ARGB64_LE, "ARGB64_LE", false, true, 1, UNPACK_PACKED4, 0, 0, 0, 0, {0, 1, 2, 3}, 3},
  {GSTAMD_VIDEO_FORMAT_ARGB64_BE, "ARGB64_BE", false, true, 1, UNPACK_PACKED64, 0, 0, 0, 0, {0, 1, 2, 3}, 10},
  {GSTAMD_VIDEO_FORMAT_RGBA64_LE, "RGBA64_LE", false, true, 1, UNPACK_PACKED64, 0, 0, 0, 0, {3, 0, 1, 2}, 9},
  /* unpack / pack_RGBA_F16LE / _F16BE (video-format.c:2828-2918): RGBA64's words as IEEE half floats (hi_depth codes 16 / 36: px16_load / px16_store) */
  {GSTAMD_VIDEO_FORMAT_RGBA_F16LE, "RGBA_F16LE", false, true, 1, UNPACK_PACKED64, 0, 0, 0, 0, {3, 0, 1, 2}, 16},
  {GSTAMD_VIDEO_FORMAT_RGBA_F16BE, "RGBA_F16BE", false, true, 1, UNPACK_PACKED64, 0, 0, 0, 0, {3, 0, 1, 2}, 36},
  {GSTAMD_VIDEO_FORMAT_RGBA64_BE, "RGBA64_BE", false, true, 1, UNPACK_PACKED64, 0, 0, 0, 0, {3, 0, 1, 2}, 10},
  {GSTAMD_VIDEO_FORMAT_BGRA64_LE, "BGRA64_LE", false, true, 1, UNPACK_PACKED64, 0, 0, 0, 0, {3, 2, 1, 0}, 9},
  {GSTAMD_VIDEO_FORMAT_BGRA64_BE, "BGRA64_BE", false, true, 1, UNPACK_PACKED64, 0, 0, 0, 0, {3, 2, 1, 0}, 10},
  {GSTAMD_VIDEO_FORMAT_ABGR64_LE, "ABGR64_LE", false, true, 1, UNPACK_PACKED64, 0, 0, 0, 0, {0, 3, 2, 1}, 9},
  {GSTAMD_VIDEO_FORMAT_ABGR64_BE, "ABGR64_BE", false, true, 1, UNPACK_PACKED64, 0, 0, 0, 0, {0, 3, 2, 1}, 10},
  /* unpack_GRAY16_LE / _BE (video-format.c:1231-1299): A = 0xffff, Y, U = V = 0x8000 */
  {GSTAMD_VIDEO_FORMAT_GRAY16_LE, "GRAY16_LE", true, false, 1, UNPACK_GRAY16, 0, 0, 0, 0, {0, 0, 0, 0}, 9},
  {GSTAMD_VIDEO_FORMAT_GRAY16_BE, "GRAY16_BE", true, false, 1, UNPACK_GRAY16, 0, 0, 0, 0, {0, 0, 0, 0}, 10},
  {GSTAMD_VIDEO_FORMAT_AYUV64, "AYUV64", true, true, 1, UNPACK_PACKED4, 0, 0, 0, 0, {0, 1, 2, 3}, 3},
};

const FormatDesc *format_desc (int format)
{
  for (const FormatDesc &f : g_formats)
    if (f.format == format)
      return &f;
  return nullptr;
}

static inline int round_up (int v, int n) { return (v + n - 1) / n * n; }

// gst_video_info_set_format's fill_planes (video-info.c:863-1100) + the caps defaults applied by
// gst_video_info_from_caps (set_default_colorimetry :165-190, set_default_chroma_site :212-225).
int video_info_set_format (GstAmdVideoInfo *info, int format, int width, int height)
{
  const FormatDesc *f = format_desc (format);
  if (!info || !f || width <= 0 || height <= 0)
    return GSTAMD_ERR_INVALID;
  memset (info, 0, sizeof (*info));
  info->format = format;
  info->width = width;
  info->height = height;
  info->n_planes = f->n_planes;
  uint64_t w = (uint64_t) width, h = (uint64_t) height;
  switch (format) {
    case GSTAMD_VIDEO_FORMAT_I420:
    case GSTAMD_VIDEO_FORMAT_YV12: {
      info->stride[0] = round_up (width, 4);
      info->stride[1] = round_up (round_up (width, 2) / 2, 4);
      info->stride[2] = info->stride[1];
      info->offset[1] = (uint64_t) info->stride[0] * round_up (height, 2);
      uint64_t cr_h = round_up (height, 2) / 2;
      info->offset[2] = info->offset[1] + info->stride[1] * cr_h;
      info->size = info->offset[2] + info->stride[2] * cr_h;
      break;
    }
    case GSTAMD_VIDEO_FORMAT_A420: {            /* video-info.c:1089-1102 */
      info->stride[0] = round_up (width, 4);
      info->stride[1] = round_up (round_up (width, 2) / 2, 4);
      info->stride[2] = info->stride[1];
      info->stride[3] = info->stride[0];
      const uint64_t h2 = (uint64_t) round_up (height, 2);
      info->offset[1] = (uint64_t) info->stride[0] * h2;
      info->offset[2] = info->offset[1] + (uint64_t) info->stride[1] * (h2 / 2);
      info->offset[3] = info->offset[2] + (uint64_t) info->stride[2] * (h2 / 2);
      info->size = info->offset[3] + (uint64_t) info->stride[0] * h2;
      break;
    }
    case GSTAMD_VIDEO_FORMAT_Y42B:
      info->stride[0] = round_up (width, 4);
      info->stride[1] = round_up (width, 8) / 2;
      info->stride[2] = info->stride[1];
      info->offset[1] = (uint64_t) info->stride[0] * h;
      info->offset[2] = info->offset[1] + info->stride[1] * h;
      info->size = info->offset[2] + info->stride[2] * h;
      break;
    case GSTAMD_VIDEO_FORMAT_GBRA:            /* video-info.c:1042-1052 */
      info->stride[0] = info->stride[1] = info->stride[2] = info->stride[3] = round_up (width, 4);
      info->offset[1] = (uint64_t) info->stride[0] * h;
      info->offset[2] = info->offset[1] * 2;
      info->offset[3] = info->offset[1] * 3;
      info->size = (uint64_t) info->stride[0] * h * 4;
      break;
    case GSTAMD_VIDEO_FORMAT_A420_10BE:
    case GSTAMD_VIDEO_FORMAT_A420_10LE:
    case GSTAMD_VIDEO_FORMAT_A420_12BE:
    case GSTAMD_VIDEO_FORMAT_A420_12LE:
    case GSTAMD_VIDEO_FORMAT_A420_16BE:
    case GSTAMD_VIDEO_FORMAT_A420_16LE:
    case GSTAMD_VIDEO_FORMAT_A422_10BE:
    case GSTAMD_VIDEO_FORMAT_A422_10LE:
    case GSTAMD_VIDEO_FORMAT_A422_12BE:
    case GSTAMD_VIDEO_FORMAT_A422_12LE:
    case GSTAMD_VIDEO_FORMAT_A422_16BE:
    case GSTAMD_VIDEO_FORMAT_A422_16LE: {      /* video-info.c:1256-1291 */
      const bool v2 = format == GSTAMD_VIDEO_FORMAT_A420_10LE || format == GSTAMD_VIDEO_FORMAT_A420_12LE || format == GSTAMD_VIDEO_FORMAT_A420_16LE ||
          format == GSTAMD_VIDEO_FORMAT_A420_10BE || format == GSTAMD_VIDEO_FORMAT_A420_12BE || format == GSTAMD_VIDEO_FORMAT_A420_16BE;
      const uint64_t h2 = (uint64_t) round_up (height, 2), ch = v2 ? h2 / 2 : h2;
      info->stride[0] = info->stride[3] = round_up (width * 2, 4);
      info->stride[1] = info->stride[2] = round_up (width, 4);
      info->offset[1] = (uint64_t) info->stride[0] * h2;
      info->offset[2] = info->offset[1] + (uint64_t) info->stride[1] * ch;
      info->offset[3] = info->offset[2] + (uint64_t) info->stride[2] * ch;
      info->size = info->offset[3] + (uint64_t) info->stride[0] * h2;
      break;
    }
    case GSTAMD_VIDEO_FORMAT_A444_10BE:
    case GSTAMD_VIDEO_FORMAT_A444_10LE:
    case GSTAMD_VIDEO_FORMAT_A444_12BE:
    case GSTAMD_VIDEO_FORMAT_A444_12LE:
    case GSTAMD_VIDEO_FORMAT_A444_16BE:
    case GSTAMD_VIDEO_FORMAT_A444_16LE:
    case GSTAMD_VIDEO_FORMAT_GBRA_10BE:
    case GSTAMD_VIDEO_FORMAT_GBRA_10LE:
    case GSTAMD_VIDEO_FORMAT_GBRA_12BE:
    case GSTAMD_VIDEO_FORMAT_GBRA_12LE:        /* video-info.c:1189-1203, 1292-1308 */
      info->stride[0] = info->stride[1] = info->stride[2] = info->stride[3] = round_up (width * 2, 4);
      info->offset[1] = (uint64_t) info->stride[0] * h;
      info->offset[2] = info->offset[1] * 2;
      info->offset[3] = info->offset[1] * 3;
      info->size = (uint64_t) info->stride[0] * h * 4;
      break;
    case GSTAMD_VIDEO_FORMAT_A422:
    case GSTAMD_VIDEO_FORMAT_A444: {           /* video-info.c:1103-1128 */
      const uint64_t h2 = (uint64_t) round_up (height, 2);
      info->stride[0] = info->stride[3] = round_up (width, 4);
      info->stride[1] = info->stride[2] = format == GSTAMD_VIDEO_FORMAT_A422 ? round_up (width, 8) / 2 : info->stride[0];
      info->offset[1] = (uint64_t) info->stride[0] * h2;
      info->offset[2] = info->offset[1] + (uint64_t) info->stride[1] * h2;
      info->offset[3] = info->offset[2] + (uint64_t) info->stride[2] * h2;
      info->size = info->offset[3] + (uint64_t) info->stride[0] * h2;
      break;
    }
    case GSTAMD_VIDEO_FORMAT_RGBP:
    case GSTAMD_VIDEO_FORMAT_BGRP:
    case GSTAMD_VIDEO_FORMAT_Y444:
    case GSTAMD_VIDEO_FORMAT_GBR:             /* video-info.c:1030-1041 */
      info->stride[0] = info->stride[1] = info->stride[2] = round_up (width, 4);
      info->offset[1] = (uint64_t) info->stride[0] * h;
      info->offset[2] = info->offset[1] * 2;
      info->size = (uint64_t) info->stride[0] * h * 3;
      break;
    case GSTAMD_VIDEO_FORMAT_GRAY10_LE32:       /* video-info.c:1327-1348 */
      info->stride[0] = (width + 2) / 3 * 4;
      info->size = (uint64_t) info->stride[0] * round_up (height, 2);
      break;
    case GSTAMD_VIDEO_FORMAT_NV12_10LE32:
      info->stride[0] = info->stride[1] = (width + 2) / 3 * 4;
      info->offset[1] = (uint64_t) info->stride[0] * round_up (height, 2);
      info->size = info->offset[1] + (uint64_t) info->stride[0] * (round_up (height, 2) / 2);
      break;
    case GSTAMD_VIDEO_FORMAT_NV16_10LE32:
      info->stride[0] = info->stride[1] = (width + 2) / 3 * 4;
      info->offset[1] = (uint64_t) info->stride[0] * height;
      info->size = (uint64_t) info->stride[0] * height * 2;
      break;
    case GSTAMD_VIDEO_FORMAT_NV12_10LE40:       /* video-info.c:1349-1365 */
      info->stride[0] = info->stride[1] = ((width * 5 >> 2) + 4) / 5 * 5;
      info->offset[1] = (uint64_t) info->stride[0] * round_up (height, 2);
      info->size = info->offset[1] + (uint64_t) info->stride[0] * (round_up (height, 2) / 2);
      break;
    case GSTAMD_VIDEO_FORMAT_NV16_10LE40:
      info->stride[0] = info->stride[1] = ((width * 5 >> 2) + 4) / 5 * 5;
      info->offset[1] = (uint64_t) info->stride[0] * round_up (height, 2);
      info->size = info->offset[1] * 2;
      break;
    case GSTAMD_VIDEO_FORMAT_UYVP:              /* video-info.c:953-957 */
      info->stride[0] = round_up (round_up (width, 2) * 5 / 2, 4);
      info->size = (uint64_t) info->stride[0] * height;
      break;
    case GSTAMD_VIDEO_FORMAT_NV12_64Z32: {      /* video-info.c:1204-1215 */
      const int w128 = round_up (width, 128), h32 = round_up (height, 32), h64 = round_up (height, 64);
      info->stride[0] = ((h32 / 32) << 16) | (w128 / 64);
      info->stride[1] = ((h64 / 64) << 16) | (w128 / 64);
      info->offset[1] = (uint64_t) w128 * h32;
      info->size = info->offset[1] + (uint64_t) w128 * (h64 / 2);
      break;
    }
    case GSTAMD_VIDEO_FORMAT_NV12_4L4:
    case GSTAMD_VIDEO_FORMAT_NV12_32L32:
    case GSTAMD_VIDEO_FORMAT_NV12_8L128: {      /* video-info.c:1216-1233, 1366-1388 */
      const int ts = format == GSTAMD_VIDEO_FORMAT_NV12_4L4 ? 4 : (format == GSTAMD_VIDEO_FORMAT_NV12_32L32 ? 32 : 8);
      const int th = format == GSTAMD_VIDEO_FORMAT_NV12_8L128 ? 128 : ts;
      const int nx = round_up (width, ts) / ts, ny = round_up (height, th) / th, nuv = round_up (ny, 2) / 2;
      info->stride[0] = (ny << 16) | nx;
      info->stride[1] = (nuv << 16) | nx;
      info->offset[1] = (uint64_t) nx * ny * (ts * th);
      info->size = info->offset[1] + (uint64_t) nx * nuv * (ts * th);
      break;
    }
    case GSTAMD_VIDEO_FORMAT_NV12_10LE40_4L4: { /* video-info.c:1216-1233 with 20-byte tiles */
      const int nx = round_up (width, 4) / 4, ny = round_up (height, 4) / 4, nuv = round_up (ny, 2) / 2;
      info->stride[0] = (ny << 16) | nx;
      info->stride[1] = (nuv << 16) | nx;
      info->offset[1] = (uint64_t) nx * ny * 20;
      info->size = info->offset[1] + (uint64_t) nx * nuv * 20;
      break;
    }
    case GSTAMD_VIDEO_FORMAT_NV12_16L32S: {     /* video-info.c:1234-1255 */
      const int nx = round_up (width, 16) / 16, ny = round_up (height, 32) / 32;
      info->stride[0] = info->stride[1] = (ny << 16) | nx;
      info->offset[1] = (uint64_t) nx * ny * 512;
      info->size = info->offset[1] + (uint64_t) nx * ny * 256;
      break;
    }
    case GSTAMD_VIDEO_FORMAT_IYU1:              /* video-info.c:965-970 */
      info->stride[0] = round_up (round_up (width, 4) + round_up (width, 4) / 2, 4);
      info->size = (uint64_t) info->stride[0] * height;
      break;
    case GSTAMD_VIDEO_FORMAT_Y41B:              /* video-info.c:1010-1019 */
      info->stride[0] = round_up (width, 4);
      info->stride[1] = info->stride[2] = round_up (width, 16) / 4;
      info->offset[1] = (uint64_t) info->stride[0] * height;
      info->offset[2] = info->offset[1] + (uint64_t) info->stride[1] * height;
      info->size = ((uint64_t) info->stride[0] + round_up (width, 16) / 2) * height;
      break;
    case GSTAMD_VIDEO_FORMAT_AV12: {            /* video-info.c:1064-1073 */
      const uint64_t h2 = (uint64_t) round_up (height, 2);
      info->stride[0] = info->stride[1] = info->stride[2] = round_up (width, 4);
      info->offset[1] = (uint64_t) info->stride[0] * h2;
      info->offset[2] = info->offset[1] + (uint64_t) info->stride[1] * h2 / 2;
      info->size = info->offset[2] + (uint64_t) info->stride[2] * h2;
      break;
    }
    case GSTAMD_VIDEO_FORMAT_NV12:
    case GSTAMD_VIDEO_FORMAT_NV21: {
      info->stride[0] = round_up (width, 4);
      info->stride[1] = info->stride[0];
      info->offset[1] = (uint64_t) info->stride[0] * round_up (height, 2);
      uint64_t cr_h = round_up (height, 2) / 2;
      info->size = info->offset[1] + info->stride[0] * cr_h;
      break;
    }
    case GSTAMD_VIDEO_FORMAT_I420_12BE:
    case GSTAMD_VIDEO_FORMAT_I420_12LE:
    case GSTAMD_VIDEO_FORMAT_I420_10BE:
    case GSTAMD_VIDEO_FORMAT_I420_10LE: {       /* video-info.c:1142-1156 */
      info->stride[0] = round_up (width * 2, 4);
      info->stride[1] = info->stride[2] = round_up (width, 4);
      info->offset[1] = (uint64_t) info->stride[0] * round_up (height, 2);
      uint64_t cr_h = round_up (height, 2) / 2;
      info->offset[2] = info->offset[1] + info->stride[1] * cr_h;
      info->size = info->offset[2] + info->stride[2] * cr_h;
      break;
    }
    case GSTAMD_VIDEO_FORMAT_I422_12BE:
    case GSTAMD_VIDEO_FORMAT_I422_12LE:
    case GSTAMD_VIDEO_FORMAT_I422_10BE:
    case GSTAMD_VIDEO_FORMAT_I422_10LE:         /* video-info.c:1157-1169 */
      info->stride[0] = round_up (width * 2, 4);
      info->stride[1] = info->stride[2] = round_up (width, 4);
      info->offset[1] = (uint64_t) info->stride[0] * round_up (height, 2);
      info->offset[2] = info->offset[1] + (uint64_t) info->stride[1] * round_up (height, 2);
      info->size = info->offset[2] + (uint64_t) info->stride[2] * round_up (height, 2);
      break;
    case GSTAMD_VIDEO_FORMAT_GBR_10BE:
    case GSTAMD_VIDEO_FORMAT_GBR_10LE:
    case GSTAMD_VIDEO_FORMAT_GBR_12BE:
    case GSTAMD_VIDEO_FORMAT_GBR_12LE:
    case GSTAMD_VIDEO_FORMAT_GBR_16BE:
    case GSTAMD_VIDEO_FORMAT_GBR_16LE:
    case GSTAMD_VIDEO_FORMAT_Y444_12BE:
    case GSTAMD_VIDEO_FORMAT_Y444_12LE:
    case GSTAMD_VIDEO_FORMAT_Y444_16BE:
    case GSTAMD_VIDEO_FORMAT_Y444_16LE:
    case GSTAMD_VIDEO_FORMAT_Y444_10BE:
    case GSTAMD_VIDEO_FORMAT_Y444_10LE:         /* video-info.c:1170-1188 */
      info->stride[0] = info->stride[1] = info->stride[2] = round_up (width * 2, 4);
      info->offset[1] = (uint64_t) info->stride[0] * h;
      info->offset[2] = info->offset[1] * 2;
      info->size = (uint64_t) info->stride[0] * h * 3;
      break;
    case GSTAMD_VIDEO_FORMAT_P012_BE:
    case GSTAMD_VIDEO_FORMAT_P012_LE:
    case GSTAMD_VIDEO_FORMAT_P016_BE:
    case GSTAMD_VIDEO_FORMAT_P016_LE:
    case GSTAMD_VIDEO_FORMAT_P010_10BE:
    case GSTAMD_VIDEO_FORMAT_P010_10LE: {       /* video-info.c:1309-1321 */
      info->stride[0] = info->stride[1] = round_up (width * 2, 4);
      info->offset[1] = (uint64_t) info->stride[0] * round_up (height, 2);
      info->size = info->offset[1] + (uint64_t) info->stride[0] * (round_up (height, 2) / 2);
      break;
    }
    case GSTAMD_VIDEO_FORMAT_NV16:
    case GSTAMD_VIDEO_FORMAT_NV61:
      info->stride[0] = info->stride[1] = round_up (width, 4);
      info->offset[1] = (uint64_t) info->stride[0] * h;
      info->size = (uint64_t) info->stride[0] * h * 2;
      break;
    case GSTAMD_VIDEO_FORMAT_NV24:
      info->stride[0] = round_up (width, 4);
      info->stride[1] = round_up (width * 2, 4);
      info->offset[1] = (uint64_t) info->stride[0] * h;
      info->size = info->offset[1] + (uint64_t) info->stride[1] * h;
      break;
    case GSTAMD_VIDEO_FORMAT_YUY2:
    case GSTAMD_VIDEO_FORMAT_UYVY:
    case GSTAMD_VIDEO_FORMAT_YVYU:
    case GSTAMD_VIDEO_FORMAT_VYUY:
      info->stride[0] = round_up (width * 2, 4);
      info->size = (uint64_t) info->stride[0] * h;
      break;
    case GSTAMD_VIDEO_FORMAT_v210:              /* video-info.c:927-931 */
      info->stride[0] = ((width + 47) / 48) * 128;
      info->size = (uint64_t) info->stride[0] * h;
      break;
    case GSTAMD_VIDEO_FORMAT_Y210:
    case GSTAMD_VIDEO_FORMAT_v216:
    case GSTAMD_VIDEO_FORMAT_Y216_BE:
    case GSTAMD_VIDEO_FORMAT_Y216_LE:
    case GSTAMD_VIDEO_FORMAT_Y212_BE:
    case GSTAMD_VIDEO_FORMAT_Y212_LE:           /* video-info.c:932-941 */
      info->stride[0] = round_up (width * 4, 8);
      info->size = (uint64_t) info->stride[0] * h;
      break;
    case GSTAMD_VIDEO_FORMAT_GRAY8:             /* video-info.c:942-946 */
      info->stride[0] = round_up (width, 4);
      info->size = (uint64_t) info->stride[0] * h;
      break;
    case GSTAMD_VIDEO_FORMAT_RGB:
    case GSTAMD_VIDEO_FORMAT_BGR:
    case GSTAMD_VIDEO_FORMAT_v308:
    case GSTAMD_VIDEO_FORMAT_IYU2:
      info->stride[0] = round_up (width * 3, 4);
      info->size = (uint64_t) info->stride[0] * h;
      break;
    case GSTAMD_VIDEO_FORMAT_RGB16:
    case GSTAMD_VIDEO_FORMAT_BGR16:
    case GSTAMD_VIDEO_FORMAT_RGB15:
    case GSTAMD_VIDEO_FORMAT_BGR15:             /* video-info.c:911-918 */
    case GSTAMD_VIDEO_FORMAT_GRAY10_LE16:
    case GSTAMD_VIDEO_FORMAT_GRAY16_LE:
    case GSTAMD_VIDEO_FORMAT_GRAY16_BE:         /* video-info.c:947-952 */
      info->stride[0] = round_up (width * 2, 4);
      info->size = (uint64_t) info->stride[0] * h;
      break;
    case GSTAMD_VIDEO_FORMAT_Y412_BE:
    case GSTAMD_VIDEO_FORMAT_Y412_LE:
    case GSTAMD_VIDEO_FORMAT_Y416_BE:
    case GSTAMD_VIDEO_FORMAT_Y416_LE:
    case GSTAMD_VIDEO_FORMAT_ARGB64_LE:
    case GSTAMD_VIDEO_FORMAT_ARGB64_BE:
    case GSTAMD_VIDEO_FORMAT_RGBA_F16LE:
    case GSTAMD_VIDEO_FORMAT_RGBA_F16BE:
    case GSTAMD_VIDEO_FORMAT_RGBA64_LE:
    case GSTAMD_VIDEO_FORMAT_RGBA64_BE:
    case GSTAMD_VIDEO_FORMAT_BGRA64_LE:
    case GSTAMD_VIDEO_FORMAT_BGRA64_BE:
    case GSTAMD_VIDEO_FORMAT_ABGR64_LE:
    case GSTAMD_VIDEO_FORMAT_ABGR64_BE:
    case GSTAMD_VIDEO_FORMAT_ARGB64:
    case GSTAMD_VIDEO_FORMAT_AYUV64:
      info->stride[0] = width * 8;
      info->size = w * 8 * h;
      break;
    default:                   /* 4-byte packed */
      info->stride[0] = width * 4;
      info->size = w * 4 * h;
      break;
  }
  if (f->kind == UNPACK_GRAY || f->kind == UNPACK_GRAY16 || f->kind == UNPACK_GRAY_LE32) {         /* set_default_colorimetry (video-info.c:175-176): DEFAULT_GRAY = 0 .. 255, bt601, unknown, unknown */
    info->color_range = GSTAMD_COLOR_RANGE_0_255;
    info->color_matrix = GSTAMD_COLOR_MATRIX_BT601;
    info->chroma_site = GSTAMD_CHROMA_SITE_UNKNOWN;
    info->color_transfer = GSTAMD_TRANSFER_UNKNOWN;
    info->color_primaries = GSTAMD_PRIMARIES_UNKNOWN;
  } else if (f->yuv) {
    info->color_range = GSTAMD_COLOR_RANGE_16_235;
    info->color_matrix = height > 576 ? GSTAMD_COLOR_MATRIX_BT709 : GSTAMD_COLOR_MATRIX_BT601;
    info->chroma_site = height > 576 ? GSTAMD_CHROMA_SITE_H_COSITED : GSTAMD_CHROMA_SITE_NONE;
    /* the bt709 / bt601 colorimetry rows of video-color.c:72-73 */
    info->color_transfer = height > 576 ? GSTAMD_TRANSFER_BT709 : GSTAMD_TRANSFER_BT601;
    info->color_primaries = height > 576 ? GSTAMD_PRIMARIES_BT709 : GSTAMD_PRIMARIES_SMPTE170M;
  } else {
    /* (float formats: DEFAULT_RGB_FLOAT, video-info.c:153-180 - range 0_1, which gst_video_color_range_offsets treats like 0_255, video-color.c:215) */
    info->color_range = f->hi_depth == 16 || f->hi_depth == 36 ? GSTAMD_COLOR_RANGE_0_1 : GSTAMD_COLOR_RANGE_0_255;
    info->color_matrix = GSTAMD_COLOR_MATRIX_RGB;
    info->chroma_site = GSTAMD_CHROMA_SITE_UNKNOWN;
    info->color_transfer = GSTAMD_TRANSFER_SRGB;         /* sRGB row (:75) */
    info->color_primaries = GSTAMD_PRIMARIES_BT709;
  }
  return GSTAMD_OK;
}

// defaults of video-converter.c:778-796 and video-resampler.c:63-70
void converter_config_init (GstAmdVideoConverterConfig *c)
{
  memset (c, 0, sizeof (*c));
  c->resampler_method = GSTAMD_RESAMPLER_METHOD_CUBIC;
  c->resampler_taps = 0;
  c->max_taps = 128;
  c->envelope = 2.0;
  c->sharpness = 1.0;
  c->sharpen = 0.0;
  c->cubic_b = 1.0 / 3.0;
  c->cubic_c = 1.0 / 3.0;
  c->alpha_mode = GSTAMD_ALPHA_MODE_COPY;
  c->alpha_value = 1.0;
  c->chroma_mode = GSTAMD_CHROMA_MODE_FULL;
  c->matrix_mode = GSTAMD_MATRIX_MODE_FULL;
  c->dither_quantization = 1;
  c->dither_method = GSTAMD_DITHER_BAYER;                            /* DEFAULT_OPT_DITHER_METHOD */
  c->chroma_resampler_method = GSTAMD_RESAMPLER_METHOD_LINEAR;
  c->fill_border = 1;                                                /* DEFAULT_OPT_FILL_BORDER */
  c->border_argb = 0xff000000u;                                      /* DEFAULT_OPT_BORDER_ARGB */
      /* DEFAULT_OPT_CHROMA_RESAMPLER_METHOD (:786) */
}

// ------------------------------------------------------------------------------------------------
// colour matrix (video-converter.c:901-1066, 1323-1442, 1719-1838)
// ------------------------------------------------------------------------------------------------
typedef double M44[4][4];

static void m_identity (M44 m)
{
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      m[i][j] = (i == j);
}

/* dst = a * b, dst may alias (video-converter.c:921-938) */
static void m_multiply (M44 dst, M44 a, M44 b)
{
  M44 tmp;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double x = 0;
      for (int k = 0; k < 4; k++)
        x += a[i][k] * b[k][j];
      tmp[i][j] = x;
    }
  memcpy (dst, tmp, sizeof (M44));
}

static void m_offset_components (M44 m, double a1, double a2, double a3)
{
  M44 a;
  m_identity (a);
  a[0][3] = a1;
  a[1][3] = a2;
  a[2][3] = a3;
  m_multiply (m, a, m);
}

static void m_scale_components (M44 m, double a1, double a2, double a3)
{
  M44 a;
  m_identity (a);
  a[0][0] = a1;
  a[1][1] = a2;
  a[2][2] = a3;
  m_multiply (m, a, m);
}

static void m_YCbCr_to_RGB (M44 m, double Kr, double Kb)
{
  double Kg = 1.0 - Kr - Kb;
  M44 k = {
    {1., 0., 2 * (1 - Kr), 0.},
    {1., -2 * Kb * (1 - Kb) / Kg, -2 * Kr * (1 - Kr) / Kg, 0.},
    {1., 2 * (1 - Kb), 0., 0.},
    {0., 0., 0., 1.},
  };
  m_multiply (m, k, m);
}

static void m_RGB_to_YCbCr (M44 m, double Kr, double Kb)
{
  double Kg = 1.0 - Kr - Kb;
  M44 k;
  double x;
  k[0][0] = Kr;
  k[0][1] = Kg;
  k[0][2] = Kb;
  k[0][3] = 0;
  x = 1 / (2 * (1 - Kb));
  k[1][0] = -x * Kr;
  k[1][1] = -x * Kg;
  k[1][2] = x * (1 - Kb);
  k[1][3] = 0;
  x = 1 / (2 * (1 - Kr));
  k[2][0] = x * (1 - Kr);
  k[2][1] = -x * Kg;
  k[2][2] = -x * Kb;
  k[2][3] = 0;
  k[3][0] = 0;
  k[3][1] = 0;
  k[3][2] = 0;
  k[3][3] = 1;
  m_multiply (m, k, m);
}

/* gst_video_color_matrix_get_Kr_Kb (video-color.c:423-459) */
static bool get_Kr_Kb (int matrix, double *Kr, double *Kb)
{
  switch (matrix) {
    case GSTAMD_COLOR_MATRIX_FCC: *Kr = 0.30; *Kb = 0.11; return true;
    case GSTAMD_COLOR_MATRIX_BT709: *Kr = 0.2126; *Kb = 0.0722; return true;
    case GSTAMD_COLOR_MATRIX_BT601: *Kr = 0.2990; *Kb = 0.1140; return true;
    case GSTAMD_COLOR_MATRIX_SMPTE240M: *Kr = 0.212; *Kb = 0.087; return true;
    case GSTAMD_COLOR_MATRIX_BT2020: *Kr = 0.2627; *Kb = 0.0593; return true;
    default: return false;
  }
}

/* gst_video_color_range_offsets for the unpack formats AYUV / ARGB (depth 8) and AYUV64 / ARGB64 (depth 16) (video-color.c:204-252) */
static void range_offsets (int range, bool yuv, int offset[3], int scale[3], int depth = 8)
{
  if (range == GSTAMD_COLOR_RANGE_16_235) {
    offset[0] = 1 << (depth - 4);
    scale[0] = 219 << (depth - 8);
    if (yuv) {
      offset[1] = offset[2] = 1 << (depth - 1);
      scale[1] = scale[2] = 224 << (depth - 8);
    } else {
      offset[1] = offset[2] = 1 << (depth - 4);
      scale[1] = scale[2] = 219 << (depth - 8);
    }
  } else {
    offset[0] = 0;
    offset[1] = offset[2] = yuv ? 1 << (depth - 1) : 0;
    scale[0] = scale[1] = scale[2] = (1 << depth) - 1;
  }
}

static void compute_convert_matrix_depth (int in_range, int in_matrix, int out_range, int out_matrix, bool in_yuv, bool out_yuv,
    int matrix_mode, int in_depth, double dm[4][4], const double (*start)[4] = nullptr, int out_depth = 8);

void compute_convert_matrix (const VideoPlan &, int in_range, int in_matrix, int out_range,
    int out_matrix, bool in_yuv, bool out_yuv, int matrix_mode, double dm[4][4])
{
  compute_convert_matrix_depth (in_range, in_matrix, out_range, out_matrix, in_yuv, out_yuv, matrix_mode, 8, dm);
}

/* chain_convert's "no gamma, combine all conversions into 1" (video-converter.c:1803-1830) for an input unpack format of in_depth
 * bits and an 8-bit output: the input side's offsets / scales are those of the 16-bit unpack format, and with in_bits > out_bits the
 * whole matrix is scaled by 1 << (in_bits - out_bits) so that it produces 16-bit values again */
static void compute_convert_matrix_depth (int in_range, int in_matrix, int out_range, int out_matrix, bool in_yuv, bool out_yuv,
    int matrix_mode, int in_depth, double dm[4][4], const double (*start)[4], int out_depth)
{
  int offset[3], scale[3];
  double Kr = 0, Kb = 0;
  m_identity (dm);
  /* with differing primaries convert_matrix already holds RGB_in -> XYZ -> RGB_out when the other factors are multiplied on
   * from the left (chain_convert :1750-1822) */
  if (start)
    memcpy (dm, start, sizeof (M44));
  if (in_depth < out_depth) {           /* :1811-1815: the widened input counts as in_bits values with a fraction */
    const int down = 1 << (out_depth - in_depth);
    m_scale_components (dm, 1 / (float) down, 1 / (float) down, 1 / (float) down);
  }
  /* compute_matrix_to_RGB (video-converter.c:1372-1402) */
  range_offsets (in_range, in_yuv, offset, scale, in_depth);
  m_offset_components (dm, -offset[0], -offset[1], -offset[2]);
  m_scale_components (dm, 1 / ((float) scale[0]), 1 / ((float) scale[1]), 1 / ((float) scale[2]));
  if (in_yuv && matrix_mode != GSTAMD_MATRIX_MODE_NONE) {
    int mtx = matrix_mode == GSTAMD_MATRIX_MODE_OUTPUT_ONLY ? out_matrix : in_matrix;
    if (get_Kr_Kb (mtx, &Kr, &Kb))
      m_YCbCr_to_RGB (dm, Kr, Kb);
  }
  /* compute_matrix_to_YUV (video-converter.c:1405-1441) */
  if (out_yuv && matrix_mode != GSTAMD_MATRIX_MODE_NONE) {
    int mtx = matrix_mode == GSTAMD_MATRIX_MODE_INPUT_ONLY ? in_matrix : out_matrix;
    if (get_Kr_Kb (mtx, &Kr, &Kb))
      m_RGB_to_YCbCr (dm, Kr, Kb);
  }
  range_offsets (out_range, out_yuv, offset, scale, out_depth);
  m_scale_components (dm, (float) scale[0], (float) scale[1], (float) scale[2]);
  m_offset_components (dm, offset[0], offset[1], offset[2]);
  if (in_depth > out_depth) {
    const int up = 1 << (in_depth - out_depth);
    m_scale_components (dm, (float) up, (float) up, (float) up);
  }
}

/* prepare_matrix (video-converter.c:1323-1370) for current_bits == 8 */
static void prepare_matrix8 (M44 dm, bool unpack_rgb, bool pack_rgb, MatrixParams *mp, int im_raw[3][4])
{
  int im[4][4];
  m_scale_components (dm, 256.0f, 256.0f, 256.0f);      /* SCALE_F */
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      im[i][j] = (int) rint (dm[i][j]);
  memset (mp, 0, sizeof (*mp));
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 4; j++)
      mp->im[i][j] = im_raw[i][j] = im[i][j];
  bool ayuv_to_rgb = (im[0][0] == im[1][0] && im[1][0] == im[2][0]) && im[0][1] == 0 && im[2][2] == 0;
  if (!unpack_rgb && pack_rgb && ayuv_to_rgb) {
    mp->kind = MATRIX_AYUV_ARGB;     /* video_converter_matrix8_AYUV_ARGB (:1209) */
    mp->p[0] = im[0][0];
    mp->p[1] = im[0][2];
    mp->p[2] = im[2][1];
    mp->p[3] = im[1][1];
    mp->p[4] = im[1][2];
    return;
  }
  /* is_no_clip_matrix (:1252-1293) */
  static const uint8_t test[8][3] = {
    {0, 0, 0}, {0, 0, 255}, {0, 255, 0}, {0, 255, 255},
    {255, 0, 0}, {255, 0, 255}, {255, 255, 0}, {255, 255, 255}
  };
  bool no_clip = true;
  for (int i = 0; i < 8 && no_clip; i++) {
    int r = test[i][0], g = test[i][1], b = test[i][2];
    for (int k = 0; k < 3; k++) {
      int v = (im[k][0] * r + im[k][1] * g + im[k][2] * b + im[k][3]) >> 8;
      if (v < 0 || v > 255)
        no_clip = false;
    }
  }
  if (no_clip) {
    mp->kind = MATRIX_TABLE;         /* video_converter_matrix8_table (:1187) */
  } else {
    mp->kind = MATRIX_8;             /* _custom_video_orc_matrix8 (:1139): coefficients as gint16 */
    for (int k = 0; k < 3; k++) {
      for (int j = 0; j < 3; j++)
        mp->im[k][j] = (int16_t) (uint16_t) im[k][j];
      mp->im[k][3] = (int16_t) (uint16_t) (im[k][3] >> 8);
    }
  }
}


// ------------------------------------------------------------------------------------------------
// primaries and transfer functions (video-color.c:307-391, 498-722; video-converter.c:940-964, 1068-1110)
// ------------------------------------------------------------------------------------------------
struct PrimariesInfo { double Wx, Wy, Rx, Ry, Gx, Gy, Bx, By; };

static const PrimariesInfo *primaries_info (int primaries)
{
  /* color_primaries[] (video-color.c:312-336); WP_CENTRE is (1/3), (1/3) there, i.e. integer 0, 0 */
  static const PrimariesInfo t[] = {
    {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0},
    {0.31271, 0.32902, 0.64, 0.33, 0.30, 0.60, 0.15, 0.06},
    {0.31006, 0.31616, 0.67, 0.33, 0.21, 0.71, 0.14, 0.08},
    {0.31271, 0.32902, 0.64, 0.33, 0.29, 0.60, 0.15, 0.06},
    {0.31271, 0.32902, 0.63, 0.34, 0.31, 0.595, 0.155, 0.07},
    {0.31271, 0.32902, 0.63, 0.34, 0.31, 0.595, 0.155, 0.07},
    {0.31006, 0.31616, 0.681, 0.319, 0.243, 0.692, 0.145, 0.049},
    {0.31271, 0.32902, 0.708, 0.292, 0.170, 0.797, 0.131, 0.046},
    {0.31271, 0.32902, 0.64, 0.33, 0.21, 0.71, 0.15, 0.06},
    {0, 0, 1.0, 0.0, 0.0, 1.0, 0.0, 0.0},
    {0.314, 0.351, 0.68, 0.32, 0.265, 0.69, 0.15, 0.06},
    {0.31271, 0.32902, 0.68, 0.32, 0.265, 0.69, 0.15, 0.06},
    {0.31271, 0.32902, 0.63, 0.34, 0.295, 0.605, 0.155, 0.077},
  };
  return primaries >= 0 && primaries < (int) (sizeof (t) / sizeof (t[0])) ? &t[primaries] : &t[0];
}

/* gst_video_color_primaries_is_equivalent (video-color.c:368-386) */
static bool primaries_equivalent (int a, int b)
{
  if (a == b)
    return true;
  return (a == 4 || a == 5) && (b == 4 || b == 5);      /* smpte170m / smpte240m */
}

/* map_equivalent_transfer + gst_video_transfer_function_is_equivalent (video-color.c:996-1042) */
static int transfer_map (int func, unsigned bpp)
{
  if (func == 11 && bpp >= 12)          /* BT2020_12 at 12 bits and more stays itself */
    return func;
  if (func == 11 || func == 5 || func == 16 || func == 13)
    return 5;                           /* BT709 */
  return func;
}

static bool transfer_equivalent (int from, unsigned from_bpp, int to, unsigned to_bpp)
{
  from = transfer_map (from, from_bpp);
  to = transfer_map (to, to_bpp);
  if (from == 11 && to_bpp < 12 && to == 5)
    return true;
  return from == to;
}

/* gst_video_transfer_function_encode (video-color.c:498-591): the expressions as they stand there */
static double transfer_encode (int func, double val)
{
  switch (func) {
    case 2: return pow (val, 1.0 / 1.8);
    case 3: return pow (val, 1.0 / 2.0);
    case 4: return pow (val, 1.0 / 2.2);
    case 16: case 5: case 13: return val < 0.018 ? 4.5 * val : 1.099 * pow (val, 0.45) - 0.099;
    case 6: return val < 0.0228 ? val * 4.0 : 1.1115 * pow (val, 0.45) - 0.1115;
    case 7: return val <= 0.0031308 ? 12.92 * val : 1.055 * pow (val, 1.0 / 2.4) - 0.055;
    case 8: return pow (val, 1 / 2.8);
    case 9: return val < 0.01 ? 0.0 : 1.0 + log10 (val) / 2.0;
    case 10: return val < 0.0031622777 ? 0.0 : 1.0 + log10 (val) / 2.5;
    case 11: return val < 0.0181 ? 4.5 * val : 1.0993 * pow (val, 0.45) - 0.0993;
    case 12: return pow (val, 1.0 / 2.19921875);
    case 14: {
      const double c1 = 0.8359375, c2 = 18.8515625, c3 = 18.6875, m1 = 0.1593017578125, m2 = 78.84375;
      const double Ln = pow (val, m1);
      return pow ((c1 + c2 * Ln) / (1.0 + c3 * Ln), m2);
    }
    case 15: {
      const double a = 0.17883277, b = 0.28466892, c = 0.55991073;
      return val > (1.0 / 12.0) ? a * log (12.0 * val - b) + c : sqrt (3.0 * val);
    }
    default: return val;
  }
}

/* gst_video_transfer_function_decode (video-color.c:630-722) */
static double transfer_decode (int func, double val)
{
  switch (func) {
    case 2: return pow (val, 1.8);
    case 3: return pow (val, 2.0);
    case 4: return pow (val, 2.2);
    case 16: case 5: case 13: return val < 0.081 ? val / 4.5 : pow ((val + 0.099) / 1.099, 1.0 / 0.45);
    case 6: return val < 0.0913 ? val / 4.0 : pow ((val + 0.1115) / 1.1115, 1.0 / 0.45);
    case 7: return val <= 0.04045 ? val / 12.92 : pow ((val + 0.055) / 1.055, 2.4);
    case 8: return pow (val, 2.8);
    case 9: return val == 0.0 ? 0.0 : pow (10.0, 2.0 * (val - 1.0));
    case 10: return val == 0.0 ? 0.0 : pow (10.0, 2.5 * (val - 1.0));
    case 11: return val < 0.08145 ? val / 4.5 : pow ((val + 0.0993) / 1.0993, 1.0 / 0.45);
    case 12: return pow (val, 2.19921875);
    case 14: {
      const double c1 = 0.8359375, c2 = 18.8515625, c3 = 18.6875, m1 = 0.1593017578125, m2 = 78.84375;
      const double tmp = pow (val, 1 / m2);
      const double tmp2 = tmp - c1 > 0.0f ? tmp - c1 : 0.0f;
      return pow (tmp2 / (c2 - c3 * tmp), 1 / m1);
    }
    case 15: {
      const double a = 0.17883277, b = 0.28466892, c = 0.55991073;
      return val > 0.5 ? (exp ((val - c) / a) + b) / 12.0 : val * val / 3.0;
    }
    default: return val;
  }
}

/* color_matrix_invert (video-converter.c:940-964): adjugate / determinant of the 3 x 3 part, the rest identity */
static void m_invert (M44 d, M44 s)
{
  M44 tmp;
  m_identity (tmp);
  for (int j = 0; j < 3; j++)
    for (int i = 0; i < 3; i++)
      tmp[j][i] = s[(i + 1) % 3][(j + 1) % 3] * s[(i + 2) % 3][(j + 2) % 3] - s[(i + 1) % 3][(j + 2) % 3] * s[(i + 2) % 3][(j + 1) % 3];
  const double det = tmp[0][0] * s[0][0] + tmp[0][1] * s[1][0] + tmp[0][2] * s[2][0];
  for (int j = 0; j < 3; j++)
    for (int i = 0; i < 3; i++)
      tmp[i][j] /= det;
  memcpy (d, tmp, sizeof (M44));
}

/* color_matrix_RGB_to_XYZ (video-converter.c:1068-1110) */
static void m_RGB_to_XYZ (M44 dst, const PrimariesInfo &pi)
{
  M44 m, im;
  m_identity (m);
  m[0][0] = pi.Rx; m[1][0] = pi.Ry; m[2][0] = (1.0 - pi.Rx - pi.Ry);
  m[0][1] = pi.Gx; m[1][1] = pi.Gy; m[2][1] = (1.0 - pi.Gx - pi.Gy);
  m[0][2] = pi.Bx; m[1][2] = pi.By; m[2][2] = (1.0 - pi.Bx - pi.By);
  m_invert (im, m);
  const double wx = pi.Wx / pi.Wy, wy = 1.0, wz = (1.0 - pi.Wx - pi.Wy) / pi.Wy;
  const double sx = im[0][0] * wx + im[0][1] * wy + im[0][2] * wz;
  const double sy = im[1][0] * wx + im[1][1] * wy + im[1][2] * wz;
  const double sz = im[2][0] * wx + im[2][1] * wy + im[2][2] * wz;
  for (int r = 0; r < 3; r++) {
    m[r][0] *= sx;
    m[r][1] *= sy;
    m[r][2] *= sz;
  }
  memcpy (dst, m, sizeof (M44));
}

/* chain_convert :1752-1800: RGB_input -> XYZ -> RGB_output as one matrix */
static void primaries_matrix (int in_primaries, int out_primaries, M44 dm)
{
  M44 p1, p2;
  m_identity (dm);
  m_RGB_to_XYZ (p1, *primaries_info (in_primaries));
  m_multiply (dm, dm, p1);
  m_RGB_to_XYZ (p2, *primaries_info (out_primaries));
  m_invert (p2, p2);
  m_multiply (dm, p2, dm);
}

static bool m_is_identity (M44 m)
{
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      if (m[i][j] != (i == j ? 1.0 : 0.0))
        return false;
  return true;
}

// ------------------------------------------------------------------------------------------------
// resampler taps (video-resampler.c:144-429) and their int16 quantisation (video-scaler.c:339-449)
// ------------------------------------------------------------------------------------------------
struct TapParams {
  int method;
  double b, c;
  double ex, fx, dx;
  double envelope, sharpness, sharpen;
};

static double sinc (double x)
{
  if (x == 0)
    return 1;
  return sin (M_PI * x) / (M_PI * x);
}

static double envelope_fn (double x)
{
  if (x <= -1 || x >= 1)
    return 0;
  return sinc (x);
}

static double get_tap (const TapParams *p, int l, int xi, double x)
{
  int xl = xi + l;
  switch (p->method) {
    case GSTAMD_RESAMPLER_METHOD_NEAREST:
      return 1.0;
    case GSTAMD_RESAMPLER_METHOD_LINEAR: {
      double a = fabs (x - xl) * p->fx;
      return a < 1.0 ? 1.0 - a : 0.0;
    }
    case GSTAMD_RESAMPLER_METHOD_CUBIC: {
      double a = fabs (x - xl) * p->fx, a2 = a * a, a3 = a2 * a, b = p->b, c = p->c;
      if (a <= 1.0)
        return ((12.0 - 9.0 * b - 6.0 * c) * a3 + (-18.0 + 12.0 * b + 6.0 * c) * a2 + (6.0 - 2.0 * b)) / 6.0;
      else if (a <= 2.0)
        return ((-b - 6.0 * c) * a3 + (6.0 * b + 30.0 * c) * a2 + (-12.0 * b - 48.0 * c) * a +
            (8.0 * b + 24.0 * c)) / 6.0;
      return 0.0;
    }
    case GSTAMD_RESAMPLER_METHOD_SINC:
      return sinc ((x - xl) * p->fx);
    case GSTAMD_RESAMPLER_METHOD_LANCZOS: {
      double env = envelope_fn ((x - xl) * p->ex);
      return (sinc ((x - xl) * p->fx) - p->sharpen) * env;
    }
  }
  return 0.0;
}

/* gst_video_resampler_init + resampler_calculate_taps; returns max_taps, fills offset + double taps */
static int resampler_init (int method, unsigned n_taps, const GstAmdVideoConverterConfig &cfg, int in_size,
    int out_size, std::vector<uint32_t> &offset, std::vector<double> &taps, double shift_arg = 0.0, bool half_taps = false)
{
  TapParams params;
  memset (&params, 0, sizeof (params));
  params.method = method;
  params.sharpness = cfg.sharpness;
  params.sharpen = cfg.sharpen;
  double scale_factor = in_size / (double) out_size;
  if (scale_factor > 1.0)
    params.fx = (1.0 / scale_factor) * params.sharpness;
  else
    params.fx = (1.0) * params.sharpness;
  int max_taps_opt = cfg.max_taps;
  n_taps = std::min<unsigned> (n_taps, (unsigned) max_taps_opt);
  switch (method) {
    case GSTAMD_RESAMPLER_METHOD_NEAREST:
      params.envelope = cfg.envelope;
      if (n_taps == 0)
        n_taps = 1;
      break;
    case GSTAMD_RESAMPLER_METHOD_LINEAR:
      params.envelope = 1.0;
      break;
    case GSTAMD_RESAMPLER_METHOD_CUBIC:
      params.b = cfg.cubic_b;
      params.c = cfg.cubic_c;
      params.envelope = 2.0;
      break;
    case GSTAMD_RESAMPLER_METHOD_SINC:
    case GSTAMD_RESAMPLER_METHOD_LANCZOS:
      params.envelope = cfg.envelope;
      break;
    default:
      break;
  }
  if (n_taps == 0) {
    params.dx = ceil (2.0 * params.envelope / params.fx);
    double v = params.dx;
    if (v < 0)
      v = 0;
    if (v > max_taps_opt)
      v = max_taps_opt;
    n_taps = (unsigned) v;            /* CLAMP (params.dx, 0, max_taps) assigned to guint */
  }
  if (half_taps && n_taps > 3)        /* GST_VIDEO_RESAMPLER_FLAG_HALF_TAPS (video-resampler.c:414): the top field's resampler of an interlaced scaler */
    n_taps /= 2;
  params.fx = 2.0 * params.envelope / n_taps;
  params.ex = 2.0 / n_taps;
  if (n_taps > (unsigned) in_size)
    n_taps = in_size;
  int max_taps = (int) n_taps;

  int tap_offs = (max_taps - 1) / 2;
  double corr = (max_taps == 1 ? 0.0 : 0.5);
  double shift = shift_arg;
  offset.assign (out_size, 0);
  taps.assign ((size_t) out_size * max_taps, 0.0);
  for (int j = 0; j < out_size; j++) {
    double ox = (0.5 + (double) j - shift) / out_size;
    double x = ox * (double) in_size - corr;
    if (x < 0)
      x = 0;
    else if (x > in_size - 1)
      x = in_size - 1;
    int xi = (int) floor (x - tap_offs);
    uint32_t off = (uint32_t) xi;
    double weight = 0;
    double *t = &taps[(size_t) j * max_taps];
    for (int l = 0; l < max_taps; l++) {
      t[l] = get_tap (&params, l, xi, x);
      weight += t[l];
    }
    for (int l = 0; l < max_taps; l++)
      t[l] /= weight;
    if (xi < 0) {
      int sh = -xi, l;
      for (l = 0; l < sh; l++)
        t[sh] += t[l];
      for (l = 0; l < max_taps - sh; l++)
        t[l] = t[sh + l];
      for (; l < max_taps; l++)
        t[l] = 0;
      off += sh;
    }
    if (xi > in_size - max_taps) {
      int sh = xi - (in_size - max_taps), l;
      for (l = 0; l < sh; l++)
        t[max_taps - sh - 1] += t[max_taps - sh + l];
      for (l = 0; l < max_taps - sh; l++)
        t[max_taps - 1 - l] = t[max_taps - 1 - sh - l];
      for (l = 0; l < sh; l++)
        t[l] = 0;
      off -= sh;
    }
    offset[j] = off;
  }
  return max_taps;
}

/* resampler_convert_coeff (video-scaler.c:339-388) */
static void convert_coeff (const double *src, int16_t *dest, int n, int precision)
{
  double multiplier = (1 << precision);
  double l_offset = 0.0, h_offset = 1.0, offset = 0.5;
  for (int i = 0; i < 64; i++) {
    int sum = 0;
    for (int j = 0; j < n; j++) {
      int16_t tap = (int16_t) floor (offset + src[j] * multiplier);
      dest[j] = tap;
      sum += tap;
    }
    if (sum == (1 << precision))
      break;
    if (l_offset == h_offset)
      break;
    if (sum < (1 << precision)) {
      if (offset > l_offset)
        l_offset = offset;
      offset += (h_offset - l_offset) / 2;
    } else {
      if (offset < h_offset)
        h_offset = offset;
      offset -= (h_offset - l_offset) / 2;
    }
  }
}

/* One gst_video_scaler_new + the function the reference would pick in get_functions
 * (video-scaler.c:1202-1342) for 4x8-bit pixels. */
static bool finish_scale_pass (int max_taps, const std::vector<double> &dtaps, int in_size, int out_size, bool horizontal, ScalePass *pass, bool h2_as_ntap, bool deep16);

bool make_scale_pass (int method, unsigned n_taps_opt, const GstAmdVideoConverterConfig &cfg, int in_size,
    int out_size, bool horizontal, ScalePass *pass, bool h2_as_ntap, bool deep16)
{
  std::vector<double> dtaps;
  int max_taps = resampler_init (method, n_taps_opt, cfg, in_size, out_size, pass->offset, dtaps);
  return finish_scale_pass (max_taps, dtaps, in_size, out_size, horizontal, pass, h2_as_ntap, deep16);
}

/* The vertical scaler of an INTERLACED frame (gst_video_scaler_new with GST_VIDEO_SCALER_FLAG_INTERLACED, video-scaler.c:229-249): one resampler per
 * field - top: (in + 1) / 2 -> (out + 1) / 2 lines, shifted by 0.5 * out / in, HALF_TAPS; bottom: the rest, the top's tap count, shifted the other way -
 * zipped line by line (resampler_zip :109-146: output i takes field i & 1, offset * 2 + (i & 1), taps over every OTHER line).  `field` 0 / 1: that
 * field's resampler as a pass over the field's own lines.  *zip_last: the zipped scaler's offset of the frame's last output line (gst_video_scaler_2d's
 * order rule); *zip_offsets (frame lines, for the line-cache simulation) when asked for.  false: the reference cannot make this scaler (a field
 * without lines, or the two resamplers disagreeing on their tap count: g_return_if_fail (r1->max_taps == r2->max_taps)). */
static bool make_field_vpass (int method, unsigned n_taps_opt, const GstAmdVideoConverterConfig &cfg, int frame_in, int frame_out, int field,
    ScalePass *pass, bool deep16, int *zip_last, std::vector<uint32_t> *zip_offsets)
{
  const int tin = (frame_in + 1) / 2, tout = (frame_out + 1) / 2, bin = frame_in - tin, bout = frame_out - tout;
  if (tin <= 0 || tout <= 0 || bin <= 0 || bout <= 0)
    return false;
  const double shift = (0.5 * frame_out) / frame_in;
  std::vector<uint32_t> toff, boff;
  std::vector<double> ttaps, btaps;
  const int tmax = resampler_init (method, n_taps_opt, cfg, tin, tout, toff, ttaps, shift, true);
  const int bmax = resampler_init (method, (unsigned) tmax, cfg, bin, bout, boff, btaps, -shift, false);
  if (tmax != bmax)
    return false;
  if (zip_last) {
    const int i = frame_out - 1;
    *zip_last = (int) ((i & 1) ? boff[(size_t) (i / 2)] : toff[(size_t) (i / 2)]) * 2 + (i & 1);
  }
  if (zip_offsets) {
    zip_offsets->assign ((size_t) frame_out, 0);
    for (int i = 0; i < frame_out; i++)
      (*zip_offsets)[(size_t) i] = ((i & 1) ? boff[(size_t) (i / 2)] : toff[(size_t) (i / 2)]) * 2 + (uint32_t) (i & 1);
  }
  if (pass) {
    pass->offset = field ? boff : toff;
    return finish_scale_pass (tmax, field ? btaps : ttaps, field ? bin : tin, field ? bout : tout, false, pass, false, deep16);
  }
  return true;
}

static bool finish_scale_pass (int max_taps, const std::vector<double> &dtaps, int in_size, int out_size, bool horizontal, ScalePass *pass, bool h2_as_ntap, bool deep16)
{
  pass->horizontal = horizontal;
  pass->in_size = in_size;
  pass->out_size = out_size;
  pass->n_taps = max_taps;
  pass->inc = out_size == 1 ? 0 : (int) ((((unsigned) (in_size - 1)) << 16) / (unsigned) (out_size - 1)) - 1;
  pass->precision = 0;
  pass->taps.clear ();
  pass->dot4_ok = false;
  pass->nw = pass->nw4 = 0;
  pass->tapw.clear ();
  if (max_taps == 1) {
    pass->kind = SCALE_NEAREST;               /* h_near_u32 / v_near: copies s[offset[i]] */
    return true;
  }
  if (deep16) {
    /* 16-bit lines (get_functions, video-scaler.c:1314-1340): h_ntap_u16 (its 2-tap branch rounds with + 4096, the others with
     * + 4095), v_2tap_u16, v_ntap_u16 - all with taps at SCALE_U16 = 12 fractional bits */
    pass->kind = max_taps == 2 ? SCALE_2TAP : SCALE_NTAP;
    pass->precision = 12;
    pass->taps.assign ((size_t) out_size * max_taps, 0);
    for (int i = 0; i < out_size; i++)
      convert_coeff (&dtaps[(size_t) i * max_taps], &pass->taps[(size_t) i * max_taps], max_taps, pass->precision);
    return true;
  }
  if (max_taps == 2 && horizontal && !h2_as_ntap) {
    pass->kind = SCALE_2TAP;                  /* video_scale_h_2tap_4u8 -> ldreslinl with scale->inc */
    return true;
  }
  if (max_taps == 2 && !horizontal) {
    pass->kind = SCALE_2TAP;                  /* video_scale_v_2tap_u8: taps at SCALE_U8_LQ + 2 bits */
    pass->precision = 8;
  } else {
    pass->kind = SCALE_NTAP;                  /* h_ntap_u8 / v_4tap_u8 / v_ntap_u8 at SCALE_U8_LQ bits */
    pass->precision = 6;
  }
  pass->taps.assign ((size_t) out_size * max_taps, 0);
  for (int i = 0; i < out_size; i++)
    convert_coeff (&dtaps[(size_t) i * max_taps], &pass->taps[(size_t) i * max_taps], max_taps, pass->precision);
  /* byte-dot-product form of a horizontal N-tap pass: the filter window is read from a 4-byte aligned LDS address,
   * so the taps are shifted right by (offset & 3) places inside nw zero-padded words (a zero tap adds nothing to
   * the 16-bit wrapping sum, so the result is unchanged) */
  if (horizontal && pass->kind == SCALE_NTAP) {
    bool ok = true;
    for (int i = 0; i < out_size && ok; i++) {
      int sum = 0;
      for (int l = 0; l < max_taps; l++) {
        const int t = pass->taps[(size_t) i * max_taps + l];
        ok = ok && t >= -128 && t <= 127;
        sum += t;
      }
      ok = ok && sum == (1 << pass->precision);
    }
    if (ok) {
      pass->dot4_ok = true;
      pass->nw = (max_taps + 3 + 3) / 4;
      pass->nw4 = (pass->nw + 3) & ~3;
      pass->tapw.assign ((size_t) out_size * pass->nw4, 0);
      for (int i = 0; i < out_size; i++) {
        const int shift = (int) (pass->offset[i] & 3);
        for (int l = 0; l < max_taps; l++) {
          const int j = l + shift;
          const uint32_t b = (uint32_t) (uint8_t) (int8_t) pass->taps[(size_t) i * max_taps + l];
          pass->tapw[(size_t) i * pass->nw4 + (j >> 2)] |= b << (8 * (j & 3));
        }
      }
    }
  }
  return true;
}

bool make_fused420_tables (const ScalePass &v, int height, Fused420Tables *t)
{
  if (v.horizontal || v.kind != SCALE_NTAP || v.precision != 6 || v.n_taps < 2)
    return false;
  const int n = v.n_taps, out_h = v.out_size;
  t->ngv = (n + 3 + 3) / 4;
  t->n_groups = (height / 2 + 2) / 2;
  t->vgroup.assign (out_h, 0);
  t->vtapw.assign ((size_t) out_h * t->ngv, 0);
  for (int j = 0; j < out_h; j++) {
    const int off = (int) v.offset[j];
    if (off < 0 || off + n > height || (j > 0 && off < (int) v.offset[j - 1]))
      return false;
    int sum = 0;
    const int g = (off + 1) >> 2, shift = (off + 1) & 3;
    t->vgroup[j] = g;
    for (int l = 0; l < n; l++) {
      const int tap = v.taps[(size_t) j * n + l];
      if (tap < -128 || tap > 127)
        return false;
      sum += tap;
      const int k = l + shift;
      t->vtapw[(size_t) j * t->ngv + (k >> 2)] |= (uint32_t) (uint8_t) (int8_t) tap << (8 * (k & 3));
    }
    if (sum != 64)
      return false;
  }
  return true;
}

// ---- column-walk scaler (video_scale_col.h) ------------------------------------------------------------------------------------------
namespace {
struct ColWin { int lo, hi, sum; };         // first / last source index with a nonzero tap, sum of the taps

// nonzero extent of every output's window; false: a tap outside int8, a sum outside [64, 128] (below 64 the alpha byte 0xff would
// not survive the pass: (255 s + 32) >> 6 < 255; above 128 the 16-bit sum of the reference wraps for alpha), an extent that runs backwards
bool col_windows (const ScalePass &p, int in_size, std::vector<ColWin> *w)
{
  const int n = p.n_taps, out = p.out_size;
  w->assign ((size_t) out, ColWin ());
  for (int i = 0; i < out; i++) {
    int f = -1, l = -1, sum = 0;
    for (int k = 0; k < n; k++) {
      const int t = p.taps[(size_t) i * n + k];
      if (t < -128 || t > 127)
        return false;
      sum += t;
      if (t != 0) {
        if (f < 0)
          f = k;
        l = k;
      }
    }
    if (f < 0 || sum < 64 || sum > 128)
      return false;
    ColWin &cw = (*w)[(size_t) i];
    cw.lo = (int) p.offset[(size_t) i] + f;
    cw.hi = (int) p.offset[(size_t) i] + l;
    cw.sum = sum;
    if (cw.lo < 0 || cw.hi >= in_size || (i > 0 && (cw.lo < (*w)[(size_t) i - 1].lo || cw.hi < (*w)[(size_t) i - 1].hi)))
      return false;
  }
  return true;
}

int col_mod (int a, int m) { const int r = a % m; return r < 0 ? r + m : r; }

struct ColHTry {
  bool ok = false;
  int nw = 0, wstep = -1, a8 = 0, words = 0;
  std::vector<int32_t> tiles;
  std::vector<int> wbase;         // per output (shared mode: the even output's base for both)
};

// greedy tiling for one alignment phase; share: the two outputs of a lane read one window (opl = 2)
ColHTry col_try_h (const std::vector<ColWin> &hw, int width, int opl, int ph, int wstep, int cap = 0)
{
  ColHTry r;
  const int out_w = (int) hw.size (), pxl = 4 * opl, span = 64 * pxl, align = opl == 2 ? 8 : 4;
  const bool share = wstep >= 0;
  r.wbase.assign ((size_t) out_w, 0);
  r.wstep = wstep;
  r.a8 = share ? 1 : 0;
  int x = 0;
  while (x < out_w) {
    int s0 = hw[(size_t) x].lo - col_mod (hw[(size_t) x].lo - ph, align);
    if (share && x + 1 < out_w) {
      /* the odd output's words start wstep words after the even one's: room for that in front of the tile's first pair */
      const int b1 = (hw[(size_t) x + 1].lo - s0) & ~3;
      if (b1 - 4 * wstep < 0)
        s0 -= align;
    }
    int p0 = s0 > 0 ? s0 : 0;
    if (p0 + span > width && col_mod (width - p0, 4) != 0) {
      /* the tile reaches the picture's right edge: reads past the last row's end are range-checked per dword, so the loads start a
         multiple of 4 pixels from the edge - no dword straddles it (the staged bytes then sit off their natural alignment in LDS, like
         the first tile's) */
      s0 -= align;
      p0 = s0 + col_mod (width - s0, 4);
      if (p0 < 0)
        return r;
    }
    if (p0 - s0 > 16)            /* a staged plane is the span + 16 bytes */
      return r;
    int n = 0;
    while (x + n < out_w && n < 64 * opl && (cap <= 0 || n < cap) && hw[(size_t) (x + n)].hi < p0 + span)
      n++;
    if (opl == 2 && x + n < out_w)
      n &= ~1;
    if (n <= 0)
      return r;
    for (int i = 0; i < n; i += share ? 2 : 1) {
      const ColWin &w0 = hw[(size_t) (x + i)];
      const int b0 = (w0.lo - s0) & ~3, e0 = (w0.hi - s0) & ~3;
      if (!share) {
        r.wbase[(size_t) (x + i)] = b0;
        r.nw = std::max (r.nw, (e0 - b0) / 4 + 1);
        continue;
      }
      int ub = b0, top = e0;
      if (x + i + 1 < out_w && i + 1 < n) {
        const ColWin &w1 = hw[(size_t) (x + i + 1)];
        const int b1 = (w1.lo - s0) & ~3, e1 = (w1.hi - s0) & ~3;
        ub = std::min (b0, b1 - 4 * wstep);
        top = std::max (e0, e1 - 4 * wstep);
        r.wbase[(size_t) (x + i + 1)] = ub;
      }
      if (ub < 0)
        return r;
      if (ub & 7)
        r.a8 = 0;
      r.wbase[(size_t) (x + i)] = ub;
      r.nw = std::max (r.nw, (top - ub) / 4 + 1);
    }
    r.tiles.push_back (x);
    r.tiles.push_back (n);
    r.tiles.push_back (s0);
    r.tiles.push_back (p0);
    x += n;
  }
  r.words = share ? r.nw + wstep : opl * r.nw;      /* LDS words a lane reads per line and plane */
  r.ok = r.nw <= 6;
  return r;
}
}  // namespace

bool make_col_tables (const ScalePass &h, const ScalePass &v, int width, int height, int opl, bool share, ColTables *t)
{
  if (!h.horizontal || v.horizontal || h.kind != SCALE_NTAP || v.kind != SCALE_NTAP || h.precision != 6 || v.precision != 6)
    return false;
  if ((opl != 1 && opl != 2) || (width % 8) != 0 || width < 16 * opl || height < 2 || h.in_size != width || v.in_size != height)
    return false;
  const int out_w = h.out_size, out_h = v.out_size;
  if (opl == 2 && (out_w & 1))
    return false;
  std::vector<ColWin> hw, vw;
  if (!col_windows (h, width, &hw) || !col_windows (v, height, &vw))
    return false;
  /* horizontal: the alignment phase and the window form with the fewest dot products per output, then the fewest LDS words, then tiles */
  ColHTry best;
  for (int ph = 0; ph < (opl == 2 ? 8 : 4); ph += 2)
    for (int ws = -1; ws <= (opl == 2 && share ? 2 : -1); ws++) {
      ColHTry c = col_try_h (hw, width, opl, ph, ws);
      if (!c.ok)
        continue;
      /* the same number of tiles, evenly wide (the last tile of the greedy run is a stub, and every tile costs a wave column) */
      const int nt = (int) c.tiles.size () / 4, even = ((out_w + nt - 1) / nt + opl - 1) / opl * opl;
      ColHTry b = col_try_h (hw, width, opl, ph, ws, even);
      if (b.ok && b.tiles.size () == c.tiles.size () && b.nw <= c.nw)
        c = b;
      const auto key = [](const ColHTry &a) { return std::make_tuple (a.nw, a.words - (a.a8 ? 1 : 0), (int) a.tiles.size ()); };
      if (!best.ok || key (c) < key (best))
        best = c;
    }
  if (!best.ok)
    return false;
  t->opl = opl;
  t->nw = best.nw;
  t->wstep = best.wstep;
  t->a8 = best.a8;
  t->tiles = best.tiles;
  /* register windows (video_scale_col.h col_hfilter_regs): in every tile unit u's shared window starts 8 u bytes after unit 0's, and that one at
     byte 0 or 8 of the tile's window space - then lane u's window begins with the two words lane u itself holds, and the rest comes over the lanes */
  if (opl == 2 && best.wstep == 1 && best.a8 && best.nw <= 4) {
    bool rw = true;
    for (size_t ti = 0; ti < best.tiles.size () && rw; ti += 4) {
      const int o0 = best.tiles[ti], n = best.tiles[ti + 1];
      const int w0 = best.wbase[(size_t) o0];
      rw = (w0 == 0 || w0 == 8) && (n % 2) == 0;
      for (int u = 0; u < n / 2 && rw; u++)
        rw = best.wbase[(size_t) (o0 + 2 * u)] == w0 + 8 * u;
    }
    if (rw)
      t->a8 = 2;
  }
  t->hout.assign ((size_t) out_w * 8, 0);
  for (size_t ti = 0; ti < best.tiles.size (); ti += 4) {
    const int o0 = best.tiles[ti], n = best.tiles[ti + 1], s0 = best.tiles[ti + 2];
    for (int x = o0; x < o0 + n; x++) {
      uint32_t *e = &t->hout[(size_t) x * 8];
      int base = best.wbase[(size_t) x];
      if (best.wstep >= 0 && ((x - o0) & 1))
        base += 4 * best.wstep;           /* the odd output's own first word; the kernel reads from the even one's */
      e[0] = (uint32_t) best.wbase[(size_t) x];
      e[1] = (uint32_t) (128 * hw[(size_t) x].sum + 32);
      for (int l = 0; l < h.n_taps; l++) {
        const int tap = h.taps[(size_t) x * h.n_taps + l];
        if (tap == 0)
          continue;
        const int b = (int) h.offset[(size_t) x] + l - s0 - base;
        if (b < 0 || b >= 4 * best.nw)
          return false;
        e[2 + (b >> 2)] |= (uint32_t) (uint8_t) (int8_t) tap << (8 * (b & 3));
      }
    }
  }
  /* vertical: line y sits in byte (y + 1) & 3 of group (y + 1) >> 2 */
  t->n_groups = (height >> 2) + 1;
  t->ngv = 0;
  t->vrow.assign ((size_t) out_h * 8, 0);
  for (int j = 0; j < out_h; j++) {
    const int gf = (vw[(size_t) j].lo + 1) >> 2, gl = (vw[(size_t) j].hi + 1) >> 2;
    t->ngv = std::max (t->ngv, gl - gf + 1);
    if (gl - gf + 1 > 5)
      return false;
    uint32_t *e = &t->vrow[(size_t) j * 8];
    e[0] = (uint32_t) gf;
    e[1] = (uint32_t) gl;
    e[2] = (uint32_t) (128 * vw[(size_t) j].sum + 32);
    for (int l = 0; l < v.n_taps; l++) {
      const int tap = v.taps[(size_t) j * v.n_taps + l];
      if (tap == 0)
        continue;
      const int b = (int) v.offset[(size_t) j] + l + 1 - 4 * gf;
      e[3 + (b >> 2)] |= (uint32_t) (uint8_t) (int8_t) tap << (8 * (b & 3));
    }
  }
  t->ngv_aligned = 0;               /* tw[k] <-> group gfirst + k until col_align_rows */
  t->pubn = 0;
  for (int j = 1; j < out_h; j++)
    t->pubn = std::max (t->pubn, (int) t->vrow[(size_t) (j - 1) * 8 + 1] - (int) t->vrow[(size_t) j * 8] + 1);
  /* a wave owning rows [a, a + m) makes the groups [gfirst (a), gfirst (a + m)) and publishes [gfirst (a), glast (a - 1)] */
  int m = 1;
  for (; m < out_h; m++) {
    bool ok = true;
    for (int j = 1; j + m < out_h && ok; j++)
      ok = (int) t->vrow[(size_t) (j + m) * 8] > (int) t->vrow[(size_t) (j - 1) * 8 + 1];
    if (ok)
      break;
  }
  t->min_rows_per_wave = m;
  return true;
}

void col_align_rows (ColTables *t, int ngv_form)
{
  const size_t out_h = t->vrow.size () / 8;
  for (size_t j = 0; j < out_h; j++) {
    uint32_t *e = &t->vrow[j * 8];
    const int n = (int) e[1] - (int) e[0] + 1, first_now = t->ngv_aligned ? t->ngv_aligned - n : 0, first_new = ngv_form - n;
    uint32_t w[5] = {0, 0, 0, 0, 0};
    for (int k = 0; k < n; k++)
      w[first_new + k] = e[3 + first_now + k];
    for (int k = 0; k < 5; k++)
      e[3 + k] = w[k];
  }
  t->ngv_aligned = ngv_form;
}

bool col_plan_regular (const VideoPlan &p, int *crow_lo, int *crow_hi)
{
  if (p.passes.size () != 2 || !p.passes[0].horizontal || p.passes[0].kind != SCALE_NTAP || p.passes[1].horizontal || p.passes[1].kind != SCALE_NTAP)
    return false;
  if (!kind_has_planes (p.front.kind) || p.front.w_sub != 1 || p.front.h_sub != 1 || p.front.chroma_v2 != 1 || p.front.hi_depth != 0 || p.matrix_before_scale ||
      p.front.swap_k >= 0 || (int) p.vpair.size () < 2 * p.front.height)
    return false;
  const int lo = -(p.rect.in_y >> 1), hi = ((p.rect.in_maxh + 1) >> 1) - 1 - (p.rect.in_y >> 1);
  for (int y = 0; y < p.front.height; y++) {
    const int u = (y + 1) >> 1;
    const int ra = std::min (std::max (u - 1, lo), hi), rb = std::min (std::max (u, lo), hi);
    const int heavy = (y & 1) ? ra : rb, light = (y & 1) ? rb : ra;
    const int e0 = p.vpair[(size_t) 2 * y], ta = vpair_row (e0), tb = p.vpair[(size_t) 2 * y + 1];
    const int th = vpair_role (e0) == 0 ? ta : tb, tl = vpair_role (e0) == 0 ? tb : ta;
    if (th != heavy || tl != light)
      return false;
  }
  *crow_lo = lo;
  *crow_hi = hi;
  return true;
}

int fused420_ring_groups (const Fused420Tables &t, int rows_per_chunk, int nwaves, int first_rows)
{
  const int out_h = (int) t.vgroup.size ();
  int worst = 0;
  if (first_rows <= 0)
    first_rows = nwaves;
  for (int j0 = 0; j0 < out_h; j0 += rows_per_chunk) {
    const int j1 = std::min (j0 + rows_per_chunk, out_h);
    int rows = first_rows;
    for (int jr = j0; jr < j1; jr += rows, rows = nwaves) {
      const int jl = std::min (jr + rows, j1) - 1;
      const int gh = std::min (t.vgroup[jl] + t.ngv - 1, t.n_groups - 1);
      worst = std::max (worst, gh - t.vgroup[jr] + 1);
    }
  }
  return worst;
}

// schedule 2 (one barrier per round): the ring holds a round's window AND the groups of the next round
int fused420_ring_groups2 (const Fused420Tables &t, int rows_per_chunk, int nwaves, int first_rows)
{
  const int out_h = (int) t.vgroup.size ();
  int worst = 0;
  if (first_rows <= 0)
    first_rows = nwaves;
  for (int j0 = 0; j0 < out_h; j0 += rows_per_chunk) {
    const int j1 = std::min (j0 + rows_per_chunk, out_h);
    int rows = first_rows;
    for (int jr = j0; jr < j1; jr += rows, rows = nwaves) {
      const int jr2 = jr + rows;
      const int jl = std::min (jr2 < j1 ? jr2 + nwaves : jr2, j1) - 1;        /* last row of the next round (or of this one) */
      const int gh = std::min (t.vgroup[jl] + t.ngv - 1, t.n_groups - 1);
      worst = std::max (worst, gh - t.vgroup[jr] + 1);
    }
  }
  return worst;
}

int fused420_first_rows (const Fused420Tables &t, int nwaves)
{
  const int out_h = (int) t.vgroup.size ();
  /* judged in the middle of the picture (the edges fold their windows) */
  const int j0 = out_h / 2;
  int best = 1;
  for (int r = 1; r <= nwaves && j0 + r <= out_h; r++)
    if (t.vgroup[j0 + r - 1] + t.ngv - 1 - t.vgroup[j0] + 1 <= nwaves)
      best = r;
  return best;
}

// ------------------------------------------------------------------------------------------------
// line-cache simulation: which two source lines does the vertical chroma upsampler pair?
// Mirrors gst_line_cache_get_lines / _add_line (video-converter.c:571-629) and the need_line
// callbacks do_unpack_lines / do_upsample_lines / do_hscale_lines / do_vscale_lines (:2966-3094)
// on line INDICES only, for a fresh converter's first frame with one thread.
// ------------------------------------------------------------------------------------------------
namespace {
struct SimCache {
  int first = 0, len = 0, backlog = 0;
  SimCache *prev = nullptr;
  int kind = 0;    // 0 unpack, 1 upsample, 2 passthrough (hscale/convert/alpha), 3 vscale
  const ScalePass *vpass = nullptr;
  int out_height = 0;
  const std::vector<uint32_t> *zip = nullptr;      // interlaced frames: the zipped scaler's first line per output line; its windows are 2 * taps lines
  int zip_lines = 0;
};

struct Sim {
  std::vector<int32_t> *vpair;
  int in_height;
  int line_lo, line_hi;     // lines the unpacker can deliver, relative to the crop origin: do_unpack_lines (:2966) clamps to the FRAME
  int h_sub;
  int up_n_lines, up_offset;
  bool ilace = false;       // an interlaced frame: 4:2:0 lines take the chroma row of their FIELD (GET_UV_420, video-format.c:1045), and the table's
                            // second entries carry video_chroma_up_vi2's weights (vpair_pack_w8)

  void clear (SimCache *c) { c->len = 0; c->first = 0; }
  void add_line (SimCache *c, int idx)
  {
    if (c->first + c->len != idx) {
      clear (c);
      c->first = idx;
    }
    c->len++;
  }
  bool get_lines (SimCache *c, int out_line, int in_line, int n_lines)
  {
    if (c->first + c->backlog < in_line) {
      int to_remove = std::min (in_line - (c->first + c->backlog), c->len);
      if (to_remove > 0)
        c->len -= to_remove;
      c->first += to_remove;
    } else if (in_line < c->first) {
      clear (c);
      c->first = in_line;
    }
    for (int guard = 0; guard < (1 << 24); guard++) {
      if (c->first <= in_line && in_line + n_lines <= c->first + c->len)
        return true;
      if (c->len == 0 && c->first + c->backlog < in_line)
        c->first = in_line - c->backlog;
      int oline = out_line + c->first + c->len - in_line;
      if (!need_line (c, oline, c->first + c->len))
        break;
    }
    return false;
  }
  int chroma_row (int line) const
  {
    int cl = std::min (std::max (line, line_lo), line_hi);
    if (ilace && h_sub == 1)
      return ((cl & ~2) >> 1) | (cl & 1);         /* GET_UV_420 with GST_VIDEO_PACK_FLAG_INTERLACED */
    return cl >> h_sub;                   /* arithmetic: line -1 -> row -1 */
  }
  bool need_line (SimCache *c, int out_line, int in_line)
  {
    switch (c->kind) {
      case 0:
        add_line (c, in_line);
        return true;
      case 1: {
        int n_lines = up_n_lines, start_line = in_line;
        if (start_line < n_lines + up_offset) {
          start_line += up_offset;
          out_line += up_offset;
        }
        if (!get_lines (c->prev, out_line, start_line, n_lines))
          return false;
        if (n_lines == 4) {
          /* video_chroma_up_vi2_u8 (video-chroma.c:347-384) over lines[0 .. 3]: lines 0 and 2 blend the rows of lines 0 and 2 (5:3, 1:7), lines 1 and 3
             those of lines 1 and 3 (7:1, 3:5) */
          static const int wa[4] = {5, 7, 1, 3};
          for (int i = 0; i < 4; i++) {
            const int line = start_line + i;
            if (line >= 0 && line < in_height) {
              (*vpair)[2 * line + 0] = vpair_pack (chroma_row (start_line + (i & 1)), 0);
              (*vpair)[2 * line + 1] = vpair_pack_w8 (chroma_row (start_line + (i & 1) + 2), wa[i]);
            }
          }
        }
        if (n_lines == 2) {
          int ra = chroma_row (start_line), rb = chroma_row (start_line + 1);
          for (int i = 0; i < 2; i++) {
            int line = start_line + i;
            if (line >= 0 && line < in_height) {
              (*vpair)[2 * line + 0] = vpair_pack (ra, i);
              (*vpair)[2 * line + 1] = rb;
            }
          }
        }
        for (int i = 0; i < n_lines; i++)
          add_line (c, start_line + i);
        return true;
      }
      case 2:
        if (!get_lines (c->prev, out_line, in_line, 1))
          return false;
        add_line (c, in_line);
        return true;
      case 3: {
        int cline = std::min (std::max (in_line, 0), c->out_height - 1);
        int sline = c->zip ? (int) (*c->zip)[(size_t) cline] : (int) c->vpass->offset[cline], n = c->zip ? c->zip_lines : c->vpass->n_taps;
        if (!get_lines (c->prev, out_line, sline, n))
          return false;
        add_line (c, in_line);
        return true;
      }
    }
    return false;
  }
};
}  // namespace

/* extra: the chain is also asked for the line PAST the picture (the vertical chroma downsampler's last pair of an odd-height 4:2:0
 * destination, do_downsample_lines :3192): one more table entry, for source line H */
static void simulate_vpairs (VideoPlan *plan, int out_height, bool extra = false)
{
  const int H = plan->front.height, HT = H + (extra ? 1 : 0);
  plan->vpair.assign ((size_t) HT * 2, 0);
  for (int y = 0; y < HT; y++) {
    int r = (y < H ? y : H - 1) >> plan->front.h_sub;
    plan->vpair[2 * y] = r;
    plan->vpair[2 * y + 1] = r;
  }
  if (!plan->front.chroma_v2)
    return;
  Sim sim;
  sim.vpair = &plan->vpair;
  sim.in_height = HT;
  sim.line_lo = -plan->rect.in_y;
  sim.line_hi = plan->rect.in_maxh - 1 - plan->rect.in_y;
  sim.h_sub = plan->front.h_sub;
  sim.up_n_lines = 2;
  sim.up_offset = -1;
  std::vector<SimCache> caches (2 + plan->passes.size ());
  caches[0].kind = 0;
  caches[1].kind = 1;
  caches[1].prev = &caches[0];
  SimCache *prev = &caches[1];
  size_t ci = 2;
  for (const ScalePass &p : plan->passes) {
    SimCache *c = &caches[ci++];
    c->prev = prev;
    if (p.horizontal) {
      c->kind = 2;
    } else {
      c->kind = 3;
      c->vpass = &p;
      c->out_height = p.out_size;
      prev->backlog = 0;     /* chain_vscale: prev->backlog = taps_i (interlaced only) */
    }
    prev = c;
  }
  /* convert_generic_task: one thread, lines 0 .. out_height-1, one at a time */
  for (int i = 0; i < out_height; i++)
    sim.get_lines (prev, i, i, 1);
  if (extra)
    sim.get_lines (prev, out_height, out_height, 1);
}

/* The pair table of ONE FIELD of an interlaced frame: the frame's chain simulated as the reference runs it on a frame with
 * GST_VIDEO_FRAME_FLAG_INTERLACED (video_converter_generic :3303-3312: upsample_i / v_scaler_i) - the chroma upsampler in groups of four lines from
 * the line it is asked for (video_chroma_up_vi2: n_lines 4, offset -2; none when the site is vertically cosited or the upsampler is off), the
 * vertical scaler asking for 2 * taps lines from its zipped offsets with that many lines of backlog (chain_vscale :1651-1660, 1673) - then the
 * entries of the field's lines, in rows of the FRAME's chroma planes.  up_v: the vertical upsampler runs. */
static void simulate_vpairs_field (VideoPlan *plan, int frame_in_h, int frame_out_h, int field, bool up_v, const std::vector<uint32_t> *zip, int zip_lines,
    bool dest_rows)
{
  const int H = frame_in_h;
  std::vector<int32_t> tab ((size_t) H * 2, 0);
  Sim sim;
  sim.vpair = &tab;
  sim.in_height = H;
  sim.line_lo = 0;
  sim.line_hi = H - 1;
  sim.h_sub = plan->front.h_sub;
  sim.ilace = true;
  sim.up_n_lines = up_v ? 4 : 1;
  sim.up_offset = up_v ? -2 : 0;
  for (int y = 0; y < H; y++) {
    const int r = sim.chroma_row (y);
    tab[(size_t) 2 * y] = vpair_pack (r, 0);
    tab[(size_t) 2 * y + 1] = vpair_pack_w8 (r, 4);
  }
  if (up_v && zip) {
    /* behind a vertical scaler the reference's own pairing is not observable (its windows read aliased lines: plan_video_converter's note): the groups
       the chain makes when its lines are asked for in frame order - what the chain run stage by stage makes (tests/staged.py staged_expected_interlaced) */
    (void) zip_lines;
    std::vector<SimCache> caches (2);
    caches[0].kind = 0;
    caches[1].kind = 1;
    caches[1].prev = &caches[0];
    for (int i = 0; i < H; i++)
      sim.get_lines (&caches[1], i, i, 1);
  } else if (up_v) {
    std::vector<SimCache> caches (2 + plan->passes.size ());
    caches[0].kind = 0;
    caches[1].kind = 1;
    caches[1].prev = &caches[0];
    SimCache *prev = &caches[1];
    size_t ci = 2;
    for (const ScalePass &p : plan->passes) {          /* (horizontal passes only) */
      SimCache *c = &caches[ci++];
      c->prev = prev;
      c->kind = 2;
      prev = c;
    }
    for (int i = 0; i < frame_out_h; i++)
      sim.get_lines (prev, i, i, 1);
  }
  if (up_v && dest_rows) {
    /* the chain's lines are the destination frame's own rows (a destination in its unpack format, nothing between the unpacker and the packer that makes
       lines of its own: get_dest_line :2925 clamps the row): the lines above the picture of the first group (-2, -1) ARE row 0 and the lines below it of
       the last group are the last row, and video_chroma_up_vi2 filters a group only `if (l0 != l1 && l2 != l3)` (video-chroma.c:368) - the two picture
       lines of those groups keep the chroma row they were unpacked with */
    for (int y = 0; y < H; y++) {
      const int start = ((y + 2) & ~3) - 2;          /* the group (4 m - 2 .. 4 m + 1) of line y, lines asked for in order */
      if (start < 0 || start + 3 > H - 1) {
        const int r = sim.chroma_row (y);
        tab[(size_t) 2 * y] = vpair_pack (r, 0);
        tab[(size_t) 2 * y + 1] = vpair_pack_w8 (r, 4);
      }
    }
  }
  const int FH = plan->front.height;
  plan->vpair.assign ((size_t) FH * 2, 0);
  for (int k = 0; k < FH; k++) {
    const int y = std::min (2 * k + field, H - 1);
    plan->vpair[(size_t) 2 * k] = tab[(size_t) 2 * y];
    plan->vpair[(size_t) 2 * k + 1] = tab[(size_t) 2 * y + 1];
  }
}

// ------------------------------------------------------------------------------------------------
// fastpath detection (video-converter.c:8907-9017 + transforms[] :8413-8905), restricted to the
// formats this library knows.  Returns a short name of the reference fastpath or nullptr.
// ------------------------------------------------------------------------------------------------
// The reference's transforms[] (video-converter.c:8413-8905) restricted to the formats of this library, one row per
// conversion function: which format pairs it serves and the conditions video_converter_lookup_fastpath (:8907-9010) tests.
namespace {
enum : unsigned {
  FP_MATRIX = 1,      /* needs_color_matrix: usable when the matrices differ */
  FP_SIZE = 2,        /* keeps_size: only when the FULL input size equals the destination rectangle */
  FP_CROP = 4,        /* do_crop */
  FP_BORDER = 8,      /* do_border */
  FP_ACOPY = 16, FP_ASET = 32, FP_AMULT = 64,
  FP_WEVEN = 128, FP_HEVEN = 256,   /* width_align / height_align == 1 */
  FP_ILACE = 512,     /* keeps_interlaced: usable on interlaced infos (the convert_scale_planes rows: see lookup_fastpath) */
};
/* one bit per format, the enum value (128 bits: every format transforms[] names lies below 128; the later ones - GBR_16LE, RBGA, Y216_LE, Y416_LE -
   have no rows and no bit) */
typedef unsigned __int128 fmask;
constexpr fmask fbit (int f) { return f >= 0 && f < 128 ? (fmask) 1 << f : (fmask) 0; }
constexpr fmask F_I420 = fbit (GSTAMD_VIDEO_FORMAT_I420), F_YV12 = fbit (GSTAMD_VIDEO_FORMAT_YV12), F_420 = F_I420 | F_YV12;
constexpr fmask F_Y42B = fbit (GSTAMD_VIDEO_FORMAT_Y42B), F_Y444 = fbit (GSTAMD_VIDEO_FORMAT_Y444);
constexpr fmask F_AYUV = fbit (GSTAMD_VIDEO_FORMAT_AYUV), F_YUY2 = fbit (GSTAMD_VIDEO_FORMAT_YUY2), F_UYVY = fbit (GSTAMD_VIDEO_FORMAT_UYVY);
constexpr fmask F_RGB4X = fbit (GSTAMD_VIDEO_FORMAT_RGBx) | fbit (GSTAMD_VIDEO_FORMAT_BGRx) | fbit (GSTAMD_VIDEO_FORMAT_xRGB) | fbit (GSTAMD_VIDEO_FORMAT_xBGR);
constexpr fmask F_RGB4A = fbit (GSTAMD_VIDEO_FORMAT_RGBA) | fbit (GSTAMD_VIDEO_FORMAT_BGRA) | fbit (GSTAMD_VIDEO_FORMAT_ARGB) | fbit (GSTAMD_VIDEO_FORMAT_ABGR);
constexpr fmask F_RGB3 = fbit (GSTAMD_VIDEO_FORMAT_RGB) | fbit (GSTAMD_VIDEO_FORMAT_BGR);
constexpr fmask F_YUV3 = fbit (GSTAMD_VIDEO_FORMAT_v308) | fbit (GSTAMD_VIDEO_FORMAT_IYU2);
constexpr fmask F_GRAY8 = fbit (GSTAMD_VIDEO_FORMAT_GRAY8);
constexpr fmask F_NV12 = fbit (GSTAMD_VIDEO_FORMAT_NV12), F_NV16 = fbit (GSTAMD_VIDEO_FORMAT_NV16), F_NV24 = fbit (GSTAMD_VIDEO_FORMAT_NV24);
constexpr fmask F_A420 = fbit (GSTAMD_VIDEO_FORMAT_A420);
constexpr fmask F_Y41B = fbit (GSTAMD_VIDEO_FORMAT_Y41B);
constexpr fmask F_RGB16S = fbit (GSTAMD_VIDEO_FORMAT_RGB16) | fbit (GSTAMD_VIDEO_FORMAT_BGR16) | fbit (GSTAMD_VIDEO_FORMAT_RGB15) | fbit (GSTAMD_VIDEO_FORMAT_BGR15);
struct FastRow {
  fmask in, out;        /* format sets; `same` rows need in == out on top */
  bool same;
  unsigned flags;
  const char *name;
};
const FastRow g_fast_rows[] = {
  {F_420, F_RGB4X | F_RGB4A | F_RGB3 | F_RGB16S, false, FP_MATRIX | FP_SIZE | FP_CROP | FP_BORDER, "convert_I420_xRGB"},       /* (... _pack_ARGB into RGB15 / 16: :8797-8800, 8809-8812) */
  {F_AYUV, F_RGB4X, false, FP_ILACE | FP_MATRIX | FP_SIZE | FP_CROP | FP_BORDER, "convert_AYUV_xRGB"},
  {F_AYUV, F_RGB4A, false, FP_ILACE | FP_MATRIX | FP_SIZE | FP_CROP | FP_BORDER | FP_ACOPY, "convert_AYUV_xRGB"},
  {F_420, F_AYUV, false, FP_ILACE | FP_SIZE | FP_ASET, "convert_I420_AYUV"},
  {F_Y42B, F_AYUV, false, FP_ILACE | FP_SIZE | FP_CROP | FP_BORDER | FP_ASET | FP_WEVEN, "convert_Y42B_AYUV"},
  {F_Y444, F_AYUV, false, FP_ILACE | FP_SIZE | FP_CROP | FP_BORDER | FP_ASET, "convert_Y444_AYUV"},
  {F_YUY2, F_AYUV, false, FP_ILACE | FP_SIZE | FP_CROP | FP_BORDER | FP_ASET | FP_WEVEN, "convert_YUY2_AYUV"},
  {F_UYVY, F_AYUV, false, FP_ILACE | FP_SIZE | FP_CROP | FP_BORDER | FP_ASET, "convert_UYVY_AYUV"},
  {F_AYUV, F_420, false, FP_SIZE | FP_CROP | FP_BORDER | FP_WEVEN | FP_HEVEN, "convert_AYUV_I420"},
  {F_AYUV, F_Y42B | F_YUY2 | F_UYVY, false, FP_ILACE | FP_SIZE | FP_CROP | FP_BORDER | FP_WEVEN, "convert_AYUV_422"},
  {F_AYUV, F_Y444, false, FP_ILACE | FP_SIZE | FP_CROP | FP_BORDER, "convert_AYUV_Y444"},
  {F_420, F_YUY2 | F_UYVY, false, FP_ILACE | FP_SIZE, "convert_I420_YUY2"},
  {F_YUY2 | F_UYVY, F_420, false, FP_ILACE | FP_SIZE, "convert_YUY2_I420"},
  {F_Y42B, F_YUY2 | F_UYVY, false, FP_ILACE | FP_SIZE | FP_CROP | FP_BORDER, "convert_Y42B_YUY2"},
  {F_Y444, F_YUY2 | F_UYVY, false, FP_ILACE | FP_SIZE | FP_CROP | FP_BORDER | FP_WEVEN, "convert_Y444_YUY2"},
  {F_YUY2 | F_UYVY, F_Y42B | F_Y444, false, FP_ILACE | FP_SIZE | FP_CROP | FP_BORDER, "convert_YUY2_planar"},
  {F_YUY2, F_UYVY, false, FP_ILACE | FP_SIZE | FP_CROP | FP_BORDER, "convert_UYVY_YUY2"},
  {F_UYVY, F_GRAY8, false, FP_ILACE | FP_MATRIX | FP_SIZE | FP_CROP | FP_BORDER, "convert_UYVY_GRAY8"},          /* :8501 */
  {F_UYVY, F_YUY2, false, FP_ILACE | FP_SIZE | FP_CROP | FP_BORDER, "convert_UYVY_YUY2"},
  /* the v210 rows (:8433-8543; convert_I420_v210, convert_v210_I420_10 ...: their own arithmetic - 8-bit samples shifted by two, no widening):
     not built, such pairs are refused at their own size (every one of these rows wants keeps_size and the same colour matrix; crop / border /
     alpha flags are matched generously - refusing a conversion the reference would run through the chain costs nothing but coverage) */
  /* round 5: the 8-bit ones on whole frames (video_v210_fast.h); with a crop or a rectangle the generous rows below still refuse */
  {F_420 | F_Y42B | F_YUY2 | F_UYVY | fbit (GSTAMD_VIDEO_FORMAT_I420_10LE) | fbit (GSTAMD_VIDEO_FORMAT_I422_10LE), fbit (GSTAMD_VIDEO_FORMAT_v210), false, FP_ILACE | FP_SIZE, "convert_8bit_v210"},
  {fbit (GSTAMD_VIDEO_FORMAT_v210), F_420 | F_Y42B | F_YUY2 | F_UYVY | fbit (GSTAMD_VIDEO_FORMAT_I420_10LE) | fbit (GSTAMD_VIDEO_FORMAT_I422_10LE), false, FP_ILACE | FP_SIZE, "convert_v210_8bit"},
  {F_420 | F_Y42B | F_YUY2 | F_UYVY | fbit (GSTAMD_VIDEO_FORMAT_I420_10LE) | fbit (GSTAMD_VIDEO_FORMAT_I422_10LE), fbit (GSTAMD_VIDEO_FORMAT_v210), false,
        FP_ILACE | FP_SIZE | FP_CROP | FP_BORDER | FP_ACOPY | FP_ASET | FP_AMULT, "convert_to_v210"},
  {fbit (GSTAMD_VIDEO_FORMAT_v210), F_420 | F_Y42B | F_YUY2 | F_UYVY | fbit (GSTAMD_VIDEO_FORMAT_I420_10LE) | fbit (GSTAMD_VIDEO_FORMAT_I422_10LE), false,
        FP_ILACE | FP_SIZE | FP_CROP | FP_BORDER | FP_ACOPY | FP_ASET | FP_AMULT, "convert_from_v210"},
  /* convert_scale_planes: every same-format pair, planar <-> planar and the NV12 / NV16 / NV24 family */
  {F_RGB4A | F_AYUV | fbit (GSTAMD_VIDEO_FORMAT_ARGB64) | fbit (GSTAMD_VIDEO_FORMAT_AYUV64), ~(fmask) 0, true, FP_CROP | FP_BORDER | FP_ACOPY, "convert_scale_planes"},
  {F_RGB4X | F_RGB3 | F_YUV3 | F_420 | F_Y42B | F_Y444 | fbit (GSTAMD_VIDEO_FORMAT_GBR) | F_NV12 | F_NV16 | F_NV24 | fbit (GSTAMD_VIDEO_FORMAT_NV21) | fbit (GSTAMD_VIDEO_FORMAT_NV61) |
        F_YUY2 | F_UYVY | fbit (GSTAMD_VIDEO_FORMAT_YVYU), ~(fmask) 0, true, FP_CROP | FP_BORDER, "convert_scale_planes"},
  {F_420 | F_Y42B | F_Y444 | F_Y41B, F_420 | F_Y42B | F_Y444 | F_Y41B, false, FP_CROP | FP_BORDER, "convert_scale_planes"},          /* Y41B: :8554-8630 */
  {F_Y41B, F_Y41B, true, FP_CROP | FP_BORDER, "convert_scale_planes"},
  /* the GRAY8 rows (:8560-8661): the luma plane from / to the planar formats, chroma planes filled with 0x80; GRAY8 -> GRAY8 */
  {F_420 | F_Y42B | F_Y444 | F_Y41B, F_GRAY8, false, FP_CROP | FP_BORDER, "convert_scale_planes"},
  {F_GRAY8, F_420 | F_Y42B | F_Y444 | F_Y41B | F_GRAY8, false, FP_CROP | FP_BORDER, "convert_scale_planes"},
  {F_NV12 | F_NV16 | F_NV24, F_NV12 | F_NV16 | F_NV24, false, FP_CROP | FP_BORDER, "convert_scale_planes"},
  /* the A420 rows (:8560-8700, 8820-8842): into 4-byte RGB with the alpha plane copied (convert_A420_pack_ARGB / _BGRA: ABGR, RGBA, BGRA only), into the
     alpha-less RGB formats through the I420 functions (BGRx, xBGR, RGBx, RGB, BGR, RGB15, BGR16 - not xRGB, ARGB, RGB16, BGR15), the plane scaler */
  {F_A420, fbit (GSTAMD_VIDEO_FORMAT_ABGR) | fbit (GSTAMD_VIDEO_FORMAT_RGBA) | fbit (GSTAMD_VIDEO_FORMAT_BGRA), false,
        FP_MATRIX | FP_SIZE | FP_CROP | FP_BORDER | FP_ACOPY, "convert_I420_xRGB"},
  {F_A420, fbit (GSTAMD_VIDEO_FORMAT_BGRx) | fbit (GSTAMD_VIDEO_FORMAT_xBGR) | fbit (GSTAMD_VIDEO_FORMAT_RGBx) | F_RGB3 | fbit (GSTAMD_VIDEO_FORMAT_RGB15) |
        fbit (GSTAMD_VIDEO_FORMAT_BGR16), false, FP_MATRIX | FP_SIZE | FP_CROP | FP_BORDER, "convert_I420_xRGB(opaque)"},
  {F_420 | F_Y42B | F_Y444 | F_Y41B | F_GRAY8, F_A420, false, FP_CROP | FP_BORDER | FP_ASET, "convert_scale_planes"},
  {F_A420, F_420 | F_Y42B | F_Y444 | F_Y41B | F_GRAY8, false, FP_CROP | FP_BORDER, "convert_scale_planes"},
  {F_A420, F_A420, true, FP_CROP | FP_BORDER | FP_ACOPY, "convert_scale_planes"},
  /* GBRA / RGBP / BGRP onto themselves (:8851-8856; GBR is in the long row above) */
  {fbit (GSTAMD_VIDEO_FORMAT_RGBP) | fbit (GSTAMD_VIDEO_FORMAT_BGRP), ~(fmask) 0, true, FP_CROP | FP_BORDER, "convert_scale_planes"},
  {fbit (GSTAMD_VIDEO_FORMAT_GBRA), ~(fmask) 0, true, FP_CROP | FP_BORDER | FP_ACOPY, "convert_scale_planes"},
  /* RGB15 / RGB16 / BGR15 / BGR16 onto themselves (:8879-8886; setup_scale serves them with nearest only, :7985-8003) */
  {F_RGB16S, ~(fmask) 0, true, FP_CROP | FP_BORDER, "convert_scale_planes"},
  /* GRAY16_LE -> GRAY16_LE, GRAY16_BE -> GRAY16_BE (:8901-8904) */
  {fbit (GSTAMD_VIDEO_FORMAT_GRAY16_LE) | fbit (GSTAMD_VIDEO_FORMAT_GRAY16_BE), ~(fmask) 0, true, FP_CROP | FP_BORDER, "convert_scale_planes"},
};
}  // namespace

static const char *lookup_fastpath (const VideoPlan &p, int alpha_mode_bits, bool same_matrix)
{
  const int in = p.in_info.format, out = p.out_info.format;
  /* interlaced infos: rows with keeps_interlaced only (:8985).  Of the convert_scale_planes rows those are every same-format pair, I420 <-> YV12 and the
     pairs among NV12 / NV16 / NV24 (:8547-8904) */
  const bool ilace = p.field != 0;
  const fmask nv = F_NV12 | F_NV16 | F_NV24;
  const bool planes_ilace_ok = in == out || ((fbit (in) & F_420) && (fbit (out) & F_420)) || ((fbit (in) & nv) && (fbit (out) & nv));
  const bool same_size = p.ref_same_size;
  const bool need_copy = alpha_mode_bits & 1, need_set = alpha_mode_bits & 2, need_mult = alpha_mode_bits & 4;
  if (p.config.dither_quantization != 1)
    return nullptr;
  const RectPlan &rc = p.rect;
  const bool crop = rc.in_x || rc.in_y || p.in_info.width < rc.in_maxw || p.in_info.height < rc.in_maxh;
  const bool border = rc.out_x || rc.out_y || (rc.out_maxw && p.out_info.width < rc.out_maxw) || (rc.out_maxh && p.out_info.height < rc.out_maxh);
  const int full_w = rc.in_maxw ? rc.in_maxw : p.in_info.width, full_h = ilace ? p.in_info.frame_height : (rc.in_maxh ? rc.in_maxh : p.in_info.height);
  for (const FastRow &r : g_fast_rows) {
    if (!(r.in & fbit (in)) || !(r.out & fbit (out)) || (r.same && in != out))
      continue;
    const unsigned f = r.flags;
    if (ilace && !(strcmp (r.name, "convert_scale_planes") == 0 ? planes_ilace_ok : (f & FP_ILACE) != 0))
      continue;
    if (((f & FP_MATRIX) || same_matrix) && (!(f & FP_SIZE) || same_size) && (!(f & FP_WEVEN) || !(full_w & 1)) &&
        (!(f & FP_HEVEN) || !(full_h & 1)) && ((f & FP_CROP) || !crop) && ((f & FP_BORDER) || !border) &&
        ((f & FP_ACOPY) || !need_copy) && ((f & FP_ASET) || !need_set) && ((f & FP_AMULT) || !need_mult))
      return r.name;
    /* the reference's loop goes on to later rows of the same pair; there is one row per pair here except the same-format
     * rows, which come last */
  }
  return nullptr;
}

// ------------------------------------------------------------------------------------------------
// convert_scale_planes on planar / semi-planar formats: setup_scale (video-converter.c:7958-8200)
// ------------------------------------------------------------------------------------------------
static int plan_planes (VideoPlan *plan, const char *fastpath)
{
  const GstAmdVideoInfo &in = plan->in_info, &out = plan->out_info;
  const GstAmdVideoConverterConfig &cfg = plan->config;
  const FormatDesc *fi = plan->fin, *fo = plan->fout;
  const int method = cfg.resampler_method;
  const int cr_method = method == GSTAMD_RESAMPLER_METHOD_NEAREST ? method : cfg.chroma_resampler_method;
  plan->plane_mode = true;
  plan->planes.clear ();
  plan->ref_fastpath = fastpath;
  uint64_t bytes = 0;
  auto up = [](int v, int sub) { return -((-v) >> sub); };       /* GST_VIDEO_SUB_SCALE */
  /* a field of an interlaced frame (setup_scale :7977, 8075, 8236): every decision is the FRAME plane's, the vertical scalers are interlaced ones
     (this field's resampler over the field's lines), and none of the halve / double shortcuts applies (`!interlaced &&`, :8171-8206) */
  const int field = plan->field;
  const int frame_ih = field ? in.frame_height : in.height, frame_oh = field ? out.frame_height : out.height;
  for (int i = 0; i < fo->n_planes; i++) {
    PlanePlan pp;
    pp.dst_plane = i;
    pp.n_elems = 1;
    int isub_w = 0, isub_h = 0, osub_w = 0, osub_h = 0;
    bool fill = false;
    if (i == 0) {
      pp.src_plane = 0;
    } else if (fo->kind == UNPACK_PLANAR_A && i == 3) {
      /* A420's alpha plane: the source's when it has one, else convert_plane_fill - with 0x80, not the alpha value: setup_scale's
         `if (i == 3) ffill = alpha_value; if (i == 0) ffill = 0; else ffill = 0x80;` lacks an else (:8141-8147) */
      pp.src_plane = 3;
      fill = fi->kind != UNPACK_PLANAR_A;
      if (fill)
        pp.src_plane = 0;
    } else if (fi->kind == UNPACK_GRAY) {
      /* the source has no such component (setup_scale :8108-8149): convert_plane_fill with 0x80 */
      fill = true;
      pp.src_plane = 0;
      osub_w = fo->w_sub, osub_h = fo->h_sub;
    } else if (fo->kind == UNPACK_SEMI) {
      pp.src_plane = 1;
      pp.n_elems = 2;
      isub_w = fi->w_sub, isub_h = fi->h_sub, osub_w = fo->w_sub, osub_h = fo->h_sub;
    } else {
      /* component of this destination plane -> the source plane holding the same component */
      const bool is_u = i == fo->u_plane;
      pp.src_plane = is_u ? fi->u_plane : fi->v_plane;
      isub_w = fi->w_sub, isub_h = fi->h_sub, osub_w = fo->w_sub, osub_h = fo->h_sub;
    }
    pp.iw = up (in.width, isub_w);
    pp.ih = up (in.height, isub_h);
    pp.ow = up (out.width, osub_w);
    pp.oh = up (out.height, osub_h);
    const int fpih = up (frame_ih, isub_h), fpoh = up (frame_oh, osub_h);          /* the frame's plane heights */
    if (field) {
      pp.ih = field == 1 ? (fpih + 1) / 2 : fpih - (fpih + 1) / 2;
      pp.oh = field == 1 ? (fpoh + 1) / 2 : fpoh - (fpoh + 1) / 2;
      if (pp.ih <= 0 || pp.oh <= 0)
        return GSTAMD_ERR_UNSUPPORTED;
    }
    if (fo->kind == UNPACK_GRAY16 || fo->kind == UNPACK_RGB16) {
      pp.n_elems = 2;          /* copies and nearest passes only (plan_core): a 16-bit sample moves as two bytes */
    } else if (fo->kind == UNPACK_PACKED3) {
      pp.n_elems = 3;          /* get_functions (video-scaler.c:1222): RGB / BGR are 3 x u8 pixels */
    } else if (fo->kind == UNPACK_PACKED422) {
      /* get_functions (:1215): the line is ROUND_UP_4 (width * 2) single bytes */
      pp.iw = round_up (in.width * 2, 4);
      pp.ow = round_up (out.width * 2, 4);
    }
    /* setup_scale (:8165): resample_method = (i == 0 ? method : cr_method) on the FRAME's plane order - GBR's first plane is G, plane 1 of a plan */
    int perm_o[4], first_plane = 0;
    format_plane_perm (fo->format, perm_o);
    for (int q = 0; q < 4; q++)
      if (perm_o[q] == 0)
        first_plane = q;
    const int rm = i == first_plane ? method : cr_method;
    /* the halve / double shortcuts exist for the planes of multi-plane formats only (setup_scale :8092-8180) */
    if (fill) {
      pp.iw = pp.ih = 0;
      pp.kind = PLANE_FILL;
      bytes += (uint64_t) pp.ow * pp.oh;
      plan->planes.push_back (pp);
      continue;
    }
    const bool p1 = pp.n_elems == 1 && (fo->n_planes > 1 || fo->kind == UNPACK_GRAY), lin = rm == GSTAMD_RESAMPLER_METHOD_LINEAR, near = rm == GSTAMD_RESAMPLER_METHOD_NEAREST;
    bool need_h = false, need_v = false;
    pp.kind = PLANE_SCALE;
    const int dih = field ? fpih : pp.ih, doh = field ? fpoh : pp.oh;          /* the heights the decisions look at */
    /* the planes of multi-plane and GRAY formats (setup_scale's plane loop, :8088): `if (iw == ow) { if (!interlaced && ih == oh) copy ...  else
       need_v_scaler }` - an interlaced plane of unchanged WIDTH always goes through the interlaced vertical scaler, which is no copy at equal heights
       either (its two resamplers are shifted by half a line, video-scaler.c:233).  One-plane formats (:8016-8087) scale vertically iff the heights differ. */
    const bool loop_plane = fo->n_planes > 1 || fo->kind == UNPACK_GRAY || fo->kind == UNPACK_GRAY16;
    if (pp.iw == pp.ow) {
      if (dih == doh && !(field && loop_plane))
        pp.kind = PLANE_COPY;
      else if (!field && dih == 2 * doh && p1 && lin)
        pp.kind = PLANE_V_HALVE;
      else if (!field && 2 * dih == doh && p1 && near)
        pp.kind = PLANE_V_DOUBLE;
      else
        need_v = true;
    } else if (dih == doh) {
      if (!field && pp.iw == 2 * pp.ow && p1 && lin)
        pp.kind = PLANE_H_HALVE;
      else if (!field && 2 * pp.iw == pp.ow && p1 && near)
        pp.kind = PLANE_H_DOUBLE;
      else
        need_h = true;
    } else {
      if (!field && pp.iw == 2 * pp.ow && dih == 2 * doh && p1 && lin)
        pp.kind = PLANE_HV_HALVE;
      else if (!field && 2 * pp.iw == pp.ow && 2 * dih == doh && p1 && near)
        pp.kind = PLANE_HV_DOUBLE;
      else
        need_h = need_v = true;
    }
    /* a packed 4:2:2 line: setup_scale compares the PIXEL widths (:8020); 63 and 64 pixels are the same 32 macropixels and still
       go through the merged scaler */
    if (fo->kind == UNPACK_PACKED422 && in.width != out.width && !need_h) {
      need_h = true;
      pp.kind = PLANE_SCALE;
    }
    if (pp.kind == PLANE_SCALE) {
      ScalePass hp, vp;
      /* get_functions (video-scaler.c:1202-1342): a 2-tap horizontal pass is ldreslin only for 1- and 4-byte pixels;
       * the 2-byte UV pixels and the 3-byte RGB pixels take video_scale_h_ntap_u8 with two 6-bit taps */
      if (need_h && fo->kind == UNPACK_PACKED422) {
        /* setup_scale (:8020-8046) + gst_video_scaler_combine_packed_YUV (video-scaler.c:1134-1199): a luma scaler over the pixels and a
           chroma scaler over the pixel pairs with the SAME tap count, merged into one scaler over the line's bytes - output byte i is luma
           i / 2 (source byte offset * 2 + the format's luma position) or chroma byte i & 3 of macropixel i / 4 (offset * 4 + (i & 3)) */
        ScalePass yp, cp;
        make_scale_pass (rm, cfg.resampler_taps, cfg, in.width, out.width, true, &yp, true);
        make_scale_pass (rm, (unsigned) yp.n_taps, cfg, (in.width + 1) / 2, (out.width + 1) / 2, true, &cp, true);
        if (cp.n_taps != yp.n_taps)
          return GSTAMD_ERR_UNSUPPORTED;        /* g_return_val_if_fail (uv max_taps == y max_taps) */
        const int y_off = (in.format == GSTAMD_VIDEO_FORMAT_YUY2 || in.format == GSTAMD_VIDEO_FORMAT_YVYU) ? 0 : 1;        /* get_y_offset */
        hp = yp;
        hp.in_size = pp.iw;
        hp.out_size = pp.ow;
        hp.merged = 1 + y_off;
        hp.dot4_ok = false;
        hp.tapw.clear ();
        hp.offset.assign ((size_t) pp.ow, 0);
        if (yp.n_taps > 1)
          hp.taps.assign ((size_t) pp.ow * yp.n_taps, 0);
        for (int b = 0; b < pp.ow; b++) {
          const bool luma = (b & 1) == y_off;
          const ScalePass &src = luma ? yp : cp;
          const int ic = luma ? std::min (b / 2, yp.out_size - 1) : std::min (b / 4, cp.out_size - 1);
          hp.offset[(size_t) b] = luma ? src.offset[(size_t) ic] * 2 + (uint32_t) y_off : src.offset[(size_t) ic] * 4 + (uint32_t) (b & 3);
          for (int l = 0; l < yp.n_taps && yp.n_taps > 1; l++)
            hp.taps[(size_t) b * yp.n_taps + l] = src.taps[(size_t) ic * yp.n_taps + l];
        }
      } else if (need_h)
        make_scale_pass (rm, cfg.resampler_taps, cfg, pp.iw, pp.ow, true, &hp, pp.n_elems != 1);
      int zip_last = 0;
      if (need_v && field) {
        if (!make_field_vpass (rm, cfg.resampler_taps, cfg, fpih, fpoh, field - 1, &vp, false, &zip_last, nullptr))
          return GSTAMD_ERR_UNSUPPORTED;
      } else if (need_v)
        make_scale_pass (rm, cfg.resampler_taps, cfg, pp.ih, pp.oh, false, &vp, false);
      hp.max_span = vp.max_span = 1 << 30;
      if (need_h && need_v) {
        /* gst_video_scaler_2d: horizontal first iff width * voffset[height - 1] <= width * height */
        const bool h_first = field ? (long) pp.ow * (long) zip_last <= (long) pp.ow * fpoh : (long) pp.ow * (long) vp.offset[pp.oh - 1] <= (long) pp.ow * pp.oh;
        if (h_first) {
          pp.passes.push_back (hp);
          pp.passes.push_back (vp);
        } else {
          pp.passes.push_back (vp);
          pp.passes.push_back (hp);
        }
      } else {
        pp.passes.push_back (need_h ? hp : vp);
      }
    }
    bytes += (uint64_t) pp.iw * pp.ih * pp.n_elems + (uint64_t) pp.ow * pp.oh * pp.n_elems;
    plan->planes.push_back (pp);
  }
  plan->algorithmic_bytes = bytes;
  std::string d = std::string ("scale_planes[") + fi->name + "->" + fo->name;
  static const char *kn[] = {"copy", "h/2", "hx2", "v/2", "vx2", "hv/2", "hvx2", "scale", "fill"};
  for (const PlanePlan &pp : plan->planes) {
    d += std::string (",") + kn[pp.kind];
    for (const ScalePass &ps : pp.passes)
      d += std::string (ps.horizontal ? ":H" : ":V") + std::to_string (ps.n_taps);
  }
  plan->description = d + "]";
  return GSTAMD_OK;
}

// ------------------------------------------------------------------------------------------------
// Source span of the outputs [t0, t1) of a horizontal pass (same rule as hscale_span on the device)
static void pass_span (const ScalePass &pass, int t0, int t1, int *lo, int *hi)
{
  if (pass.kind == SCALE_2TAP) {
    *lo = (t0 * pass.inc) >> 16;
    *hi = (((t1 - 1) * pass.inc) >> 16) + 2;
  } else {
    *lo = (int) pass.offset[t0];
    *hi = (int) pass.offset[t1 - 1] + (pass.kind == SCALE_NEAREST ? 1 : pass.n_taps);
  }
}

// Pick the tile width whose staging step wastes the fewest lane slots: a wave stages ceil (span / 8) 8-pixel groups
// in rounds of 64 lanes, so e.g. a 2:1 reduction is best served by 252 outputs (span 506 -> 64 groups, one round)
// rather than 256 (span 514 -> 65 groups, two rounds).
TileGeom pass_tile_geom (const ScalePass &pass)
{
  TileGeom best = {0, 0, 0};
  double best_cost = 1e30;
  const int osz = pass.out_size;
  for (int tw = 256; tw >= 64; tw -= 4) {
    int worst = 0;
    for (int t0 = 0; t0 < osz; t0 += tw) {
      int lo, hi;
      pass_span (pass, t0, std::min (t0 + tw, osz), &lo, &hi);
      worst = std::max (worst, hi - (lo & ~7));
    }
    const int groups = (worst + 7) / 8;
    const int rounds = (groups + 63) / 64;
    const double cost = (double) rounds * 64.0 / (double) std::min (tw, osz);
    if (cost < best_cost - 1e-12) {
      best_cost = cost;
      best.tile_w = tw;
      best.lds_px = groups * 8;
    }
    if (tw >= osz && rounds == 1)
      break;
  }
  /* 16-pixel staging (video_hscale420.h): the fewest tiles whose spans fit one round of 64 lanes x 16 pixels, evenly wide */
  best.tile16_w = 0;
  if (pass.kind == SCALE_NTAP) {
    for (int tiles = (osz + 255) / 256; tiles <= (osz + 63) / 64; tiles++) {
      const int tw = std::min (256, ((osz + tiles - 1) / tiles + 3) & ~3);
      int worst = 0;
      for (int t0 = 0; t0 < osz; t0 += tw) {
        int lo, hi;
        pass_span (pass, t0, std::min (t0 + tw, osz), &lo, &hi);
        worst = std::max (worst, hi - (lo & ~15));
      }
      if (worst <= 1024) {
        best.tile16_w = tw;
        break;
      }
    }
  }
  return best;
}



// bytes of the picture's samples in a frame of format f (SURVEY.md 8d's accounting: every plane once)
static uint64_t picture_bytes (const FormatDesc *f, int w, int h)
{
  if (f->kind == UNPACK_PACKED4)
    return (uint64_t) w * h * (f->hi_depth == 3 ? 8 : 4);
  if (f->kind == UNPACK_PACKED3)
    return (uint64_t) w * h * 3;
  if (f->kind == UNPACK_GRAY)
    return (uint64_t) w * h;
  if (f->kind == UNPACK_Y410)
    return (uint64_t) w * h * 4;
  if (f->kind == UNPACK_PACKED64)
    return (uint64_t) w * h * 8;
  if (f->kind == UNPACK_GRAY16 || f->kind == UNPACK_RGB16)
    return (uint64_t) w * h * 2;
  if (f->kind == UNPACK_P422_16)
    return (uint64_t) ((w + 1) / 2) * 8 * h;
  if (f->kind == UNPACK_V210)
    return (uint64_t) ((w + 5) / 6) * 16 * h;
  if (f->kind == UNPACK_PACKED411)
    return (uint64_t) ((w + 3) / 4) * 6 * h;
  if (f->kind == UNPACK_GRAY_LE32)
    return (uint64_t) ((w + 2) / 3) * 4 * h;
  if (f->kind == UNPACK_SEMI_TILED)
    return (uint64_t) w * h + 2 * (((uint64_t) w + 1) / 2) * (((uint64_t) h + 1) / 2);
  if (f->kind == UNPACK_P422_UYVP)
    return (uint64_t) ((w + 1) / 2) * 5 * h;
  if (f->kind == UNPACK_SEMI_LE40_TILED)
    return (uint64_t) ((w + 3) / 4) * 5 * h + (uint64_t) ((w + 3) / 4) * 5 * ((h + 1) / 2);
  if (f->kind == UNPACK_SEMI_LE40)
    return (uint64_t) ((10 * (uint64_t) w + 7) / 8) * h + (uint64_t) ((20 * (((uint64_t) w + 1) / 2) + 7) / 8) * ((h + (1 << f->h_sub) - 1) >> f->h_sub);
  if (f->kind == UNPACK_SEMI_LE32)
    return (uint64_t) ((w + 2) / 3) * 4 * (h + ((h + (1 << f->h_sub) - 1) >> f->h_sub));
  const uint64_t cw = ((uint64_t) w + (1 << f->w_sub) - 1) >> f->w_sub, ch = ((uint64_t) h + (1 << f->h_sub) - 1) >> f->h_sub;
  const uint64_t n = (uint64_t) w * h + 2 * cw * ch + (GSTAMD_KIND_ALPHA_PLANE (f->kind) >= 0 ? (uint64_t) w * h : 0);
  return f->hi_depth ? 2 * n : n;
}

static int plan_core (const GstAmdVideoInfo *in, const GstAmdVideoInfo *out, const GstAmdVideoConverterConfig *config, VideoPlan *plan, std::string *error);
static void fill_pack_params (const GstAmdVideoInfo *in, const GstAmdVideoInfo *out, const FormatDesc *fi, const FormatDesc *fo, const GstAmdVideoConverterConfig &cfg,
    int full_in_w, int full_in_h, int full_out_w, int full_out_h, PackPlanarParams *pkp);

/* chain_dither (:2035-2085) on 16-bit lines ahead of a 10 / 12 / 16-bit planar pack: quantiser 1 << (16 - depth), or the target
 * quantiser when that is coarser; shift[] in unpack order (A, Y, U, V) */
static void setup_dither16 (const GstAmdVideoConverterConfig &cfg, const FormatDesc *fo, DitherParams *d)
{
  memset (d, 0, sizeof (*d));
  if (cfg.dither_method == GSTAMD_DITHER_NONE)
    return;
  const int depth = hi_depth_bits (fo->hi_depth);
  unsigned q = 1u << (16 - depth);
  if (cfg.dither_quantization > q)
    q = cfg.dither_quantization;
  int shift = 0;
  for (unsigned v = q; v > 1; v >>= 1)
    shift++;
  if (shift > 16)
    shift = 16;                 /* guint16 masks */
  d->on = shift > 0;
  d->method = cfg.dither_method;
  /* the planar formats have no alpha component (depth 0: quantiser 0); ARGB64 / AYUV64 carry 16 bits of it like the other three - their own
     quantiser is 1, so a stage only exists for dither-quantization > 1, run as a pass over the finished frame (k_dither16_image) */
  d->shift[0] = fo->hi_depth == 3 || fo->kind == UNPACK_PACKED64 || fo->kind == UNPACK_PLANAR_A ? shift : 0;
  if (fo->kind == UNPACK_Y410 && fo->hi_depth != 27 && fo->format != GSTAMD_VIDEO_FORMAT_BGR10x2_LE && fo->format != GSTAMD_VIDEO_FORMAT_RGB10x2_LE) {
    /* 2 bits of alpha: quantiser 1 << 14 (chain_dither :2060-2075 per component depth; the x2 formats declare no fourth component: depth 0) */
    unsigned qa = 1u << 14;
    if (cfg.dither_quantization > qa)
      qa = cfg.dither_quantization;
    int sa = 0;
    for (unsigned v = qa; v > 1; v >>= 1)
      sa++;
    d->shift[0] = sa > 16 ? 16 : sa;
  }
  d->shift[1] = d->shift[2] = d->shift[3] = shift;
}

// gamma-mode = remap: the composite plan of GammaPlan (planner.h).  `in` / `out` are the crop / the destination rectangle.
static int plan_gamma (const GstAmdVideoInfo *in, const GstAmdVideoInfo *out, VideoPlan *plan, int alpha_bits, bool same_primaries, M44 prim_dm,
    std::string *error)
{
  auto fail = [&](int code, const std::string &msg) {
    if (error)
      *error = msg;
    return code;
  };
  const GstAmdVideoConverterConfig &cfg = plan->config;
  const FormatDesc *fi = plan->fin, *fo = plan->fout;
  GammaPlan &g = plan->gamma;
  const bool in16 = fi->hi_depth != 0, out16 = fo->hi_depth != 0;
  if (out16 && !kind_has_planes (fo->kind) && fo->kind != UNPACK_PLANAR_A && fo->hi_depth != 3 && fo->kind != UNPACK_P422_16 && !GSTAMD_KIND_PX16 (fo->kind) && fo->kind != UNPACK_V210 &&
      !GSTAMD_KIND_LE32 (fo->kind))
    return fail (GSTAMD_ERR_UNSUPPORTED, "10-bit destination layout not implemented on the GPU path");
  /* a 10 / 12 / 16-bit planar SOURCE: the 16-bit front (unpack + chroma upsampler, k_front16) of the conversion into an AYUV64 frame of
   * the same size - planned like any other conversion, its front / pair table taken over */
  FrontParams front16;
  std::vector<int32_t> vpair16;
  memset (&front16, 0, sizeof (front16));
  if (in16 && fi->hi_depth != 3) {
    const int fw = plan->rect.in_maxw ? plan->rect.in_maxw : in->width, fh = plan->rect.in_maxh ? plan->rect.in_maxh : in->height;
    const int ow_full = plan->rect.out_maxw ? plan->rect.out_maxw : out->width, oh_full = plan->rect.out_maxh ? plan->rect.out_maxh : out->height;
    const bool differs16 = fi->w_sub != fo->w_sub || fi->h_sub != fo->h_sub || in->chroma_site != out->chroma_site || fw != ow_full || fh != oh_full;
    const bool up16 = differs16 && cfg.chroma_mode != GSTAMD_CHROMA_MODE_NONE && cfg.chroma_mode != GSTAMD_CHROMA_MODE_DOWNSAMPLE_ONLY;
    GstAmdVideoInfo a64;
    /* the unpack format of the source: AYUV64, or ARGB64 for the RGB formats of the 16-bit chain (RGB10A2_LE, BGR10A2_LE) */
    if (video_info_set_format (&a64, fi->yuv ? GSTAMD_VIDEO_FORMAT_AYUV64 : GSTAMD_VIDEO_FORMAT_ARGB64, in->width, in->height) != GSTAMD_OK)
      return fail (GSTAMD_ERR_INVALID, "bad frame size");
    a64.color_range = in->color_range;
    a64.color_matrix = in->color_matrix;
    a64.color_transfer = in->color_transfer;
    a64.color_primaries = in->color_primaries;
    a64.chroma_site = in->chroma_site;
    GstAmdVideoConverterConfig fc;
    converter_config_init (&fc);
    fc.chroma_mode = up16 ? GSTAMD_CHROMA_MODE_FULL : GSTAMD_CHROMA_MODE_NONE;
    fc.matrix_mode = GSTAMD_MATRIX_MODE_NONE;
    fc.dither_method = GSTAMD_DITHER_NONE;
    fc.internal_flags = 1;
    VideoPlan tmp;
    std::string terr;
    memset (&tmp.rect, 0, sizeof (tmp.rect));
    /* the crop inside its frame: the chroma upsampler pairs the FRAME's rows (the rows above and below a crop are read, do_unpack_lines :2966) -
       round 4 planned this front as if the crop were the frame (found by the round-5 fuzz draws of gamma-mode = remap with a vertical crop) */
    tmp.rect.in_x = plan->rect.in_x, tmp.rect.in_y = plan->rect.in_y;
    tmp.rect.in_maxw = fw, tmp.rect.in_maxh = fh;
    tmp.rect.out_maxw = in->width, tmp.rect.out_maxh = in->height;
    tmp.orig_in = plan->orig_in;
    tmp.orig_out = a64;
    const int tr = plan_core (in, &a64, &fc, &tmp, &terr);
    if (tr != GSTAMD_OK)
      return fail (tr, "16-bit front of a gamma remap: " + terr);
    if (tmp.gamma.planes_fast || !tmp.gamma.src16)
      return fail (GSTAMD_ERR_UNSUPPORTED, "16-bit front of a gamma remap: unexpected plan for the front");
    front16 = tmp.front;
    vpair16 = tmp.vpair;
  }
  g.on = true;
  g.in16 = in16;
  g.out16 = out16;
  g.src64 = fi->hi_depth == 3;
  g.src16 = in16 && !g.src64;
  g.store64 = fo->hi_depth == 3;
  g.pack16 = out16 && !g.store64;
  g.planes_fast = false;
  /* nothing of the single-converter plan is used by a composite */
  plan->plane_mode = plan->relayout = plan->fast_pair = plan->fast_enc420 = plan->fast_420p = plan->fast_422 = plan->fast_422_ayuv = plan->fast_post = plan->deep16 = false;
  plan->matrix_before_scale = false;
  plan->vpair.clear ();
  plan->planes.clear ();
  memset (&plan->front, 0, sizeof (plan->front));
  memset (&plan->matrix, 0, sizeof (plan->matrix));
  memset (&plan->post, 0, sizeof (plan->post));
  memset (&plan->pack, 0, sizeof (plan->pack));
  memset (&plan->deep, 0, sizeof (plan->deep));
  memset (&plan->dither, 0, sizeof (plan->dither));
  if (g.src16) {
    plan->front = front16;
    plan->vpair = vpair16;
  }
  memset (&g.to_rgb16, 0, sizeof (g.to_rgb16));
  memset (&g.to_yuv16, 0, sizeof (g.to_yuv16));
  memset (&g.dither16, 0, sizeof (g.dither16));
  memset (&g.pack, 0, sizeof (g.pack));
  g.pack_hi_depth = fo->hi_depth;
  g.dec16.clear ();
  g.enc16.clear ();
  const bool unpack_rgb = !fi->yuv, pack_rgb = !fo->yuv;
  const int in_matrix = unpack_rgb ? GSTAMD_COLOR_MATRIX_RGB : in->color_matrix, out_matrix = pack_rgb ? GSTAMD_COLOR_MATRIX_RGB : out->color_matrix;
  /* the two 8-bit unpack-format images around the 16-bit part */
  if (video_info_set_format (&g.mid_in, fi->yuv ? GSTAMD_VIDEO_FORMAT_AYUV : GSTAMD_VIDEO_FORMAT_ARGB, in->width, in->height) != GSTAMD_OK ||
      video_info_set_format (&g.mid_out, fo->yuv ? GSTAMD_VIDEO_FORMAT_AYUV : GSTAMD_VIDEO_FORMAT_ARGB, out->width, out->height) != GSTAMD_OK)
    return fail (GSTAMD_ERR_INVALID, "bad frame size");
  g.mid_in.color_range = in->color_range;
  g.mid_in.color_matrix = in->color_matrix;
  g.mid_in.chroma_site = in->chroma_site;
  g.mid_out.color_range = out->color_range;
  g.mid_out.color_matrix = out->color_matrix;
  g.mid_out.chroma_site = out->chroma_site;
  g.sub_in_info = plan->orig_in;
  g.sub_out_info = plan->orig_out;
  /* video_converter_compute_resample (:2850-2895) on the FULL frames decides whether the chroma resamplers exist at all */
  const int full_in_w = plan->rect.in_maxw ? plan->rect.in_maxw : in->width, full_in_h = plan->rect.in_maxh ? plan->rect.in_maxh : in->height;
  const int full_out_w = plan->rect.out_maxw ? plan->rect.out_maxw : out->width, full_out_h = plan->rect.out_maxh ? plan->rect.out_maxh : out->height;
  const bool differs = fi->w_sub != fo->w_sub || fi->h_sub != fo->h_sub || in->chroma_site != out->chroma_site || full_in_w != full_out_w ||
      full_in_h != full_out_h;
  const bool up = differs && cfg.chroma_mode != GSTAMD_CHROMA_MODE_NONE && cfg.chroma_mode != GSTAMD_CHROMA_MODE_DOWNSAMPLE_ONLY;
  const bool down = differs && cfg.chroma_mode != GSTAMD_CHROMA_MODE_NONE && cfg.chroma_mode != GSTAMD_CHROMA_MODE_UPSAMPLE_ONLY;
  GstAmdVideoConverterConfig base = cfg;
  base.gamma_mode = GSTAMD_GAMMA_MODE_NONE;
  base.primaries_mode = GSTAMD_PRIMARIES_MODE_NONE;
  base.matrix_mode = GSTAMD_MATRIX_MODE_NONE;
  base.alpha_mode = GSTAMD_ALPHA_MODE_COPY;
  base.alpha_value = 1.0;
  base.internal_flags = 1;
  g.cfg_in = base;
  g.cfg_in.dither_quantization = 1;
  g.cfg_in.chroma_mode = up ? GSTAMD_CHROMA_MODE_FULL : GSTAMD_CHROMA_MODE_NONE;
  g.cfg_in.dest_x = g.cfg_in.dest_y = g.cfg_in.dest_width = g.cfg_in.dest_height = 0;
  g.cfg_out = base;
  g.cfg_out.chroma_mode = down ? GSTAMD_CHROMA_MODE_FULL : GSTAMD_CHROMA_MODE_NONE;
  g.cfg_out.src_x = g.cfg_out.src_y = g.cfg_out.src_width = g.cfg_out.src_height = 0;
  /* chain_convert_to_RGB (:1566-1610): the matrix to R'G'B' in 0 .. 1, scaled to 1 << 8, at 8 bits */
  memset (&g.to_rgb, 0, sizeof (g.to_rgb));
  memset (&g.to_yuv, 0, sizeof (g.to_yuv));
  int scratch[3][4];
  if (!unpack_rgb) {
    M44 dm;
    int offset[3], scale[3];
    double Kr = 0, Kb = 0;
    m_identity (dm);
    range_offsets (in->color_range, true, offset, scale);
    m_offset_components (dm, -offset[0], -offset[1], -offset[2]);
    m_scale_components (dm, 1 / ((float) scale[0]), 1 / ((float) scale[1]), 1 / ((float) scale[2]));
    if (cfg.matrix_mode != GSTAMD_MATRIX_MODE_NONE && get_Kr_Kb (cfg.matrix_mode == GSTAMD_MATRIX_MODE_OUTPUT_ONLY ? out_matrix : in_matrix, &Kr, &Kb))
      m_YCbCr_to_RGB (dm, Kr, Kb);
    m_scale_components (dm, (float) 256, (float) 256, (float) 256);
    if (!in16 && !m_is_identity (dm))
      prepare_matrix8 (dm, unpack_rgb, pack_rgb, &g.to_rgb, scratch);
  }
  if (in16 && !unpack_rgb) {
    /* the same at current_bits 16: the unpack format's 16-bit range offsets, scaled to 1 << 16, prepare_matrix -> video_converter_matrix16 */
    M44 dm;
    int offset[3], scale[3];
    double Kr = 0, Kb = 0;
    m_identity (dm);
    range_offsets (in->color_range, true, offset, scale, 16);
    m_offset_components (dm, -offset[0], -offset[1], -offset[2]);
    m_scale_components (dm, 1 / ((float) scale[0]), 1 / ((float) scale[1]), 1 / ((float) scale[2]));
    if (cfg.matrix_mode != GSTAMD_MATRIX_MODE_NONE && get_Kr_Kb (cfg.matrix_mode == GSTAMD_MATRIX_MODE_OUTPUT_ONLY ? out_matrix : in_matrix, &Kr, &Kb))
      m_YCbCr_to_RGB (dm, Kr, Kb);
    m_scale_components (dm, (float) 65536, (float) 65536, (float) 65536);
    if (!m_is_identity (dm)) {
      m_scale_components (dm, 256.0f, 256.0f, 256.0f);
      g.to_rgb16.has_matrix = 1;
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++)
          g.to_rgb16.im[i][j] = (int) rint (dm[i][j]);
    }
  }
  /* setup_gamma_decode (:1495-1530), 8 -> 16 bits, or 16 -> 16 */
  g.dec.clear ();
  if (in16) {
    g.dec16.resize (65536);
    for (int i = 0; i < 65536; i++)
      g.dec16[i] = (uint16_t) rint (transfer_decode (in->color_transfer, i / 65535.0) * 65535.0);
  } else {
    g.dec.resize (256);
    for (int i = 0; i < 256; i++)
      g.dec[i] = (uint16_t) rint (transfer_decode (in->color_transfer, i / 255.0) * 65535.0);
  }
  /* chain_convert with gamma (:1845-1854): only the primaries, on 16-bit values */
  memset (&g.prim, 0, sizeof (g.prim));
  if (!same_primaries && !m_is_identity (prim_dm)) {
    M44 dm;
    memcpy (dm, prim_dm, sizeof (M44));
    m_scale_components (dm, 256.0f, 256.0f, 256.0f);
    g.prim.has_matrix = 1;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 4; j++)
        g.prim.im[i][j] = (int) rint (dm[i][j]);
  }
  /* chain_alpha on 16-bit lines (:1870-1953) */
  g.alpha_kind = alpha_bits == 2 ? ALPHA_SET : alpha_bits == 4 ? ALPHA_MULT : ALPHA_NONE;
  g.alpha_value = (unsigned) (int) (255 * cfg.alpha_value);
  /* setup_gamma_encode (:1532-1565), 16 -> pack_bits */
  g.enc.clear ();
  if (out16) {
    g.enc16.resize (65536);
    for (int i = 0; i < 65536; i++)
      g.enc16[i] = (uint16_t) rint (transfer_encode (out->color_transfer, i / 65535.0) * 65535.0);
  } else {
    g.enc.resize (65536);
    for (int i = 0; i < 65536; i++)
      g.enc[i] = (uint8_t) rint (transfer_encode (out->color_transfer, i / 65535.0) * 255.0);
  }
  if (out16 && !pack_rgb) {
    /* chain_convert_to_YUV at pack_bits 16: identity / (1 << 16), the matrix to Y'CbCr, the pack format's 16-bit range */
    M44 dm;
    int offset[3], scale[3];
    double Kr = 0, Kb = 0;
    m_identity (dm);
    m_scale_components (dm, 1 / (float) 65536, 1 / (float) 65536, 1 / (float) 65536);
    if (cfg.matrix_mode != GSTAMD_MATRIX_MODE_NONE && get_Kr_Kb (cfg.matrix_mode == GSTAMD_MATRIX_MODE_INPUT_ONLY ? in_matrix : out_matrix, &Kr, &Kb))
      m_RGB_to_YCbCr (dm, Kr, Kb);
    range_offsets (out->color_range, true, offset, scale, 16);
    m_scale_components (dm, (float) scale[0], (float) scale[1], (float) scale[2]);
    m_offset_components (dm, offset[0], offset[1], offset[2]);
    if (!m_is_identity (dm)) {
      m_scale_components (dm, 256.0f, 256.0f, 256.0f);
      g.to_yuv16.has_matrix = 1;
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++)
          g.to_yuv16.im[i][j] = (int) rint (dm[i][j]);
    }
  }
  if (g.pack16)
    fill_pack_params (in, out, fi, fo, cfg, full_in_w, full_in_h, full_out_w, full_out_h, &g.pack);
  if (out16)
    setup_dither16 (cfg, fo, &g.dither16);
  /* chain_convert_to_YUV (:1955-2015): identity / (1 << 8), then the matrix to Y'CbCr and the output range */
  if (!pack_rgb && !out16) {
    M44 dm;
    int offset[3], scale[3];
    double Kr = 0, Kb = 0;
    m_identity (dm);
    m_scale_components (dm, 1 / (float) 256, 1 / (float) 256, 1 / (float) 256);
    if (cfg.matrix_mode != GSTAMD_MATRIX_MODE_NONE && get_Kr_Kb (cfg.matrix_mode == GSTAMD_MATRIX_MODE_INPUT_ONLY ? in_matrix : out_matrix, &Kr, &Kb))
      m_RGB_to_YCbCr (dm, Kr, Kb);
    range_offsets (out->color_range, true, offset, scale);
    m_scale_components (dm, (float) scale[0], (float) scale[1], (float) scale[2]);
    m_offset_components (dm, offset[0], offset[1], offset[2]);
    if (!m_is_identity (dm))
      prepare_matrix8 (dm, unpack_rgb, pack_rgb, &g.to_yuv, scratch);
  }
  /* chain_scale (:1685-1717) on the ARGB64 lines: u16 scalers, taps at 12 bits */
  plan->passes.clear ();
  const int in_w = in->width, in_h = in->height, out_w = out->width, out_h = out->height;
  const long s0 = (long) in_w * in_h, s3 = (long) out_w * out_h;
  g.shrink = s3 <= s0;
  if (in_w != out_w || in_h != out_h) {
    const long s1 = (long) out_w * in_h, s2 = (long) in_w * out_h;
    const bool h_first = s1 <= s2;
    for (int step = 0; step < 2; step++) {
      const bool horizontal = (step == 0) == h_first;
      const int isz = horizontal ? in_w : in_h, osz = horizontal ? out_w : out_h;
      if (isz == osz)
        continue;
      ScalePass pass;
      make_scale_pass (cfg.resampler_method, cfg.resampler_taps, cfg, isz, osz, horizontal, &pass, false, true);
      pass.max_span = 1 << 30;
      plan->passes.push_back (pass);
    }
  }
  plan->algorithmic_bytes = picture_bytes (fi, in_w, in_h) + picture_bytes (fo, out_w, out_h);
  g.fused = plan->passes.empty () && fo->kind == UNPACK_PACKED4 && !in16 && !out16;
  g.comp.clear ();
  if (g.fused && !g.prim.has_matrix && g.alpha_kind == ALPHA_NONE && g.dec.size () == 256 && g.enc.size () == 65536) {
    g.comp.resize (256);
    for (int i = 0; i < 256; i++)
      g.comp[i] = g.enc[g.dec[i]];
  }
  g.lut_direct = false;
  if (g.fused) {
    /* the direct conversion (crop, chroma upsampler as the real chain decides it, destination rectangle, borders, dither) minus matrix and
       alpha, which are the gamma chain's */
    g.cfg_in = base;
    g.cfg_in.chroma_mode = cfg.chroma_mode;
    g.mid_in = plan->orig_out;
  }
  if (g.fused && !g.comp.empty () && g.to_yuv.kind == MATRIX_NONE && !plan->rect.fill) {
    /* an RGB destination, nothing between the two tables: if the chain's own one-step plan for these frames (same options, no gamma, the
       chain rather than a transforms[] fastpath) has exactly to_rgb as its matrix and no dither stage, the remap is that conversion followed
       by the composed table on the colour bytes */
    GstAmdVideoConverterConfig direct = cfg;
    direct.gamma_mode = GSTAMD_GAMMA_MODE_NONE;
    direct.primaries_mode = GSTAMD_PRIMARIES_MODE_NONE;
    direct.internal_flags = 1;
    direct.src_x = direct.src_y = direct.src_width = direct.src_height = 0;
    direct.dest_x = direct.dest_y = direct.dest_width = direct.dest_height = 0;
    VideoPlan tmp;
    std::string terr;
    memset (&tmp.rect, 0, sizeof (tmp.rect));
    tmp.rect.in_maxw = in->width, tmp.rect.in_maxh = in->height;
    tmp.rect.out_maxw = out->width, tmp.rect.out_maxh = out->height;
    tmp.orig_in = *in;
    tmp.orig_out = *out;
    const int dbg_r = plan_core (in, out, &direct, &tmp, &terr);
    if (getenv ("GSTAMD_PLAN_DEBUG")) {
      fprintf (stderr, "lut_direct probe: r=%d (%s) gamma %d dither %d planar %d plane %d passes %zu alpha %d | kind %d/%d p %d %d %d %d %d / %d %d %d %d %d\n", dbg_r, terr.c_str (), tmp.gamma.on,
          tmp.dither.on, tmp.out_planar, tmp.plane_mode, tmp.passes.size (), tmp.post.alpha_kind, tmp.matrix.kind, g.to_rgb.kind, tmp.matrix.p[0], tmp.matrix.p[1], tmp.matrix.p[2],
          tmp.matrix.p[3], tmp.matrix.p[4], g.to_rgb.p[0], g.to_rgb.p[1], g.to_rgb.p[2], g.to_rgb.p[3], g.to_rgb.p[4]);
      for (int i = 0; i < 3; i++)
        fprintf (stderr, "  im %d %d %d %d / %d %d %d %d\n", tmp.matrix.im[i][0], tmp.matrix.im[i][1], tmp.matrix.im[i][2], tmp.matrix.im[i][3], g.to_rgb.im[i][0], g.to_rgb.im[i][1],
            g.to_rgb.im[i][2], g.to_rgb.im[i][3]);
    }
    if (dbg_r == GSTAMD_OK && !tmp.gamma.on && !tmp.dither.on && !tmp.out_planar && !tmp.plane_mode &&
        tmp.passes.empty () && tmp.post.alpha_kind == ALPHA_NONE && tmp.matrix.kind == g.to_rgb.kind && g.to_rgb.kind != MATRIX_NONE) {
      /* the chain's to_RGB matrix maps to 0 .. 256, the one-step conversion's to 0 .. 255: same form (kind), other coefficients - the
         direct plan is made with to_rgb in place of its own matrix (internal_flags & 2) */
      g.lut_direct = true;
      g.cfg_in = cfg;
      g.cfg_in.gamma_mode = GSTAMD_GAMMA_MODE_NONE;
      g.cfg_in.primaries_mode = GSTAMD_PRIMARIES_MODE_NONE;
      g.cfg_in.internal_flags = 3;
    }
  }
  plan->description = std::string (g.lut_direct ? "gamma_lut[" : g.fused ? "gamma_fused[" : "gamma_remap[") + fi->name + "->" + fo->name + (g.src16 ? ",front16" : "") + (in16 ? ",dec16" : "") +
      (g.to_rgb.kind || g.to_rgb16.has_matrix ? ",to_rgb" : "") + (g.prim.has_matrix ? ",primaries" : "") +
      (g.to_yuv.kind || g.to_yuv16.has_matrix ? ",to_yuv" : "") + (out16 ? ",enc16" : "") + (g.pack16 ? ",pack16" : g.store64 ? ",store64" : "") + (plan->passes.empty () ? "" : g.shrink ? ",scale16(first)" : ",scale16(last)") + "]";
  return GSTAMD_OK;
}



// GammaPlan::planes_fast: the chain degenerates to per-sample work when both formats are planar / semi-planar YUV on the same chroma
// grid, the size stays, no chroma resampler exists, no matrix, no crop
static void deep_planes_try (const GstAmdVideoInfo *in, const GstAmdVideoInfo *out, VideoPlan *plan, bool resampler)
{
  GammaPlan &g = plan->gamma;
  const FormatDesc *fi = plan->fin, *fo = plan->fout;
  g.planes_fast = false;
  if (!fi->yuv || !fo->yuv || !kind_has_planes (fi->kind) || !kind_has_planes (fo->kind) || hi_depth_be (fi->hi_depth) || hi_depth_be (fo->hi_depth) ||
      fi->w_sub != fo->w_sub || fi->h_sub != fo->h_sub ||
      in->width != out->width || in->height != out->height || resampler || g.prim.has_matrix || g.alpha_kind != ALPHA_NONE || plan->rect.in_x ||
      plan->rect.in_y || (plan->rect.in_maxw && (plan->rect.in_maxw != in->width || plan->rect.in_maxh != in->height)))
    return;
  /* pack_NV61 writes the last chroma pair of an odd-width line in NV16 order (video-format.c:2005-2011; PackPlanarParams::tail_swap):
     the general packer knows, this per-sample form does not */
  if (fo->format == GSTAMD_VIDEO_FORMAT_NV61 && (out->width & 1))
    return;
  DeepPlanesParams &d = g.planes;
  memset (&d, 0, sizeof (d));
  d.width = in->width;
  d.height = in->height;
  d.w_sub = fi->w_sub;
  d.h_sub = fi->h_sub;
  d.in_kind = fi->kind;
  d.out_kind = fo->kind;
  d.in_hi = fi->hi_depth;
  d.out_hi = fo->hi_depth;
  d.in_u = fi->u_plane;
  d.in_v = fi->v_plane;
  d.out_u = fo->u_plane;
  d.out_v = fo->v_plane;
  if (fo->hi_depth)
    d.dither = g.dither16;
  if (fo->hi_depth && g.dither16.on && g.dither16.method != GSTAMD_DITHER_BAYER)
    return;                     /* error diffusion runs over whole AYUV64 lines (launch_pack16_ed), not sample by sample */
  g.planes_fast = true;
}


// chain_downsample's decision + the geometry of a planar / semi-planar / 3-byte / packed 4:2:2 destination (the block plan_core fills
// plan->pack with), for plans that do not run through plan_core's tail
static void fill_pack_params (const GstAmdVideoInfo *in, const GstAmdVideoInfo *out, const FormatDesc *fi, const FormatDesc *fo, const GstAmdVideoConverterConfig &cfg,
    int full_in_w, int full_in_h, int full_out_w, int full_out_h, PackPlanarParams *pkp)
{
  PackPlanarParams &pk = *pkp;
  memset (&pk, 0, sizeof (pk));
  pk.width = out->width;
  pk.height = out->height;
  pk.kind = fo->kind;
  memcpy (pk.pos, fo->pos, sizeof (pk.pos));
  pk.tail_swap = (out->format == GSTAMD_VIDEO_FORMAT_VYUY || out->format == GSTAMD_VIDEO_FORMAT_NV61) && (out->width & 1);
  pk.w_sub = fo->w_sub;
  pk.h_sub = fo->h_sub;
  pk.u_plane = fo->u_plane;
  pk.v_plane = fo->v_plane;
  const bool differs = fi->w_sub != fo->w_sub || fi->h_sub != fo->h_sub || in->chroma_site != out->chroma_site || full_in_w != full_out_w ||
      full_in_h != full_out_h;
  if (differs && cfg.chroma_mode != GSTAMD_CHROMA_MODE_NONE && cfg.chroma_mode != GSTAMD_CHROMA_MODE_UPSAMPLE_ONLY) {
    if (fo->w_sub == 1)
      pk.down_h = (out->chroma_site & GSTAMD_CHROMA_SITE_H_COSITED) ? 2 : 1;
    if (fo->w_sub == 2)
      pk.down_h = (out->chroma_site & GSTAMD_CHROMA_SITE_H_COSITED) ? 4 : 3;
    if (fo->h_sub == 1 && !(out->chroma_site & GSTAMD_CHROMA_SITE_V_COSITED))
      pk.down_v = 1;
  }
}

// An ARGB64 / AYUV64 SOURCE (GammaPlan::src64): the frame is the first 16-bit image of the chain; what follows is the composite of the
// other 16-bit plans - u16 scalers, the convert matrix on 16-bit values, then by destination: the frame itself (store64), the 10-bit packer
// (pack16), or narrowing + a sub-conversion for the 8-bit tail.  plane_scale: the reference's convert_scale_planes fastpath (same format, no
// matrix): the 2-D scaler's own pass order.
static int plan_src64 (const GstAmdVideoInfo *in, const GstAmdVideoInfo *out, VideoPlan *plan, int alpha_bits, bool same_matrix, bool same_primaries, M44 prim_dm,
    int in_matrix, int out_matrix, bool plane_scale, std::string *error)
{
  auto fail = [&](int code, const std::string &msg) {
    if (error)
      *error = msg;
    return code;
  };
  const GstAmdVideoConverterConfig &cfg = plan->config;
  const FormatDesc *fi = plan->fin, *fo = plan->fout;
  GammaPlan &g = plan->gamma;
  if (cfg.gamma_mode == GSTAMD_GAMMA_MODE_REMAP && !plane_scale)
    return fail (GSTAMD_ERR_UNSUPPORTED, "gamma-mode = remap with a 16-bit unpack format is not implemented on the GPU path");
  g.on = true;
  g.src64 = true;
  g.src16 = false;
  g.store64 = fo->hi_depth == 3;
  g.pack16 = fo->hi_depth != 0 && !g.store64;
  g.fused = g.planes_fast = false;
  plan->plane_mode = plan->relayout = plan->fast_pair = plan->fast_enc420 = plan->fast_420p = plan->fast_422 = plan->fast_422_ayuv = plan->fast_post = false;
  plan->deep16 = plan->deep_out = false;
  plan->matrix_before_scale = false;
  plan->vpair.clear ();
  plan->planes.clear ();
  memset (&plan->front, 0, sizeof (plan->front));
  memset (&plan->matrix, 0, sizeof (plan->matrix));
  memset (&plan->post, 0, sizeof (plan->post));
  memset (&plan->pack, 0, sizeof (plan->pack));
  memset (&plan->deep, 0, sizeof (plan->deep));
  memset (&g.to_rgb, 0, sizeof (g.to_rgb));
  memset (&g.to_yuv, 0, sizeof (g.to_yuv));
  memset (&g.prim, 0, sizeof (g.prim));
  memset (&g.dither16, 0, sizeof (g.dither16));
  g.dec.clear ();
  g.enc.clear ();
  const int in_w = in->width, in_h = in->height, out_w = out->width, out_h = out->height;
  g.shrink = (long) out_w * out_h <= (long) in_w * in_h;
  /* the convert stage on 16-bit values (chain_convert with in_bits 16) */
  if (!plane_scale && (!same_matrix || !same_primaries)) {
    M44 dm;
    compute_convert_matrix_depth (in->color_range, in_matrix, out->color_range, out_matrix, fi->yuv, fo->yuv, cfg.matrix_mode, 16, dm,
        same_primaries ? nullptr : prim_dm, fo->hi_depth ? 16 : 8);
    if (!m_is_identity (dm)) {
      m_scale_components (dm, 256.0f, 256.0f, 256.0f);
      g.prim.has_matrix = 1;
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++)
          g.prim.im[i][j] = (int) rint (dm[i][j]);
    }
  }
  /* the alpha stage works on the lines the convert stage leaves: 16-bit ones for a 16-bit destination (here), 8-bit ones otherwise (the
     sub-conversion's) */
  g.alpha_kind = ALPHA_NONE;
  g.alpha_value = (unsigned) (int) (255 * cfg.alpha_value);
  if (fo->hi_depth != 0 && !plane_scale)         /* (alpha_bits is 0 already where the destination declares no alpha: planes, Y410) */
    g.alpha_kind = alpha_bits == 2 ? ALPHA_SET : alpha_bits == 4 ? ALPHA_MULT : ALPHA_NONE;
  /* scalers on the 16-bit lines: all of them for a 16-bit destination, the shrinking ones otherwise (an 8-bit tail scales what grows) */
  plan->passes.clear ();
  const bool scale_here = (in_w != out_w || in_h != out_h) && (fo->hi_depth != 0 || g.shrink);
  if (scale_here) {
    const long s1 = (long) out_w * in_h, s2 = (long) in_w * out_h;
    bool h_first = s1 <= s2;
    ScalePass hp, vp;
    if (in_w != out_w)
      make_scale_pass (cfg.resampler_method, cfg.resampler_taps, cfg, in_w, out_w, true, &hp, false, true);
    if (in_h != out_h)
      make_scale_pass (cfg.resampler_method, cfg.resampler_taps, cfg, in_h, out_h, false, &vp, false, true);
    hp.max_span = vp.max_span = 1 << 30;
    if (plane_scale && in_w != out_w && in_h != out_h)
      h_first = (long) out_w * (long) vp.offset[out_h - 1] <= (long) out_w * out_h;      /* gst_video_scaler_2d */
    for (int step = 0; step < 2; step++) {
      const bool horizontal = (step == 0) == h_first;
      if (horizontal ? in_w != out_w : in_h != out_h)
        plan->passes.push_back (horizontal ? hp : vp);
    }
  }
  video_info_set_format (&g.mid_in, GSTAMD_VIDEO_FORMAT_AYUV, in_w, in_h);      /* dimensions of the source image */
  const int full_out_w = plan->rect.out_maxw ? plan->rect.out_maxw : out_w, full_out_h = plan->rect.out_maxh ? plan->rect.out_maxh : out_h;
  if (fo->hi_depth == 0) {
    /* narrowing (video_orc_convert_u16_to_u8) into the 8-bit unpack format, then the sub-conversion */
    const int mw = scale_here || (in_w == out_w && in_h == out_h) ? out_w : in_w, mh = scale_here || (in_w == out_w && in_h == out_h) ? out_h : in_h;
    if (video_info_set_format (&g.mid_out, fo->yuv ? GSTAMD_VIDEO_FORMAT_AYUV : GSTAMD_VIDEO_FORMAT_ARGB, mw, mh) != GSTAMD_OK)
      return GSTAMD_ERR_INVALID;
    g.mid_out.color_range = out->color_range;
    g.mid_out.color_matrix = out->color_matrix;
    g.mid_out.chroma_site = out->chroma_site;
    g.sub_out_info = plan->orig_out;
    GstAmdVideoConverterConfig sub = cfg;
    sub.gamma_mode = GSTAMD_GAMMA_MODE_NONE;
    sub.primaries_mode = GSTAMD_PRIMARIES_MODE_NONE;
    sub.matrix_mode = GSTAMD_MATRIX_MODE_NONE;
    sub.internal_flags = 1;
    sub.src_x = sub.src_y = sub.src_width = sub.src_height = 0;
    const bool differs = fi->w_sub != fo->w_sub || fi->h_sub != fo->h_sub || in->chroma_site != out->chroma_site ||
        (plan->rect.in_maxw ? plan->rect.in_maxw : in_w) != full_out_w || (plan->rect.in_maxh ? plan->rect.in_maxh : in_h) != full_out_h;
    const bool down = differs && cfg.chroma_mode != GSTAMD_CHROMA_MODE_NONE && cfg.chroma_mode != GSTAMD_CHROMA_MODE_UPSAMPLE_ONLY;
    sub.chroma_mode = down ? GSTAMD_CHROMA_MODE_FULL : GSTAMD_CHROMA_MODE_NONE;
    g.cfg_out = sub;
    g.enc.resize (65536);
    for (int i = 0; i < 65536; i++)
      g.enc[i] = (uint8_t) (i >> 8);
  } else {
    video_info_set_format (&g.mid_out, GSTAMD_VIDEO_FORMAT_AYUV, out_w, out_h);     /* dimensions only */
    if (g.pack16) {
      fill_pack_params (in, out, fi, fo, cfg, in_w, in_h, full_out_w, full_out_h, &g.pack);
      g.pack_hi_depth = fo->hi_depth;
    }
    if (!plane_scale)
      setup_dither16 (cfg, fo, &g.dither16);          /* chain_dither :2035-2085 (the plane-scaling fastpath has no chain) */
  }
  plan->algorithmic_bytes = picture_bytes (fi, in_w, in_h) + picture_bytes (fo, out_w, out_h);
  plan->description = std::string ("deep64[") + fi->name + "->" + fo->name + (g.prim.has_matrix ? ",matrix16" : "") +
      (plan->passes.empty () ? "" : g.shrink ? ",scale16(first)" : ",scale16(last)") +
      (g.store64 ? ",store64" : g.pack16 ? ",pack16" : ",narrow") + (plane_scale ? "]{as convert_scale_planes}" : "]");
  return GSTAMD_OK;
}

// A 10-bit destination (GammaPlan with pack16): called at the end of plan_core, whose front / vpair / passes / pack it keeps where they
// apply (10-bit source) and replaces by a sub-conversion into the 8-bit unpack format where they do not (8-bit source).
static int finalize_deep_out (const GstAmdVideoInfo *in, const GstAmdVideoInfo *out, VideoPlan *plan, bool same_matrix, bool same_primaries, M44 prim_dm,
    int in_matrix, int out_matrix, int alpha_bits, std::string *error)
{
  const GstAmdVideoConverterConfig &cfg = plan->config;
  const FormatDesc *fi = plan->fin, *fo = plan->fout;
  GammaPlan &g = plan->gamma;
  if (!kind_has_planes (fo->kind) && fo->kind != UNPACK_PLANAR_A && fo->hi_depth != 3 && fo->kind != UNPACK_P422_16 && !GSTAMD_KIND_PX16 (fo->kind) && fo->kind != UNPACK_V210 &&
      !GSTAMD_KIND_LE32 (fo->kind)) {
    if (error)
      *error = "10-bit destination layout not implemented on the GPU path";
    return GSTAMD_ERR_UNSUPPORTED;
  }
  g.on = true;
  g.store64 = fo->hi_depth == 3;
  g.pack16 = !g.store64;
  g.src16 = fi->hi_depth != 0;
  g.pack = plan->pack;
  g.pack_hi_depth = fo->hi_depth;
  const int in_w = in->width, in_h = in->height, out_w = out->width, out_h = out->height;
  g.shrink = (long) out_w * out_h <= (long) in_w * in_h;
  memset (&g.to_rgb, 0, sizeof (g.to_rgb));
  memset (&g.to_yuv, 0, sizeof (g.to_yuv));
  /* chain_convert (:1803-1838) with out_bits 16: the matrix exists when the colour matrices or the primaries differ, on 16-bit values */
  memset (&g.prim, 0, sizeof (g.prim));
  if (!same_matrix || !same_primaries) {
    M44 dm;
    compute_convert_matrix_depth (in->color_range, in_matrix, out->color_range, out_matrix, fi->yuv, fo->yuv, cfg.matrix_mode, g.src16 ? 16 : 8, dm,
        same_primaries ? nullptr : prim_dm, 16);
    if (!m_is_identity (dm)) {
      m_scale_components (dm, 256.0f, 256.0f, 256.0f);
      g.prim.has_matrix = 1;
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++)
          g.prim.im[i][j] = (int) rint (dm[i][j]);
    }
  }
  g.alpha_kind = alpha_bits == 2 ? ALPHA_SET : alpha_bits == 4 ? ALPHA_MULT : ALPHA_NONE;
  g.alpha_value = (unsigned) (int) (255 * cfg.alpha_value);
  /* chain_dither (:2035-2085) on 16-bit lines: quantiser 1 << (16 - depth), or the target quantiser when that is coarser */
  setup_dither16 (cfg, fo, &g.dither16);
  g.dec.clear ();
  g.enc.clear ();
  video_info_set_format (&g.mid_out, fo->yuv ? GSTAMD_VIDEO_FORMAT_AYUV : GSTAMD_VIDEO_FORMAT_ARGB, out_w, out_h);      /* dimensions only */
  if (g.src16) {
    video_info_set_format (&g.mid_in, GSTAMD_VIDEO_FORMAT_AYUV, in_w, in_h);        /* dimensions of the front image */
  } else {
    /* the 8-bit part - unpack, chroma upsampler, the scalers when the picture shrinks (chain_scale's first call sits before the
       convert stage) - is a sub-conversion into the 8-bit unpack format */
    const int mw = g.shrink ? out_w : in_w, mh = g.shrink ? out_h : in_h;
    if (video_info_set_format (&g.mid_in, fi->yuv ? GSTAMD_VIDEO_FORMAT_AYUV : GSTAMD_VIDEO_FORMAT_ARGB, mw, mh) != GSTAMD_OK)
      return GSTAMD_ERR_INVALID;
    g.mid_in.color_range = in->color_range;
    g.mid_in.color_matrix = in->color_matrix;
    g.mid_in.chroma_site = in->chroma_site;
    g.sub_in_info = plan->orig_in;
    const int full_in_w = plan->rect.in_maxw ? plan->rect.in_maxw : in_w, full_in_h = plan->rect.in_maxh ? plan->rect.in_maxh : in_h;
    const bool differs = fi->w_sub != fo->w_sub || fi->h_sub != fo->h_sub || in->chroma_site != out->chroma_site ||
        full_in_w != (plan->rect.out_maxw ? plan->rect.out_maxw : out_w) || full_in_h != (plan->rect.out_maxh ? plan->rect.out_maxh : out_h);          /* FRAME sizes (video_converter_compute_resample :2866-2870) */
    const bool up = differs && cfg.chroma_mode != GSTAMD_CHROMA_MODE_NONE && cfg.chroma_mode != GSTAMD_CHROMA_MODE_DOWNSAMPLE_ONLY;
    g.cfg_in = cfg;
    g.cfg_in.gamma_mode = GSTAMD_GAMMA_MODE_NONE;
    g.cfg_in.primaries_mode = GSTAMD_PRIMARIES_MODE_NONE;
    g.cfg_in.matrix_mode = GSTAMD_MATRIX_MODE_NONE;
    g.cfg_in.alpha_mode = GSTAMD_ALPHA_MODE_COPY;
    g.cfg_in.alpha_value = 1.0;
    g.cfg_in.internal_flags = 1;
    g.cfg_in.dither_quantization = 1;
    g.cfg_in.chroma_mode = up ? GSTAMD_CHROMA_MODE_FULL : GSTAMD_CHROMA_MODE_NONE;
    g.cfg_in.dest_x = g.cfg_in.dest_y = g.cfg_in.dest_width = g.cfg_in.dest_height = 0;
    g.dec.resize (256);
    for (int i = 0; i < 256; i++)
      g.dec[i] = (uint16_t) (i * 257);                  /* video_orc_convert_u8_to_u16: mergebw d, s, s */
    /* the picture grows: the u16 scalers after the convert stage */
    plan->passes.clear ();
    if (!g.shrink && (in_w != out_w || in_h != out_h)) {
      const long s1 = (long) out_w * in_h, s2 = (long) in_w * out_h;
      const bool h_first = s1 <= s2;
      for (int step = 0; step < 2; step++) {
        const bool horizontal = (step == 0) == h_first;
        const int isz = horizontal ? in_w : in_h, osz = horizontal ? out_w : out_h;
        if (isz == osz)
          continue;
        ScalePass pass;
        make_scale_pass (cfg.resampler_method, cfg.resampler_taps, cfg, isz, osz, horizontal, &pass, false, true);
        pass.max_span = 1 << 30;
        plan->passes.push_back (pass);
      }
    }
    plan->vpair.clear ();
  }
  plan->algorithmic_bytes = picture_bytes (fi, in_w, in_h) + picture_bytes (fo, out_w, out_h);
  deep_planes_try (in, out, plan, g.pack.down_h || g.pack.down_v || plan->front.chroma_h != CHROMA_H_NONE || plan->front.chroma_v2);
  /* k_deep_planes writes whole planes from (0, 0) and knows no border: a destination rectangle (an origin, a frame wider / taller than the
     picture, borders to fill) goes through the pack16 tail below, which places the picture and fills convert_fill_border's lines */
  const bool out_rect = cfg.dest_x || cfg.dest_y || plan->rect.fill || plan->rect.out_x || plan->rect.out_y ||
      (plan->rect.out_maxw && (plan->rect.out_maxw != out_w || plan->rect.out_maxh != out_h));
  if (g.planes_fast && !out_rect) {
    plan->description = std::string ("deep_planes[") + fi->name + "->" + fo->name + (g.dither16.on ? ",dither" : "") + "]";
    return GSTAMD_OK;
  }
  g.planes_fast = false;
  if (g.store64) {
    plan->description = std::string ("deep_out[") + fi->name + "->" + fo->name + (g.src16 ? ",front16" : ",widen") + (g.prim.has_matrix ? ",matrix16" : "") +
        (plan->passes.empty () ? "" : g.shrink ? ",scale16(first)" : ",scale16(last)") + ",store64]";
    return GSTAMD_OK;
  }
  plan->description = std::string ("deep_out[") + fi->name + "->" + fo->name + (g.src16 ? ",front16" : ",widen") + (g.prim.has_matrix ? ",matrix16" : "") +
      (plan->passes.empty () ? "" : g.shrink ? ",scale16(first)" : ",scale16(last)") + (g.dither16.on ? ",dither" : "") + ",pack16[h" +
      std::to_string (g.pack.down_h) + ",v" + std::to_string (g.pack.down_v) + "]]";
  return GSTAMD_OK;
}


// A 10-bit source into an 8-bit planar / semi-planar / 3-byte destination: the 16-bit front and - when the picture shrinks - the u16
// scalers of this plan, the convert stage (matrix16 of plan->deep, then video_orc_convert_u16_to_u8 = the encode stage with the table
// v >> 8) into an 8-bit unpack-format image, and a sub-conversion for what the reference does on 8-bit lines afterwards: the scalers
// when the picture grows, chroma downsampler, dither, pack, destination rectangle.
static int finalize_deep_to_planar (const GstAmdVideoInfo *in, const GstAmdVideoInfo *out, VideoPlan *plan, int alpha_bits, std::string *error)
{
  (void) error;
  const GstAmdVideoConverterConfig &cfg = plan->config;
  const FormatDesc *fi = plan->fin, *fo = plan->fout;
  GammaPlan &g = plan->gamma;
  g.on = true;
  g.src16 = true;
  g.pack16 = false;
  const int in_w = in->width, in_h = in->height, out_w = out->width, out_h = out->height;
  g.shrink = (long) out_w * out_h <= (long) in_w * in_h;
  memset (&g.to_rgb, 0, sizeof (g.to_rgb));
  memset (&g.to_yuv, 0, sizeof (g.to_yuv));
  g.prim = plan->deep;
  g.alpha_kind = alpha_bits == 2 ? ALPHA_SET : alpha_bits == 4 ? ALPHA_MULT : ALPHA_NONE;
  g.alpha_value = (unsigned) (int) (255 * cfg.alpha_value);
  /* chain_alpha comes after the convert stage, on 8-bit lines here: (a8 * alpha) / 255 and the high byte of the 16-bit form agree only
     for set; mult is left to the sub-conversion */
  GstAmdVideoConverterConfig sub = cfg;
  sub.gamma_mode = GSTAMD_GAMMA_MODE_NONE;
  sub.primaries_mode = GSTAMD_PRIMARIES_MODE_NONE;
  sub.matrix_mode = GSTAMD_MATRIX_MODE_NONE;
  sub.internal_flags = 1;
  sub.src_x = sub.src_y = sub.src_width = sub.src_height = 0;
  g.alpha_kind = ALPHA_NONE;                            /* the destination has no alpha channel (planar / 3-byte formats) - or it is A420: */
  /* the alpha stage is the sub-conversion's then, which sees an AYUV / ARGB image where the real source may have no alpha: what
     convert_get_alpha_mode decided for the real pair is handed on as the mode that decides the same on the image */
  sub.alpha_mode = alpha_bits == 2 ? GSTAMD_ALPHA_MODE_SET : alpha_bits == 4 ? GSTAMD_ALPHA_MODE_MULT : GSTAMD_ALPHA_MODE_COPY;
  if (alpha_bits == 0 || alpha_bits == 1)
    sub.alpha_value = 1.0;
  const int full_in_w = plan->rect.in_maxw ? plan->rect.in_maxw : in_w, full_in_h = plan->rect.in_maxh ? plan->rect.in_maxh : in_h;
  const int full_out_w = plan->rect.out_maxw ? plan->rect.out_maxw : out_w, full_out_h = plan->rect.out_maxh ? plan->rect.out_maxh : out_h;
  const bool differs = fi->w_sub != fo->w_sub || fi->h_sub != fo->h_sub || in->chroma_site != out->chroma_site || full_in_w != full_out_w ||
      full_in_h != full_out_h;
  const bool down = differs && cfg.chroma_mode != GSTAMD_CHROMA_MODE_NONE && cfg.chroma_mode != GSTAMD_CHROMA_MODE_UPSAMPLE_ONLY;
  sub.chroma_mode = down ? GSTAMD_CHROMA_MODE_FULL : GSTAMD_CHROMA_MODE_NONE;
  g.cfg_out = sub;
  video_info_set_format (&g.mid_in, GSTAMD_VIDEO_FORMAT_AYUV, in_w, in_h);
  const int mw = g.shrink ? out_w : in_w, mh = g.shrink ? out_h : in_h;
  if (video_info_set_format (&g.mid_out, fo->yuv ? GSTAMD_VIDEO_FORMAT_AYUV : GSTAMD_VIDEO_FORMAT_ARGB, mw, mh) != GSTAMD_OK)
    return GSTAMD_ERR_INVALID;
  g.mid_out.color_range = out->color_range;
  g.mid_out.color_matrix = out->color_matrix;
  g.mid_out.chroma_site = out->chroma_site;
  g.sub_out_info = plan->orig_out;
  g.dec.clear ();
  g.enc.resize (65536);
  for (int i = 0; i < 65536; i++)
    g.enc[i] = (uint8_t) (i >> 8);                      /* video_orc_convert_u16_to_u8 */
  if (!g.shrink)
    plan->passes.clear ();                              /* the 8-bit scalers of the sub-conversion */
  deep_planes_try (in, out, plan, down || plan->front.chroma_h != CHROMA_H_NONE || plan->front.chroma_v2);
  if (g.planes_fast && !cfg.dest_x && !cfg.dest_y && !plan->rect.fill && !plan->dither.on && !plan->pack.dither.on) {
    plan->description = std::string ("deep_planes[") + fi->name + "->" + fo->name + "]";
    return GSTAMD_OK;
  }
  g.planes_fast = false;
  plan->description = std::string ("deep_in[") + fi->name + "->" + fo->name + ",front16" + (g.prim.has_matrix ? ",matrix16" : "") +
      (plan->passes.empty () ? "" : ",scale16(first)") + ",narrow]";
  return GSTAMD_OK;
}

static int plan_core (const GstAmdVideoInfo *in, const GstAmdVideoInfo *out,
    const GstAmdVideoConverterConfig *config, VideoPlan *plan, std::string *error)
{
  auto fail = [&](int code, const std::string &msg) {
    if (error)
      *error = msg;
    return code;
  };
  if (!in || !out || !plan)
    return fail (GSTAMD_ERR_INVALID, "NULL argument");
  plan->in_info = *in;
  plan->out_info = *out;
  if (config)
    plan->config = *config;
  else
    converter_config_init (&plan->config);
  const GstAmdVideoConverterConfig &cfg = plan->config;
  plan->fin = format_desc (in->format);
  plan->fout = format_desc (out->format);
  format_plan_planes (plan->fin, &plan->in_info);
  format_plan_planes (plan->fout, &plan->out_info);
  if (!plan->fin || !plan->fout)
    return fail (GSTAMD_ERR_UNSUPPORTED, "format not implemented on the GPU path");
  if (in->width <= 0 || in->height <= 0 || out->width <= 0 || out->height <= 0)
    return fail (GSTAMD_ERR_INVALID, "bad frame size");
  plan->out_planar = plan->fout->kind != UNPACK_PACKED4;
  plan->plane_mode = false;
  plan->relayout = false;
  plan->fast_422_ayuv = false;
  /* one field of an interlaced frame (plan_field_infos): the frame's heights decide what the reference's converter is made of */
  const int field = in->interlace_mode == GSTAMD_INTERLACE_FIELD_TOP ? 1 : (in->interlace_mode == GSTAMD_INTERLACE_FIELD_BOTTOM ? 2 : 0);
  plan->field = field;
  plan->interlaced = false;
  plan->field_src_chroma_frame = false;
  const int frame_in_h = field ? in->frame_height : in->height, frame_out_h = field ? out->frame_height : out->height;
  if (field) {
    if (out->interlace_mode != in->interlace_mode || frame_in_h < 2 || frame_out_h < 2)
      return fail (GSTAMD_ERR_INVALID, "field conversion: both infos must name the same field of frames of two lines or more");
    const RectPlan &rc = plan->rect;
    if (rc.in_x || rc.in_y || rc.out_x || rc.out_y || rc.fill || (rc.in_maxw && (rc.in_maxw != in->width || rc.in_maxh != in->height)) ||
        (rc.out_maxw && (rc.out_maxw != out->width || rc.out_maxh != out->height)))
      return fail (GSTAMD_ERR_UNSUPPORTED, "interlaced frames with a source crop or a destination rectangle are not implemented on the GPU path");
    /* GET_UV_420 / IS_CHROMA_LINE_420 with GST_VIDEO_PACK_FLAG_INTERLACED (video-format.c:1045-1052) address chroma row ((y & ~2) >> 1) | (y & 1): for a
       frame height that is not a multiple of four the last lines of the bottom field read (unpack) and write (pack) a row past the chroma planes */
    if ((plan->fin->h_sub == 1 && (frame_in_h & 3)) || (plan->fout->h_sub == 1 && (frame_out_h & 3)))
      return fail (GSTAMD_ERR_UNSUPPORTED, "interlaced 4:2:0 frames whose height is not a multiple of four: the reference's unpack / pack functions address a "
          "chroma row past the planes there; not reproduced");
    if (cfg.gamma_mode == GSTAMD_GAMMA_MODE_REMAP)
      return fail (GSTAMD_ERR_UNSUPPORTED, "gamma-mode = remap on interlaced frames is not implemented on the GPU path");
  }
  /* chain_dither (:2035-2100) on an 8-bit chain: a stage exists when dither-quantization asks for a coarser quantiser than the
   * format's own (1 at 8 bits) and the method is not NONE - NONE returns before anything is set up, quantisation included.  Every
   * component the destination has (depth > 0) gets the quantiser, rounded down to a power of two (count_power). */
  memset (&plan->dither, 0, sizeof (plan->dither));
  DitherParams planar_dither;
  memset (&planar_dither, 0, sizeof (planar_dither));
  if (plan->fout->hi_depth) {
    /* a 10-bit destination always has a dither stage unless the method is none (16-bit lines into 10-bit samples: quantiser 64):
       finalize_deep_out */
    /* the error-diffusion methods on 16-bit lines (video_dither_ed.h ed16_*): the same frame-line-0 rule as on 8-bit lines below */
    if (cfg.dither_method != GSTAMD_DITHER_NONE && cfg.dither_method != GSTAMD_DITHER_BAYER && plan->rect.out_y != 0)
      return fail (GSTAMD_ERR_UNSUPPORTED, "error-diffusion dither into a destination rectangle below the frame's first line: the reference never clears the "
          "error line there (y == 0 is the frame's line 0), so every frame depends on the one before");
  } else if ((cfg.dither_quantization > 1 || plan->fout->kind == UNPACK_RGB16) && cfg.dither_method != GSTAMD_DITHER_NONE) {
    /* verterr / floyd-steinberg / sierra-lite (video_dither_ed.h) clear their error line when the FRAME's line 0 comes by (video-dither.c:82,
       124, 192: y == 0, and y counts from the frame's top): with a destination rectangle that starts lower the errors of one frame's last
       line run into the next frame's first - the output would depend on the frames converted before */
    if (cfg.dither_method != GSTAMD_DITHER_BAYER && plan->rect.out_y != 0)
      return fail (GSTAMD_ERR_UNSUPPORTED, "error-diffusion dither into a destination rectangle below the frame's first line: the reference never clears the "
          "error line there (y == 0 is the frame's line 0), so every frame depends on the one before");
    int shift = 0;
    for (unsigned q = cfg.dither_quantization; q > 1; q >>= 1)
      shift++;
    /* every method masks with a guint16 (GstVideoDither::mask; the ordered one with FLAG_QUANTIZE runs on 16-bit sums of pixel + matrix value,
       video_orc_dither_ordered_4u8_mask): a quantiser of 512 and more clears every component it applies to.  (Round 4 capped the ordered
       method at 256 - found by the round-5 fuzz draws of dither-quantization 1024.) */
    if (shift > 16)
      shift = 16;
    if (plan->out_planar) {
      /* planar / semi-planar / 3-byte / packed 4:2:2 destinations: the stage sits between chroma downsampling and packing - it is
         part of the pack kernel (PackPlanarParams::dither, shift[] in unpack order); none of these formats has an alpha component */
      planar_dither.on = 1;
      planar_dither.y0 = plan->rect.out_y;
      planar_dither.method = cfg.dither_method;
      planar_dither.shift[0] = GSTAMD_KIND_ALPHA_PLANE (plan->fout->kind) >= 0 ? shift : 0;          /* A420's alpha plane is a component of depth 8 like the others */
      planar_dither.shift[1] = planar_dither.shift[2] = planar_dither.shift[3] = shift;
      if (plan->fout->kind == UNPACK_RGB16) {
        /* components of 5 / 6 bits on 8-bit lines: their own quantiser 1 << (8 - depth) (:2070-2078) where the target is not coarser */
        const int nat_g = 8 - plan->fout->pos[0];
        planar_dither.shift[1] = planar_dither.shift[3] = std::max (shift, 3);
        planar_dither.shift[2] = std::max (shift, nat_g);
      }
    } else {
      plan->dither.on = 1;
      plan->dither.y0 = plan->rect.out_y;
      plan->dither.method = cfg.dither_method;
      for (int comp = 0; comp < 4; comp++)
        plan->dither.shift[plan->fout->pos[comp]] = (comp == 0 && !plan->fout->alpha) ? 0 : shift;
    }
  }

  const bool unpack_rgb = !plan->fin->yuv, pack_rgb = !plan->fout->yuv;
  /* gst_video_converter_init_from_config (:2380-2404): RGB formats force the RGB matrix */
  int in_matrix = unpack_rgb ? GSTAMD_COLOR_MATRIX_RGB : in->color_matrix;
  int out_matrix = pack_rgb ? GSTAMD_COLOR_MATRIX_RGB : out->color_matrix;
  if (!unpack_rgb && (in_matrix == GSTAMD_COLOR_MATRIX_UNKNOWN || in_matrix == GSTAMD_COLOR_MATRIX_RGB))
    return fail (GSTAMD_ERR_INVALID, "YUV input needs a YUV colour matrix");
  if (!pack_rgb && (out_matrix == GSTAMD_COLOR_MATRIX_UNKNOWN || out_matrix == GSTAMD_COLOR_MATRIX_RGB))
    return fail (GSTAMD_ERR_INVALID, "YUV output needs a YUV colour matrix");

  /* convert_get_alpha_mode (:2264-2294); bits: COPY 1, SET 2, MULT 4 */
  int alpha_bits = 0;
  if (plan->fout->alpha) {
    bool decided = false;
    if (plan->fin->alpha) {
      if (cfg.alpha_mode == GSTAMD_ALPHA_MODE_COPY) {
        alpha_bits = 1;
        decided = true;
      } else if (cfg.alpha_mode == GSTAMD_ALPHA_MODE_MULT) {
        alpha_bits = cfg.alpha_value == 1.0 ? 1 : 4;
        decided = true;
      }
    }
    if (!decided)
      alpha_bits = cfg.alpha_value == 1.0 ? 0 : 2;
  }

  bool same_matrix = cfg.matrix_mode == GSTAMD_MATRIX_MODE_NONE ? true : in_matrix == out_matrix;
  bool same_primaries = cfg.primaries_mode == GSTAMD_PRIMARIES_MODE_NONE ? true : primaries_equivalent (in->color_primaries, out->color_primaries);
  M44 prim_dm;
  m_identity (prim_dm);
  if (!same_primaries)
    primaries_matrix (in->color_primaries, out->color_primaries, prim_dm);
  /* 10-bit sources: the reference unpacks them to AYUV64 and runs the chain on 16-bit lines until the convert stage narrows to
   * the 8-bit pack format.  Built so far: the chain into a 4-byte 8-bit destination (decoder output -> display), unscaled, shrunk
   * (scaled on the 16-bit lines, then converted) or enlarged (converted, then scaled on 8-bit lines). */
  plan->deep16 = plan->fin->hi_depth != 0;
  plan->deep_out = plan->fout->hi_depth != 0;
  /* Fastpaths of the reference that are the generic chain with two decisions forced, reproduced by forcing them here:
   *  - convert_I420_BGRA / _ARGB / _pack_ARGB (:6772-6990) and convert_I420_AYUV / Y42B_AYUV / Y444_AYUV (:3563, ..):
   *    chroma is sampled nearest (row y >> 1, loadupdb), no interpolation;
   *  - convert_I420_* and convert_AYUV_ARGB / _BGRA / _ABGR / _RGBA (:6544-6770): video_orc_convert_{I420,AYUV}_* with
   *    (im[0][0], im[0][2], im[2][1], im[1][1], im[1][2]) of video_converter_compute_matrix (:1444), whatever the
   *    matrix looks like - the arithmetic of video_orc_convert_AYUV_ARGB (video-orc.orc:1634, 1859);
   *  - convert_AYUV_I420 / _Y42B / _Y444 (:5575-5800; video-orc.orc:1445-1540): plain avgub of the chroma of the two lines,
   *    then of the pixel pair - the non-cosited downsampler whatever the chroma sites say;
   *  - convert_scale_planes on a one-plane 4-byte format (setup_scale :7958-8075, convert_plane_hv :7693): the plane
   *    goes through gst_video_scaler_2d as raw 4 x u8 pixels - no unpack / matrix / alpha / pack - and the order of the
   *    two passes is the 2-D scaler's own rule (video-scaler.c:1542-1545), not chain_scale's. */
  bool force_nearest = false, force_ayuv_argb = false, plane_scale = false, force_avg_down = false, order_2d = false, force_opaque = false;
  const char *fp = (cfg.internal_flags & 1) ? nullptr : lookup_fastpath (*plan, alpha_bits, same_matrix && same_primaries);
  /* setup_scale (:7985-8003), called by the lookup for every row without keeps_size before it looks at the sizes: RGB15 / 16 and the foreign-endian
     GRAY16 "only with nearest resampling" - the lookup fails and the whole chain runs (with its gamma / dither stages), even for a copy */
  if (fp && cfg.resampler_method != GSTAMD_RESAMPLER_METHOD_NEAREST) {
    if (plan->fin->kind == UNPACK_RGB16 && plan->fin == plan->fout)
      fp = nullptr;
    else if (plan->fin->kind == UNPACK_GRAY16 && plan->fin->hi_depth == 10)
      fp = nullptr;
  }
  /* (no fastpath has a dither stage: the keeps_interlaced ones - convert_UYVY_v210, convert_I420_v210 ... - run as they are) */
  if (field && !fp && (plan->dither.on || planar_dither.on || (plan->fout->hi_depth && cfg.dither_method != GSTAMD_DITHER_NONE && plan->fin->format != plan->fout->format)))
    return fail (GSTAMD_ERR_UNSUPPORTED, "a dither stage on interlaced frames (its pattern follows the frame's line numbers) is not implemented on the GPU path");
  plan->gamma.on = false;
  if (cfg.gamma_mode == GSTAMD_GAMMA_MODE_REMAP) {
    /* video_converter_lookup_fastpath :8940-8944: "fastpaths don't do gamma" - they are only looked at for a same-size conversion
     * between equivalent transfer functions, and when one matches it runs (without any gamma step); everything else takes the chain
     * with the decode / encode tables */
    const bool eq = transfer_equivalent (in->color_transfer, hi_depth_bits (plan->fin->hi_depth), out->color_transfer, hi_depth_bits (plan->fout->hi_depth));
    if (!(plan->ref_same_size && eq))
      fp = nullptr;
    if (!fp)
      return plan_gamma (in, out, plan, alpha_bits, same_primaries, prim_dm, error);
  }
  if (fp && plan->fin->kind == UNPACK_GRAY16 && (in->width != out->width || in->height != out->height) &&
      cfg.resampler_method != GSTAMD_RESAMPLER_METHOD_NEAREST) {
    /* native 16-bit samples go through the u16 plane scalers, which this library has not built: the chain's u16 passes in gst_video_scaler_2d's
       order, nothing else between unpack and pack */
    order_2d = true;
    fp = nullptr;
  }
  if (field && plan->fin->hi_depth == 3 && !(fp && strcmp (fp, "convert_scale_planes") == 0 && in->width == out->width && frame_in_h == frame_out_h))
    return fail (GSTAMD_ERR_UNSUPPORTED, "interlaced ARGB64 / AYUV64 sources are not implemented on the GPU path");
  if (plan->fin->hi_depth == 3)
    return plan_src64 (in, out, plan, alpha_bits, same_matrix, same_primaries, prim_dm, in_matrix, out_matrix, fp != nullptr, error);
  if (fp) {
    const std::string name = fp;
    const int ki = plan->fin->kind, ko = plan->fout->kind;
    /* no fastpath converts primaries: the rows without needs_color_matrix are only taken when they are equivalent, and the others build
     * their matrix with video_converter_compute_matrix (:2837-2847) - to RGB, to YUV, nothing between - whatever primaries-mode says */
    same_primaries = true;
    m_identity (prim_dm);
    if (name == "convert_8bit_v210" || name == "convert_v210_8bit") {
      plan->v210_fast = true;
      plan->plane_mode = plan->relayout = plan->fast_pair = plan->fast_enc420 = plan->fast_420p = plan->fast_422 = plan->fast_422_ayuv = plan->fast_post = false;
      plan->deep16 = plan->deep_out = plan->matrix_before_scale = false;
      plan->planes.clear ();
      memset (&plan->front, 0, sizeof (plan->front));
      memset (&plan->matrix, 0, sizeof (plan->matrix));
      memset (&plan->post, 0, sizeof (plan->post));
      memset (&plan->pack, 0, sizeof (plan->pack));
      memset (&plan->deep, 0, sizeof (plan->deep));
      memset (&plan->dither, 0, sizeof (plan->dither));
      plan->ref_fastpath = name == "convert_8bit_v210" ? std::string ("convert_") + plan->fin->name + "_v210" : std::string ("convert_v210_") + plan->fout->name;
      plan->passes.clear ();
      plan->vpair.clear ();
      plan->algorithmic_bytes = picture_bytes (plan->fin, in->width, in->height) + picture_bytes (plan->fout, out->width, out->height);
      plan->description = std::string ("v210_fast[") + plan->fin->name + "->" + plan->fout->name + "]{as " + plan->ref_fastpath + "}";
      return GSTAMD_OK;
    }
    if (name == "convert_I420_xRGB")
      force_nearest = force_ayuv_argb = true;
    else if (name == "convert_I420_xRGB(opaque)") {
      /* video_orc_convert_I420_BGRA & co on A420's first three planes: the destination's fourth byte is 0xff, whatever the alpha plane holds */
      force_nearest = force_ayuv_argb = force_opaque = true;
      fp = "convert_I420_xRGB";
    }
    else if (name == "convert_AYUV_xRGB")
      force_ayuv_argb = true;
    else if (name == "convert_I420_AYUV" || name == "convert_Y42B_AYUV" || name == "convert_Y444_AYUV" || name == "convert_YUY2_AYUV" ||
        name == "convert_UYVY_AYUV")
      force_nearest = true;
    else if (name == "convert_AYUV_I420" || name == "convert_AYUV_422" || name == "convert_AYUV_Y444")
      force_avg_down = true;
    else if (name == "convert_I420_YUY2" || name == "convert_YUY2_I420" || name == "convert_Y42B_YUY2" || name == "convert_Y444_YUY2" ||
        name == "convert_YUY2_planar" || name == "convert_UYVY_YUY2")
      /* video_orc_convert_{I420,Y42B,Y444}_{YUY2,UYVY}, _{YUY2,UYVY}_{I420,Y42B,Y444}, _UYVY_YUY2 (video-orc.orc:1193-1620): chroma is
       * duplicated where the destination has more of it and avgub'ed (lines, then pixel pairs) where it has less */
      force_nearest = force_avg_down = true;
    else if (name == "convert_UYVY_GRAY8") {
      /* video_orc_convert_UYVY_GRAY8 (video-orc.orc:2937-2946; :5565-5606): the luma bytes as they are, whatever the two colour matrices say
         (needs_color_matrix is set on the row).  The function starts at the two FRAMES' first pixels - crop and rectangle origins are not
         applied - and then fills the border around the rectangle over part of what it wrote: only the uncropped form is reproduced */
      const RectPlan &rc = plan->rect;
      if (rc.in_x || rc.in_y || rc.out_x || rc.out_y || rc.fill || (rc.in_maxw && (rc.in_maxw != in->width || rc.in_maxh != in->height)) ||
          (rc.out_maxw && (rc.out_maxw != out->width || rc.out_maxh != out->height)))
        return fail (GSTAMD_ERR_UNSUPPORTED, "the reference's convert_UYVY_GRAY8 ignores the crop and rectangle origins; not reproduced");
      same_matrix = same_primaries = true;
    } else if (name == "convert_scale_planes" && ki == UNPACK_PACKED4)
      plane_scale = true;
    else if (name == "convert_scale_planes" && (ki == ko || (kind_has_planes (ki) && kind_has_planes (ko)) || (ki == UNPACK_GRAY && ko == UNPACK_PLANAR) ||
            (ki == UNPACK_PLANAR && ko == UNPACK_GRAY) || ki == UNPACK_PLANAR_A || ko == UNPACK_PLANAR_A || ki == UNPACK_PLANAR_H4 || ko == UNPACK_PLANAR_H4))
      return plan_planes (plan, fp);
    else
      return fail (GSTAMD_ERR_UNSUPPORTED, std::string ("reference takes fastpath ") + fp +
          " for this conversion; no GPU kernel for it yet");
    plan->ref_fastpath = fp;
    if ((force_nearest || force_ayuv_argb || force_avg_down) && (in->width != out->width || frame_in_h != frame_out_h))
      return fail (GSTAMD_ERR_UNSUPPORTED, "the reference selects a same-size fastpath by the uncropped input size while the crop differs from the "
          "destination rectangle; not reproduced");
  }

  // ---- front: unpack + chroma upsample (chain_unpack_line, chain_upsample) ----------------------
  const int full_in_w = plan->rect.in_maxw ? plan->rect.in_maxw : in->width, full_in_h = field ? frame_in_h : (plan->rect.in_maxh ? plan->rect.in_maxh : in->height);
  const int full_out_w = plan->rect.out_maxw ? plan->rect.out_maxw : out->width, full_out_h = field ? frame_out_h : (plan->rect.out_maxh ? plan->rect.out_maxh : out->height);
  FrontParams &fr = plan->front;
  memset (&fr, 0, sizeof (fr));
  fr.kind = plan->fin->kind;
  fr.width = in->width;
  fr.height = in->height;
  fr.w_sub = plan->fin->w_sub;
  fr.h_sub = plan->fin->h_sub;
  fr.u_plane = plan->fin->u_plane;
  fr.v_plane = plan->fin->v_plane;
  memcpy (fr.pos, plan->fin->pos, sizeof (fr.pos));
  fr.chroma_h = CHROMA_H_NONE;
  fr.chroma_v2 = 0;
  fr.swap_k = in->format == GSTAMD_VIDEO_FORMAT_VYUY && (in->width & 1) ? (in->width - 1) >> 1 : -1;
  fr.hi_depth = plan->fin->hi_depth;
  fr.luma_last = (plan->rect.in_maxh ? plan->rect.in_maxh : in->height) - 1 - plan->rect.in_y;
  /* video_converter_compute_resample (:2850-2895) + gst_video_chroma_resample_new (video-chroma.c:1041-1109) */
  if (cfg.chroma_mode != GSTAMD_CHROMA_MODE_NONE && cfg.chroma_mode != GSTAMD_CHROMA_MODE_DOWNSAMPLE_ONLY
      && plan->fin->yuv) {
    /* the FULL frame sizes are compared (in_info / out_info), not the crop / destination rectangles */
    bool differs = plan->fin->w_sub != plan->fout->w_sub || plan->fin->h_sub != plan->fout->h_sub ||
        in->chroma_site != out->chroma_site || full_in_w != full_out_w || full_in_h != full_out_h;
    if (differs && (fr.w_sub || fr.h_sub)) {
      if (fr.w_sub == 1)
        fr.chroma_h = (in->chroma_site & GSTAMD_CHROMA_SITE_H_COSITED) ? CHROMA_H_H2_CS : CHROMA_H_H2;
      if (fr.w_sub == 2)
        fr.chroma_h = (in->chroma_site & GSTAMD_CHROMA_SITE_H_COSITED) ? CHROMA_H_H4_CS : CHROMA_H_H4;
      if (fr.h_sub == 1 && !(in->chroma_site & GSTAMD_CHROMA_SITE_V_COSITED))
        fr.chroma_v2 = 1;      /* video_chroma_up_v2_u8; the cosited variant is an h-only stub */
    }
  }
  if (force_nearest) {
    fr.chroma_h = CHROMA_H_NONE;
    fr.chroma_v2 = 0;
  }
  /* a field: video_chroma_up_vi2_u8 where the frame's chain has the v2 upsampler (v_index + 16, video-chroma.c:1087); the cosited one is a stub as well */
  const bool field_up_v = field && fr.chroma_v2 != 0;

  // ---- scaling (chain_scale :1685-1717 decides WHERE and in which ORDER) -------------------------
  plan->passes.clear ();
  const int in_w = in->width, in_h = in->height, out_w = out->width, out_h = out->height;
  /* (a field: the sizes chain_scale and gst_video_scaler_2d compare are the frame's) */
  const int dec_in_h = field ? frame_in_h : in_h, dec_out_h = field ? frame_out_h : out_h;
  const long s0 = (long) in_w * dec_in_h, s3 = (long) out_w * dec_out_h;
  const bool need_scale = in_w != out_w || dec_in_h != dec_out_h;
  plan->matrix_before_scale = need_scale && !(s3 <= s0);
  std::vector<uint32_t> field_zip;          /* the frame's zipped vertical scaler: first line of every output line */
  int field_zip_lines = 0;
  if (need_scale) {
    const long s1 = (long) out_w * dec_in_h, s2 = (long) in_w * dec_out_h;
    bool h_first = s1 <= s2;
    if ((plane_scale || order_2d) && in_w != out_w && dec_in_h != dec_out_h) {
      /* gst_video_scaler_2d: horizontal first iff width * voffset[height - 1] <= width * height */
      if (field) {
        int zip_last = 0;
        if (!make_field_vpass (cfg.resampler_method, cfg.resampler_taps, cfg, frame_in_h, frame_out_h, 0, nullptr, false, &zip_last, nullptr))
          return fail (GSTAMD_ERR_UNSUPPORTED, "the reference cannot make an interlaced scaler for these heights");
        h_first = (long) out_w * (long) zip_last <= (long) out_w * frame_out_h;
      } else {
      ScalePass vp;
      make_scale_pass (cfg.resampler_method, cfg.resampler_taps, cfg, in_h, out_h, false, &vp);
      h_first = (long) out_w * (long) vp.offset[out_h - 1] <= (long) out_w * out_h;
      }
    }
    for (int step = 0; step < 2; step++) {
      bool horizontal = (step == 0) == h_first;
      int isz = horizontal ? in_w : dec_in_h, osz = horizontal ? out_w : dec_out_h;
      if (isz == osz)
        continue;
      ScalePass pass;
      if (field && !horizontal) {
        /* chain_vscale (:1651-1660) / setup_scale (:8075): GST_VIDEO_SCALER_FLAG_INTERLACED - this field's resampler over the field's lines */
        if (!make_field_vpass (cfg.resampler_method, cfg.resampler_taps, cfg, frame_in_h, frame_out_h, field - 1, &pass,
                plan->deep16 && (!plan->matrix_before_scale || plan->deep_out), nullptr, &field_zip))
          return fail (GSTAMD_ERR_UNSUPPORTED, "the reference cannot make an interlaced scaler for these heights");
        field_zip_lines = 2 * pass.n_taps;
        if (pass.in_size != in_h || pass.out_size != out_h)
          return fail (GSTAMD_ERR_INVALID, "field conversion: the infos' heights are not this field's");
      } else
      make_scale_pass (cfg.resampler_method, cfg.resampler_taps, cfg, isz, osz, horizontal, &pass, false,
          plan->deep16 && (!plan->matrix_before_scale || plan->deep_out));
      pass.max_span = 1 << 30;
      if (horizontal) {
        int worst = 0;
        for (int t0 = 0; t0 < osz; t0 += 256) {
          const int t1 = std::min (t0 + 256, osz);
          int lo, hi;
          if (pass.kind == SCALE_2TAP) {
            lo = (t0 * pass.inc) >> 16;
            hi = (((t1 - 1) * pass.inc) >> 16) + 2;
          } else {
            lo = (int) pass.offset[t0];
            hi = (int) pass.offset[t1 - 1] + (pass.kind == SCALE_NEAREST ? 1 : pass.n_taps);
          }
          worst = std::max (worst, hi - lo);
        }
        pass.max_span = worst;
      }
      plan->passes.push_back (pass);
    }
  }

  /* unpack_VYUY hands lines that are not 8-byte aligned to a C loop (video-format.c: `if (IS_ALIGNED (d, 8)) video_orc_unpack_VYUY ... else`) which
     stores V where U belongs and U where V does - on EVERY macropixel, like the odd-width tail does on the last one.  With a border to fill the
     destination-side lines start out_x pixels into their allocation (get_border_temp_line :717-730: out_x * pack_pstride, 4 bytes a pixel on an 8-bit
     pack), and setup_allocators hands that allocator back through every stage that works in place - up to the unpacker unless a scaler that makes
     new lines sits in between (every filter; the nearest VERTICAL scaler hands its input line on).  An odd out_x is then such a line: unscaled
     chains and chains with only a nearest vertical pass.  (Found by the device fuzz, seed 4832; the boundary from sweeps over the pass kinds.) */
  {
    bool new_lines = false;
    for (const ScalePass &sp : plan->passes)
      new_lines = new_lines || sp.horizontal || sp.kind != SCALE_NEAREST;
    /* a destination in its unpack format (ARGB / AYUV: identity_pack) lends its OWN rows instead (get_dest_line: row + out_x * 4, border or not); with
       rows on 8 bytes that is the same rule, otherwise the alignment changes from row to row (announced below) */
    const bool own_rows = plan->fout->format == GSTAMD_VIDEO_FORMAT_ARGB || plan->fout->format == GSTAMD_VIDEO_FORMAT_AYUV;
    const bool rows_on_8 = (out->stride[0] % 8) == 0 && (out->offset[0] % 8) == 0;
    if (in->format == GSTAMD_VIDEO_FORMAT_VYUY && (plan->rect.out_x & 1) && plan->fout->hi_depth == 0 && !new_lines && (own_rows ? rows_on_8 : plan->rect.fill)) {
      std::swap (fr.pos[2], fr.pos[3]);
      fr.swap_k = -1;
    }
  }
  /* Reference quirk: when the vertical scaler comes first and pulls straight from the 2-line chroma
   * upsampler, the unpack temp-line ring (setup_allocators :2115-2187, sized MAX(n_taps, 5)) is one
   * line short: the pair mate of the window's last line overwrites the window's first line, so the
   * reference output depends on buffer reuse (verified against oracle/_ref: >15 % of bytes differ
   * from the intended filter).  There is nothing to reproduce: the plan computes the intended filter (what the reference gives when
   * the same chain is run as two conversions, in -> AYUV at the source size and AYUV -> out) and says so in plan->divergence. */
  if (!field && !plan->passes.empty () && !plan->passes[0].horizontal && fr.chroma_v2 && plan->passes[0].n_taps >= 5)
    plan->divergence += "vertical-first N-tap scaling fed by the 4:2:0 chroma upsampler: the reference's unpack ring is one line short "
        "(temp-line aliasing, its output depends on buffer reuse); this library applies the filter to the lines the chain intends. ";

  // ---- colour matrix (chain_convert :1719-1868) --------------------------------------------------
  memset (&plan->matrix, 0, sizeof (plan->matrix));
  memset (&plan->deep, 0, sizeof (plan->deep));
  if (plan->deep16) {
    /* chain_convert with in_bits 16, out_bits 8: the matrix (when the colour matrices differ) is prepared for current_bits 16 ->
     * video_converter_matrix16 on integers rint (m * 256) (prepare_matrix :1323-1370); without it the stage only narrows */
    if (!same_matrix || !same_primaries) {
      M44 dm;
      compute_convert_matrix_depth (in->color_range, in_matrix, out->color_range, out_matrix, plan->fin->yuv, plan->fout->yuv, cfg.matrix_mode, 16, dm,
          same_primaries ? nullptr : prim_dm, plan->deep_out ? 16 : 8);
      bool identity = true;
      for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
          identity = identity && dm[i][j] == (i == j ? 1.0 : 0.0);
      if (!identity) {
        m_scale_components (dm, 256.0f, 256.0f, 256.0f);      /* SCALE_F */
        plan->deep.has_matrix = 1;
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 4; j++)
            plan->deep.im[i][j] = plan->im_raw[i][j] = (int) rint (dm[i][j]);
      }
    }
  } else if (!same_matrix || !same_primaries || force_ayuv_argb) {
    M44 dm;
    compute_convert_matrix_depth (in->color_range, in_matrix, out->color_range, out_matrix, plan->fin->yuv, plan->fout->yuv, cfg.matrix_mode, 8, dm,
        same_primaries ? nullptr : prim_dm);
    prepare_matrix8 (dm, unpack_rgb, pack_rgb, &plan->matrix, plan->im_raw);
    if (force_ayuv_argb) {             /* the ORC parameters are 16-bit (.param 2) */
      const int (*im)[4] = plan->im_raw;
      plan->matrix.kind = MATRIX_AYUV_ARGB;
      plan->matrix.p[0] = (int16_t) im[0][0];
      plan->matrix.p[1] = (int16_t) im[0][2];
      plan->matrix.p[2] = (int16_t) im[2][1];
      plan->matrix.p[3] = (int16_t) im[1][1];
      plan->matrix.p[4] = (int16_t) im[1][2];
    }
  }

  // ---- alpha + pack ------------------------------------------------------------------------------
  PostParams &post = plan->post;
  memset (&post, 0, sizeof (post));
  if ((cfg.internal_flags & 2) && g_matrix_override)
    plan->matrix = *g_matrix_override;          /* GammaPlan::lut_direct: the direct plan with the gamma chain's to_RGB matrix in place of its own */
  post.matrix = plan->matrix;
  post.alpha_kind = alpha_bits == 2 ? ALPHA_SET : alpha_bits == 4 ? ALPHA_MULT : ALPHA_NONE;
  post.alpha_value = (int) (255 * cfg.alpha_value);
  if (force_opaque)
    post.alpha_kind = ALPHA_SET, post.alpha_value = 255;
  memcpy (post.pack_pos, plan->fout->pos, sizeof (post.pack_pos));
  memset (&plan->pack, 0, sizeof (plan->pack));
  if (plan->out_planar) {
    /* the chain ends in AYUV (the unpack format of every planar YUV format); chain_downsample (:2040) +
     * video_converter_compute_resample (:2850-2895): the downsampler exists when anything about the chroma grid
     * differs between input and output and the chroma mode allows it */
    const FormatDesc *fo = plan->fout;
    PackPlanarParams &pk = plan->pack;
    pk.width = out->width;
    pk.height = out->height;
    pk.kind = fo->kind;
    memcpy (pk.pos, fo->pos, sizeof (pk.pos));
    pk.tail_swap = (out->format == GSTAMD_VIDEO_FORMAT_VYUY || out->format == GSTAMD_VIDEO_FORMAT_NV61) && (out->width & 1);
    pk.w_sub = fo->w_sub;
    pk.h_sub = fo->h_sub;
    pk.u_plane = fo->u_plane;
    pk.v_plane = fo->v_plane;
    const bool differs = plan->fin->w_sub != fo->w_sub || plan->fin->h_sub != fo->h_sub ||
        in->chroma_site != out->chroma_site || full_in_w != full_out_w || full_in_h != full_out_h;
    if (differs && cfg.chroma_mode != GSTAMD_CHROMA_MODE_NONE && cfg.chroma_mode != GSTAMD_CHROMA_MODE_UPSAMPLE_ONLY) {
      if (fo->w_sub == 1)
        pk.down_h = (out->chroma_site & GSTAMD_CHROMA_SITE_H_COSITED) ? 2 : 1;
      if (fo->w_sub == 2)
        pk.down_h = (out->chroma_site & GSTAMD_CHROMA_SITE_H_COSITED) ? 4 : 3;
      if (fo->h_sub == 1 && !(out->chroma_site & GSTAMD_CHROMA_SITE_V_COSITED))
        pk.down_v = 1;          /* the cosited vertical variant is an h-only stub (video-chroma.c:996) */
    }
    if (field)
      pk.down_v = 0;          /* video_chroma_down_vi2_u8 is a stub (video-chroma.c:461-470): every line keeps its chroma, pack_planar_420 & co take the field's
                                 even lines' (IS_CHROMA_LINE_420 with the interlaced flag) - this field's lines 0, 2, ... into its chroma rows */
    if (force_avg_down) {
      pk.down_h = fo->w_sub == 1 ? 1 : 0;
      pk.down_v = fo->h_sub == 1 ? 1 : 0;          /* (a field: GET_LINE_OFFSETS pairs lines l and l + 2, two consecutive lines of the field, :3383) */
    }
    for (int i = 0; i < 4; i++)
      post.pack_pos[i] = i;
    pk.dither = planar_dither;
    if (!plan->ref_fastpath.empty ())
      memset (&pk.dither, 0, sizeof (pk.dither));          /* no fastpath has a dither stage (they are only looked up at dither-quantization 1, where RGB15 / 16 alone still have one in the chain) */
  }
  if (plane_scale) {                   /* raw bytes in, raw bytes out */
    for (int i = 0; i < 4; i++)
      fr.pos[i] = post.pack_pos[i] = i;
    post.alpha_kind = ALPHA_NONE;
    plan->matrix_before_scale = false;
  }

  /* an odd-height 4:2:0 destination fed by the vertical chroma upsampler at the same size: video_orc_chroma_down_v2 averages the last
     line with the line PAST the picture, which the chain makes like any other - do_unpack_lines clamps it to the last line (:2966), the
     upsampler pairs it with the (clamped) line after it - so its chroma is the last chroma row unblended.  The AYUV image gets that line
     as one more row (k_convert's grid is one row taller), the pack kernel reads it (PackPlanarParams::virtual_line). */
  /* With a source crop that ends above the frame's last line the line past the picture is a real one (do_unpack_lines clamps to the FRAME),
     whatever the source format: the same extra row then. */
  plan->pack.virtual_line = plan->out_planar && plan->passes.empty () && (fr.chroma_v2 || fr.luma_last >= in_h) && plan->pack.down_v && plan->pack.h_sub == 1 &&
      (out_h & 1) && in_h == out_h && !plan->deep16 && !plan->deep_out && !field ? 1 : 0;
  if (field && fr.h_sub == 1 && (kind_has_planes (fr.kind) || GSTAMD_KIND_ALPHA_PLANE (fr.kind) >= 0)) {
    const bool dest_rows = plan->passes.empty () && !plan->deep16 && (plan->fout->format == GSTAMD_VIDEO_FORMAT_ARGB || plan->fout->format == GSTAMD_VIDEO_FORMAT_AYUV);
    simulate_vpairs_field (plan, frame_in_h, frame_out_h, field - 1, field_up_v, field_zip.empty () ? nullptr : &field_zip, field_zip_lines, dest_rows);
    fr.chroma_v2 = 2;
    plan->field_src_chroma_frame = true;
  } else
  simulate_vpairs (plan, out_h, plan->pack.virtual_line != 0);

  plan->algorithmic_bytes = 0;
  {
    /* source planes read once + destination written once */
    uint64_t rd = 0;
    const FormatDesc *f = plan->fin;
    if (f->kind == UNPACK_PACKED4)
      rd = (uint64_t) in_w * in_h * 4;
    else if (f->kind == UNPACK_PACKED3)
      rd = (uint64_t) in_w * in_h * 3;
    else if (f->kind == UNPACK_GRAY)
      rd = (uint64_t) in_w * in_h;
    else {
      uint64_t cw = ((uint64_t) in_w + (1 << f->w_sub) - 1) >> f->w_sub, ch = ((uint64_t) in_h + (1 << f->h_sub) - 1) >> f->h_sub;
      rd = (uint64_t) in_w * in_h + 2 * cw * ch;
      if (f->hi_depth)
        rd *= 2;                /* 16-bit words */
    }
    uint64_t wr = (uint64_t) out_w * out_h * 4;
    if (plan->out_planar) {
      const FormatDesc *fo = plan->fout;
      uint64_t cw = ((uint64_t) out_w + (1 << fo->w_sub) - 1) >> fo->w_sub, ch = ((uint64_t) out_h + (1 << fo->h_sub) - 1) >> fo->h_sub;
      wr = fo->kind == UNPACK_PACKED3 ? (uint64_t) out_w * out_h * 3 : fo->kind == UNPACK_GRAY ? (uint64_t) out_w * out_h : (uint64_t) out_w * out_h + 2 * cw * ch;
    }
    plan->algorithmic_bytes = rd + wr;
  }
  /* the line-pair kernel (video_fast.h): unscaled 4:2:0 semi-planar -> 4-byte RGB through the AYUV_ARGB
   * matrix, with ORC's 16-bit addw wrap provably out of reach: |mulhsw (s, p)| <= (32896 * |p| >> 16) + 1 */
  bool matrix_no_wrap = false, matrix_no_wrap_core = false;          /* _core: whatever the destination's byte order */
  if (plan->matrix.kind == MATRIX_AYUV_ARGB && post.alpha_kind == ALPHA_NONE) {
    long t[5];
    bool fits = true;
    for (int i = 0; i < 5; i++) {
      long ap = labs ((long) plan->matrix.p[i]);
      fits = fits && ap < 32768;
      t[i] = ((32896L * ap) >> 16) + 1;
    }
    long worst = t[0] + std::max (std::max (t[1], t[2]), t[3] + t[4]);
    matrix_no_wrap = fits && worst < 32000;
    matrix_no_wrap_core = matrix_no_wrap;
    /* ... and the kernels built on it are instantiated for the four byte orders with the colour bytes in sequence (BGRx, RGBx, xRGB, xBGR and
       their alpha forms - GSTAMD_FOR_LAYOUTS in video_kernels.hip); RBGA goes through the generic kernels */
    if (plan->fout->kind == UNPACK_PACKED4 && plan->fout->hi_depth == 0) {
      const int pr = post.pack_pos[1], pg = post.pack_pos[2], pb = post.pack_pos[3];
      const bool seq = (pr == 2 && pg == 1 && pb == 0) || (pr == 0 && pg == 1 && pb == 2) || (pr == 1 && pg == 2 && pb == 3) || (pr == 3 && pg == 2 && pb == 1);
      matrix_no_wrap = matrix_no_wrap && seq;
    }
  }
  plan->fast_pair = plan->passes.empty () && fr.kind == UNPACK_SEMI && fr.chroma_v2 == 1 && (in_w % 4) == 0 && in_h >= 2 && matrix_no_wrap;
  /* capture direction (video_422_fast.h): unscaled packed 4:2:2 -> 4-byte RGB, whole 8-pixel groups, no odd-width tail quirk */
  plan->fast_422 = plan->passes.empty () && fr.kind == UNPACK_PACKED422 && !plan->out_planar && matrix_no_wrap && (in_w % 8) == 0 &&
      fr.swap_k < 0 && !fr.chroma_v2;
  plan->fast_422_ayuv = plan->passes.empty () && fr.kind == UNPACK_PACKED422 && plan->matrix.kind == MATRIX_NONE && post.alpha_kind == ALPHA_NONE &&
      (in_w % 8) == 0 && fr.swap_k < 0 && !fr.chroma_v2 && post.pack_pos[0] == 0 && post.pack_pos[1] == 1 && post.pack_pos[2] == 2 && post.pack_pos[3] == 3;
  /* decoder-output direction: the reference's I420 / YV12 -> RGB same-size fastpaths (nearest chroma), whole 8-pixel groups */
  plan->fast_420p = plan->passes.empty () && fr.kind == UNPACK_PLANAR && fr.w_sub == 1 && fr.h_sub == 1 && fr.chroma_h == CHROMA_H_NONE &&
      !fr.chroma_v2 && !plan->out_planar && matrix_no_wrap && (in_w % 8) == 0;
  /* the encoder-facing block kernel (video_encode_fast.h): unscaled 4-byte RGB -> 4:2:0 planar / semi-planar through the table
   * matrix (which the reference only picks when no pixel can clip, so every row sum stays inside 16 bits) with byte coefficients */
  plan->fast_enc420 = false;
  if (plan->passes.empty () && plan->out_planar && kind_has_planes (plan->fout->kind) && plan->fout->w_sub == 1 && plan->fout->h_sub == 1 &&
      fr.kind == UNPACK_PACKED4 && !plan->fin->yuv && plan->matrix.kind == MATRIX_TABLE && (in_w % 4) == 0 && !plan->pack.virtual_line) {
    bool fits = true;
    for (int k = 0; k < 3; k++)
      for (int j = 0; j < 3; j++)
        fits = fits && abs (plan->matrix.im[k][j]) <= 255;
    plan->fast_enc420 = fits;
  }
  /* scaled plans: the same matrix code after the scaler, if the alpha channel provably arrives as 0xff: opaque source
   * and every N-tap phase summing to exactly 1.0 (the 2-tap and nearest kernels keep a constant channel as it is) */
  plan->fast_post = !plan->passes.empty () && !plan->matrix_before_scale && matrix_no_wrap && kind_has_planes (fr.kind);
  for (const ScalePass &ps : plan->passes) {
    if (ps.kind != SCALE_NTAP)
      continue;
    for (int i = 0; i < ps.out_size && plan->fast_post; i++) {
      int sum = 0;
      for (int l = 0; l < ps.n_taps; l++)
        sum += ps.taps[(size_t) i * ps.n_taps + l];
      if (sum != (1 << ps.precision))
        plan->fast_post = false;
    }
  }
  /* enlarging from NV12 / NV21: the colour stage's source-size image (A, R, G, B bytes: one of the line-pair kernel's byte orders) when the pairing the
     scaler's requests produce is the closed form that kernel computes - lines (2 p - 1, 2 p) over chroma rows (p - 1, p), clamped */
  plan->fast_pre = false;
  if (!plan->passes.empty () && plan->matrix_before_scale && matrix_no_wrap_core && fr.kind == UNPACK_SEMI && fr.chroma_v2 == 1 && (in_w % 4) == 0 && in_h >= 2 &&
      fr.hi_depth == 0 && (int) plan->vpair.size () >= 2 * in_h) {
    const int lo = -(plan->rect.in_y >> 1), hi = ((plan->rect.in_maxh + 1) >> 1) - 1 - (plan->rect.in_y >> 1);
    bool regular = true;
    for (int y = 0; y < in_h && regular; y++) {
      const int u = (y + 1) >> 1;
      const int ra = std::min (std::max (u - 1, lo), hi), rb = std::min (std::max (u, lo), hi);
      const int heavy = (y & 1) ? ra : rb, light = (y & 1) ? rb : ra;
      const int e0 = plan->vpair[(size_t) 2 * y], ta = vpair_row (e0), tb = plan->vpair[(size_t) 2 * y + 1];
      const int th = vpair_role (e0) == 0 ? ta : tb, tl = vpair_role (e0) == 0 ? tb : ta;
      regular = th == heavy && tl == light;
    }
    plan->fast_pre = regular;
  }
  if (plan->deep16 || plan->deep_out)
    plan->fast_pair = plan->fast_420p = plan->fast_422 = plan->fast_422_ayuv = plan->fast_enc420 = plan->fast_post = plan->fast_pre = false;
  if (field)            /* the generic kernels (their chroma rows and weights come from the field's pair table); the per-line 4:2:2 kernels stay */
    plan->fast_pair = plan->fast_420p = plan->fast_enc420 = false;
  if (plan->pack.dither.on)             /* the dither stage lives in the pack kernel: the fused kernels that write planes / 3-byte pixels themselves have none */
    plan->fast_pair = plan->fast_enc420 = false;
  /* planar / semi-planar 8-bit YUV on both sides with the same subsampling, and a chain that neither filters nor mixes: the front hands
     every pixel the chroma sample of its own position (no upsampling filter, no vertical pairing), the pack takes it back as it is or as
     the average of equal values - decoder output to encoder input (I420 -> NV12) is a re-arrangement of the planes */
  plan->relayout = plan->passes.empty () && plan->out_planar && !plan->deep16 && !plan->deep_out && kind_has_planes (fr.kind) && kind_has_planes (plan->fout->kind) &&
      plan->fin->hi_depth == 0 && plan->fout->hi_depth == 0 && plan->fin->w_sub == plan->fout->w_sub && plan->fin->h_sub == plan->fout->h_sub &&
      fr.chroma_h == CHROMA_H_NONE && !fr.chroma_v2 && plan->pack.down_h != 2 && !plan->pack.virtual_line && !plan->pack.tail_swap &&
      !plan->pack.dither.on && plan->matrix.kind == MATRIX_NONE && post.alpha_kind == ALPHA_NONE && in_w == out->width && in_h == out->height &&
      !plan->rect.in_x && !plan->rect.in_y && !plan->rect.out_x && !plan->rect.out_y;

  if (field && (plan->deep_out || (plan->deep16 && plan->out_planar)))
    return fail (GSTAMD_ERR_UNSUPPORTED, "interlaced frames through the 16-bit part of the chain into a planar or 10 / 12 / 16-bit destination are not implemented on the GPU path");
  std::string d = plan->passes.empty () ? (plan->deep16 ? "convert16" : plan->fast_pair ? "fused_convert_pair" : plan->relayout ? "planes_relayout" : plan->fast_enc420 ? "fused_encode_420" : plan->fast_422 ? "fused_convert_422" : plan->fast_420p ? "fused_convert_420p" : "fused_convert") : "scale";
  d += std::string ("[") + plan->fin->name + "->" + plan->fout->name;
  d += fr.chroma_h == CHROMA_H_H2_CS ? ",h2cs" : fr.chroma_h == CHROMA_H_H2 ? ",h2" : fr.chroma_h == CHROMA_H_H4 ? ",h4" : fr.chroma_h == CHROMA_H_H4_CS ? ",h4cs" : "";
  d += fr.chroma_v2 == 2 ? (field_up_v ? ",vi2" : ",vi") : fr.chroma_v2 ? ",v2" : "";
  if (field)
    d += field == 1 ? ",top" : ",bottom";
  for (const ScalePass &p : plan->passes)
    d += std::string (p.horizontal ? ",H" : ",V") + std::to_string (p.n_taps) + (p.dot4_ok ? "b" : "");    /* b: byte-dot-product taps */
  static const char *mk[] = {"none", "ayuv_argb", "table", "matrix8"};
  d += std::string (",matrix=") + mk[plan->matrix.kind] + (plan->matrix_before_scale ? "(pre)" : "") + "]";
  if (!plan->ref_fastpath.empty ())
    d += "{as " + plan->ref_fastpath + "}";
  if (plan->out_planar)
    d += std::string ("+pack_planar[h") + std::to_string (plan->pack.down_h) + ",v" + std::to_string (plan->pack.down_v) + (plan->pack.dither.on ? ",dither" : "") + "]";
  plan->description = d;
  if (plan->deep_out)
    return finalize_deep_out (in, out, plan, same_matrix, same_primaries, prim_dm, in_matrix, out_matrix, alpha_bits, error);
  if (plan->deep16 && plan->out_planar)
    return finalize_deep_to_planar (in, out, plan, alpha_bits, error);
  return GSTAMD_OK;
}

// gst_video_converter_new's rectangle options (:2307-2366), then the plan for the cropped picture
int plan_video_converter (const GstAmdVideoInfo *in, const GstAmdVideoInfo *out,
    const GstAmdVideoConverterConfig *config, VideoPlan *plan, std::string *error)
{
  if (!in || !out || !plan) {
    if (error)
      *error = "NULL argument";
    return GSTAMD_ERR_INVALID;
  }
  GstAmdVideoConverterConfig cfg;
  if (config)
    cfg = *config;
  else
    converter_config_init (&cfg);
  /* interlaced frames.  gst_video_converter_new insists on one mode for both infos (:2435).  An interleaved frame (GST_VIDEO_FRAME_IS_INTERLACED) becomes
     two field conversions; this plan only speaks for them. */
  if (in->interlace_mode != out->interlace_mode) {
    if (error)
      *error = "the two infos carry different interlace modes (gst_video_converter_new refuses that: \"we won't ever do deinterlace\")";
    return GSTAMD_ERR_INVALID;
  }
  if (in->interlace_mode != GSTAMD_INTERLACE_MODE_PROGRESSIVE && !info_is_interleaved (*in) && !info_is_field (*in)) {
    if (error)
      *error = "interlace-mode fields / alternate is not implemented on the GPU path";
    return GSTAMD_ERR_UNSUPPORTED;
  }
  if (info_is_interleaved (*in)) {
    VideoPlan fp[2];
    GstAmdVideoConverterConfig fcfg;
    if (!plan_field_config (in, out, &cfg, &fcfg)) {
      if (error)
        *error = "interlaced frames with a source crop or a destination rectangle are not implemented on the GPU path";
      return GSTAMD_ERR_UNSUPPORTED;
    }
    for (int f = 0; f < 2; f++) {
      GstAmdVideoInfo fin, fout;
      plan_field_infos (in, out, f, &fin, &fout);
      const int r = plan_video_converter (&fin, &fout, &fcfg, &fp[f], error);
      if (r != GSTAMD_OK)
        return r;
    }
    *plan = VideoPlan ();
    plan->in_info = plan->orig_in = *in;
    plan->out_info = plan->orig_out = *out;
    plan->config = cfg;
    plan->fin = format_desc (in->format);
    plan->fout = format_desc (out->format);
    format_plan_planes (plan->fin, &plan->in_info);
    format_plan_planes (plan->fout, &plan->out_info);
    plan->interlaced = true;
    plan->plane_mode = plan->out_planar = plan->fast_pair = plan->relayout = plan->fast_enc420 = plan->fast_420p = plan->fast_422_ayuv = plan->fast_422 = false;
    plan->fast_post = plan->deep_out = plan->deep16 = plan->matrix_before_scale = plan->ref_same_size = false;
    memset (&plan->rect, 0, sizeof (plan->rect));
    memset (&plan->front, 0, sizeof (plan->front));
    memset (&plan->matrix, 0, sizeof (plan->matrix));
    memset (&plan->post, 0, sizeof (plan->post));
    memset (&plan->pack, 0, sizeof (plan->pack));
    memset (&plan->deep, 0, sizeof (plan->deep));
    memset (&plan->dither, 0, sizeof (plan->dither));
    plan->ref_fastpath = fp[0].ref_fastpath;
    plan->description = "interlaced{" + fp[0].description + " | " + fp[1].description + "}";
    plan->divergence = fp[0].divergence;
    if (fp[1].divergence != fp[0].divergence)
      plan->divergence += fp[1].divergence;
    plan->algorithmic_bytes = fp[0].algorithmic_bytes + fp[1].algorithmic_bytes;
    return GSTAMD_OK;
  }
  const bool is_field = info_is_field (*in);
  if (is_field && (cfg.src_x || cfg.src_y || cfg.src_width || cfg.src_height || cfg.dest_x || cfg.dest_y || cfg.dest_width || cfg.dest_height)) {
    if (error)
      *error = "interlaced frames with a source crop or a destination rectangle are not implemented on the GPU path";
    return GSTAMD_ERR_UNSUPPORTED;
  }
  const FormatDesc *fi = format_desc (in->format), *fo = format_desc (out->format);
  RectPlan rc;
  memset (&rc, 0, sizeof (rc));
  GstAmdVideoInfo ein = *in, eout = *out;
  if (fi && fo && in->width > 0 && in->height > 0 && out->width > 0 && out->height > 0) {
    const int in_maxw = in->width, in_maxh = in->height, out_maxw = out->width, out_maxh = out->height;
    int in_x = cfg.src_x & ~((1 << fi->w_sub) - 1), in_y = cfg.src_y & ~((1 << fi->h_sub) - 1);
    int in_w = cfg.src_width > 0 ? cfg.src_width : in_maxw - in_x, in_h = cfg.src_height > 0 ? cfg.src_height : in_maxh - in_y;
    in_w = std::min (in_w, in_maxw - in_x);
    if (in_w + in_x < 0 || in_w + in_x > in_maxw)
      in_w = 0;
    in_h = std::min (in_h, in_maxh - in_y);
    if (in_h + in_y < 0 || in_h + in_y > in_maxh)
      in_h = 0;
    int out_x = cfg.dest_x & ~((1 << fo->w_sub) - 1), out_y = cfg.dest_y & ~((1 << fo->h_sub) - 1);
    int out_w = cfg.dest_width > 0 ? cfg.dest_width : out_maxw - out_x, out_h = cfg.dest_height > 0 ? cfg.dest_height : out_maxh - out_y;
    if (out_w > out_maxw - out_x)
      out_w = out_maxw - out_x;
    out_w = std::max (0, std::min (out_w, out_maxw));
    if (out_w + out_x < 0 || out_w + out_x > out_maxw)
      out_w = 0;
    if (out_h > out_maxh - out_y)
      out_h = out_maxh - out_y;
    out_h = std::max (0, std::min (out_h, out_maxh));
    if (out_h + out_y < 0 || out_h + out_y > out_maxh)
      out_h = 0;
    if (in_x < 0 || in_y < 0 || out_x < 0 || out_y < 0 || in_w <= 0 || in_h <= 0 || out_w <= 0 || out_h <= 0) {
      if (error)
        *error = "empty or negative source / destination rectangle";
      return GSTAMD_ERR_UNSUPPORTED;
    }
    rc.in_x = in_x;
    rc.in_y = in_y;
    rc.out_x = out_x;
    rc.out_y = out_y;
    rc.out_maxw = out_maxw;
    rc.out_maxh = out_maxh;
    rc.in_maxw = in_maxw;
    rc.in_maxh = in_maxh;
    rc.fill = cfg.fill_border != 0 && (out_h < out_maxh || out_w < out_maxw);
    ein.width = in_w;
    ein.height = in_h;
    eout.width = out_w;
    eout.height = out_h;
    plan->ref_same_size = in_maxw == out_w && in_maxh == out_h;
    /* setup_borderline (:2189-2262): the border pixel in unpack order */
    const uint32_t argb = cfg.border_argb;
    const int a = argb >> 24, r = (argb >> 16) & 0xff, g = (argb >> 8) & 0xff, b = argb & 0xff;
    rc.border[0] = (uint8_t) a;
    rc.border[1] = (uint8_t) r;
    rc.border[2] = (uint8_t) g;
    rc.border[3] = (uint8_t) b;
    if (fo->yuv && fo->kind != UNPACK_GRAY && fo->kind != UNPACK_GRAY16 && fo->kind != UNPACK_GRAY_LE32) {          /* GST_VIDEO_INFO_IS_YUV: a GRAY8 frame keeps the ARGB bytes, pack_GRAY8 then stores R */
      /* identity -> compute_matrix_to_YUV (force) -> rint; then 16 / 128 / 128 are added whatever the range */
      M44 dm;
      m_identity (dm);
      double Kr = 0, Kb = 0;
      const int mtx = cfg.matrix_mode == GSTAMD_MATRIX_MODE_INPUT_ONLY ? (fi->yuv ? in->color_matrix : GSTAMD_COLOR_MATRIX_RGB) : out->color_matrix;
      if (get_Kr_Kb (mtx, &Kr, &Kb))
        m_RGB_to_YCbCr (dm, Kr, Kb);
      int offset[4], scale[4];
      /* the range of the PACK format (compute_matrix_to_YUV :1430-1438): 16-bit scales for a 10 / 12 / 16-bit destination, which the 8-bit
         formula below then clamps to 0 or 255 for most colours - the reference's border in such frames */
      range_offsets (out->color_range, true, offset, scale, fo->hi_depth ? 16 : 8);
      m_scale_components (dm, (float) scale[0], (float) scale[1], (float) scale[2]);
      m_offset_components (dm, offset[0], offset[1], offset[2]);
      int im[3][3];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
          im[i][j] = (int) rint (dm[i][j]);
      auto clamp8 = [](int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); };
      rc.border[1] = (uint8_t) clamp8 (16 + ((r * im[0][0] + g * im[0][1] + b * im[0][2]) >> 8));
      rc.border[2] = (uint8_t) clamp8 (128 + ((r * im[1][0] + g * im[1][1] + b * im[1][2]) >> 8));
      rc.border[3] = (uint8_t) clamp8 (128 + ((r * im[2][0] + g * im[2][1] + b * im[2][2]) >> 8));
    }
    if (g_border_override && (cfg.internal_flags & 1))
      memcpy (rc.border, g_border_override, 4);
  } else {
    plan->ref_same_size = in->width == out->width && in->height == out->height;
  }
  if (rc.in_maxh == 0) {
    rc.in_maxw = in->width;
    rc.in_maxh = in->height;
  }
  /* unpack_v210 takes no horizontal offset ("Horizontal offsets are not supported for v210", video-format.c:570-573): the generic chain's crop of a v210
     source is src_width pixels from the line's FIRST pixel on, whatever src-x says (do_unpack_lines :2966 passes in_x, the unpacker drops it) */
  if (fi && fi->kind == UNPACK_V210)
    rc.in_x = 0;
  /* IYU1: whole frames.  unpack_IYU1 advances by x * 4 BYTES for a horizontal offset ("FIXME", video-format.c:2382) - a crop starts inside another group's
     bytes; rectangles and borders inside its six-byte groups are not built either */
  if (fi && fo && ((fi->kind == UNPACK_PACKED411 && (rc.in_x || rc.in_y || ein.width != rc.in_maxw || ein.height != rc.in_maxh)) ||
          (fo->kind == UNPACK_PACKED411 && (rc.out_x || rc.out_y || rc.fill || eout.width != rc.out_maxw || eout.height != rc.out_maxh)))) {
    if (error)
      *error = "source crops and destination rectangles on IYU1 frames (six-byte groups of four pixels; the reference's unpacker misplaces a horizontal offset) are not implemented on the GPU path";
    return GSTAMD_ERR_UNSUPPORTED;
  }
  if (is_field && fi && fo && (fi->kind == UNPACK_PACKED411 || fo->kind == UNPACK_PACKED411)) {
    if (error)
      *error = "interlaced IYU1 frames are not implemented on the GPU path";
    return GSTAMD_ERR_UNSUPPORTED;
  }
  if (fi && fo && ((fi->kind == UNPACK_SEMI_TILED && (rc.in_x || rc.in_y || ein.width != rc.in_maxw || ein.height != rc.in_maxh)) ||
          (fo->kind == UNPACK_SEMI_TILED && (rc.out_x || rc.out_y || rc.fill || eout.width != rc.out_maxw || eout.height != rc.out_maxh)) ||
          (is_field && (fi->kind == UNPACK_SEMI_TILED || fo->kind == UNPACK_SEMI_TILED)))) {
    if (error)
      *error = "source crops, destination rectangles and interlaced frames on tiled NV12 frames are not implemented on the GPU path";
    return GSTAMD_ERR_UNSUPPORTED;
  }
  /* the 10LE32 formats: whole frames (unpack_NV12_10LE32 & co skip the pixels left of x but keep writing from the line's first slot on, video-format.c:5620,
     5646-5655); a source whose width is 6 n + 3 makes the unpacker read the chroma word AFTER its row for the last pixel's V - past the plane in the last row */
  if (fi && fo && ((GSTAMD_KIND_LE32 (fi->kind) && (rc.in_x || rc.in_y || ein.width != rc.in_maxw || ein.height != rc.in_maxh)) ||
          (GSTAMD_KIND_LE32 (fo->kind) && (rc.out_x || rc.out_y || rc.fill || eout.width != rc.out_maxw || eout.height != rc.out_maxh)))) {
    if (error)
      *error = "source crops and destination rectangles on frames with three 10-bit samples per 32-bit word (GRAY10_LE32, NV12_10LE32, NV16_10LE32) are not implemented on the GPU path";
    return GSTAMD_ERR_UNSUPPORTED;
  }
  if (fi && fi->kind == UNPACK_SEMI_LE32 && (rc.in_maxw % 6) == 3) {
    if (error)
      *error = "NV12_10LE32 / NV16_10LE32 sources of width 6 n + 3: the reference's unpacker reads the word after the chroma row for the last pixel (past the plane in the last row); not reproduced";
    return GSTAMD_ERR_UNSUPPORTED;
  }
  if (is_field && fi && fo && (GSTAMD_KIND_LE32 (fi->kind) || GSTAMD_KIND_LE32 (fo->kind))) {
    if (error)
      *error = "interlaced frames with three 10-bit samples per 32-bit word are not implemented on the GPU path";
    return GSTAMD_ERR_UNSUPPORTED;
  }
  /* unpack_RGBA64_LE and its seven siblings advance a guint16 pointer by x * 8 (video-format.c:2483, 2532 ...): 16 bytes per pixel of a horizontal
     crop offset - the picture starts at pixel 2 x (reproduced), and runs into the next row when that is too far right (refused) */
  if (fi && fi->kind == UNPACK_PACKED64 && !fi->yuv && rc.in_x != 0 && fi->hi_depth != 16 && fi->hi_depth != 36) {         /* (unpack_Y412_LE / _Y416_LE step x * 4, unpack_RGBA_F16LE x * 8 BYTES: no quirk) */
    if (2 * rc.in_x + ein.width > rc.in_maxw) {
      if (error)
        *error = "the reference's 64-bit unpackers misplace a horizontal source crop (x * 8 on a 16-bit pointer) and read past the row here; not reproduced";
      return GSTAMD_ERR_UNSUPPORTED;
    }
    rc.in_x *= 2;               /* inside the row: the crop simply starts twice as far right */
  }
  if (is_field)
    plan->ref_same_size = in->width == out->width && in->frame_height == out->frame_height;          /* (the lookup's same_size: the frames') */
  plan->rect = rc;
  plan->orig_in = *in;
  plan->orig_out = *out;
  const int r = plan_core (&ein, &eout, &cfg, plan, error);
  if (r != GSTAMD_OK || !fi || !fo)
    return r;
  if (is_field) {
    /* (the notes below are what the reference's PROGRESSIVE chain does with lines it hands out twice; the interlaced chain's own cases of that kind
       have not been mapped: the one stage known to alias - a nearest vertical scaler, which hands its input line on - is refused in the chain) */
    if (!plan->plane_mode && plan->ref_fastpath.empty ())
      for (const ScalePass &sp : plan->passes)
        if (!sp.horizontal && sp.kind == SCALE_NEAREST) {
          if (error)
            *error = "nearest vertical scaling of interlaced frames through the generic chain (lines handed out more than once, in place stages behind them) is not implemented on the GPU path";
          return GSTAMD_ERR_UNSUPPORTED;
        }
    /* Observed with the reference (scripts/ilace_probe.py; oracle/_ref): the generic chain's vertical scaler of an interlaced frame asks the line cache
       for 2 * taps lines and keeps as many behind it (chain_vscale :1651-1675: backlog = taps_i), more lines than the temporary-line rings below it hold
       (setup_allocators :2115-2187).  Whenever a window reaches back to a line whose buffer has been handed out again the scaler reads ANOTHER line's
       pixels - luma included: every enlargement (22 x 12 -> 22 x 20: the top field's rows come from source lines four further down), reductions whose
       windows overlap (32 -> 24 lines with three taps: five rows), every 4:2:0 source (its upsampler's groups of four sit in the same rings).  The same
       frame through the reference's plane scaler (same scaler object, same taps: Y444 -> Y444) has none of it.  Nothing to reproduce: the field plans
       apply the interlaced scaler's taps to the lines the chain intends - tests/test_video_interlaced.py pins that against the reference run stage by
       stage, the vertical pass through its plane scaler. */
    if (!plan->plane_mode && plan->ref_fastpath.empty ())
      for (const ScalePass &sp : plan->passes)
        if (!sp.horizontal) {
          plan->divergence += "interlaced frames through the generic chain's vertical scaler: the reference keeps 2 * taps lines of backlog over temporary-line rings "
              "that are shorter (line aliasing: its windows read other lines' pixels, luma included; its own plane scaler with the same taps does not); "
              "this library applies the interlaced scaler's taps to the lines the chain intends. ";
          break;
        }
    /* (convert_scale_planes on packed 4:2:2 of odd width with the vertical pass first: the note at the end of this function applies to a field as it is) */
    if (plan->plane_mode && fi->kind == UNPACK_PACKED422 && (ein.width & 1) && !plan->planes.empty () && plan->planes[0].passes.size () == 2 && !plan->planes[0].passes[0].horizontal)
      plan->divergence += "packed 4:2:2 of odd width scaled vertically, then horizontally: the reference's vertical pass copies 2 * width bytes of a line, the V sample of the "
          "last half macropixel stays uninitialised in its temporary line and the horizontal pass reads it; this library takes that sample from the source row. ";
    return r;
  }
  if (rc.fill && (fo->kind == UNPACK_PACKED422 || fo->kind == UNPACK_P422_16) && ((rc.out_x | eout.width | rc.out_maxw) & 1) &&
      (plan->plane_mode || !plan->ref_fastpath.empty ())) {
    /* (the generic chain packs frame lines pair by pair: border_picture_positions; rectangles on whole macropixels are filled by every path.
       convert_fill_border's group 42 (:7277-7285) writes 2-byte units from the rectangle's odd right edge on and swaps its U and V for EVERY border
       line then, and each fastpath / plane scaler has its own tail rule: not built) */
    if (error)
      *error = "borders of the reference's fastpaths and plane scaler (convert_fill_border, group 42) on a packed 4:2:2 destination whose rectangle or frame ends inside a macropixel are not implemented on the GPU path";
    return GSTAMD_ERR_UNSUPPORTED;
  }
  /* Two places where the reference's generic chain reads lines it has not (or has already) converted; its output there is
   * uninitialised memory or a matrix applied twice, so there is nothing to reproduce - such plans are refused:
   *  (a) do_convert_lines converts MIN (in_width, out_width) pixels (:3112) into a FRESH line when the bit depth changes (:3117-3136).
   *      When the line is wider than that at this point - the horizontal pass already enlarged it (scaling first, the picture
   *      shrinks overall) or will only shrink it afterwards (scaling last) - the scaler / packer reads the rest uninitialised.
   *  (b) the nearest vertical scaler hands out the SAME line for every output row it repeats (video_scale_v_near), and the
   *      convert / alpha stages after it work in place (:3127-3141, do_alpha_lines): a repeated row gets the matrix once more
   *      per repetition. */
  /* the odd-width tail of pack_VYUY / pack_NV61 is the FRAME line's (the reference packs out_maxwidth pixels a line, borders included): a
     rectangle that ends before the frame's right edge packs its last pixel the ordinary way */
  if (!(plan->rect.out_x + eout.width == plan->rect.out_maxw && (plan->rect.out_maxw & 1)))
    plan->pack.tail_swap = plan->gamma.pack.tail_swap = 0;
  /* a rectangle inside a v210 frame: the packer works on the frame line's groups (PackPlanarParams::frame_on) */
  if (fo->kind == UNPACK_V210 && (rc.out_x || rc.out_y || rc.fill || (rc.out_maxw && (rc.out_maxw != eout.width || rc.out_maxh != eout.height)))) {
    if (!plan->gamma.on || !plan->gamma.pack16 || !rc.out_maxw) {
      if (error)
        *error = "a destination rectangle inside a v210 frame is only implemented for the 16-bit chain's packer";
      return GSTAMD_ERR_UNSUPPORTED;
    }
    if (plan->gamma.dither16.on && plan->gamma.dither16.method != GSTAMD_DITHER_BAYER) {
      if (error)
        *error = "error-diffusion dither into a rectangle inside a v210 frame is not implemented on the GPU path";
      return GSTAMD_ERR_UNSUPPORTED;
    }
    PackPlanarParams &pk = plan->gamma.pack;
    pk.frame_on = rc.fill ? 2 : 1;
    pk.frame_x = rc.out_x, pk.frame_y = rc.out_y, pk.frame_w = rc.out_maxw, pk.frame_h = rc.out_maxh;
    for (int k = 0; k < 3; k++)
      pk.border10[k] = ((uint32_t) rc.border[k + 1] * 257u) >> 6;          /* setup_borderline's video_orc_splat2_u64 of the 8-bit border, then pack_v210's >> 6 */
  }
  /* pack_Y210 / pack_Y212_LE repeat the first luma of an odd-width line's last macropixel as its second (video-format.c:849-850) - of the FRAME line: a
     rectangle of odd width that ends before the frame's right edge has the border's luma there (border_picture_positions); 2 = the packer leaves it */
  if (fo->kind == UNPACK_P422_16 && rc.fill && (eout.width & 1) && rc.out_x + eout.width < rc.out_maxw)
    plan->gamma.pack.tail_swap = 2;
  plan->gamma.dither16.y0 = plan->rect.out_y;          /* the 16-bit dither stage counts frame lines (do_dither_lines: out_line = i + out_y) */
  const VideoPlan &pl = *plan;
  const bool chain = !pl.plane_mode && pl.ref_fastpath.empty ();
  const int iw = ein.width, ih = ein.height, ow = eout.width, oh = eout.height;
  const bool scale_first = (long) ow * oh <= (long) iw * ih;
  const bool remap = pl.gamma.on && cfg.gamma_mode == GSTAMD_GAMMA_MODE_REMAP;
  if (chain && (iw != ow || ih != oh)) {
    const int in_bits = fi->hi_depth ? 16 : 8, out_bits = fo->hi_depth ? 16 : 8;
    if (in_bits != out_bits && !remap && (scale_first ? ow : iw) > std::min (iw, ow)) {
      plan->divergence += "the reference converts only MIN (in_width, out_width) pixels of a line when the bit depth changes (do_convert_lines); with the "
          "horizontal pass on the other side of that step the rest of its line is uninitialised memory; this library converts the whole line. ";
    }
    bool v_near_up = false;
    for (const ScalePass &sp : pl.passes)
      v_near_up = v_near_up || (!sp.horizontal && sp.kind == SCALE_NEAREST && sp.out_size > sp.in_size);
    const GammaPlan &g = pl.gamma;
    const bool in_place_op = pl.post.matrix.kind != MATRIX_NONE || pl.matrix.kind != MATRIX_NONE || pl.post.alpha_kind == ALPHA_MULT || pl.deep.has_matrix ||
        (g.on && (g.prim.has_matrix || g.to_rgb.kind != MATRIX_NONE || g.to_yuv.kind != MATRIX_NONE || g.alpha_kind == ALPHA_MULT || remap));
    /* the stages after the scalers that write into their input line: chroma downsampling (video_chroma_down_h2 / v2 work on the
     * lines they are given) and the dither stage (do_dither_lines, write_input) */
    const bool late_in_place = pl.dither.on || pl.pack.dither.on || (g.on && g.dither16.on) || ((pl.out_planar || fo->kind == UNPACK_PACKED422) && (pl.pack.down_h || pl.pack.down_v)) ||
        (g.on && g.pack16 && (g.pack.down_h || g.pack.down_v));
    /* (the colour / alpha stage only shows it when the lines are the destination frame's own rows - a destination in its unpack
     * format, identity_pack :2105, get_dest_line; with temporary lines the repeated row is converted from a fresh copy) */
    const bool identity_pack = fo->format == GSTAMD_VIDEO_FORMAT_AYUV || fo->format == GSTAMD_VIDEO_FORMAT_ARGB ||
        fo->format == GSTAMD_VIDEO_FORMAT_AYUV64 || fo->format == GSTAMD_VIDEO_FORMAT_ARGB64;
    /* with temporary lines it shows once a row is repeated more than twice (the ring hands the converted line out again:
       7 -> 14 rows match, 7 -> 49 do not) */
    bool many_repeats = false;
    for (const ScalePass &sp : pl.passes)
      many_repeats = many_repeats || (!sp.horizontal && sp.kind == SCALE_NEAREST && sp.out_size > 2 * sp.in_size);
    /* and when the bit depth changes the matrix runs on the SOURCE line before it is narrowed into a fresh one (:3127-3136): the
       repeated row's source line has already been through it */
    /* any nearest vertical pass into an odd-height 4:2:0 destination: the line past the picture (the vertical chroma downsampler's last pair) is the
       scaler's last line handed out AGAIN - the in-place stages have been through it once already */
    bool v_near = false;
    for (const ScalePass &sp : pl.passes)
      v_near = v_near || (!sp.horizontal && sp.kind == SCALE_NEAREST);
    if (v_near && !v_near_up && fo->h_sub == 1 && (oh & 1) && (in_place_op || late_in_place) && cfg.chroma_mode != GSTAMD_CHROMA_MODE_NONE &&
        cfg.chroma_mode != GSTAMD_CHROMA_MODE_UPSAMPLE_ONLY) {
      plan->divergence += "nearest vertical scaling into an odd-height 4:2:0 destination: the line past the picture is the scaler's last line handed out a second "
          "time, after the in-place stages (colour, dither, chroma downsampling) have already changed it (line aliasing); this library pairs the last line "
          "with itself. ";
    }
    /* observed with the reference (scripts/fuzz_video.py, a sweep of 8 -> 9 .. 69 rows): a 4:2:0 source enlarged by the nearest vertical scaler ALONE
       (no horizontal pass between the in-place stages and the scaler) comes out with other chroma on the first line of a pair once the enlargement
       exceeds five - 8 -> 42 rows match, 8 -> 43 do not, with a horizontal pass anywhere in the chain every ratio matches: its temporary lines are
       reused before the repetitions have been served.  Announced, not reproduced */
    if (pl.front.chroma_v2 && pl.passes.size () == 1 && !pl.passes[0].horizontal && pl.passes[0].kind == SCALE_NEAREST && pl.passes[0].out_size > 5 * pl.passes[0].in_size &&
        !(v_near_up && ((scale_first && in_place_op && (identity_pack || many_repeats || in_bits != out_bits || (remap && (pl.gamma.prim.has_matrix || pl.gamma.alpha_kind == ALPHA_MULT)))) ||
                late_in_place || (identity_pack && pl.front.chroma_v2)))) {
      plan->divergence += "nearest vertical enlargement by more than five of a 4:2:0 source with no horizontal pass: the reference's temporary lines are reused before "
          "every repetition of a line has been served (line aliasing: the first line of a chroma pair changes); this library serves every repetition from the same line. ";
    }
    /* observed the same way (150 000 device fuzz draws late in round 4, then sweeps of small crops of NV12 / I420 into VUYA / BGRA): the nearest scaler in
       both directions, the VERTICAL one first (the order chain_scale picks when in_width x out_height < out_width x in_height) - once the FIRST line of an
       upsampler pair (lines 1, 3, ... of the crop) has been handed out MORE THAN FOUR times, the pair's second line comes out with other content, luma
       included: a temporary line reused, not a pairing rule (4 -> 16 rows match, 4 -> 18 do not; 3 -> 14 matches - its line 1 is repeated four times -
       3 -> 13 and 3 -> 15 do not; two source lines never show it; horizontal first every ratio matches; a 4:2:2 source is exact).  Announced, not reproduced */
    if (pl.front.chroma_v2 && pl.passes.size () == 2 && !pl.passes[0].horizontal && pl.passes[0].kind == SCALE_NEAREST && pl.passes[1].horizontal &&
        pl.passes[1].kind == SCALE_NEAREST) {
      const ScalePass &pv = pl.passes[0];
      std::vector<int> repeats ((size_t) pv.in_size + 1, 0);
      for (uint32_t o : pv.offset)
        if ((int) o <= pv.in_size)
          repeats[o]++;
      bool reused = false;
      for (int l = 1; l + 1 < pv.in_size; l += 2)          /* the pair (l, l + 1): its first line asked for more than four times, its second comes out wrong */
        reused = reused || repeats[(size_t) l] > 4;
      if (reused)
        plan->divergence += "nearest enlargement of a 4:2:0 source, vertical pass first, with the first line of a chroma pair handed out more than four times: the "
            "reference delivers the pair's second line with the content of another line (a temporary line reused; line aliasing); this library replicates the source's pixels. ";
    }
    /* (a 4:2:0 source into a frame in its unpack format: the chroma upsampler itself works in the destination's rows, which the scaler hands out again) */
    /* under gamma-mode = remap the stages between the decode and encode tables (the primaries matrix, the alpha multiply) work in place on 16-bit
       temporary lines whose allocator the two scalers share (pass_alloc, one line): the vertical scaler's output IS the horizontal scaler's cached
       line, so the second hand-out of a line is already converted (observed: 20x1 -> 3x3 drifts row by row; two taps and two source lines are exact) */
    const bool gamma_in_place16 = remap && (g.prim.has_matrix || g.alpha_kind == ALPHA_MULT);
    if (v_near_up && ((scale_first && in_place_op && (identity_pack || many_repeats || in_bits != out_bits || gamma_in_place16)) || late_in_place ||
            (identity_pack && pl.front.chroma_v2))) {
      plan->divergence += "nearest vertical enlargement ahead of a stage that works in place (colour / alpha, chroma downsampling, dither): the reference "
          "processes a repeated line once more per repetition (line aliasing); this library applies every stage once per output row. ";
    }
  }
  /* the composite plans (8-bit source into the 16-bit part of the chain) unpack and upsample the whole frame in line order before
   * anything else; the reference's vertical chroma upsampler pairs lines in the order the nearest vertical scaler asks for them
   * (it skips / repeats lines), which the pair-table simulation of the direct plans follows and the composite does not */
  /* (a 10 / 12 / 16-bit 4:2:0 source under gamma-mode = remap goes the same way: its 16-bit front is planned as an unscaled conversion, plan_gamma) */
  if (chain && pl.gamma.on && (!pl.gamma.src16 || remap) && !pl.gamma.src64 && fi->h_sub == 1 && (kind_has_planes (fi->kind) || GSTAMD_KIND_ALPHA_PLANE (fi->kind) >= 0 || kind_is_tiled (fi->kind))) {
    for (const ScalePass &sp : pl.passes)
      if (!sp.horizontal && sp.kind == SCALE_NEAREST) {
        if (error)
          *error = "nearest vertical scaling of a 4:2:0 source through the 16-bit part of the chain (line pairing of the chroma upsampler follows the scaler's requests) is not implemented";
        return GSTAMD_ERR_UNSUPPORTED;
      }
    /* ... and the same for a filter whose windows leave source lines out (a shrink by more than its taps: 12 -> 2 rows with two taps asks for lines
     * 2, 3, 8, 9): do_upsample_lines pairs (L, L + 1) from the line it is ASKED for (:2991-3021, start_line = in_line), so a window that starts on
     * an even line after a gap flips the pairing against the in-order (odd, even) pairs the composite's front makes.  Found by the device fuzz once
     * the comparison of 16-bit frames covered their chroma planes (tests/cases.py visible_planes); windows that cover every line from the top
     * keep the in-order pairs */
    for (const ScalePass &sp : pl.passes)
      if (!sp.horizontal && sp.kind != SCALE_NONE && !sp.offset.empty ()) {
        const int taps = sp.kind == SCALE_NTAP ? sp.n_taps : (sp.kind == SCALE_2TAP ? 2 : 1);
        bool gaps = (int) sp.offset[0] > 1;
        for (size_t y = 0; y + 1 < sp.offset.size () && !gaps; y++)
          gaps = (int) sp.offset[y + 1] > (int) sp.offset[y] + taps;
        if (gaps) {
          if (error)
            *error = "vertical scaling that leaves source lines out, of a 4:2:0 source through the 16-bit part of the chain (line pairing of the chroma upsampler follows the scaler's requests), is not implemented";
          return GSTAMD_ERR_UNSUPPORTED;
        }
      }
  }
  /* do_alpha_lines sets / multiplies MIN (in_width, out_width) pixels (video-converter.c do_alpha_lines); when the line is wider at
   * that point (the horizontal pass on the other side of the alpha stage) the rest keeps the alpha it had.  Well defined, but not
   * built here: refused rather than silently different */
  {
    const bool alpha_op = pl.post.alpha_kind != ALPHA_NONE || (pl.gamma.on && pl.gamma.alpha_kind != ALPHA_NONE);
    if (chain && alpha_op && (scale_first ? ow : iw) > std::min (iw, ow)) {
      if (error)
        *error = "alpha-mode set / mult with the horizontal pass on the other side of the alpha stage (the reference touches MIN (in_width, out_width) "
            "pixels of a wider line) is not implemented";
      return GSTAMD_ERR_UNSUPPORTED;
    }
  }
  /* an odd-height 4:2:0 destination, no vertical scaler, and a source crop that ends above the frame's last line: the line past the picture the
   * vertical chroma downsampler averages the last line with is a REAL source line then (do_unpack_lines clamps to the frame, :2966).  The
   * unscaled direct plans make it (PackPlanarParams::virtual_line); behind a horizontal scaler or inside the composite plans it is not built */
  {
    const bool line_below = plan->rect.in_maxh && plan->rect.in_y + ih < plan->rect.in_maxh;
    const bool down_v = (pl.out_planar && pl.pack.down_v) || (pl.gamma.on && pl.gamma.pack16 && pl.gamma.pack.down_v) ||
        (pl.gamma.on && !pl.gamma.pack16 && !pl.gamma.store64 && !pl.gamma.fused && !pl.gamma.planes_fast &&
            cfg.chroma_mode != GSTAMD_CHROMA_MODE_NONE && cfg.chroma_mode != GSTAMD_CHROMA_MODE_UPSAMPLE_ONLY);
    if (chain && line_below && fo->h_sub == 1 && (oh & 1) && ih == oh && down_v && !pl.pack.virtual_line) {
      if (error)
        *error = "an odd-height 4:2:0 destination from a source crop with frame lines below it (the chroma downsampler's line past the picture is a real "
            "line there) is only implemented for unscaled 8-bit conversions";
      return GSTAMD_ERR_UNSUPPORTED;
    }
  }
  /* a 4:2:0 source into a 4:2:0 destination of odd height through the composite plans: the reference's last chroma row averages the
   * last line with a line past the picture, which its upsampler makes from the clamped last rows (do_unpack_lines clamps :2966); the
   * composite's sub-conversions end at the last line - and the direct plans' pair table has no entry for that line either */
  const bool both_v = !pl.gamma.on && pl.front.chroma_v2 && pl.out_planar && pl.pack.down_v;       /* direct plans: both vertical chroma resamplers run */
  bool has_vpass = false;
  for (const ScalePass &sp : pl.passes)
    has_vpass = has_vpass || !sp.horizontal;
  /* composite plans: the same question asked of the chain (video_converter_compute_resample: both resamplers exist once subsampling, chroma site or
     frame size differ, chroma-mode permitting; the vertical downsampler of a vertically cosited destination is a stub).  The line past the picture is
     only CONSUMED by a vertical downsampler, and only differs from the last line when a vertical upsampler pairs it anew: with either of them out of
     the chain the last chroma row is the last line's own and the composite is exact (host fuzz: 0 bad of 120 000 draws with this rule, the
     conversions it lets through were 1.5 % of the draws) */
  const bool chroma_differs = fi->w_sub != fo->w_sub || fi->h_sub != fo->h_sub || in->chroma_site != out->chroma_site || in->width != out->width || in->height != out->height;
  const bool comp_up_v = chroma_differs && cfg.chroma_mode != GSTAMD_CHROMA_MODE_NONE && cfg.chroma_mode != GSTAMD_CHROMA_MODE_DOWNSAMPLE_ONLY;
  const bool comp_down_v = chroma_differs && cfg.chroma_mode != GSTAMD_CHROMA_MODE_NONE && cfg.chroma_mode != GSTAMD_CHROMA_MODE_UPSAMPLE_ONLY &&
      !(out->chroma_site & GSTAMD_CHROMA_SITE_V_COSITED);
  const bool comp_both_v = pl.gamma.on && !pl.gamma.planes_fast && comp_up_v && comp_down_v;
  has_vpass = has_vpass || (pl.gamma.on && ih != oh);          /* (a composite may keep its scaler passes in a sub-conversion) */
  if (chain && comp_both_v && has_vpass && fi->h_sub == 1 && fo->h_sub == 1 && (oh & 1)) {
    /* composite + vertical size change: the composite's last pair (last line, last line) IS what the reference delivers - its scaler hands out its last
       line for the line past the picture and, on 16-bit lines, the upsampler's re-pairing does not reach it (host fuzz with the refusal lifted: 0 of the
       ~1 % of 165 000 draws that land here differ; they are compared like any other plan, no divergence note) */
  } else if (chain && both_v && has_vpass && fi->h_sub == 1 && fo->h_sub == 1 && (oh & 1)) {
    /* with a vertical scaler in the chain the line past the picture is the scaler's last line once more (do_vscale_lines clamps the
       output line, :3074) - but it asks its window from the chroma upsampler a second time, lines that cache has already let go are
       unpacked and paired anew, and what comes back differs from what the last line was made of (luma included): the reference's
       last line pair depends on which lines happened to survive.  Here the last pair is (last line, last line). */
    plan->divergence += "4:2:0 -> 4:2:0 of odd height with a vertical scaler between the chroma resamplers: the reference asks the scaler for the line past the "
        "picture, whose window is unpacked and paired anew (line aliasing); this library pairs the last line with itself. ";
  } else if (chain && both_v && pl.pack.virtual_line) {
    /* same size: the line past the picture is made and used (PackPlanarParams::virtual_line) */
  } else
  if (chain && (comp_both_v || both_v) && fi->h_sub == 1 && fo->h_sub == 1 && (oh & 1)) {
    if (error)
      *error = "4:2:0 -> 4:2:0 of odd height through the generic chain (chroma of the line past the picture) is not implemented";
    return GSTAMD_ERR_UNSUPPORTED;
  }
  /* unpack_VYUY (video-format.c:310-352) falls back to a C loop with U and V the other way round when the line it writes is not
   * 8-byte aligned; with an unpack-format destination the lines are the destination frame's own (get_dest_line), so on frames whose
   * rows are not all 8-byte aligned (odd widths) every other row comes out with swapped chroma */
  bool vyuy_new_lines = false;          /* a scaler that makes new lines between the unpacker and the destination's rows (not the nearest vertical one) */
  for (const ScalePass &sp : pl.passes)
    vyuy_new_lines = vyuy_new_lines || sp.horizontal || sp.kind != SCALE_NEAREST;
  if (chain && fi->format == GSTAMD_VIDEO_FORMAT_VYUY && !vyuy_new_lines && fo->kind == UNPACK_PACKED4 && fo->hi_depth == 0 &&
      fo->pos[0] == 0 && fo->pos[1] == 1 && fo->pos[2] == 2 && fo->pos[3] == 3 && (fo->alpha) &&
      ((out->stride[0] % 8) != 0 || (out->offset[0] % 8) != 0)) {
    plan->divergence += "VYUY unpacked straight into destination rows that are not 8-byte aligned: the reference's fallback loop (video-format.c:337-352) "
        "swaps U and V on those rows; this library unpacks every row the same way. ";
  }
  /* pack_VYUY (video-format.c:354-384) has the same fallback for a SOURCE line that is not 8-byte aligned, and its loop stores U, Y, V, Y - UYVY
   * order.  The packer's source line is the source frame's own row when nothing between unpack and pack makes a line of its own: an AYUV source
   * (identity unpack), no scaler, no matrix, no chroma downsampler, no dither, no border - then rows of a frame of odd width (or every row behind an
   * odd src-x) come out with U and V swapped.  (Found in round 5 when the new formats reshuffled the fuzz draws: seed 101.) */
  {
    const RectPlan &rcv = plan->rect;
    const bool whole = !rcv.out_x && !rcv.out_y && !rcv.fill && (!rcv.out_maxw || (rcv.out_maxw == out->width && rcv.out_maxh == out->height));
    if (chain && fi->format == GSTAMD_VIDEO_FORMAT_AYUV && fo->format == GSTAMD_VIDEO_FORMAT_VYUY && pl.passes.empty () && !pl.gamma.on &&
        pl.matrix.kind == MATRIX_NONE && pl.post.matrix.kind == MATRIX_NONE && pl.out_planar && !pl.pack.down_h && !pl.pack.dither.on && whole &&
        ((in->stride[0] % 8) != 0 || (in->offset[0] % 8) != 0 || (rcv.in_x & 1)))
      plan->divergence += "AYUV rows that are not 8-byte aligned packed straight into VYUY: the reference's fallback loop (video-format.c:366-373) stores "
          "UYVY order on those rows; this library packs every row the same way. ";
  }
  /* a ONE-line 4:2:0 source enlarged into a rectangle below the first row of a frame in its unpack format (ARGB / AYUV: the chain's lines are the
   * destination's own rows, get_dest_line): the chroma upsampler makes its line pair (-1, 0) in the rows (out_y - 1, out_y) and the row above the
   * rectangle keeps what it wrote there */
  /* (the device fuzz's seed 5484 widened it: the same with one output row, and with the 16-bit unpack formats as destinations - the pair's other line
   * still lands in a frame row, or in front of the frame when the rectangle starts at row 0, and the picture's own row differs) */
  const bool one_line_420 = pl.front.chroma_v2 || (pl.gamma.on && fi->h_sub == 1 && (kind_has_planes (fi->kind) || GSTAMD_KIND_ALPHA_PLANE (fi->kind) >= 0 || kind_is_tiled (fi->kind)) && cfg.chroma_mode != GSTAMD_CHROMA_MODE_NONE &&
      cfg.chroma_mode != GSTAMD_CHROMA_MODE_DOWNSAMPLE_ONLY);
  /* ... and every chain whose unpacker works in the destination's rows (no scaler that makes new lines in between) does it to the row above a rectangle
   * that starts below row 0: a crop copied unscaled into a rectangle of an ARGB frame keeps picture-derived pixels in that border row */
  /* ... and so does a source CROP (frame lines above or below it: the pair mates of the crop's first / last line are real lines, unpacked into the rows
   * next to the picture - or past the frame's last row) */
  const bool cropped_v = plan->rect.in_y > 0 || (plan->rect.in_maxh && plan->rect.in_y + ih < plan->rect.in_maxh);
  if (chain && one_line_420 && (ih == 1 || (!vyuy_new_lines && (plan->rect.out_y > 0 || cropped_v))) &&
      (fo->format == GSTAMD_VIDEO_FORMAT_AYUV || fo->format == GSTAMD_VIDEO_FORMAT_ARGB || fo->format == GSTAMD_VIDEO_FORMAT_AYUV64 ||
          fo->format == GSTAMD_VIDEO_FORMAT_ARGB64)) {
    plan->divergence += "a 4:2:0 source whose chroma upsampler works in the rows of an ARGB / AYUV (or ARGB64 / AYUV64) frame - one source line, a rectangle below row 0, "
        "or a crop with frame lines above / below it: the reference unpacks the pair mates of the picture's first / last line into the frame rows next to the picture "
        "(or past the frame); this library leaves those rows to the border and pairs inside the crop. ";
  }
  /* convert_scale_planes on packed 4:2:2 with the vertical pass first: that pass moves `width` pixels = 2 * width bytes of a line (video-converter.c
   * convert_plane_v / _hv), which on an odd width leaves out the V sample of the last, half macropixel; the horizontal pass then reads it from a
   * temporary line nobody wrote */
  if (pl.plane_mode && fi->kind == UNPACK_PACKED422 && (iw & 1) && !pl.planes.empty () && pl.planes[0].passes.size () == 2 && !pl.planes[0].passes[0].horizontal) {
    plan->divergence += "packed 4:2:2 of odd width scaled vertically, then horizontally: the reference's vertical pass copies 2 * width bytes of a line, the V sample of the "
        "last half macropixel stays uninitialised in its temporary line and the horizontal pass reads it; this library takes that sample from the source row. ";
  }
  return r;
}

bool plan_field_config (const GstAmdVideoInfo *in, const GstAmdVideoInfo *out, const GstAmdVideoConverterConfig *cfg, GstAmdVideoConverterConfig *fcfg)
{
  if (cfg)
    *fcfg = *cfg;
  else
    converter_config_init (fcfg);
  const GstAmdVideoConverterConfig &c = *fcfg;
  const bool whole_src = c.src_x == 0 && c.src_y == 0 && (c.src_width == 0 || c.src_width >= in->width) && (c.src_height == 0 || c.src_height >= in->height);
  const bool whole_dst = c.dest_x == 0 && c.dest_y == 0 && (c.dest_width == 0 || c.dest_width >= out->width) && (c.dest_height == 0 || c.dest_height >= out->height);
  if (!whole_src || !whole_dst)
    return false;
  fcfg->src_x = fcfg->src_y = fcfg->src_width = fcfg->src_height = 0;
  fcfg->dest_x = fcfg->dest_y = fcfg->dest_width = fcfg->dest_height = 0;
  return true;
}

void plan_field_infos (const GstAmdVideoInfo *in, const GstAmdVideoInfo *out, int f, GstAmdVideoInfo *fin, GstAmdVideoInfo *fout)
{
  const GstAmdVideoInfo *src[2] = {in, out};
  GstAmdVideoInfo *dst[2] = {fin, fout};
  for (int k = 0; k < 2; k++) {
    GstAmdVideoInfo &d = *dst[k];
    d = *src[k];
    d.interlace_mode = f ? GSTAMD_INTERLACE_FIELD_BOTTOM : GSTAMD_INTERLACE_FIELD_TOP;
    d.frame_height = src[k]->height;
    d.height = f ? src[k]->height - (src[k]->height + 1) / 2 : (src[k]->height + 1) / 2;
    for (int i = 0; i < GSTAMD_VIDEO_MAX_PLANES; i++) {
      d.offset[i] = src[k]->offset[i] + (uint64_t) (f ? src[k]->stride[i] : 0);
      d.stride[i] = src[k]->stride[i] * 2;
    }
  }
}

bool plan_is_pad_scaler (const VideoPlan &p, int *h, int *v)
{
  *h = *v = -1;
  if (p.gamma.on || p.plane_mode || p.out_planar || p.deep_out || p.deep16 || p.dither.on || p.passes.empty () || p.passes.size () > 2)
    return false;
  if (!p.fin || p.fin != p.fout || p.fin->kind != UNPACK_PACKED4 || p.fin->hi_depth != 0)
    return false;
  if (p.front.kind != UNPACK_PACKED4 || p.matrix.kind != MATRIX_NONE || p.post.matrix.kind != MATRIX_NONE || p.post.alpha_kind != ALPHA_NONE)
    return false;
  for (int k = 0; k < 4; k++)
    if (p.front.pos[k] != k || p.post.pack_pos[k] != k)
      return false;
  const RectPlan &r = p.rect;
  if (r.in_x || r.in_y || r.out_x || r.out_y || r.fill || r.in_maxw != p.in_info.width || r.in_maxh != p.in_info.height ||
      r.out_maxw != p.out_info.width || r.out_maxh != p.out_info.height)
    return false;
  for (size_t i = 0; i < p.passes.size (); i++) {
    const ScalePass &sp = p.passes[i];
    if (sp.merged || sp.kind == SCALE_NONE || (sp.horizontal ? *h : *v) >= 0)
      return false;
    (sp.horizontal ? *h : *v) = (int) i;
  }
  return true;
}

int scaled_tile_rows_for (const VideoPlan &p)
{
  int h = -1, v = -1;
  if (!plan_is_pad_scaler (p, &h, &v) || h < 0 || v < 0)
    return 16;
  const ScalePass &ph = p.passes[h], &pv = p.passes[v];
  const int cols = std::min (64, ph.out_size);
  const int ntap_h = ph.kind == SCALE_NTAP ? ph.n_taps : (ph.kind == SCALE_2TAP ? 2 : 1), ntap_v = pv.kind == SCALE_NTAP ? pv.n_taps : (pv.kind == SCALE_2TAP ? 2 : 1);
  int best = 16;
  double best_eff = -1.0;
  for (int th = 16; th >= 12; th--) {
    const int rows = std::min (th, pv.out_size);
    int items;
    if (h < v) {                /* horizontal first: the vertical source span of the rows x the tile's columns */
      const int c0 = pv.out_size / 2 >= rows ? pv.out_size / 2 - rows / 2 : 0;
      items = ((int) pv.offset[c0 + rows - 1] + ntap_v - (int) pv.offset[c0]) * cols;
    } else {
      const int c0 = ph.out_size / 2 >= cols ? ph.out_size / 2 - cols / 2 : 0;
      const int span = ph.kind == SCALE_2TAP ? (((c0 + cols - 1) * ph.inc) >> 16) + 2 - ((c0 * ph.inc) >> 16)
          : (int) ph.offset[c0 + cols - 1] + ntap_h - (int) ph.offset[c0];
      items = ((span + 3) / 4) * rows;
    }
    const double eff = (double) items / (double) ((items + 255) / 256 * 256);
    if (eff > best_eff + 1e-9) {
      best_eff = eff;
      best = th;
    }
  }
  return best;
}

}  // namespace gstamd

// ------------------------------------------------------------------------------------------------
// GstVideoTestSrc (gst/videotestsrc/videotestsrc.c): the colours its painters use and the per-frame values of video_testsrc.h - host side of
// gstamd_video_test_pattern_* (video_testsrc.hip) and of the host emulator
// ------------------------------------------------------------------------------------------------
#include "video_testsrc.h"

namespace gstamd {

namespace {
struct VtsColor { int Y, U, V, A, R, G, B; };
/* vts_colors_bt709_ycbcr_100 / _75 and vts_colors_bt601_ycbcr_100 / _75 (videotestsrc.c:59-155): published colour-bar values (white, yellow, cyan, green,
   magenta, red, blue, black, -I, +Q, super black, dark grey) */
const VtsColor k709_100[12] = {{235, 128, 128, 255, 255, 255, 255}, {219, 16, 138, 255, 255, 255, 0}, {188, 154, 16, 255, 0, 255, 255}, {173, 42, 26, 255, 0, 255, 0},
  {78, 214, 230, 255, 255, 0, 255}, {63, 102, 240, 255, 255, 0, 0}, {32, 240, 118, 255, 0, 0, 255}, {16, 128, 128, 255, 0, 0, 0}, {16, 198, 21, 255, 0, 0, 128},
  {16, 235, 198, 255, 0, 128, 255}, {0, 128, 128, 255, 0, 0, 0}, {32, 128, 128, 255, 19, 19, 19}};
const VtsColor k709_75[8] = {{180, 128, 128, 255, 191, 191, 191}, {168, 44, 136, 255, 191, 191, 0}, {145, 147, 44, 255, 0, 191, 191}, {133, 63, 52, 255, 0, 191, 0},
  {63, 193, 204, 255, 191, 0, 191}, {51, 109, 212, 255, 191, 0, 0}, {28, 212, 120, 255, 0, 0, 191}, {16, 128, 128, 255, 0, 0, 0}};
const VtsColor k601_100[12] = {{235, 128, 128, 255, 255, 255, 255}, {210, 16, 146, 255, 255, 255, 0}, {170, 166, 16, 255, 0, 255, 255}, {145, 54, 34, 255, 0, 255, 0},
  {106, 202, 222, 255, 255, 0, 255}, {81, 90, 240, 255, 255, 0, 0}, {41, 240, 110, 255, 0, 0, 255}, {16, 128, 128, 255, 0, 0, 0}, {16, 198, 21, 255, 0, 0, 128},
  {16, 235, 198, 255, 0, 128, 255}, {0, 128, 128, 255, 0, 0, 0}, {32, 128, 128, 255, 19, 19, 19}};
const VtsColor k601_75[8] = {{180, 128, 128, 255, 191, 191, 191}, {162, 44, 142, 255, 191, 191, 0}, {131, 156, 44, 255, 0, 191, 191}, {112, 72, 58, 255, 0, 191, 0},
  {84, 184, 198, 255, 191, 0, 191}, {65, 100, 212, 255, 191, 0, 0}, {35, 212, 114, 255, 0, 0, 191}, {16, 128, 128, 255, 0, 0, 0}};

uint32_t painted (const VtsColor &c, bool rgb)
{
  return rgb ? (uint32_t) c.A | ((uint32_t) c.R << 8) | ((uint32_t) c.G << 16) | ((uint32_t) c.B << 24)
      : (uint32_t) c.A | ((uint32_t) (c.Y & 0xff) << 8) | ((uint32_t) (c.U & 0xff) << 16) | ((uint32_t) (c.V & 0xff) << 24);
}

/* RGB_TO_Y_CCIR & co (videotestsrc.c:164-202): 10 fractional bits */
int fix10 (double x) { return (int) (x * 1024 + 0.5); }
VtsColor user_color (uint32_t argb, bool bt601)
{
  VtsColor c;
  const int r = (argb >> 16) & 0xff, g = (argb >> 8) & 0xff, b = argb & 0xff;
  c.A = (argb >> 24) & 0xff, c.R = r, c.G = g, c.B = b;
  const double kr = bt601 ? 0.29900 : 0.212600, kg = bt601 ? 0.58700 : 0.715200, kb = bt601 ? 0.11400 : 0.072200;
  const double ur = bt601 ? 0.16874 : 0.114572, ug = bt601 ? 0.33126 : 0.385427, vg = bt601 ? 0.41869 : 0.454153, vb = bt601 ? 0.08131 : 0.045847;
  c.Y = (fix10 (kr * 219.0 / 255.0) * r + fix10 (kg * 219.0 / 255.0) * g + fix10 (kb * 219.0 / 255.0) * b + (512 + (16 << 10))) >> 10;
  c.U = ((-fix10 (ur * 224.0 / 255.0) * r - fix10 (ug * 224.0 / 255.0) * g + fix10 (0.50000 * 224.0 / 255.0) * b + 512 - 1) >> 10) + 128;
  c.V = ((fix10 (0.50000 * 224.0 / 255.0) * r - fix10 (vg * 224.0 / 255.0) * g - fix10 (vb * 224.0 / 255.0) * b + 512 - 1) >> 10) + 128;
  return c;
}
}  // namespace

void test_pattern_setup (TestPatternParams *p, const GstAmdVideoInfo *info, int pattern, uint32_t foreground_argb, uint32_t background_argb)
{
  memset ((void *) p, 0, sizeof (*p));
  const FormatDesc *f = format_desc (info->format);
  const bool rgb = f && !f->yuv;          /* GST_VIDEO_INFO_IS_RGB: GRAY frames are painted as AYUV */
  const bool bt601 = info->color_matrix == GSTAMD_COLOR_MATRIX_BT601;
  p->pattern = pattern;
  p->w = info->width;
  p->h = info->height;
  for (int i = 0; i < 12; i++)
    p->colors[i] = painted (bt601 ? k601_100[i] : k709_100[i], rgb);
  for (int i = 0; i < 8; i++)
    p->colors75[i] = painted (bt601 ? k601_75[i] : k709_75[i], rgb);
  p->fg = painted (user_color (foreground_argb, bt601), rgb);
  p->bg = painted (user_color (background_argb, bt601), rgb);
}

void test_pattern_frame (TestPatternParams *p, uint64_t n_frames)
{
  p->odd_frame = (int) (n_frames & 1);
  p->rand_state = testsrc_lcg_skip (0u, (uint32_t) (n_frames * (uint64_t) testsrc_draws_per_frame (p->pattern, p->w, p->h)));
  /* gst_video_test_src_ball with animation-mode = frames, motion = wavy (:1486-1513): libm on the host, as in the reference */
  const int radius = 20;
  const double rad = 2 * 3.1415926535897932384626433832795028841971693993751 * ((double) n_frames / 200);
  p->ball_radius = radius;
  p->ball_x = radius + (0.5 + 0.5 * sin (rad)) * (p->w - 2 * radius);
  p->ball_y = radius + (0.5 + 0.5 * sin (rad * sqrt (2))) * (p->h - 2 * radius);
}

void test_pattern_conversion (const GstAmdVideoInfo *info, GstAmdVideoInfo *pi, GstAmdVideoConverterConfig *cfg)
{
  const FormatDesc *f = format_desc (info->format);
  const bool rgb = f && !f->yuv;
  video_info_set_format (pi, rgb ? GSTAMD_VIDEO_FORMAT_ARGB : GSTAMD_VIDEO_FORMAT_AYUV, info->width, info->height);
  pi->color_range = info->color_range;
  pi->color_matrix = rgb ? GSTAMD_COLOR_MATRIX_RGB : info->color_matrix;
  pi->color_transfer = info->color_transfer;
  pi->color_primaries = info->color_primaries;
  pi->chroma_site = info->chroma_site;
  converter_config_init (cfg);
  cfg->internal_flags = 1;                          /* the generic chain: chroma downsampler + pack, what convert_hline_generic (:1626-1683) calls */
  cfg->dither_method = GSTAMD_DITHER_NONE;          /* (TO_16 and the pack function: no dither stage) */
  cfg->matrix_mode = GSTAMD_MATRIX_MODE_NONE;
  cfg->chroma_mode = GSTAMD_CHROMA_MODE_FULL;
  cfg->alpha_mode = GSTAMD_ALPHA_MODE_COPY;
}

}  // namespace gstamd
