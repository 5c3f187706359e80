// tuning.h - every development / support knob of the library in ONE table.
//
// The library's kernel choice does not depend on the process environment at the time of a call: the knobs below are read from the
// environment once, when the library first looks at one of them, and afterwards change only through gstamd_tuning_set () (the tests
// switch fast paths off with it to reach the general kernels).  A knob that is not set means "the library's own choice"; none of
// them changes a result - every path they select is held to the same byte-exact tests.
//
//   name                     meaning (value)
//   GSTAMD_NO_FAST_PRE       enlarging NV12 / NV21 plans make their source-size colour image with the per-pixel kernel instead of the line-pair kernel (set)
//   GSTAMD_STORE_POLICY, GSTAMD_STORE_WT_BELOW
//                            how k_convert_strip's pixels leave the CU (video_fast.h store16_policy: 0 streaming, 1 write-through sc0 sc1, 2 sc1, 3 sc0 sc1 nt,
//                            4 plain; default: 1 for launches of fewer than GSTAMD_STORE_WT_BELOW = 9 frames, 3 for longer lists; profiles/r06/store_policy.md)
//   GSTAMD_NO_COL            the older fused / two-pass kernels instead of the column-walk scaler k_scale_col (set)
//   GSTAMD_COL_OPL, GSTAMD_COL_SHARE, GSTAMD_COL_WAVES, GSTAMD_COL_CHUNKS, GSTAMD_COL_DEBUG
//                            form and geometry of the column-walk scaler: outputs per lane (1 / 2), shared windows (0: off), waves per
//                            workgroup, workgroups down a frame; DEBUG prints the choice
//   GSTAMD_NO_PLANE_QUAD     k_plane_direct / k_plane_tiles instead of k_plane_quad for planes of two short passes and the pass-free planes next to them (set)
//   GSTAMD_PLANE_QUAD_MODE, GSTAMD_PLANE_QUAD_ROWS, GSTAMD_PLANE_QUAD_NT, GSTAMD_PLANE_QUAD_ONLY, GSTAMD_PLANE_QUAD_NO_DSTEP
//                            k_plane_quad: widest form allowed (0: 4 output bytes per lane, 1: 8, 2: + 16 for pass-free planes, 3: 8 for those), rows a
//                            wave walks, nontemporal loads (bit 0) / stores (bit 1), one plane of the frame only (timing; the others are NOT
//                            converted), the general selectors for 2:1 planes too
//   GSTAMD_NO_DEEP_PLANES16  k_deep_planes (8 samples per lane, every format switch at run time) instead of k_deep_planes16 (set)
//   GSTAMD_NO_ENCODE16       the three-stage composite instead of k_encode16 for 4-byte pixels -> deep planar YUV (set)
//   GSTAMD_ENCODE16_NARROW, GSTAMD_ENCODE16_WIDE
//                            k_encode16 with four / eight pixels per lane whatever the launch (default: eight in frame lists)
//   GSTAMD_LIST_DEBUG        print, per frame list, how many launches took the list (set)
//   GSTAMD_NO_FUSED420       two-pass form instead of the fused 4:2:0 N-tap scaler (set)
//   GSTAMD_NO_H420_REG       general horizontal 4:2:0 kernel (pair table) instead of the regular-pairs one (set)
//   GSTAMD_H420_ROWS         lines per wave of the horizontal 4:2:0 kernels (n; 0: kernel off)
//   GSTAMD_VSCALE_ROWS       output rows per lane group of the vertical N-tap pass (n)
//   GSTAMD_NO_FAST420P       generic kernel instead of the I420 / YV12 -> RGB fastpath kernel (set)
//   GSTAMD_NO_FAST422        generic kernel instead of the packed 4:2:2 -> RGB kernel (set)
//   GSTAMD_NO_PLANE_FRAME    one kernel per plane and pass (intermediate plane in HBM) instead of k_plane_frame (set)
//   GSTAMD_NO_BILINEAR4_UP   k_bilinear4_rows instead of k_bilinear4_up for horizontal-first 2-tap x 2-tap plans of 4-byte pixels (set)
//   GSTAMD_BIL4_UP_ROWS      output rows per wave of k_bilinear4_up (n, default 8)
//   GSTAMD_NO_BILINEAR4      wave-tile scaler instead of the four-outputs-per-lane nearest / 2-tap scaler of 4-byte sources (set)
//   GSTAMD_NO_CONVERT16_FAST the general 16-bit convert kernel instead of the ones specialised by plane layout and chroma filter (set)
//   GSTAMD_NO_GAMMA_COMP     decode and encode tables separately in the fused gamma kernel, not their composition (set)
//   GSTAMD_NO_SWIZZLE34      the chain's kernels instead of the 3- / 4-byte pixel permutation kernel (set)
//   GSTAMD_NO_RELAYOUT       the chain's kernels instead of the plane re-arrangement kernel for I420 <-> NV12 & co (set)
//   GSTAMD_NO_CONVERT_PACK   AYUV image + k_pack_planar instead of the packer fed by the chain itself, k_convert_pack (set)
//   GSTAMD_NO_BILINEAR420    wave-tile scaler instead of the 4:2:0 bilinear kernels (set)
//   GSTAMD_NO_BILINEAR_ROWS  k_bilinear420 instead of k_bilinear420_rows (set)
//   GSTAMD_NO_BILINEAR_HALF  k_bilinear420_rows instead of k_bilinear420_half where the picture shrinks by exactly two (set)
//   GSTAMD_BIL_HALF_SMALL    k_bilinear420_half also for single frames of less than 4 M outputs (set)
//   GSTAMD_BIL_HALF_STORE    tuning builds: 1 the halves traded through LDS (default), 2 plain direct stores, 3 streaming direct stores, 5 / 6 ablations (n)
//   GSTAMD_BIL_HALF_ROWS     output rows per wave of k_bilinear420_half (n; default: one resident round for a frame, 8 in lists)
//   GSTAMD_NO_CONVERT_PACK_422UP  AYUV image + k_pack_planar for packed 4:2:2 sources whose chain upsamples the chroma horizontally (set)
//   GSTAMD_NO_CONVERT_PACK_WIDE  the byte-store form of k_convert_pack instead of its whole-block form (set)
//   GSTAMD_BIL_TILE, GSTAMD_BIL_TABLE, GSTAMD_BIL_ROWS, GSTAMD_BIL_ROWS_TILE, GSTAMD_BIL_SLOTS, GSTAMD_BIL_WG, GSTAMD_BIL_VERBOSE
//                            geometry of the bilinear kernels (n)
//   GSTAMD_FUSED_WAVES, GSTAMD_FUSED_ROWS, GSTAMD_FUSED_SCHED, GSTAMD_FUSED_FIRST, GSTAMD_FUSED_DEBUG
//                            geometry / schedule of the fused scaler (n)
//   GSTAMD_NO_FIR_LDS        one-lane-per-sample FIR instead of the LDS-staged one (set)
//   GSTAMD_SCALED_TILE_ROWS  rows per tile of k_aggregate_scaled (4 .. 16)
//   GSTAMD_AGG_BX            workgroup width of k_aggregate (n)
// -DGSTAMD_TUNING builds only (profiling sessions): GSTAMD_ABLATE, GSTAMD_AGG_ABLATE, GSTAMD_AGG_ROWS, GSTAMD_AGG_DEPTH,
//   GSTAMD_AGG_STRIP_ROWS, GSTAMD_AGG_STRIP_PX, GSTAMD_AGG_NT, GSTAMD_FAST_VARIANT (text), GSTAMD_FUSED_TRACE (a path).
#pragma once

namespace gstamd {

// value of a knob, or `unset` (default -1) when neither the environment at first use nor gstamd_tuning_set gave it one
int tuning_int (const char *name, int unset = -1);
inline bool tuning_on (const char *name) { return tuning_int (name, -1) >= 0; }
// text-valued knobs of tuning builds (read from the environment once, too); NULL when unset
const char *tuning_text (const char *name);

}  // namespace gstamd

extern "C" {
/* value < 0 removes the knob (back to the library's own choice).  Returns 0, or -1 for a name the table does not have. */
int gstamd_tuning_set (const char *name, int value);
int gstamd_tuning_get (const char *name);
}
