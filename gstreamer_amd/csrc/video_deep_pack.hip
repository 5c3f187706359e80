// video_deep_pack.hip - k_deep_scale_pack (video_deep_pack.h): a shrinking 10-bit planar / semi-planar source into an 8-bit planar / semi-planar
// destination in one launch; takes frame lists (the grid's third dimension).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "planner.h"
#include "tuning.h"
#include "video_kernels.h"
#include "video_device.h"
#include "video_deep_pack.h"

namespace gstamd {

// One wave per workgroup; a lane = a 4-pixel wide block of the destination (two lines of a 4:2:0 one).  Lanes 1 .. 62 store, lanes 0 and 63 make the blocks
// left and right of them for the chroma the cosited downsampler reads across the block's edges (DeepScaledSrc::row4n trades it through the wave): a
// workgroup covers 62 blocks of a line.  No lane leaves before the trades (blocks past the line's ends are clamped onto it and store nothing).
template <int SEMI, int CH>
__global__ __launch_bounds__ (64) void k_deep_scale_pack (PackPlanarParams pk, DeepPackParams dp, DstPlanes d, int nblk, FrameDeltas fl)
{
  GSTAMD_FRAME_Z;
  dp.pl.p[0] += fls_, dp.pl.p[1] += fls_;
  if (!(SEMI & 1))
    dp.pl.p[2] += fls_;
  const int lane = (int) threadIdx.x, blk = (int) blockIdx.x * 62 + lane - 1;
  const bool store = lane >= 1 && lane <= 62 && blk < nblk;
  const int bc = blk < 0 ? 0 : (blk > nblk - 1 ? nblk - 1 : blk);
  deep_scale_pack_lane<SEMI, CH> (pk, dp, d, 4 * bc, (int) blockIdx.y, fld_, store);
}

// what the kernel asks of the frames beyond deep_scale_pack_plan_ok: its 16-byte loads (source rows) and 4-byte stores (plane rows of the rectangle)
bool deep_scale_pack_usable (const PackPlanarParams &pk, const DeepPackParams &dp, uint8_t *const planes[3], const int strides[3])
{
  const bool semi = deep_front4_variant (dp.f) >= 3;
  bool ok = dp.hx2 && (pk.width % 4) == 0 && !pk.dither.on && (pk.kind == UNPACK_PLANAR || pk.kind == UNPACK_SEMI) && deep_front4_variant (dp.f) >= 0;
  for (int i = 0; ok && i < (semi ? 2 : 3); i++)
    ok = ((uintptr_t) dp.pl.p[i] % 16) == 0 && (dp.pl.stride[i] % 16) == 0;
  for (int i = 0; ok && i < (pk.kind == UNPACK_SEMI ? 2 : 3); i++)
    ok = ((uintptr_t) planes[i] % 4) == 0 && (strides[i] % 4) == 0;
  return ok;
}

hipError_t launch_deep_scale_pack (const PackPlanarParams &pk, const DeepPackParams &dp, uint8_t *const planes[3], const int strides[3], hipStream_t stream)
{
  if (!deep_scale_pack_usable (pk, dp, planes, strides))
    return hipErrorNotSupported;          /* (the caller asked the same question before it came here) */
  DstPlanes d;
  for (int i = 0; i < 3; i++) {
    d.p[i] = planes[i];
    d.stride[i] = strides[i];
  }
  const int variant = deep_pack_variant (dp.f);
  const int nblk = pk.width / 4, rows = (pk.height + (1 << pk.h_sub) - 1) >> pk.h_sub;
  int nz;
  const FrameDeltas &fl = video_frame_list_for (dp.pl.p[0], planes[0], &nz);
  const dim3 grid ((nblk + 61) / 62, rows, nz);
  switch (variant) {
    case 0: hipLaunchKernelGGL ((k_deep_scale_pack<0, CHROMA_H_NONE>), grid, dim3 (64), 0, stream, pk, dp, d, nblk, fl); break;
    case 1: hipLaunchKernelGGL ((k_deep_scale_pack<0, CHROMA_H_H2>), grid, dim3 (64), 0, stream, pk, dp, d, nblk, fl); break;
    case 2: hipLaunchKernelGGL ((k_deep_scale_pack<0, CHROMA_H_H2_CS>), grid, dim3 (64), 0, stream, pk, dp, d, nblk, fl); break;
    case 3: hipLaunchKernelGGL ((k_deep_scale_pack<1, CHROMA_H_NONE>), grid, dim3 (64), 0, stream, pk, dp, d, nblk, fl); break;
    case 4: hipLaunchKernelGGL ((k_deep_scale_pack<1, CHROMA_H_H2>), grid, dim3 (64), 0, stream, pk, dp, d, nblk, fl); break;
    case 5: hipLaunchKernelGGL ((k_deep_scale_pack<1, CHROMA_H_H2_CS>), grid, dim3 (64), 0, stream, pk, dp, d, nblk, fl); break;
    case 6: hipLaunchKernelGGL ((k_deep_scale_pack<2, CHROMA_H_NONE>), grid, dim3 (64), 0, stream, pk, dp, d, nblk, fl); break;
    case 7: hipLaunchKernelGGL ((k_deep_scale_pack<2, CHROMA_H_H2>), grid, dim3 (64), 0, stream, pk, dp, d, nblk, fl); break;
    case 8: hipLaunchKernelGGL ((k_deep_scale_pack<2, CHROMA_H_H2_CS>), grid, dim3 (64), 0, stream, pk, dp, d, nblk, fl); break;
    case 9: hipLaunchKernelGGL ((k_deep_scale_pack<3, CHROMA_H_NONE>), grid, dim3 (64), 0, stream, pk, dp, d, nblk, fl); break;
    case 10: hipLaunchKernelGGL ((k_deep_scale_pack<3, CHROMA_H_H2>), grid, dim3 (64), 0, stream, pk, dp, d, nblk, fl); break;
    case 11: hipLaunchKernelGGL ((k_deep_scale_pack<3, CHROMA_H_H2_CS>), grid, dim3 (64), 0, stream, pk, dp, d, nblk, fl); break;
    default: return hipErrorNotSupported;
  }
  return hipGetLastError ();
}

// ---- k_deep_scale_pack16: the same chain into a 10 / 12 / 16-bit planar or semi-planar destination (P010 -> P010 / I420_10LE at half the size); k_deep_scale_pack's
// lane layout (62 storing lanes a workgroup)
template <int SEMI, int CH>
__global__ __launch_bounds__ (64) void k_deep_scale_pack16 (PackPlanarParams pk, int hi_depth, DitherParams dt, DeepPackParams dp, DstPlanes16 d, int nblk, FrameDeltas fl)
{
  GSTAMD_FRAME_Z;
  dp.pl.p[0] += fls_, dp.pl.p[1] += fls_;
  if (!(SEMI & 1))
    dp.pl.p[2] += fls_;
  d.p[0] += fld_, d.p[1] += fld_, d.p[2] += fld_;
  const int lane = (int) threadIdx.x, blk = (int) blockIdx.x * 62 + lane - 1;
  const bool store = lane >= 1 && lane <= 62 && blk < nblk;
  const int bc = blk < 0 ? 0 : (blk > nblk - 1 ? nblk - 1 : blk);
  deep_scale_pack16_lane<SEMI, CH> (pk, hi_depth, dt, dp, d, 4 * bc, (int) blockIdx.y, store);
}

bool deep_scale_pack16_usable (const PackPlanarParams &pk, const DeepPackParams &dp, uint8_t *const planes[3], const int strides[3])
{
  const int variant = deep_front4_variant (dp.f);
  bool ok = dp.hx2 && variant >= 0 && (pk.width % 4) == 0 && (pk.kind == UNPACK_PLANAR || pk.kind == UNPACK_SEMI);
  for (int i = 0; ok && i < (variant >= 3 ? 2 : 3); i++)
    ok = ((uintptr_t) dp.pl.p[i] % 16) == 0 && (dp.pl.stride[i] % 16) == 0;
  for (int i = 0; ok && i < (pk.kind == UNPACK_SEMI ? 2 : 3); i++)
    ok = planes[i] != nullptr && ((uintptr_t) planes[i] % 2) == 0 && (strides[i] % 2) == 0;
  return ok;
}

hipError_t launch_deep_scale_pack16 (const PackPlanarParams &pk, int hi_depth, const DitherParams &dt, const DeepPackParams &dp, uint8_t *const planes[3],
    const int strides[3], hipStream_t stream)
{
  if (!deep_scale_pack16_usable (pk, dp, planes, strides))
    return hipErrorNotSupported;
  DstPlanes16 d;
  for (int i = 0; i < 3; i++) {
    d.p[i] = planes[i];
    d.stride[i] = strides[i];
  }
  const int variant = deep_pack_variant (dp.f);
  const int nblk = pk.width / 4, rows = (pk.height + (1 << pk.h_sub) - 1) >> pk.h_sub;
  int nz;
  const FrameDeltas &fl = video_frame_list_for (dp.pl.p[0], planes[0], &nz);
  const dim3 grid ((nblk + 61) / 62, rows, nz);
  switch (variant) {
#define P16(v, sh, ch) case v: hipLaunchKernelGGL ((k_deep_scale_pack16<sh, ch>), grid, dim3 (64), 0, stream, pk, hi_depth, dt, dp, d, nblk, fl); break;
    P16 (0, 0, CHROMA_H_NONE) P16 (1, 0, CHROMA_H_H2) P16 (2, 0, CHROMA_H_H2_CS) P16 (3, 1, CHROMA_H_NONE) P16 (4, 1, CHROMA_H_H2) P16 (5, 1, CHROMA_H_H2_CS)
    P16 (6, 2, CHROMA_H_NONE) P16 (7, 2, CHROMA_H_H2) P16 (8, 2, CHROMA_H_H2_CS) P16 (9, 3, CHROMA_H_NONE) P16 (10, 3, CHROMA_H_H2) P16 (11, 3, CHROMA_H_H2_CS)
#undef P16
    default: return hipErrorNotSupported;
  }
  return hipGetLastError ();
}

// ---- k_deep_scale4: the same chain into a 4-byte 8-bit destination (P010 -> BGRA at half the size): a lane = four pixels of a line, one 16-byte store
template <int SEMI, int CH>
__global__ __launch_bounds__ (64) void k_deep_scale4 (DeepPackParams dp, Deep16Params dd, PostParams post, uint8_t *__restrict__ dst, int dstride, FrameDeltas fl)
{
  GSTAMD_FRAME_Z;
  dp.pl.p[0] += fls_, dp.pl.p[1] += fls_;
  if (!(SEMI & 1))
    dp.pl.p[2] += fls_;
  deep_scale4_lane<SEMI, CH> (dp, dd, post, dst + fld_, dstride, (int) (blockIdx.x * 64 + threadIdx.x) * 4, (int) blockIdx.y);
}

bool deep_scale4_usable (const DeepPackParams &dp, const uint8_t *dst, int dstride)
{
  const int variant = deep_front4_variant (dp.f);
  bool ok = dp.hx2 && variant >= 0 && (dp.out_w % 4) == 0 && ((uintptr_t) dst % 4) == 0 && (dstride % 4) == 0;
  for (int i = 0; ok && i < (variant >= 3 ? 2 : 3); i++)
    ok = ((uintptr_t) dp.pl.p[i] % 16) == 0 && (dp.pl.stride[i] % 16) == 0;
  return ok;
}

hipError_t launch_deep_scale4 (const DeepPackParams &dp, const Deep16Params &dd, const PostParams &post, uint8_t *dst, int dstride, hipStream_t stream)
{
  if (!deep_scale4_usable (dp, dst, dstride))
    return hipErrorNotSupported;
  int nz;
  const FrameDeltas &fl = video_frame_list_for (dp.pl.p[0], dst, &nz);
  const dim3 grid ((dp.out_w / 4 + 63) / 64, dp.out_h, nz);
  switch (deep_pack_variant (dp.f)) {
    case 0: hipLaunchKernelGGL ((k_deep_scale4<0, CHROMA_H_NONE>), grid, dim3 (64), 0, stream, dp, dd, post, dst, dstride, fl); break;
    case 1: hipLaunchKernelGGL ((k_deep_scale4<0, CHROMA_H_H2>), grid, dim3 (64), 0, stream, dp, dd, post, dst, dstride, fl); break;
    case 2: hipLaunchKernelGGL ((k_deep_scale4<0, CHROMA_H_H2_CS>), grid, dim3 (64), 0, stream, dp, dd, post, dst, dstride, fl); break;
    case 3: hipLaunchKernelGGL ((k_deep_scale4<1, CHROMA_H_NONE>), grid, dim3 (64), 0, stream, dp, dd, post, dst, dstride, fl); break;
    case 4: hipLaunchKernelGGL ((k_deep_scale4<1, CHROMA_H_H2>), grid, dim3 (64), 0, stream, dp, dd, post, dst, dstride, fl); break;
    case 5: hipLaunchKernelGGL ((k_deep_scale4<1, CHROMA_H_H2_CS>), grid, dim3 (64), 0, stream, dp, dd, post, dst, dstride, fl); break;
    case 6: hipLaunchKernelGGL ((k_deep_scale4<2, CHROMA_H_NONE>), grid, dim3 (64), 0, stream, dp, dd, post, dst, dstride, fl); break;
    case 7: hipLaunchKernelGGL ((k_deep_scale4<2, CHROMA_H_H2>), grid, dim3 (64), 0, stream, dp, dd, post, dst, dstride, fl); break;
    case 8: hipLaunchKernelGGL ((k_deep_scale4<2, CHROMA_H_H2_CS>), grid, dim3 (64), 0, stream, dp, dd, post, dst, dstride, fl); break;
    case 9: hipLaunchKernelGGL ((k_deep_scale4<3, CHROMA_H_NONE>), grid, dim3 (64), 0, stream, dp, dd, post, dst, dstride, fl); break;
    case 10: hipLaunchKernelGGL ((k_deep_scale4<3, CHROMA_H_H2>), grid, dim3 (64), 0, stream, dp, dd, post, dst, dstride, fl); break;
    case 11: hipLaunchKernelGGL ((k_deep_scale4<3, CHROMA_H_H2_CS>), grid, dim3 (64), 0, stream, dp, dd, post, dst, dstride, fl); break;
    default: return hipErrorNotSupported;
  }
  return hipGetLastError ();
}

}  // namespace gstamd
