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

// one wave per workgroup: a lane = a 4-pixel wide block of the destination (two lines of a 4:2:0 one)
template <int SEMI, int CH>
__global__ __launch_bounds__ (64) void k_deep_scale_pack (PackPlanarParams pk, DeepPackParams dp, DstPlanes d, int wide, FrameDeltas fl)
{
  GSTAMD_FRAME_Z;
  dp.pl.p[0] += fls_, dp.pl.p[1] += fls_;
  if (!SEMI)
    dp.pl.p[2] += fls_;
  deep_scale_pack_lane<SEMI, CH> (pk, dp, d, wide, (int) (blockIdx.x * 64 + threadIdx.x) * 4, (int) blockIdx.y, fld_);
}

hipError_t launch_deep_scale_pack (const PackPlanarParams &pk, const DeepPackParams &dp_, uint8_t *const planes[3], const int strides[3], hipStream_t stream)
{
  DeepPackParams dp = dp_;
  DstPlanes d;
  for (int i = 0; i < 3; i++) {
    d.p[i] = planes[i];
    d.stride[i] = strides[i];
  }
  /* row4n's loads: source rows on 16 bytes; pack_planar_block4's stores: plane rows on 4 */
  const int variant = deep_front4_variant (dp.f);
  const bool semi = variant >= 3;
  dp.vec = ((uintptr_t) dp.pl.p[0] % 16) == 0 && (dp.pl.stride[0] % 16) == 0 && ((uintptr_t) dp.pl.p[1] % 16) == 0 && (dp.pl.stride[1] % 16) == 0 &&
      (semi || (((uintptr_t) dp.pl.p[2] % 16) == 0 && (dp.pl.stride[2] % 16) == 0));
  int wide = !pk.dither.on && (pk.kind == UNPACK_PLANAR || pk.kind == UNPACK_SEMI) && !tuning_on ("GSTAMD_DEEP_PACK_NARROW");
  for (int i = 0; wide && i < (pk.kind == UNPACK_SEMI ? 2 : 3); i++)
    wide = ((uintptr_t) planes[i] % 4) == 0 && (strides[i] % 4) == 0;
  const int lanes = (pk.width + 3) / 4, rows = (pk.height + (1 << pk.h_sub) - 1) >> pk.h_sub;
  int nz;
  const FrameDeltas &fl = video_frame_list_for (dp.pl.p[0], planes[0], &nz);
  const dim3 grid ((lanes + 63) / 64, rows, nz);
  switch (variant) {
    case 0: hipLaunchKernelGGL ((k_deep_scale_pack<0, CHROMA_H_NONE>), grid, dim3 (64), 0, stream, pk, dp, d, wide, fl); break;
    case 1: hipLaunchKernelGGL ((k_deep_scale_pack<0, CHROMA_H_H2>), grid, dim3 (64), 0, stream, pk, dp, d, wide, fl); break;
    case 2: hipLaunchKernelGGL ((k_deep_scale_pack<0, CHROMA_H_H2_CS>), grid, dim3 (64), 0, stream, pk, dp, d, wide, fl); break;
    case 3: hipLaunchKernelGGL ((k_deep_scale_pack<1, CHROMA_H_NONE>), grid, dim3 (64), 0, stream, pk, dp, d, wide, fl); break;
    case 4: hipLaunchKernelGGL ((k_deep_scale_pack<1, CHROMA_H_H2>), grid, dim3 (64), 0, stream, pk, dp, d, wide, fl); break;
    case 5: hipLaunchKernelGGL ((k_deep_scale_pack<1, CHROMA_H_H2_CS>), grid, dim3 (64), 0, stream, pk, dp, d, wide, fl); break;
    default: return hipErrorNotSupported;
  }
  return hipGetLastError ();
}

}  // namespace gstamd
