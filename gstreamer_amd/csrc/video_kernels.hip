// video_kernels.hip - gfx950 kernels (thin __global__ wrappers over video_device.h) + host launchers.
//
// Design: these are HBM-bound byte stencils (no MFMA).  The unscaled path is ONE fused pass:
// every source byte is read once (chroma rows are shared through L2) and every destination
// byte written once with 16-byte stores per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "planner.h"
#include "video_kernels.h"
#include "video_device.h"

namespace gstamd {

template <int CH>
__global__ __launch_bounds__ (256) void k_convert (FrontParams f, Planes pl, const int *__restrict__ vpair, ColorParams color,
    int pack0, int pack1, int pack2, int pack3, uint8_t *__restrict__ dst, int dstride, int spans_per_row, int vec_ok)
{
  convert_body<CH> (f, pl, vpair, color, pack0, pack1, pack2, pack3, dst, dstride, spans_per_row, vec_ok,
      (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

template <class SRC>
__global__ __launch_bounds__ (256) void k_hscale (SRC src, ScaleDev sd, Dst dst, int out_w, int rows)
{
  hscale_body<SRC> (src, sd, dst, out_w, rows, (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

template <class SRC>
__global__ __launch_bounds__ (256) void k_vscale (SRC src, ScaleDev sd, Dst dst, int width, int out_h)
{
  vscale_body<SRC> (src, sd, dst, width, out_h, (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
static inline bool aligned (const void *p, size_t a) { return ((uintptr_t) p & (a - 1)) == 0; }

hipError_t launch_convert (const FrontParams &f, const Planes &pl, const int *vpair_dev, const ColorParams &color,
    const int pack_pos[4], uint8_t *dst, int dstride, hipStream_t stream)
{
  const int spans = (f.width + K1_PX - 1) / K1_PX;
  int vec_ok = aligned (dst, 16) && (dstride % 16) == 0 && f.kind != UNPACK_PACKED4 && f.w_sub == 1;
  if (vec_ok) {
    vec_ok = aligned (pl.p[0], 8) && (pl.stride[0] % 8) == 0;
    if (f.kind == UNPACK_SEMI)
      vec_ok = vec_ok && aligned (pl.p[1], 8) && (pl.stride[1] % 8) == 0;
    else
      vec_ok = vec_ok && aligned (pl.p[1], 4) && aligned (pl.p[2], 4) && (pl.stride[1] % 4) == 0 && (pl.stride[2] % 4) == 0;
  }
  const int bx = spans >= 256 ? 256 : (spans > 64 ? 128 : 64);
  dim3 grid ((spans + bx - 1) / bx, f.height), block (bx);
  switch (f.chroma_h) {
    case CHROMA_H_H2_CS:
      hipLaunchKernelGGL (k_convert<CHROMA_H_H2_CS>, grid, block, 0, stream, f, pl, vpair_dev, color, pack_pos[0],
          pack_pos[1], pack_pos[2], pack_pos[3], dst, dstride, spans, vec_ok);
      break;
    case CHROMA_H_H2:
      hipLaunchKernelGGL (k_convert<CHROMA_H_H2>, grid, block, 0, stream, f, pl, vpair_dev, color, pack_pos[0],
          pack_pos[1], pack_pos[2], pack_pos[3], dst, dstride, spans, vec_ok);
      break;
    default:
      hipLaunchKernelGGL (k_convert<CHROMA_H_NONE>, grid, block, 0, stream, f, pl, vpair_dev, color, pack_pos[0],
          pack_pos[1], pack_pos[2], pack_pos[3], dst, dstride, spans, vec_ok);
      break;
  }
  return hipGetLastError ();
}

static Dst make_dst (uint8_t *p, int stride, bool final, const ColorParams &post, const int pack_pos[4])
{
  Dst d;
  d.p = p;
  d.stride = stride;
  d.final = final ? 1 : 0;
  d.post = post;
  for (int i = 0; i < 4; i++)
    d.pack_pos[i] = pack_pos[i];
  return d;
}

hipError_t launch_scale_from_front (bool horizontal, const FrontParams &f, const Planes &pl, const int *vpair_dev,
    const ColorParams &pre, const ScaleDev &sd, uint8_t *dst, int dstride, bool final, const ColorParams &post,
    const int pack_pos[4], int out_w, int out_h, hipStream_t stream)
{
  SrcFront src;
  src.f = f;
  src.pl = pl;
  src.vpair = vpair_dev;
  src.pre = pre;
  Dst d = make_dst (dst, dstride, final, post, pack_pos);
  dim3 block (256), grid ((out_w + 255) / 256, out_h);
  if (horizontal)
    hipLaunchKernelGGL (k_hscale<SrcFront>, grid, block, 0, stream, src, sd, d, out_w, out_h);
  else
    hipLaunchKernelGGL (k_vscale<SrcFront>, grid, block, 0, stream, src, sd, d, out_w, out_h);
  return hipGetLastError ();
}

hipError_t launch_scale_from_image (bool horizontal, const uint8_t *simg, int sstride, const ScaleDev &sd, uint8_t *dst,
    int dstride, bool final, const ColorParams &post, const int pack_pos[4], int out_w, int out_h, hipStream_t stream)
{
  SrcImage src;
  src.p = simg;
  src.stride = sstride;
  Dst d = make_dst (dst, dstride, final, post, pack_pos);
  dim3 block (256), grid ((out_w + 255) / 256, out_h);
  if (horizontal)
    hipLaunchKernelGGL (k_hscale<SrcImage>, grid, block, 0, stream, src, sd, d, out_w, out_h);
  else
    hipLaunchKernelGGL (k_vscale<SrcImage>, grid, block, 0, stream, src, sd, d, out_w, out_h);
  return hipGetLastError ();
}

}  // namespace gstamd
