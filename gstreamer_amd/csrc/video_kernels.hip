// video_kernels.hip - gfx950 kernels (thin __global__ wrappers over video_device.h) + host launchers.
//
// Design: these are HBM-bound byte stencils (no MFMA).  The unscaled path is ONE fused pass:
// every source byte is read once (chroma rows are shared through L2) and every destination
// byte written once with 16-byte stores per lane.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "planner.h"
#include "tuning.h"
#include "video_kernels.h"
#include "video_device.h"
#include "video_fast.h"
#include "video_scale_fast.h"
#include "video_hscale420.h"
#include "video_pack.h"
#include "video_bilinear_fast.h"
#include "video_encode_fast.h"
#include "video_planes.h"
#include "video_deep.h"
#include "video_dither.h"
#include "video_dither_ed.h"
#include "video_relayout.h"
#include "video_swizzle34.h"
#include "video_gamma.h"

namespace gstamd {

// ---- frame lists for the single-kernel plans ------------------------------------------------------------------------------------------
// gstamd_video_converter_frames hands a list of frames of ONE layout over: frame i's planes sit at frame 0's plus a constant.  A kernel
// that converts a frame in one launch takes the list as the grid's third dimension: its arguments are frame 0's, workgroup z adds
// list.s[z] to every source pointer and list.d[z] to every destination pointer (pointer + offset stays a global pointer for the
// compiler; pointers read out of an array argument would be generic ones - FLAT loads).  The list travels from capi_video.cpp to the
// launchers through a thread-local context: a launcher uses it only when ITS source and destination pointers lie inside frame 0 of
// the list (a kernel that reads or writes a scratch image must not be rebased), and says so; the caller converts the remaining
// frames one by one when nobody did.
// (GSTAMD_MAX_BATCH, FrameDeltas: video_kernels.h)
namespace {
struct FrameListCtx {
  bool armed = false;
  int n = 0, used = 0;
  bool mixed = false;
  const uint8_t *src0 = nullptr, *dst0 = nullptr;
  size_t src_size = 0, dst_size = 0;
  FrameDeltas fd;
  /* a scratch image between two list-taking kernels of a plan (video_frame_list_scratch): frame i's copy sits i * scr_stride behind frame 0's */
  const uint8_t *scr0 = nullptr;
  size_t scr_stride = 0;
  FrameDeltas tmp;
};
thread_local FrameListCtx g_frame_list;
}
void video_frame_list_begin (int n, const void *const *src, void *const *dst, size_t src_size, size_t dst_size)
{
  FrameListCtx &c = g_frame_list;
  c.armed = n > 1 && n <= GSTAMD_MAX_BATCH;
  c.n = n;
  c.used = 0;
  c.mixed = false;
  c.src0 = (const uint8_t *) src[0], c.dst0 = (const uint8_t *) dst[0];
  c.src_size = src_size, c.dst_size = dst_size;
  c.scr0 = nullptr, c.scr_stride = 0;
  memset (&c.fd, 0, sizeof (c.fd));
  for (int i = 0; c.armed && i < n; i++) {
    c.fd.s[i] = (long long) ((const uint8_t *) src[i] - c.src0);
    c.fd.d[i] = (long long) ((const uint8_t *) dst[i] - c.dst0);
  }
}
// The plan's scratch image `p` exists once per frame of the armed list, per_frame bytes apart: a list-taking kernel that writes it (or reads it) is
// rebased frame by frame like one that touches the frames themselves - a two-kernel plan (scale into the image, pack from it) serves the list in two launches.
void video_frame_list_scratch (const void *p, size_t per_frame)
{
  FrameListCtx &c = g_frame_list;
  if (c.armed)
    c.scr0 = (const uint8_t *) p, c.scr_stride = per_frame;
}
int video_frame_list_end ()
{
  g_frame_list.armed = false;
  if (tuning_on ("GSTAMD_LIST_DEBUG"))
    fprintf (stderr, "frame list of %d: %d list launches%s\n", g_frame_list.n, g_frame_list.used, g_frame_list.mixed ? ", and kernels that take no lists" : "");
  return g_frame_list.mixed ? 0 : g_frame_list.used;
}
// a launcher whose kernel does NOT take frame lists writes dp: when that is inside frame 0 of an armed list the plan is not one the
// list may serve on its own (a patched kernel before or after it would have converted every frame, this one only frame 0): the
// caller then converts the other frames one by one (conversions are pure functions of the source frame, doing one twice is harmless)
void video_frame_list_touch (const void *dp)
{
  FrameListCtx &c = g_frame_list;
  const uint8_t *d8 = (const uint8_t *) dp;
  if (c.armed && ((d8 >= c.dst0 && d8 < c.dst0 + c.dst_size) || (c.scr0 && d8 >= c.scr0 && d8 < c.scr0 + c.scr_stride)))
    c.mixed = true;
}
// the list for a launch whose source / destination pointers are sp / dp (NULL: not a frame's plane, or no list): *nz = frames
static const FrameDeltas &frame_list_for (const void *sp, const void *dp, int *nz)
{
  static const FrameDeltas none = {};
  FrameListCtx &c = g_frame_list;
  *nz = 1;
  if (!c.armed)
    return none;
  const uint8_t *s8 = (const uint8_t *) sp, *d8 = (const uint8_t *) dp;
  const bool s_frame = s8 >= c.src0 && s8 < c.src0 + c.src_size, d_frame = d8 >= c.dst0 && d8 < c.dst0 + c.dst_size;
  const bool s_scr = c.scr0 && s8 >= c.scr0 && s8 < c.scr0 + c.scr_stride, d_scr = c.scr0 && d8 >= c.scr0 && d8 < c.scr0 + c.scr_stride;
  if (s_frame && d_frame) {
    c.used++;
    *nz = c.n;
    return c.fd;
  }
  if ((s_frame && d_scr) || (s_scr && d_frame)) {          /* frame -> scratch image, scratch image -> frame */
    for (int i = 0; i < c.n; i++) {
      c.tmp.s[i] = s_frame ? c.fd.s[i] : (long long) (i * c.scr_stride);
      c.tmp.d[i] = d_frame ? c.fd.d[i] : (long long) (i * c.scr_stride);
    }
    c.used++;
    *nz = c.n;
    return c.tmp;
  }
  if (d_scr || d_frame)
    c.mixed = true;             /* a kernel that writes frame 0's picture or scratch image and cannot be rebased: the list is not served by launches alone */
  return none;
}
// (GSTAMD_FRAME_Z: video_kernels.h)
// the same for launchers of other compilation units (video_deep_pack.hip)
const FrameDeltas &video_frame_list_for (const void *sp, const void *dp, int *nz) { return frame_list_for (sp, dp, nz); }

template <int CH>
__global__ __launch_bounds__ (256) void k_convert (FrontParams f, Planes pl, const int *__restrict__ vpair, ColorParams color,
    int pack0, int pack1, int pack2, int pack3, uint8_t *__restrict__ dst, int dstride, int spans_per_row, int vec_ok)
{
  convert_body<CH> (f, pl, vpair, color, pack0, pack1, pack2, pack3, dst, dstride, spans_per_row, vec_ok,
      (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

// gamma-mode = remap, unscaled, into a 4-byte destination, in ONE kernel: k_convert's body with the gamma chain between the colour stage and
// the packer, both tables in LDS (64 KB + 512 B: the encode table is a gather per component and pixel, which HBM / L2 serve badly - the
// three-launch composite spends 50 us of its 109 on it at 4K).  A workgroup loads the tables once and walks `rows` rows.
#define GSTAMD_GAMMA_LDS_BYTES (65536 + 512)
template <int CH>
__global__ __launch_bounds__ (512) void k_convert_gamma (FrontParams f, Planes pl, const int *__restrict__ vpair, ColorParams color,
    int pack0, int pack1, int pack2, int pack3, uint8_t *__restrict__ dst, int dstride, int spans_per_row, int vec_ok, GammaDev g, int rows)
{
  extern __shared__ uint4 gamma_lds[];
  uint8_t *enc = (uint8_t *) gamma_lds;
  uint16_t *dec = (uint16_t *) (enc + 65536);
  GammaChainFn fn;
  fn.g = g;
  if (g.comp) {                 /* the composed table: 256 bytes of LDS, not 64.5 KB */
    if (threadIdx.x < 64)
      ((uint32_t *) gamma_lds)[threadIdx.x] = ((const uint32_t *) g.comp)[threadIdx.x];
    fn.g.comp = (const uint8_t *) gamma_lds;
  } else {
    for (int i = (int) threadIdx.x; i < 4096; i += (int) blockDim.x)
      gamma_lds[i] = ((const uint4 *) g.enc)[i];
    if (threadIdx.x < 256)
      dec[threadIdx.x] = g.dec[threadIdx.x];
    fn.g.enc = enc;
    fn.g.dec = dec;
  }
  __syncthreads ();
  const int span = (int) (blockIdx.x * blockDim.x + threadIdx.x);
  const int y0 = (int) blockIdx.y * rows, y1 = y0 + rows < f.height ? y0 + rows : f.height;
  for (int y = y0; y < y1; y++)
    convert_body<CH, GammaChainFn> (f, pl, vpair, color, pack0, pack1, pack2, pack3, dst, dstride, spans_per_row, vec_ok, span, y, fn);
}

// GammaPlan::lut_direct: the composed gamma table over the converted rectangle of a 4-byte RGB destination, in place; four pixels per
// lane, the table in LDS
__global__ __launch_bounds__ (256) void k_lut3 (uint8_t *__restrict__ img, int stride, int w, int h, const uint8_t *__restrict__ comp, int keep)
{
  __shared__ uint32_t tab[64];
  if (threadIdx.x < 64)
    tab[threadIdx.x] = ((const uint32_t *) comp)[threadIdx.x];
  __syncthreads ();
  const uint8_t *t = (const uint8_t *) tab;
  const int x = (int) (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = (int) blockIdx.y;
  if (x >= w)
    return;
  uint32_t *p = (uint32_t *) (img + (size_t) y * stride) + x;
  if (x + 4 <= w && (((uintptr_t) p) & 15) == 0) {
    uint4 v = *(uint4 *) p;
    v.x = gamma_lut3_px (t, v.x, keep), v.y = gamma_lut3_px (t, v.y, keep), v.z = gamma_lut3_px (t, v.z, keep), v.w = gamma_lut3_px (t, v.w, keep);
    *(uint4 *) p = v;
  } else {
    for (int i = 0; i < 4 && x + i < w; i++)
      p[i] = gamma_lut3_px (t, p[i], keep);
  }
}

hipError_t launch_lut3 (uint8_t *img, int stride, int w, int h, const uint8_t *comp_dev, int keep, hipStream_t stream)
{
  video_frame_list_touch (img);
  hipLaunchKernelGGL (k_lut3, dim3 (((w + 3) / 4 + 255) / 256, h), dim3 (256), 0, stream, img, stride, w, h, comp_dev, keep);
  return hipGetLastError ();
}

// the 16-bit chain of a 10-bit source into an 8-bit 4-byte destination (video_deep.h): a lane = 4 pixels of two rows
__global__ __launch_bounds__ (256) void k_convert16 (FrontParams f, Planes pl, const int *__restrict__ vpair, Deep16Params d, PostParams post,
    uint8_t *__restrict__ dst, int dstride)
{
  convert16_rows2 (f, pl, vpair, d, post, dst, dstride, (int) (blockIdx.x * blockDim.x + threadIdx.x) * 4, 2 * (int) blockIdx.y);
}

template <int SEMI, int CH>
__global__ __launch_bounds__ (256) void k_convert16_fast (FrontParams f, Planes pl, const int *__restrict__ vpair, Deep16Params d, PostParams post,
    uint8_t *__restrict__ dst, int dstride)
{
  convert16_fast_rows2<SEMI, CH> (f, pl, vpair, d, post, dst, dstride, (int) (blockIdx.x * blockDim.x + threadIdx.x) * 4, 2 * (int) blockIdx.y);
}

hipError_t launch_convert16 (const FrontParams &f, const Planes &pl, const int *vpair_dev, const Deep16Params &d, const PostParams &post, uint8_t *dst,
    int dstride, hipStream_t stream)
{
  video_frame_list_touch (dst);
  const int variant = (f.width % 4) == 0 && !tuning_on ("GSTAMD_NO_CONVERT16_FAST") ? deep_front4_variant (f) : -1;
  if (variant >= 0) {
    /* planes with horizontally subsampled chroma, widths in whole 4-pixel blocks: layout and chroma filter are template parameters */
    const dim3 fgrid ((f.width / 4 + 255) / 256, (f.height + 1) / 2);
    switch (variant) {
      case 0: hipLaunchKernelGGL ((k_convert16_fast<0, CHROMA_H_NONE>), fgrid, dim3 (256), 0, stream, f, pl, vpair_dev, d, post, dst, dstride); break;
      case 1: hipLaunchKernelGGL ((k_convert16_fast<0, CHROMA_H_H2>), fgrid, dim3 (256), 0, stream, f, pl, vpair_dev, d, post, dst, dstride); break;
      case 2: hipLaunchKernelGGL ((k_convert16_fast<0, CHROMA_H_H2_CS>), fgrid, dim3 (256), 0, stream, f, pl, vpair_dev, d, post, dst, dstride); break;
      case 3: hipLaunchKernelGGL ((k_convert16_fast<1, CHROMA_H_NONE>), fgrid, dim3 (256), 0, stream, f, pl, vpair_dev, d, post, dst, dstride); break;
      case 4: hipLaunchKernelGGL ((k_convert16_fast<1, CHROMA_H_H2>), fgrid, dim3 (256), 0, stream, f, pl, vpair_dev, d, post, dst, dstride); break;
      default: hipLaunchKernelGGL ((k_convert16_fast<1, CHROMA_H_H2_CS>), fgrid, dim3 (256), 0, stream, f, pl, vpair_dev, d, post, dst, dstride); break;
    }
    return hipGetLastError ();
  }
  dim3 grid ((f.width / 4 + 256) / 256, (f.height + 1) / 2);
  hipLaunchKernelGGL (k_convert16, grid, dim3 (256), 0, stream, f, pl, vpair_dev, d, post, dst, dstride);
  return hipGetLastError ();
}

// 10-bit source, scaled in 16 bits (video_deep.h): front into an AYUV64 image, u16 passes, the last one fused with the convert stage.
// Plain one-lane-per-pixel kernels with the AYUV64 images in HBM: correctness and coverage first.
__global__ __launch_bounds__ (256) void k_front16 (FrontParams f, Planes pl, const int *__restrict__ vpair, uint8_t *__restrict__ img, int istride)
{
  front16_lane4 (f, pl, vpair, img, istride, (int) (blockIdx.x * blockDim.x + threadIdx.x) * 4, (int) blockIdx.y);
}

__global__ __launch_bounds__ (256) void k_scale16 (Deep16Image im, ScaleDev sd, int horizontal, uint8_t *__restrict__ dst, int dstride, int ow, int oh)
{
  scale16_lane (im, sd, horizontal != 0, dst, dstride, ow, oh, (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

__global__ __launch_bounds__ (256) void k_scale16_final (Deep16Image im, ScaleDev sd, int horizontal, Deep16Params d, PostParams post,
    uint8_t *__restrict__ dst, int dstride, int ow, int oh)
{
  scale16_final_lane (im, sd, horizontal != 0, d, post, dst, dstride, ow, oh, (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

template <int SEMI, int CH>
__global__ __launch_bounds__ (256) void k_front_hscale16 (FrontParams f, Planes pl, const int *__restrict__ vpair, ScaleDev sd, uint8_t *__restrict__ dst, int dstride, int ow)
{
  front_hscale16_lane<SEMI, CH> (f, pl, vpair, sd, dst, dstride, ow, (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

// the 16-bit front inside the first, horizontal u16 pass; false: this front has no specialised form (the caller runs k_front16 + k_scale16)
bool front_hscale16_usable (const FrontParams &f)
{
  return deep_front4_variant (f) >= 0 && !tuning_on ("GSTAMD_NO_CONVERT16_FAST");
}

hipError_t launch_front_hscale16 (const FrontParams &f, const Planes &pl, const int *vpair_dev, const ScaleDev &sd, uint8_t *dst, int dstride, int ow, hipStream_t stream)
{
  video_frame_list_touch (dst);          /* takes no frame list: says so should it ever write a frame (today: scratch images only) */
  const dim3 grid ((ow + 255) / 256, f.height);
  switch (deep_front4_variant (f)) {
    case 0: hipLaunchKernelGGL ((k_front_hscale16<0, CHROMA_H_NONE>), grid, dim3 (256), 0, stream, f, pl, vpair_dev, sd, dst, dstride, ow); break;
    case 1: hipLaunchKernelGGL ((k_front_hscale16<0, CHROMA_H_H2>), grid, dim3 (256), 0, stream, f, pl, vpair_dev, sd, dst, dstride, ow); break;
    case 2: hipLaunchKernelGGL ((k_front_hscale16<0, CHROMA_H_H2_CS>), grid, dim3 (256), 0, stream, f, pl, vpair_dev, sd, dst, dstride, ow); break;
    case 3: hipLaunchKernelGGL ((k_front_hscale16<1, CHROMA_H_NONE>), grid, dim3 (256), 0, stream, f, pl, vpair_dev, sd, dst, dstride, ow); break;
    case 4: hipLaunchKernelGGL ((k_front_hscale16<1, CHROMA_H_H2>), grid, dim3 (256), 0, stream, f, pl, vpair_dev, sd, dst, dstride, ow); break;
    default: hipLaunchKernelGGL ((k_front_hscale16<1, CHROMA_H_H2_CS>), grid, dim3 (256), 0, stream, f, pl, vpair_dev, sd, dst, dstride, ow); break;
  }
  return hipGetLastError ();
}

template <int SEMI, int CH>
__global__ __launch_bounds__ (256) void k_front16_fast (FrontParams f, Planes pl, const int *__restrict__ vpair, uint8_t *__restrict__ img, int istride)
{
  front16_fast_lane4<SEMI, CH> (f, pl, vpair, img, istride, (int) (blockIdx.x * blockDim.x + threadIdx.x) * 4, (int) blockIdx.y);
}

hipError_t launch_front16 (const FrontParams &f, const Planes &pl, const int *vpair_dev, uint8_t *img, int istride, hipStream_t stream)
{
  video_frame_list_touch (img);
  const int variant = (f.width % 4) == 0 && ((uintptr_t) img % 16) == 0 && (istride % 16) == 0 && !tuning_on ("GSTAMD_NO_CONVERT16_FAST") ? deep_front4_variant (f) : -1;
  if (variant >= 0) {
    const dim3 fgrid ((f.width / 4 + 255) / 256, f.height);
    switch (variant) {
      case 0: hipLaunchKernelGGL ((k_front16_fast<0, CHROMA_H_NONE>), fgrid, dim3 (256), 0, stream, f, pl, vpair_dev, img, istride); break;
      case 1: hipLaunchKernelGGL ((k_front16_fast<0, CHROMA_H_H2>), fgrid, dim3 (256), 0, stream, f, pl, vpair_dev, img, istride); break;
      case 2: hipLaunchKernelGGL ((k_front16_fast<0, CHROMA_H_H2_CS>), fgrid, dim3 (256), 0, stream, f, pl, vpair_dev, img, istride); break;
      case 3: hipLaunchKernelGGL ((k_front16_fast<1, CHROMA_H_NONE>), fgrid, dim3 (256), 0, stream, f, pl, vpair_dev, img, istride); break;
      case 4: hipLaunchKernelGGL ((k_front16_fast<1, CHROMA_H_H2>), fgrid, dim3 (256), 0, stream, f, pl, vpair_dev, img, istride); break;
      default: hipLaunchKernelGGL ((k_front16_fast<1, CHROMA_H_H2_CS>), fgrid, dim3 (256), 0, stream, f, pl, vpair_dev, img, istride); break;
    }
    return hipGetLastError ();
  }
  dim3 grid ((f.width / 4 + 256) / 256, f.height);
  hipLaunchKernelGGL (k_front16, grid, dim3 (256), 0, stream, f, pl, vpair_dev, img, istride);
  return hipGetLastError ();
}

hipError_t launch_scale16 (const Deep16Image &im, const ScaleDev &sd, bool horizontal, uint8_t *dst, int dstride, int ow, int oh, const Deep16Params *d,
    const PostParams *post, hipStream_t stream)
{
  video_frame_list_touch (dst);
  dim3 grid ((ow + 255) / 256, oh);
  if (d)
    hipLaunchKernelGGL (k_scale16_final, grid, dim3 (256), 0, stream, im, sd, horizontal ? 1 : 0, *d, *post, dst, dstride, ow, oh);
  else
    hipLaunchKernelGGL (k_scale16, grid, dim3 (256), 0, stream, im, sd, horizontal ? 1 : 0, dst, dstride, ow, oh);
  return hipGetLastError ();
}

// gamma-mode = remap: the per-pixel stages of video_gamma.h over an image (one lane per pixel; the tables are gathers from HBM / L2)
// (src and dst may be the same 16-bit image: a stage that only has the middle part runs in place)
__global__ __launch_bounds__ (256) void k_gamma_stage (GammaDev g, int mask, const uint8_t *src, int sstride, uint8_t *dst, int dstride, int w, int h)
{
  const int x = (int) (blockIdx.x * blockDim.x + threadIdx.x), y = (int) blockIdx.y;
  if (x < w && y < h)
    gamma_stage_px (g, mask, src, sstride, dst, dstride, x, y);
}

hipError_t launch_gamma_stage (const GammaDev &g, int mask, const uint8_t *src, int sstride, uint8_t *dst, int dstride, int w, int h, hipStream_t stream)
{
  video_frame_list_touch (dst);
  dim3 grid ((w + 255) / 256, h);
  hipLaunchKernelGGL (k_gamma_stage, grid, dim3 (256), 0, stream, g, mask, src, sstride, dst, dstride, w, h);
  return hipGetLastError ();
}

// dither-quantization > 1 into an ARGB64 / AYUV64 frame: the stage ahead of the (copying) packer as a pass over the finished frame
__global__ __launch_bounds__ (256) void k_dither16_image (DitherParams d, uint8_t *img, int stride, int w, int h)
{
  dither16_image_px (d, img, stride, w, h, (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

hipError_t launch_dither16_image (const DitherParams &d, uint8_t *img, int stride, int w, int h, hipStream_t stream)
{
  video_frame_list_touch (img);
  hipLaunchKernelGGL (k_dither16_image, dim3 ((w + 255) / 256, h), dim3 (256), 0, stream, d, img, stride, w, h);
  return hipGetLastError ();
}

// plane to plane between 8- and 10-bit planar formats (video_deep.h deep_planes_body): grid.y = luma rows, then chroma rows
__global__ __launch_bounds__ (256) void k_deep_planes (DeepPlanesParams d, DeepPlanesPtrs pp, FrameDeltas fl)
{
  GSTAMD_FRAME_Z;
  deep_planes_body (d, pp, (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y, fls_, fld_);
}

// ... sixteen samples per lane where a row of samples goes in and a row comes out (video_deep.h: k_deep_planes16); one wave per workgroup: a
// 1080p row is 120 lanes
template <int TO_HI>
__global__ __launch_bounds__ (64) void k_deep_planes16 (DeepPlanesParams d, DeepPlanesPtrs pp, FrameDeltas fl)
{
  GSTAMD_FRAME_Z;
  deep_planes16_body<TO_HI> (d, pp, (int) (blockIdx.x * 64 + threadIdx.x), (int) blockIdx.y, fls_, fld_);
}

hipError_t launch_deep_planes (const DeepPlanesParams &d, const DeepPlanesPtrs &pp, hipStream_t stream)
{
  const int ch = (d.height + (1 << d.h_sub) - 1) >> d.h_sub;
  int nz;
  const FrameDeltas &fl = frame_list_for (pp.in[0], pp.out[0], &nz);
  if (pp.vec && deep_planes16_ok (d) && !tuning_on ("GSTAMD_NO_DEEP_PLANES16")) {
    const dim3 grid16 ((d.width / 16 + 63) / 64, deep_planes16_rows (d), nz);
    if (d.in_hi && d.out_hi)
      hipLaunchKernelGGL (k_deep_planes16<2>, grid16, dim3 (64), 0, stream, d, pp, fl);
    else if (d.out_hi)
      hipLaunchKernelGGL (k_deep_planes16<1>, grid16, dim3 (64), 0, stream, d, pp, fl);
    else
      hipLaunchKernelGGL (k_deep_planes16<0>, grid16, dim3 (64), 0, stream, d, pp, fl);
    return hipGetLastError ();
  }
  dim3 grid (((d.width + 7) / 8 + 255) / 256, d.height + ch, nz);
  hipLaunchKernelGGL (k_deep_planes, grid, dim3 (256), 0, stream, d, pp, fl);
  return hipGetLastError ();
}

// 10-bit destinations: chroma downsample + dither + pack of the final AYUV64 image (video_deep.h pack16_body), one lane per 4-pixel block
__global__ __launch_bounds__ (256) void k_pack16 (PackPlanarParams pk, int hi_depth, DitherParams dt, const uint8_t *__restrict__ src, int sstride, DstPlanes16 d)
{
  pack16_body (pk, hi_depth, dt, src, sstride, d, (int) (blockIdx.x * blockDim.x + threadIdx.x) * 4, (int) blockIdx.y);
}

__global__ __launch_bounds__ (256) void k_pack16_packed (PackPlanarParams pk, int hi_depth, DitherParams dt, const uint8_t *__restrict__ src, int sstride,
    uint8_t *__restrict__ dst, int dstride)
{
  pack16_packed_body (pk, hi_depth, dt, src, sstride, dst, dstride, (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

hipError_t launch_pack16 (const PackPlanarParams &pk, int hi_depth, const DitherParams &dt, const uint8_t *src, int sstride, uint8_t *const planes[3],
    const int strides[3], hipStream_t stream)
{
  video_frame_list_touch (planes[0]);
  if (pk.kind == UNPACK_P422_16 || pk.kind == UNPACK_P422_UYVP || GSTAMD_KIND_PX16 (pk.kind) || pk.kind == UNPACK_V210) {          /* Y210, Y212_LE, Y410: a lane per macropixel / pixel */
    hipLaunchKernelGGL (k_pack16_packed, dim3 ((pack16_units (pk) + 255) / 256, pack16_rows (pk)), dim3 (256), 0, stream, pk, hi_depth, dt, src, sstride, planes[0], strides[0]);
    return hipGetLastError ();
  }
  DstPlanes16 d;
  for (int i = 0; i < 3; i++) {
    d.p[i] = planes[i];
    d.stride[i] = strides[i];
  }
  const int rows = (pk.height + (1 << pk.h_sub) - 1) >> pk.h_sub;
  dim3 grid ((pk.width / 4 + 256) / 256, rows);
  hipLaunchKernelGGL (k_pack16, grid, dim3 (256), 0, stream, pk, hi_depth, dt, src, sstride, d);
  return hipGetLastError ();
}

// ---- error diffusion on 16-bit lines (video_dither_ed.h, round 5) -------------------------------------------------------------------------
#define GSTAMD_ED_LINES 1024
__global__ __launch_bounds__ (256) void k_dither16_verterr (DitherParams d, uint8_t *__restrict__ img, int stride, int w, int h)
{
  const int x = (int) (blockIdx.x * blockDim.x + threadIdx.x);
  if (x >= w)
    return;
  Err4 e = err4_zero ();
#pragma unroll 4
  for (int y = 0; y < h; y++) {
    Px16 *q = (Px16 *) (img + (size_t) y * stride) + x;
    *q = ed16_verterr_px (d, *q, e);
  }
}

// k_dither_ed's wavefront over 8-byte pixels: lane r = line band + r, three pixels behind lane r - 1, a line's errors in a 4-slot LDS ring
template <int METHOD>
__global__ __launch_bounds__ (GSTAMD_ED_LINES) void k_dither16_ed (DitherParams d, uint8_t *__restrict__ img, int stride, int w, int h, Err4 *__restrict__ carry)
{
  __shared__ Err4 ring[GSTAMD_ED_LINES][4];
  const int r = (int) threadIdx.x;
  for (int band = 0; band < h; band += GSTAMD_ED_LINES) {
    const int rows = h - band < GSTAMD_ED_LINES ? h - band : GSTAMD_ED_LINES;
    const int y = band + r;
    const bool active = r < rows, to_carry = r == rows - 1 && band + rows < h;
    Px16 *row = (Px16 *) (img + (size_t) (active ? y : band) * stride);
    Err4 left = err4_zero ();
    const int nsteps = w + 3 * (rows - 1);
    Px16 next_px = {0u, 0u};
    if (active && r == 0)
      next_px = row[0];
    for (int s = 0; s < nsteps; s++) {
      const int x = s - 3 * r;
      const Px16 px = next_px;
      if (active && x + 1 >= 0 && x + 1 < w)
        next_px = row[x + 1];                     /* in flight across the barrier */
      if (active && x >= 0 && x < w) {
        Err4 p[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const int j = x + k;
          if (y == 0 || j >= w)
            p[k] = err4_zero ();
          else if (r == 0)
            p[k] = carry[j];
          else
            p[k] = ring[r - 1][(j + 1) & 3];
        }
        if (x == 0)
          left = err4_zero ();
        const Px16 out = METHOD == GSTAMD_DITHER_FLOYD_STEINBERG ? ed16_floyd_px (d, px, left, p[0], p[1], p[2]) : ed16_sierra_px (d, px, left, p[1], p[2]);
        row[x] = out;
        ring[r][(x + 1) & 3] = left;
        if (to_carry)
          carry[x] = left;
      }
      __syncthreads ();
    }
  }
}

// the dither stage over a 16-bit image in place: ordered (k_dither16_image) or one of the error-diffusion methods
hipError_t launch_dither16_any (const DitherParams &d, uint8_t *img, int stride, int w, int h, hipStream_t stream, void *ed_carry)
{
  if (!dither_is_diffusion (d))
    return launch_dither16_image (d, img, stride, w, h, stream);
  video_frame_list_touch (img);
  if (d.method == GSTAMD_DITHER_VERTERR) {
    hipLaunchKernelGGL (k_dither16_verterr, dim3 ((w + 255) / 256), dim3 (256), 0, stream, d, img, stride, w, h);
    return hipGetLastError ();
  }
  if (h > GSTAMD_ED_LINES && !ed_carry)
    return hipErrorInvalidValue;
  if (d.method == GSTAMD_DITHER_FLOYD_STEINBERG)
    hipLaunchKernelGGL (k_dither16_ed<GSTAMD_DITHER_FLOYD_STEINBERG>, dim3 (1), dim3 (GSTAMD_ED_LINES), 0, stream, d, img, stride, w, h, (Err4 *) ed_carry);
  else
    hipLaunchKernelGGL (k_dither16_ed<GSTAMD_DITHER_SIERRA_LITE>, dim3 (1), dim3 (GSTAMD_ED_LINES), 0, stream, d, img, stride, w, h, (Err4 *) ed_carry);
  return hipGetLastError ();
}

__global__ __launch_bounds__ (256) void k_pack16_down_v (PackPlanarParams pk, uint8_t *__restrict__ img, int stride)
{
  pack16_down_v_px (pk, img, stride, (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

__global__ __launch_bounds__ (256) void k_pack16_down_h (PackPlanarParams pk, uint8_t *__restrict__ img, int stride)
{
  pack16_down_h_px (pk, img, stride, (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

// error-diffusion dither ahead of a 10 / 12 / 16-bit packer: the chroma downsamplers in place on the AYUV64 image (which must not be the
// caller's frame), the dither pass over every pixel of it, then the packer as a pure selection
hipError_t launch_pack16_ed (const PackPlanarParams &pk, int hi_depth, const DitherParams &dt, uint8_t *img, int sstride, uint8_t *const planes[3],
    const int strides[3], hipStream_t stream, void *ed_carry)
{
  video_frame_list_touch (planes[0]);
  const int rows = (pk.height + (1 << pk.h_sub) - 1) >> pk.h_sub;
  const dim3 grid ((pk.width + 255) / 256, rows);
  if (pk.down_v)
    hipLaunchKernelGGL (k_pack16_down_v, grid, dim3 (256), 0, stream, pk, img, sstride);
  if (pk.down_h && pk.w_sub == 1)
    hipLaunchKernelGGL (k_pack16_down_h, grid, dim3 (256), 0, stream, pk, img, sstride);
  hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return e;
  if ((e = launch_dither16_any (dt, img, sstride, pk.width, pk.height, stream, ed_carry)) != hipSuccess)
    return e;
  DitherParams off;
  memset (&off, 0, sizeof (off));
  return launch_pack16 (pack_select_only (pk), hi_depth, off, img, sstride, planes, strides, stream);
}

// 4-byte 8-bit pixels -> deep planar / semi-planar YUV in one kernel (video_deep.h: k_encode16), one wave per workgroup like k_encode420
template <int SEMI, int NB>
__global__ __launch_bounds__ (64) void k_encode16 (Enc16Params ep, const uint8_t *__restrict__ src, int sstride, DstPlanes16 d, FrameDeltas fl)
{
  GSTAMD_FRAME_Z;
  enc16_block<SEMI, NB> (ep, src, sstride, d, (int) (blockIdx.x * 64 + threadIdx.x) * 4 * NB, (int) blockIdx.y, fls_, fld_);
}

// eight pixels of a line per lane where the rows allow (width % 8, destination rows on 16 bytes): the kernel is bound by the number of
// memory instructions it issues, not by their bytes
hipError_t launch_encode16 (const Enc16Params &ep, const uint8_t *src, int sstride, uint8_t *const planes[3], const int strides[3], hipStream_t stream)
{
  DstPlanes16 d;
  bool wide = (ep.width % 8) == 0 && !tuning_on ("GSTAMD_ENCODE16_NARROW");
  for (int i = 0; i < 3; i++) {
    d.p[i] = planes[i];
    d.stride[i] = strides[i];
    if (planes[i])
      wide = wide && ((uintptr_t) planes[i] % 16) == 0 && (strides[i] % 16) == 0;
  }
  int nz;
  const FrameDeltas &fl = frame_list_for (src, planes[0], &nz);
  /* (a single 4K frame is one round of waves either way - 8640 or 17280 of them for 8192 places - and runs as a load phase, a compute phase and
     a store phase; the narrow form's second half-round overlaps a little: 23.9 against 25.7 us.  Lists of 8: 17.4 against 13.9 us per frame) */
  wide = wide && (nz > 1 || tuning_on ("GSTAMD_ENCODE16_WIDE"));
  const int nb = wide ? 2 : 1;
  const dim3 grid ((ep.width / (4 * nb) + 63) / 64, (ep.height + (1 << ep.pk.h_sub) - 1) >> ep.pk.h_sub, nz);
  if (ep.pk.kind == UNPACK_SEMI) {
    if (wide)
      hipLaunchKernelGGL ((k_encode16<1, 2>), grid, dim3 (64), 0, stream, ep, src, sstride, d, fl);
    else
      hipLaunchKernelGGL ((k_encode16<1, 1>), grid, dim3 (64), 0, stream, ep, src, sstride, d, fl);
  } else {
    if (wide)
      hipLaunchKernelGGL ((k_encode16<0, 2>), grid, dim3 (64), 0, stream, ep, src, sstride, d, fl);
    else
      hipLaunchKernelGGL ((k_encode16<0, 1>), grid, dim3 (64), 0, stream, ep, src, sstride, d, fl);
  }
  return hipGetLastError ();
}

// the dither stage as a pass over the packed destination rectangle (video_dither.h)
__global__ __launch_bounds__ (256) void k_dither4 (DitherParams d, uint8_t *__restrict__ img, int stride, int w, int h)
{
  dither_lane4 (d, img, stride, w, h, (int) (blockIdx.x * blockDim.x + threadIdx.x) * 4, (int) blockIdx.y);
}

// vertical error carry (dither_verterr_u8): a lane per column walks down the rectangle, the error in registers; the loads of a column do
// not depend on the recurrence, so they run ahead
__global__ __launch_bounds__ (256) void k_dither_verterr (DitherParams d, uint8_t *__restrict__ img, int stride, int w, int h)
{
  const int x = (int) (blockIdx.x * blockDim.x + threadIdx.x);
  if (x >= w)
    return;
  Err4 e = err4_zero ();
  uint32_t *p = (uint32_t *) img + x;
#pragma unroll 8
  for (int y = 0; y < h; y++) {
    uint32_t *q = (uint32_t *) ((uint8_t *) p + (size_t) y * stride);
    *q = ed_verterr_px (d, *q, e);
  }
}

// Floyd-Steinberg / Sierra Lite (video_dither_ed.h): lane r = line band + r, three pixels behind lane r - 1; the errors a line leaves for
// the next sit in a 4-slot LDS ring per line (slot (x + 1) & 3 = pixel x: the line below reads pixels x' .. x' + 2 one to three steps after
// they were written and the slot is rewritten at the fourth); the last line of a 1024-line band hands its errors to the next band through
// `carry` (w entries in HBM).  One workgroup: the recurrence is serial in x and y, the wavefront is all the parallelism there is.
template <int METHOD>
__global__ __launch_bounds__ (GSTAMD_ED_LINES) void k_dither_ed (DitherParams d, uint8_t *__restrict__ img, int stride, int w, int h, Err4 *__restrict__ carry)
{
  __shared__ Err4 ring[GSTAMD_ED_LINES][4];
  __shared__ Err4 a0s[GSTAMD_ED_LINES];
  __shared__ Err4 a0_carry;
  const int r = (int) threadIdx.x;
  for (int band = 0; band < h; band += GSTAMD_ED_LINES) {
    const int rows = h - band < GSTAMD_ED_LINES ? h - band : GSTAMD_ED_LINES;
    const int y = band + r;
    const bool active = r < rows, to_carry = r == rows - 1 && band + rows < h;
    uint32_t *row = (uint32_t *) (img + (size_t) (active ? y : band) * stride);
    Err4 left = err4_zero ();
    const int nsteps = w + 3 * (rows - 1);
    uint32_t next_px = active && r == 0 ? row[0] : 0u;
    for (int s = 0; s < nsteps; s++) {
      const int x = s - 3 * r;
      const uint32_t px = next_px;
      if (active && x + 1 >= 0 && x + 1 < w)
        next_px = row[x + 1];                     /* in flight across the barrier */
      if (active && x >= 0 && x < w) {
        Err4 p[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const int j = x + k;
          if (y == 0 || j >= w)
            p[k] = err4_zero ();
          else if (r == 0)
            p[k] = carry[j];
          else
            p[k] = ring[r - 1][(j + 1) & 3];
        }
        uint32_t out;
        if (METHOD == GSTAMD_DITHER_FLOYD_STEINBERG) {
          Err4 a0 = err4_zero ();
          if (x == 0)
            left = y == 0 ? err4_zero () : (r == 0 ? a0_carry : a0s[r - 1]);
          out = ed_floyd_px (d, px, left, p[0], p[1], p[2], x == 0, x == w - 1, &a0);
          if (x == 0) {
            a0s[r] = a0;
            if (to_carry)
              a0_carry = a0;
          }
        } else {
          out = ed_sierra_px (d, px, left, p[1], p[2]);
        }
        row[x] = out;
        ring[r][(x + 1) & 3] = left;
        if (to_carry)
          carry[x] = left;
      }
      __syncthreads ();
    }
  }
}

hipError_t launch_dither4 (const DitherParams &d, uint8_t *img, int stride, int w, int h, hipStream_t stream, void *ed_carry)
{
  video_frame_list_touch (img);
  if (d.method == GSTAMD_DITHER_VERTERR) {
    hipLaunchKernelGGL (k_dither_verterr, dim3 ((w + 255) / 256), dim3 (256), 0, stream, d, img, stride, w, h);
    return hipGetLastError ();
  }
  if (d.method == GSTAMD_DITHER_FLOYD_STEINBERG || d.method == GSTAMD_DITHER_SIERRA_LITE) {
    if (h > GSTAMD_ED_LINES && !ed_carry)
      return hipErrorInvalidValue;
    if (d.method == GSTAMD_DITHER_FLOYD_STEINBERG)
      hipLaunchKernelGGL (k_dither_ed<GSTAMD_DITHER_FLOYD_STEINBERG>, dim3 (1), dim3 (GSTAMD_ED_LINES), 0, stream, d, img, stride, w, h, (Err4 *) ed_carry);
    else
      hipLaunchKernelGGL (k_dither_ed<GSTAMD_DITHER_SIERRA_LITE>, dim3 (1), dim3 (GSTAMD_ED_LINES), 0, stream, d, img, stride, w, h, (Err4 *) ed_carry);
    return hipGetLastError ();
  }
  dim3 grid ((w / 4 + 256) / 256, h);
  hipLaunchKernelGGL (k_dither4, grid, dim3 (256), 0, stream, d, img, stride, w, h);
  return hipGetLastError ();
}

// up to 32 independent frames of one format per launch (blockIdx.z = frame): amortises launch ramp/tail
struct FrameBatch {
  const uint8_t *y[GSTAMD_MAX_BATCH];
  const uint8_t *uv[GSTAMD_MAX_BATCH];
  uint8_t *dst[GSTAMD_MAX_BATCH];
  int ystride, uvstride, dstride;
};

__device__ __forceinline__ Planes batch_planes (const FrameBatch &b, int f)
{
  Planes pl;
  pl.p[0] = b.y[f];
  pl.p[1] = b.uv[f];
  pl.p[2] = pl.p[3] = nullptr;
  pl.stride[0] = b.ystride;
  pl.stride[1] = b.uvstride;
  pl.stride[2] = pl.stride[3] = 0;
  return pl;
}


template <class SRC>
__global__ __launch_bounds__ (256) void k_hscale (SRC src, ScaleDev sd, Dst dst, int out_w, int rows)
{
  hscale_body<SRC> (src, sd, dst, out_w, rows, (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

// LDS-staged horizontal pass: block = 256 consecutive outputs of one row; the source span under them is
// evaluated once into LDS (<= HSCALE_LDS_PX pixels), then every lane filters from LDS.
#define HSCALE_LDS_PX 12288
#define WAVE_TILE_LDS_BYTES 16384      // per wave: above this the 256-lane LDS kernels take over
template <class SRC>
__global__ __launch_bounds__ (256) void k_hscale_lds (SRC src, ScaleDev sd, Dst dst, int out_w, int rows)
{
  extern __shared__ uint32_t lds[];     // max_span pixels (launcher sizes it)
  const int y = blockIdx.y, t0 = blockIdx.x * blockDim.x;
  const int t1 = t0 + (int) blockDim.x < out_w ? t0 + (int) blockDim.x : out_w;
  int x_lo, x_hi;
  hscale_span (sd, t0, t1, &x_lo, &x_hi);
  hscale_stage<SRC> (src, lds, x_lo, x_hi, y, (int) threadIdx.x, (int) blockDim.x);
  __syncthreads ();
  const int x = t0 + (int) threadIdx.x;
  if (x < out_w)
    hscale_from_lds (lds, x_lo, sd, dst, x, y);
}

// fused nearest/2-tap scaler with the two source lines staged in LDS (spans <= SCALE2_LDS_PX pixels each)
#define SCALE2_LDS_PX 6144
template <class SRC>
__global__ __launch_bounds__ (256) void k_scale2x2_lds (SRC src, ScaleDev sh, ScaleDev sv, int h_first, Dst dst, int out_w, int out_h,
    int span)
{
  extern __shared__ uint32_t lds2[];    // 2 x span pixels
  uint32_t *lds_a = lds2, *lds_b = lds2 + span;
  const int y = blockIdx.y, t0 = blockIdx.x * blockDim.x;
  const int t1 = t0 + (int) blockDim.x < out_w ? t0 + (int) blockDim.x : out_w;
  int x_lo, x_hi;
  hscale_span (sh, t0, t1, &x_lo, &x_hi);
  const int ya = (int) sv.offset[y];
  src.stage (lds_a, x_lo, x_hi, ya, (int) threadIdx.x, (int) blockDim.x);
  if (sv.kind == SCALE_2TAP)
    src.stage (lds_b, x_lo, x_hi, ya + 1, (int) threadIdx.x, (int) blockDim.x);
  __syncthreads ();
  const int x = t0 + (int) threadIdx.x;
  if (x < out_w)
    dst.put (x, y, scale2x2_from_lds (lds_a, lds_b, x_lo, sh, sv, h_first, x, y));
}

// ------------------------------------------------------------------------------------------------
// wave-tile scalers (video_scale_fast.h): workgroup = one wave = tile_w outputs of one row
// ------------------------------------------------------------------------------------------------
static __device__ __forceinline__ void wave_lds_sync ()
{
  __builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier ();
  __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "wavefront");
}

template <class SRC>
__global__ __launch_bounds__ (64) void k_scale2x2_wave (SRC src, ScaleDev sh, ScaleDev sv, int h_first, Dst dst, PostFast pf, int out_w,
    int out_h, int tile_w, int lds_px, int packed)
{
  extern __shared__ uint32_t lds_w[];
  uint32_t *la = lds_w, *lb = lds_w + lds_px;
  const int lane = (int) threadIdx.x, y = (int) blockIdx.y, t0 = (int) blockIdx.x * tile_w;
  const int t1 = t0 + tile_w < out_w ? t0 + tile_w : out_w;
  int x_lo, x_hi;
  hscale_span (sh, t0, t1, &x_lo, &x_hi);
  const int xa = x_lo & ~7, ya = (int) sv.offset[y];
  tile_stage_row (src, la, xa, x_hi, ya, lane, packed);
  if (sv.kind == SCALE_2TAP)
    tile_stage_row (src, lb, xa, x_hi, ya + 1, lane, packed);
  wave_lds_sync ();
  scale2x2_tile_lane (la, lb, xa, sh, sv, h_first, dst, pf, t0, t1, y, lane);
}

template <class SRC>
__global__ __launch_bounds__ (64) void k_hscale_wave (SRC src, ScaleDev sd, Dst dst, PostFast pf, int out_w, int rows, int tile_w,
    int lds_px, int packed)
{
  extern __shared__ uint32_t lds_w[];
  const int lane = (int) threadIdx.x, y = (int) blockIdx.y, t0 = (int) blockIdx.x * tile_w;
  const int t1 = t0 + tile_w < out_w ? t0 + tile_w : out_w;
  int x_lo, x_hi;
  hscale_span (sd, t0, t1, &x_lo, &x_hi);
  const int xa = x_lo & ~7;
  tile_stage_row (src, lds_w, xa, x_hi, y, lane, packed);
  wave_lds_sync ();
  hscale_tile_lane (lds_w, xa, sd, dst, pf, t0, t1, y, lane);
}

// horizontal N-tap pass from the source frame as byte dot products: three byte planes of plane_w words each in LDS
template <int NW>
__global__ __launch_bounds__ (64) void k_hscale_dot4_wave (SrcFront src, ScaleDev sd, Dst dst, PostFast pf, int out_w, int rows, int tile_w,
    int plane_w, int packed)
{
  extern __shared__ uint32_t lds_w[];
  uint32_t *py = lds_w, *pu = lds_w + plane_w, *pv = lds_w + 2 * plane_w;
  const int lane = (int) threadIdx.x, y = (int) blockIdx.y, t0 = (int) blockIdx.x * tile_w;
  const int t1 = t0 + tile_w < out_w ? t0 + tile_w : out_w;
  int x_lo, x_hi;
  hscale_span (sd, t0, t1, &x_lo, &x_hi);
  const int xa = x_lo & ~7;
  Dot4Taps<NW> ft;
  hscale_dot4_fetch<NW> (sd, xa, t0, t1, lane, ft);
#ifdef GSTAMD_TUNING
  if (!(packed & 0x200))
#endif
    tile_stage_row_planes (src, py, pu, pv, xa, x_hi, y, lane, packed & 1);
  wave_lds_sync ();
#ifdef GSTAMD_TUNING
  if (!(packed & 0x100))
#endif
    hscale_dot4_lane<NW> (py, pu, pv, ft, sd, sd.nw, dst, pf, t0, t1, y, lane);
}

// horizontal N-tap pass from a 2x horizontally subsampled planar / semi-planar frame (video_hscale420.h): a wave walks
// rows_per_wave lines of its tile, taps in registers, 16 source pixels per lane and line
template <int NW>
__global__ __launch_bounds__ (64) void k_hscale420_dot4 (SrcFront src, ScaleDev sd, Dst dst, PostFast pf, int out_w, int rows, int tile_w,
    int rows_per_wave, int ablate)
{
  extern __shared__ uint32_t lds_w[];
  const int plane_w = GSTAMD_H420_PLANE_BYTES / 4;
  uint32_t *py = lds_w, *pu = lds_w + plane_w, *pv = lds_w + 2 * plane_w;
  const int lane = (int) threadIdx.x, t0 = (int) blockIdx.x * tile_w;
  const int t1 = t0 + tile_w < out_w ? t0 + tile_w : out_w;
  const int y0 = (int) blockIdx.y * rows_per_wave, y1 = y0 + rows_per_wave < rows ? y0 + rows_per_wave : rows;
  int x_lo, x_hi;
  hscale_span (sd, t0, t1, &x_lo, &x_hi);
  const int xa = x_lo & ~15;
  Dot4Taps<NW> ft;
  hscale_dot4_fetch<NW> (sd, xa, t0, t1, lane, ft);
  H420State c;
  h420_begin (src, c, xa, x_hi, y0, lane);
#ifndef GSTAMD_TUNING
  ablate = 0;                   /* stage skipping is for -DGSTAMD_TUNING profiling builds only */
#endif
  for (int y = y0; y < y1; y++) {
    const int late = ablate & 1;
    if (!(ablate & 0x200))
      h420_stage_line_any (src, c, py, pu, pv, xa, x_hi, y, (y + 1 < y1 && !late) ? y + 1 : -1, lane);
    wave_lds_sync ();
    if (!(ablate & 0x100))
      hscale_dot4_lane<NW> (py, pu, pv, ft, sd, sd.nw, dst, pf, t0, t1, y, lane);
    if (late && y + 1 < y1)
      h420_request_line (src, c, xa, x_hi, y + 1, lane);
    wave_lds_sync ();
  }
}

// the regular 4:2:0 case of the same pass (video_hscale420.h, second half): line pairs, fixed register roles, straight-line
// memory operations (every pair: 2 luma + the chroma-row loads, then 8 stores)
template <int NW, int CH, int SEMI>
__global__ __launch_bounds__ (64) void k_hscale420_reg (H420RegParams p, int n_taps)
{
  extern __shared__ uint32_t lds_w[];
  const int lane = (int) threadIdx.x, t0 = (int) blockIdx.x * p.tile_w;
  const int t1 = t0 + p.tile_w < p.out_w ? t0 + p.tile_w : p.out_w;
  const int pairs = p.height / 2 + 1, ppw = p.lines_per_wave / 2;
  const int u0 = (int) blockIdx.y * ppw, u1 = u0 + ppw < pairs ? u0 + ppw : pairs;
  int x_lo, x_hi;
  h420r_span (p, n_taps, t0, t1, &x_lo, &x_hi);
  const int xa = x_lo & ~15, w0 = 4 * lane;
  int x0 = xa + 16 * lane;
  if (x0 + 16 > p.width)
    x0 = p.width - 16;                  // lanes past the span: harmless loads, their LDS bytes meet zero taps only
  Dot4Taps<NW> ft;
  h420r_fetch_taps<NW> (p, xa, t0, t1, lane, ft);
  uint32_t P[8], Q[8];
  {
    H420Raw r;
    h420r_load_raw<SEMI> (p, h420r_crow (p, u0 - 1), x0 >> 1, r);
    h420_filter_raw2<CH> (SEMI != 0, p.u_first != 0, r, P);
  }
  H420Pair cur, nxt;
  h420r_request<SEMI> (p, u0, x0, cur);
  for (int u = u0; u < u1; u += 2) {
    h420r_stage_pair<CH, SEMI> (p, cur, P, Q, lds_w, w0);
    h420r_request<SEMI> (p, u + 1 < u1 ? u + 1 : u1 - 1, x0, nxt);
    wave_lds_sync ();
    h420r_filter_pair<NW> (p, lds_w, ft, u, t0, t1, lane);
    wave_lds_sync ();
    if (u + 1 >= u1)
      break;
    h420r_stage_pair<CH, SEMI> (p, nxt, Q, P, lds_w, w0);
    h420r_request<SEMI> (p, u + 2 < u1 ? u + 2 : u1 - 1, x0, cur);
    wave_lds_sync ();
    h420r_filter_pair<NW> (p, lds_w, ft, u + 1, t0, t1, lane);
    wave_lds_sync ();
  }
}

// unscaled packed 4:2:2 -> 4-byte RGB (video_422_fast.h): a lane = 8 pixels of one line
__global__ __launch_bounds__ (256) void k_convert422 (Fast422Params p, const uint8_t *src, int sstride, uint8_t *dst, int dstride, FrameDeltas fl)
{
  GSTAMD_FRAME_Z;
  src += fls_, dst += fld_;
  const int x0 = (int) (blockIdx.x * blockDim.x + threadIdx.x) * 8, y = (int) blockIdx.y;
  if (x0 < p.fp.width)
    convert422_lane8_any (p, src + (size_t) y * sstride, dst + (size_t) y * dstride, x0);
}

__global__ __launch_bounds__ (256) void k_convert422_ayuv (Fast422Params p, const uint8_t *src, int sstride, uint8_t *dst, int dstride, FrameDeltas fl)
{
  GSTAMD_FRAME_Z;
  src += fls_, dst += fld_;
  const int x0 = (int) (blockIdx.x * blockDim.x + threadIdx.x) * 8, y = (int) blockIdx.y;
  if (x0 < p.fp.width)
    convert422_lane8_ayuv (p, src + (size_t) y * sstride, dst + (size_t) y * dstride, x0);
}

// unscaled planar 4:2:0 -> 4-byte RGB, nearest chroma (video_422_fast.h): a lane = 8 pixels of a line pair
__global__ __launch_bounds__ (256) void k_convert420p (Fast420pParams p, uint8_t *dst, int dstride, FrameDeltas fl)
{
  GSTAMD_FRAME_Z;
  p.y += fls_, p.u += fls_, p.v += fls_, dst += fld_;
  const int x0 = (int) (blockIdx.x * blockDim.x + threadIdx.x) * 8, r = (int) blockIdx.y;
  if (x0 < p.fp.width)
    convert420p_lane8x2 (p, dst, dstride, x0, r);
}

// vertical N-tap pass over an AYUV image, 4 pixels per lane, one wave per workgroup
__global__ __launch_bounds__ (64) void k_vscale_pk (SrcImage src, ScaleDev sd, Dst dst, PostFast pf, int width, int out_h)
{
  vscale_pk_lane (src, sd, dst, pf, width, out_h, (int) (blockIdx.x * 64 + threadIdx.x) * 4, (int) blockIdx.y);
}

template <int R>
__global__ __launch_bounds__ (64) void k_vscale_pk_rows (SrcImage src, ScaleDev sd, Dst dst, PostFast pf, int width, int out_h)
{
  vscale_pk_rows_lane<R> (src, sd, dst, pf, width, out_h, (int) (blockIdx.x * 64 + threadIdx.x) * 4, (int) blockIdx.y * R);
}

template <class SRC>
__global__ __launch_bounds__ (256) void k_scale2x2 (SRC src, ScaleDev sh, ScaleDev sv, int h_first, Dst dst, int out_w, int out_h)
{
  scale2x2_body<SRC> (src, sh, sv, h_first, dst, out_w, out_h, (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

template <class SRC>
__global__ __launch_bounds__ (256) void k_vscale (SRC src, ScaleDev sd, Dst dst, int width, int out_h)
{
  vscale_body<SRC> (src, sd, dst, width, out_h, (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

// fused bilinear scaler from semi-planar 4:2:0 (video_bilinear_fast.h): workgroup = one wave = 256 outputs of one row
template <int CH, int L>
__global__ __launch_bounds__ (64) void k_bilinear420 (BilParams bp, Planes pl, uint8_t *__restrict__ dst, int dstride, int vec, int tiles_x)
{
  extern __shared__ uint32_t lds_w[];
  const BilLds lds = bil_lds (lds_w, bp.ylen);
  /* XCD-aware block order (wide_block_map): an XCD walks 32 consecutive output rows of one column tile, so the chroma rows that
   * neighbouring output rows share (and the luma rows of overlapping windows) are found in ITS L2 - with the plain (tile, row)
   * grid vertically adjacent tiles land on different XCDs and the L2 -> fabric read traffic was 1.83x the source bytes (PMC) */
  int tile, y;
  if (!wide_block_map ((int) blockIdx.x, tiles_x, bp.out_h, &tile, &y))
    return;
  const int lane = (int) threadIdx.x, t0 = tile * bp.tile_w;
  const int t1 = t0 + bp.tile_w < bp.out_w ? t0 + bp.tile_w : bp.out_w;
  const int r0 = (int) bp.voffset[y];
  BilRegs r;
  bil_fetch (bp, pl, t0, t1, r0, lane, vec != 0, r);
  bil_commit (bp, t0, t1, lane, r, &lds);
  __syncthreads ();                       /* one wave per workgroup: orders the LDS writes before the reads */
  bil_emit<CH, L> (bp, dst, dstride, t0, t1, y, r0, lane, &lds);
}

// the same path with the chroma work done once per source pixel in byte lanes (video_bilinear_rows.h): workgroup = one wave =
// 256 outputs x `bp.rows` consecutive output rows
template <int CH, int L, int NP>
static __device__ __forceinline__ void bilinear420_rows_body (const BilParams &bp, const Planes &pl, uint8_t *__restrict__ dst, int dstride, int tiles_x, int fblock)
{
  /* the waves of a workgroup are independent (no barrier, an LDS slice each); a workgroup of several only makes the dispatcher's
   * job smaller - it places workgroups one at a time, and with thousands of single-wave groups the last waves start microseconds
   * after the first.  Wave w of workgroup b plays block (b / 8 * waves + w) * 8 + b % 8 of the XCD-aware map. */
  extern __shared__ __attribute__ ((aligned (16))) uint8_t lds_all[];
  const int wave = __builtin_amdgcn_readfirstlane ((int) threadIdx.x >> 6), waves = (int) blockDim.x >> 6;      /* wave-uniform: keep it scalar */
  uint8_t *lds = lds_all + wave * (BILR_PLANES * BILR_PLANE_BYTES);
  int tile, g;
  if (!wide_block_map (((fblock >> 3) * waves + wave) * 8 + (fblock & 7), tiles_x, bp.strips, &tile, &g))
    return;
  const int lane = (int) threadIdx.x & 63, t0 = tile * bp.rows_tile_w;
  const int t1 = t0 + bp.rows_tile_w < bp.out_w ? t0 + bp.rows_tile_w : bp.out_w;
  int x_lo, x_hi, k_lo, k_hi;
  bil_span (bp, t0, t1, &x_lo, &x_hi, &k_lo, &k_hi);
  const int xa = x_lo & ~15;
  BilrLane c;
  bilr_lane_setup<NP> (bp, t0, t1, xa, lane, c);
  BilrState st;
  bilr_state_init (st);
  uint32_t q4[4][2];
  layout_init<L> (q4);
  uint32_t (&q)[2] = q4[0];
  /* the strip's rows.  Their table entries (first source line, second vertical tap) are read once, one lane per row: a table read
   * inside the loop is a vector load followed by s_waitcnt vmcnt(0), which also waits for the prefetch and the stores of the
   * row before.  The first row's entry comes through the scalar cache on its own so that the first loads can go out sooner. */
  const int y0 = (int) ((unsigned) g * (unsigned) bp.out_h / (unsigned) bp.strips);
  const int y1 = (int) ((unsigned) (g + 1) * (unsigned) bp.out_h / (unsigned) bp.strips);
  const int first_r0 = (int) bp.voffset[y0];
  const int yl = y0 + lane < bp.out_h ? y0 + lane : bp.out_h - 1;
  const int tab_r0 = (int) bp.voffset[yl], tab_p1 = (int) bp.vtaps[(size_t) yl * 2 + 1];
  BilrReq rq;
  bilr_request (bp, pl, st, first_r0, xa, x_hi, lane, rq);
  const bool even = bilr_even_offsets (bp);       /* wave-uniform: which form of the LDS pair reads (video_bilinear_rows.h bilr_pair) */
  for (int y = y0; y < y1; y++) {
    bilr_install<CH> (bp, pl, st, rq, __builtin_amdgcn_readlane (tab_r0, y - y0), xa, x_hi, lane, lds);
    if (y + 1 < y1)
      bilr_request (bp, pl, st, __builtin_amdgcn_readlane (tab_r0, y + 1 - y0), xa, x_hi, lane, rq);   /* in flight while this row is emitted */
    __builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier ();
    __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "wavefront");
    if (even)
      bilr_emit_row<L, NP, true> (bp, c, lds, dst, dstride, y, __builtin_amdgcn_readlane (tab_p1, y - y0), q);
    else
      bilr_emit_row<L, NP, false> (bp, c, lds, dst, dstride, y, __builtin_amdgcn_readlane (tab_p1, y - y0), q);
    __builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier ();
    __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "wavefront");
  }
}

template <int CH, int L, int NP>
__global__ __launch_bounds__ (256) void k_bilinear420_rows (BilParams bp, Planes pl, uint8_t *__restrict__ dst, int dstride, int tiles_x)
{
  bilinear420_rows_body<CH, L, NP> (bp, pl, dst, dstride, tiles_x, (int) blockIdx.x);
}

// several frames per launch: workgroups [f * blocks_per_frame, (f + 1) * blocks_per_frame) are frame f (blocks_per_frame is a
// multiple of 8, so a workgroup's XCD is the same with or without the frame offset)
template <int CH, int L, int NP>
__global__ __launch_bounds__ (256) void k_bilinear420_rows_frames (BilParams bp, BilBatch fb, int dstride, int tiles_x, int blocks_per_frame)
{
  const int frame = (int) blockIdx.x / blocks_per_frame;
  /* a pointer read out of an indexed kernel-argument array is a generic pointer to the compiler: every access through it becomes a
   * FLAT instruction, which counts on both vmcnt and lgkmcnt and turns each wait of the kernel into a full drain (3 us per frame
   * here).  They are global memory: say so. */
  typedef const __attribute__ ((address_space (1))) uint8_t *gptr_t;
  Planes pl;
  pl.p[0] = (const uint8_t *) (gptr_t) fb.p[0][frame], pl.p[1] = (const uint8_t *) (gptr_t) fb.p[1][frame];
  pl.p[2] = (const uint8_t *) (gptr_t) fb.p[2][frame], pl.p[3] = nullptr;
  pl.stride[0] = fb.stride[0], pl.stride[1] = fb.stride[1], pl.stride[2] = fb.stride[2], pl.stride[3] = 0;
  uint8_t *dst = (uint8_t *) (__attribute__ ((address_space (1))) uint8_t *) fb.dst[frame];
  bilinear420_rows_body<CH, L, NP> (bp, pl, dst, dstride, tiles_x, (int) blockIdx.x - frame * blocks_per_frame);
}

// plane scaler of planar / semi-planar formats (video_planes.h): one lane per output pixel of the plane
__global__ __launch_bounds__ (256) void k_plane_simple (int kind, SrcPlane s, DstPlane d, int ow, int oh)
{
  plane_simple_body (kind, s, d, ow, oh, (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

__global__ __launch_bounds__ (256) void k_plane_hscale (SrcPlane s, ScaleDev sd, DstPlane d, int ow, int rows)
{
  plane_hscale_body (s, sd, d, ow, rows, (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

__global__ __launch_bounds__ (256) void k_plane_vscale (SrcPlane s, ScaleDev sd, DstPlane d, int width, int oh)
{
  plane_vscale_body (s, sd, d, width, oh, (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

// ... and of its 10 / 12 / 16-bit relatives from the AYUV64 image (video_deep.h pack16_alpha_plane_body)
__global__ __launch_bounds__ (256) void k_pack16_alpha_plane (PackPlanarParams pk, int hi_depth, DitherParams dt, const uint8_t *__restrict__ src, int sstride,
    uint8_t *__restrict__ plane, int stride)
{
  pack16_alpha_plane_body (pk, hi_depth, dt, src, sstride, plane, stride, 4 * (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

hipError_t launch_pack16_alpha_plane (const PackPlanarParams &pk, int hi_depth, const DitherParams &dt, const uint8_t *src, int sstride, uint8_t *plane, int stride,
    hipStream_t stream)
{
  video_frame_list_touch (plane);
  hipLaunchKernelGGL (k_pack16_alpha_plane, dim3 (((pk.width + 3) / 4 + 255) / 256, pk.height), dim3 (256), 0, stream, pk, hi_depth, dt, src, sstride, plane, stride);
  return hipGetLastError ();
}

// A420's alpha plane from the chain's AYUV image (video_pack.h pack_alpha_plane_body)
__global__ __launch_bounds__ (256) void k_pack_alpha_plane (PackPlanarParams pk, const uint8_t *__restrict__ img, int istride, uint8_t *__restrict__ plane, int stride)
{
  pack_alpha_plane_body (pk, img, istride, plane, stride, 4 * (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

hipError_t launch_pack_alpha_plane (const PackPlanarParams &pk, const uint8_t *img, int istride, uint8_t *plane, int stride, hipStream_t stream)
{
  video_frame_list_touch (plane);
  hipLaunchKernelGGL (k_pack_alpha_plane, dim3 (((pk.width + 3) / 4 + 255) / 256, pk.height), dim3 (256), 0, stream, pk, img, istride, plane, stride);
  return hipGetLastError ();
}

// the reference's v210 fastpaths (video_v210_fast.h): a lane per group of six pixels of a line / line pair; frame lists in blockIdx.z
__global__ __launch_bounds__ (256) void k_v210_fast (V210FastParams p, FrameDeltas fl)
{
  GSTAMD_FRAME_Z;
  for (int i = 0; i < 3; i++)
    p.s[i] = p.s[i] ? p.s[i] + fls_ : nullptr, p.d[i] = p.d[i] ? p.d[i] + fld_ : nullptr;
  v210_fast_body (p, (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

__global__ __launch_bounds__ (256) void k_v210_fast_vec (V210FastParams p, FrameDeltas fl)
{
  GSTAMD_FRAME_Z;
  for (int i = 0; i < 3; i++)
    p.s[i] = p.s[i] ? p.s[i] + fls_ : nullptr, p.d[i] = p.d[i] ? p.d[i] + fld_ : nullptr;
  v210_fast_block (p, (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

hipError_t launch_v210_fast (const V210FastParams &p, hipStream_t stream)
{
  int nz;
  const FrameDeltas &fl = frame_list_for (p.s[0], p.d[0], &nz);
  bool vec = v210_fast_vec_ok (p);
  for (int z = 1; z < nz && vec; z++)         /* every frame of a list on the same 16-byte phase */
    vec = (fl.s[z] % 16) == 0 && (fl.d[z] % 16) == 0;
  if (vec) {
    hipLaunchKernelGGL (k_v210_fast_vec, dim3 ((v210_fast_blocks (p) + 63) / 64, v210_fast_rows (p), nz), dim3 (64), 0, stream, p, fl);
    return hipGetLastError ();
  }
  hipLaunchKernelGGL (k_v210_fast, dim3 ((v210_fast_groups (p) + 255) / 256, v210_fast_rows (p), nz), dim3 (256), 0, stream, p, fl);
  return hipGetLastError ();
}

// 3- / 4-byte pixel permutations (video_swizzle34.h)
template <int SB, int DB>
__global__ __launch_bounds__ (256) void k_swizzle34 (Swz34Params p, FrameDeltas fl)
{
  GSTAMD_FRAME_Z;
  p.src += fls_, p.dst += fld_;
  swizzle34_body<SB, DB> (p, (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

// map[j]: which source byte of a pixel lands in destination byte j.  src_pos / dst_pos: byte of component c (A, c1, c2, c3) in the
// source / destination pixel; a 3-byte pixel has no component 0.
bool swizzle34_setup (int src_bytes, const int *src_pos, int dst_bytes, const int *dst_pos, const uint8_t *src, int sstride, uint8_t *dst, int dstride,
    int width, Swz34Params *p)
{
  if (((uintptr_t) src % 4) != 0 || (sstride % 4) != 0 || ((uintptr_t) dst % 4) != 0 || (dstride % 4) != 0 || (src_bytes == 4 && dst_bytes == 4))
    return false;
  uint8_t map[4] = {0, 0, 0, 0};
  for (int c = dst_bytes == 4 ? 0 : 1; c < 4; c++)
    map[dst_pos[c]] = c == 0 && src_bytes == 3 ? 0xff : (uint8_t) src_pos[c];
  memset ((void *) p, 0, sizeof (*p));
  if (src_bytes == 3 && dst_bytes == 4)
    swz34_selectors<3, 4> (map, p);
  else if (src_bytes == 4 && dst_bytes == 3)
    swz34_selectors<4, 3> (map, p);
  else
    swz34_selectors<3, 3> (map, p);
  p->src = src, p->sstride = sstride, p->dst = dst, p->dstride = dstride, p->width = width;
  return true;
}

hipError_t launch_swizzle34 (const Swz34Params &p, int src_bytes, int dst_bytes, int height, hipStream_t stream)
{
  int nz;
  const FrameDeltas &fl = frame_list_for (p.src, p.dst, &nz);
  const dim3 grid (((p.width + 3) / 4 + 255) / 256, height, nz);
  if (src_bytes == 3 && dst_bytes == 4)
    hipLaunchKernelGGL ((k_swizzle34<3, 4>), grid, dim3 (256), 0, stream, p, fl);
  else if (src_bytes == 4 && dst_bytes == 3)
    hipLaunchKernelGGL ((k_swizzle34<4, 3>), grid, dim3 (256), 0, stream, p, fl);
  else
    hipLaunchKernelGGL ((k_swizzle34<3, 3>), grid, dim3 (256), 0, stream, p, fl);
  return hipGetLastError ();
}

// plane re-arrangement (video_relayout.h): 16 output bytes per lane, grid.y = luma rows, then the chroma rows of the destination's planes
__global__ __launch_bounds__ (256) void k_planes_relayout (RelayoutParams p, FrameDeltas fl)
{
  GSTAMD_FRAME_Z;
  relayout_body (p, (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y, fls_, fld_);
}

bool relayout_usable (const RelayoutParams &p)
{
  for (int i = 0; i < 3; i++) {
    if (p.in[i] && (((uintptr_t) p.in[i] % 16) != 0 || (p.in_stride[i] % 16) != 0))
      return false;
    if (p.out[i] && (((uintptr_t) p.out[i] % 16) != 0 || (p.out_stride[i] % 16) != 0))
      return false;
  }
  return true;
}

hipError_t launch_planes_relayout (const RelayoutParams &p, hipStream_t stream)
{
  const int lanes = relayout_lanes (p);
  int nz;
  const FrameDeltas &fl = frame_list_for (p.in[0], p.out[0], &nz);
  hipLaunchKernelGGL (k_planes_relayout, dim3 ((lanes + 255) / 256, relayout_rows (p), nz), dim3 (256), 0, stream, p, fl);
  return hipGetLastError ();
}

// the frame's planes (video_planes.h): blockIdx.x runs over the 64 x 16 output tiles of the planes of `jobs`.  An if-chain over the jobs, not
// jobs.job[j] (indexing the by-value argument with a run-time index sends the whole struct through scratch) and not one body after
// uniform selects of every field (all three jobs live in SGPRs at once: 170 spilled).  Two kernels - the planes that stage nothing, and
// the two-pass N-tap tiles with their LDS and barriers - because one body with both cost 185 VGPRs.
template <int K>
__device__ __forceinline__ void plane_tiles_job (const PlaneJobs &jobs, uint8_t *lds)
{
  const int tile = (int) blockIdx.x - jobs.job[K].tile0;
#pragma unroll
  for (int phase = 0; phase < PLN_PHASES; phase++) {
    plane_tile_body (jobs.job[K], lds, tile, (int) threadIdx.x, phase);
    if (phase + 1 < PLN_PHASES)
      __syncthreads ();
  }
}

static __device__ __forceinline__ void plane_jobs_rebase (PlaneJobs &jobs, long long ds, long long dd)
{
#pragma unroll
  for (int k = 0; k < PLN_MAX_JOBS; k++)
    jobs.job[k].s.p += ds, jobs.job[k].d.p += dd;
}

__global__ __launch_bounds__ (PLN_THREADS) void k_plane_tiles (PlaneJobs jobs, FrameDeltas fl)
{
  GSTAMD_FRAME_Z;
  plane_jobs_rebase (jobs, fls_, fld_);
  extern __shared__ uint32_t plane_lds[];         /* the largest tile's needs (plane_job_lds_bytes), not PLN_LDS_BYTES: workgroups per CU */
  uint8_t *lds = (uint8_t *) plane_lds;
  const int b = (int) blockIdx.x;
  if (jobs.n > 2 && b >= jobs.job[2].tile0)
    plane_tiles_job<2> (jobs, lds);
  else if (jobs.n > 1 && b >= jobs.job[1].tile0)
    plane_tiles_job<1> (jobs, lds);
  else
    plane_tiles_job<0> (jobs, lds);
}

__global__ __launch_bounds__ (PLN_THREADS) void k_plane_direct (PlaneJobs jobs, FrameDeltas fl)
{
  GSTAMD_FRAME_Z;
  plane_jobs_rebase (jobs, fls_, fld_);
  const int b = (int) blockIdx.x;
  if (jobs.n > 2 && b >= jobs.job[2].tile0)
    plane_direct_body (jobs.job[2], b - jobs.job[2].tile0, (int) threadIdx.x);
  else if (jobs.n > 1 && b >= jobs.job[1].tile0)
    plane_direct_body (jobs.job[1], b - jobs.job[1].tile0, (int) threadIdx.x);
  else
    plane_direct_body (jobs.job[0], b, (int) threadIdx.x);
}

// the planes of two short passes and the pass-free ones next to them (video_planes.h: k_plane_quad): a workgroup = 64 lanes x 4 waves, a wave
// walks g.rows output rows.  An if-chain over the jobs (see k_plane_tiles); the argument is not written to - a frame list's rebase happens
// where the body forms a pointer
template <int K>
__device__ __forceinline__ void plane_quad_job (const PlaneJobs &jobs, const QuadGrid &g, int local, long long ds, long long dd)
{
  const int bxi = local % g.bx[K], byi = local / g.bx[K];
  const int y0 = __builtin_amdgcn_readfirstlane ((byi * 4 + (int) (threadIdx.x >> 6)) * g.rows[K]);
  plane_rows_body (jobs.job[K], bxi * 64 + (int) (threadIdx.x & 63), y0, g.rows[K], ds, dd, g.nt);
}

__global__ __launch_bounds__ (256) void k_plane_quad (PlaneJobs jobs, QuadGrid g, FrameDeltas fl)
{
  GSTAMD_FRAME_Z;
  int k, local;
  quad_grid_find (g, (int) blockIdx.x, &k, &local);
  if (k == 2)
    plane_quad_job<2> (jobs, g, local, fls_, fld_);
  else if (k == 1)
    plane_quad_job<1> (jobs, g, local, fls_, fld_);
  else
    plane_quad_job<0> (jobs, g, local, fls_, fld_);
}

// `jobs`: the frame's planes in any order; they are split by kind of body and each kind gets a launch of its own
hipError_t launch_plane_frame (const PlaneJobs &jobs, size_t lds_bytes, hipStream_t stream)
{
  int nz;
  const FrameDeltas &fl = frame_list_for (jobs.job[0].s.p, jobs.job[0].d.p, &nz);
  const bool quads = !tuning_on ("GSTAMD_NO_PLANE_QUAD");
  {
    PlaneJobs part;
    QuadGrid g;
    memset ((void *) &part, 0, sizeof (part));
    long long wave_rows = 0;
    for (int i = 0; i < jobs.n; i++) {
      if (!quads || !jobs.job[i].quad)
        continue;
      if (tuning_int ("GSTAMD_PLANE_QUAD_ONLY", -1) >= 0 && tuning_int ("GSTAMD_PLANE_QUAD_ONLY", -1) != i)          /* timing one plane of the frame on its own (the others are not converted) */
        continue;
      const PlaneJob &J = jobs.job[i];
      const int bytes = quad_mode_bytes (J.quad - 1);
      g.bx[part.n] = ((J.ow * J.s.n + bytes - 1) / bytes + 63) / 64;
      wave_rows += (long long) g.bx[part.n] * J.oh;
      part.job[part.n++] = J;
    }
    /* rows per wave: what a two-pass plane sets up per column (indices, weights, selectors) serves more rows the longer a wave walks, but
       the launch wants many waves more than it wants that (MI355X, NV12 4K -> 1080p in lists of 8, profiles/r04/f8scale_variants.log: 30.2 /
       29.6 / 33.6 us per launch at 1 / 2 / 4 rows; single frames 7.7 / 9.0 / 12.6); a pass-free plane has nothing to set up */
    const int rows_pin = tuning_int ("GSTAMD_PLANE_QUAD_ROWS", 0);
    g.nt = tuning_int ("GSTAMD_PLANE_QUAD_NT", 0);
    int blocks = 0;
    for (int k = 0; k < part.n; k++) {
      const bool two_pass = part.job[k].kind == PLANE_SCALE;
      g.rows[k] = !two_pass ? 1 : (rows_pin > 0 ? rows_pin : (wave_rows * nz >= 16384 ? 2 : 1));
      g.n[k] = g.bx[k] * ((part.job[k].oh + 4 * g.rows[k] - 1) / (4 * g.rows[k]));
      blocks += g.n[k];
    }
    for (int i = part.n; i < PLN_MAX_JOBS; i++)
      g.n[i] = 0, g.bx[i] = 1, g.rows[i] = 1;
    if (part.n)
      hipLaunchKernelGGL (k_plane_quad, dim3 (blocks, 1, nz), dim3 (256), 0, stream, part, g, fl);
  }
  for (int direct = 0; direct < 2; direct++) {
    PlaneJobs part;
    memset ((void *) &part, 0, sizeof (part));
    int tiles = 0;
    for (int i = 0; i < jobs.n; i++) {
      if ((quads && jobs.job[i].quad) || plane_job_is_direct (jobs.job[i]) != (direct == 1))
        continue;
      PlaneJob &J = part.job[part.n++];
      J = jobs.job[i];
      J.tile0 = tiles;
      tiles += J.tiles_x * ((J.oh + PLN_TH - 1) / PLN_TH);
    }
    if (!part.n)
      continue;
    if (direct)
      hipLaunchKernelGGL (k_plane_direct, dim3 (tiles, 1, nz), dim3 (PLN_THREADS), 0, stream, part, fl);
    else
      hipLaunchKernelGGL (k_plane_tiles, dim3 (tiles, 1, nz), dim3 (PLN_THREADS), lds_bytes ? lds_bytes : 4, stream, part, fl);
  }
  return hipGetLastError ();
}

// borders (convert_fill_border, video-converter.c:7190): every pixel of a destination plane outside the picture rectangle
// gets the plane's border value; es = bytes per pixel of the plane (1, 2, 3, 4 or 8: value_hi is the upper word)
__global__ __launch_bounds__ (256) void k_fill_border (uint8_t *__restrict__ p, int stride, int es, uint32_t value, uint32_t value_hi, int maxw, int maxh,
    int x0, int y0, int w, int h)
{
  const int x = (int) (blockIdx.x * blockDim.x + threadIdx.x), y = (int) blockIdx.y;
  if (x >= maxw || y >= maxh || (x >= x0 && x < x0 + w && y >= y0 && y < y0 + h))
    return;
  uint8_t *q = p + (size_t) y * stride + (size_t) x * es;
  if (es == 8)
    *(uint2 *) q = make_uint2 (value, value_hi);
  else if (es == 4)
    *(uint32_t *) q = value;
  else if (es == 2)
    *(uint16_t *) q = (uint16_t) value;
  else if (es == 3) {                      /* memset_u24 */
    q[0] = (uint8_t) value;
    q[1] = (uint8_t) (value >> 8);
    q[2] = (uint8_t) (value >> 16);
  } else
    *q = (uint8_t) value;
}

__global__ __launch_bounds__ (256) void k_pack_planar (PackPlanarParams pk, const uint8_t *__restrict__ src, int sstride, DstPlanes d, int wide)
{
  const int x0 = (int) (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (wide && pack_planar_block4 (pk, src, sstride, d, x0, (int) blockIdx.y))
    return;
  const SrcImage img = {src, sstride, pk.width};
  pack_planar_body (pk, img, d, x0, (int) blockIdx.y);
}

// the same with the unscaled 8-bit chain as its pixel source: YUY2 -> I420, AYUV -> NV12, I420 -> Y42B ... in one launch, no AYUV image
__global__ __launch_bounds__ (64) void k_convert_pack (PackPlanarParams pk, SrcPacked4 src, DstPlanes d, int wide, FrameDeltas fl)
{
  GSTAMD_FRAME_Z;
  src.p += fls_;
  const int x0 = (int) (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  /* whole blocks inside the picture in wide accesses (pack_planar_block4 with the frame as its row source: one 16-byte load per line, word /
     half-word stores); the picture's edge, dither and the 3-byte / packed 4:2:2 packs through the general body */
  if (wide && pack_planar_block4 (pk, src, d, x0, (int) blockIdx.y, fld_))
    return;
  pack_planar_body (pk, src, d, x0, (int) blockIdx.y, fld_);
}

// 4-byte RGB -> 4:2:0 (video_encode_fast.h): one lane = a 4 x 2 pixel block, one wave per workgroup
template <int SEMI>
__global__ __launch_bounds__ (64) void k_encode420 (Enc420Params ep, const uint8_t *__restrict__ src, int sstride, DstPlanes d, FrameDeltas fl)
{
  GSTAMD_FRAME_Z;
  src += fls_;
  enc420_block<SEMI> (ep, src, sstride, d, (int) (blockIdx.x * 64 + threadIdx.x) * 4, (int) blockIdx.y, fld_);
}

// k_swizzle4: a pure byte permutation of 4-byte pixels is a copy as far as the memory system goes, and is laid out like the copy that
// measures best on this chip (scripts/c4_probe.hip: one 16-byte element per lane, no grid-stride loop, nontemporal load and store:
// 6.4-6.6 TB/s against 4.7-5.3 for the grid-stride forms): one lane = 4 pixels, one wave = 1 KB of a row, every byte touched once.
__global__ __launch_bounds__ (256) void k_swizzle4 (const uint8_t *__restrict__ src, int sstride, uint8_t *__restrict__ dst, int dstride, int width, uint32_t sel,
    FrameDeltas fl)
{
  GSTAMD_FRAME_Z;
  src += fls_, dst += fld_;
  const int x = (int) (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = (int) blockIdx.y;
  if (x >= width)
    return;
  const uint8_t *sp = src + (size_t) y * sstride + 4 * (size_t) x;
  uint8_t *dp = dst + (size_t) y * dstride + 4 * (size_t) x;
  if (x + 4 <= width) {
    typedef unsigned int u32x4 __attribute__ ((ext_vector_type (4)));
    const u32x4 v = __builtin_nontemporal_load ((const u32x4 *) sp);
    const u32x4 o = {swizzle4_px (v.x, sel), swizzle4_px (v.y, sel), swizzle4_px (v.z, sel), swizzle4_px (v.w, sel)};
    __builtin_nontemporal_store (o, (u32x4 *) dp);
  } else {
    for (int i = 0; x + i < width; i++)
      ((uint32_t *) dp)[i] = swizzle4_px (((const uint32_t *) sp)[i], sel);
  }
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
static inline bool aligned (const void *p, size_t a) { return ((uintptr_t) p & (a - 1)) == 0; }

bool swizzle4_usable (const FrontParams &f, const Planes &pl, const ColorParams &color, const uint8_t *dst, int dstride)
{
  return f.kind == UNPACK_PACKED4 && f.hi_depth == 0 && color.matrix.kind == MATRIX_NONE && color.alpha_kind == ALPHA_NONE && aligned (dst, 16) &&
      (dstride % 16) == 0 && aligned (pl.p[0], 16) && (pl.stride[0] % 16) == 0;
}

hipError_t launch_swizzle4 (const FrontParams &f, const Planes &pl, const int pack_pos[4], uint8_t *dst, int dstride, hipStream_t stream)
{
  const int lanes = (f.width + 3) / 4;
  int nz;
  const FrameDeltas &fl = frame_list_for (pl.p[0], dst, &nz);
  hipLaunchKernelGGL (k_swizzle4, dim3 ((lanes + 255) / 256, f.height, nz), dim3 (256), 0, stream, pl.p[0], pl.stride[0], dst, dstride, f.width,
      swizzle4_selector (f.pos, pack_pos), fl);
  return hipGetLastError ();
}

hipError_t launch_convert (const FrontParams &f, const Planes &pl, const int *vpair_dev, const ColorParams &color,
    const int pack_pos[4], uint8_t *dst, int dstride, hipStream_t stream, int extra_rows)
{
  video_frame_list_touch (dst);
  const int spans = (f.width + K1_PX - 1) / K1_PX;
  int vec_ok = aligned (dst, 16) && (dstride % 16) == 0 && kind_has_planes (f.kind) && f.w_sub == 1;
  if (vec_ok) {
    vec_ok = aligned (pl.p[0], 8) && (pl.stride[0] % 8) == 0;
    if (f.kind == UNPACK_SEMI)
      vec_ok = vec_ok && aligned (pl.p[1], 8) && (pl.stride[1] % 8) == 0;
    else
      vec_ok = vec_ok && aligned (pl.p[1], 4) && aligned (pl.p[2], 4) && (pl.stride[1] % 4) == 0 && (pl.stride[2] % 4) == 0;
  }
  if (f.kind == UNPACK_PACKED4 && aligned (dst, 16) && (dstride % 16) == 0 && aligned (pl.p[0], 16) && (pl.stride[0] % 16) == 0)
    vec_ok = 2;                         /* convert_body's 16-byte path for 4-byte packed sources */
  const int bx = spans >= 256 ? 256 : (spans > 64 ? 128 : 64);
  dim3 grid ((spans + bx - 1) / bx, f.height + extra_rows), block (bx);       /* extra_rows: the line past the picture (PackPlanarParams::virtual_line) */
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

template <int CH>
static hipError_t launch_convert_gamma_ch (const FrontParams &f, const Planes &pl, const int *vpair_dev, const ColorParams &color, const int pack_pos[4],
    uint8_t *dst, int dstride, int spans, int vec_ok, const GammaDev &g, hipStream_t stream)
{
  const void *fn = (const void *) k_convert_gamma<CH>;
  hipError_t e = hipFuncSetAttribute (fn, hipFuncAttributeMaxDynamicSharedMemorySize, GSTAMD_GAMMA_LDS_BYTES);
  if (e != hipSuccess)
    return e;
  /* two workgroups of 512 lanes per CU (LDS), every one of them resident at once: rows per workgroup from the device's CU count; with the
     composed 256-byte table LDS is no limit and a workgroup takes two rows */
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    n_cu = hipGetDevice (&dev) == hipSuccess && hipGetDeviceProperties (&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int gx = (spans + 511) / 512;
  int rows = g.comp ? 2 : (f.height * gx + 2 * n_cu - 1) / (2 * n_cu);
  rows = rows < 1 ? 1 : rows;
  dim3 grid (gx, (f.height + rows - 1) / rows);
  hipLaunchKernelGGL (k_convert_gamma<CH>, grid, dim3 (512), g.comp ? 256 : GSTAMD_GAMMA_LDS_BYTES, stream, f, pl, vpair_dev, color, pack_pos[0], pack_pos[1], pack_pos[2],
      pack_pos[3], dst, dstride, spans, vec_ok, g, rows);
  return hipGetLastError ();
}

hipError_t launch_convert_gamma (const FrontParams &f, const Planes &pl, const int *vpair_dev, const ColorParams &color, const int pack_pos[4], uint8_t *dst,
    int dstride, const GammaDev &g, hipStream_t stream)
{
  video_frame_list_touch (dst);
  const int spans = (f.width + K1_PX - 1) / K1_PX;
  int vec_ok = aligned (dst, 16) && (dstride % 16) == 0 && kind_has_planes (f.kind) && f.w_sub == 1;
  if (vec_ok) {
    vec_ok = aligned (pl.p[0], 8) && (pl.stride[0] % 8) == 0;
    if (f.kind == UNPACK_SEMI)
      vec_ok = vec_ok && aligned (pl.p[1], 8) && (pl.stride[1] % 8) == 0;
    else
      vec_ok = vec_ok && aligned (pl.p[1], 4) && aligned (pl.p[2], 4) && (pl.stride[1] % 4) == 0 && (pl.stride[2] % 4) == 0;
  }
  if (f.kind == UNPACK_PACKED4 && aligned (dst, 16) && (dstride % 16) == 0 && aligned (pl.p[0], 16) && (pl.stride[0] % 16) == 0)
    vec_ok = 2;                         /* convert_body's 16-byte path for 4-byte packed sources */
  if (f.chroma_h == CHROMA_H_H2_CS)
    return launch_convert_gamma_ch<CHROMA_H_H2_CS> (f, pl, vpair_dev, color, pack_pos, dst, dstride, spans, vec_ok, g, stream);
  if (f.chroma_h == CHROMA_H_H2)
    return launch_convert_gamma_ch<CHROMA_H_H2> (f, pl, vpair_dev, color, pack_pos, dst, dstride, spans, vec_ok, g, stream);
  return launch_convert_gamma_ch<CHROMA_H_NONE> (f, pl, vpair_dev, color, pack_pos, dst, dstride, spans, vec_ok, g, stream);
}

// strip kernel: grid.x = 256-pixel columns, grid.y = strips of K line pairs, grid.z = frame; one wave per workgroup
template <int CH, int L, int ABL>
__global__ __launch_bounds__ (64) void k_convert_strip (FastParams fp, FrameBatch batch, int pairs, int K)
{
  if (ABL == GSTAMD_FAST_LUT) {
    fast_lut_lds[threadIdx.x] = ((const uint32_t *) fp.lut)[threadIdx.x];
    __syncthreads ();
  }
  const Planes pl = batch_planes (batch, blockIdx.z);
  const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
  const int p0 = blockIdx.y * K;
  if (x0 + 4 <= fp.width)
    fast_strip<CH, L, ABL> (fp, pl, batch.dst[blockIdx.z], batch.dstride, x0, p0, p0 + K < pairs ? p0 + K : pairs);
}

// the same under the XCD-aware block order of the wide kernel (wide_block_map): 1-D grid
template <int CH, int L, int ABL>
__global__ __launch_bounds__ (64) void k_convert_strip_xcd (FastParams fp, FrameBatch batch, int ncol, int pairs, int K, int strips, int total_strips)
{
  if (ABL == GSTAMD_FAST_LUT) {
    fast_lut_lds[threadIdx.x] = ((const uint32_t *) fp.lut)[threadIdx.x];
    __syncthreads ();
  }
  int col, S;
  if (!wide_block_map (blockIdx.x, ncol, total_strips, &col, &S))
    return;
  const int z = S / strips, p0 = (S - z * strips) * K;
  const Planes pl = batch_planes (batch, z);
  const int x0 = (col * 64 + threadIdx.x) * 4;
  if (x0 + 4 <= fp.width)
    fast_strip<CH, L, ABL> (fp, pl, batch.dst[z], batch.dstride, x0, p0, p0 + K < pairs ? p0 + K : pairs);
}

// destination byte order -> layout template argument (the four orders of the eight 4-byte RGB formats)
static int fast_layout (const FastParams &fp)
{
  return fp.ayuv ? GSTAMD_LAYOUT_AYUV : GSTAMD_LAYOUT (fp.pack_pos[1], fp.pack_pos[2], fp.pack_pos[3]);
}

#define GSTAMD_FOR_LAYOUTS(W) \
    W (2, 1, 0)      /* BGRA, BGRx */ \
    W (0, 1, 2)      /* RGBA, RGBx */ \
    W (1, 2, 3)      /* ARGB, xRGB */ \
    W (3, 2, 1)      /* ABGR, xBGR */

template <int CH, int ABL>
static bool launch_strip_variant (const FastParams &fp, const FrameBatch &batch, int n, int K, bool xcd_order, hipStream_t stream)
{
  const int pairs = fp.height / 2 + 1, strips = (pairs + K - 1) / K, ncol = (fp.width + 255) / 256;
#define W(pr, pg, pb) case GSTAMD_LAYOUT (pr, pg, pb): \
    if (xcd_order) \
      hipLaunchKernelGGL ((k_convert_strip_xcd<CH, GSTAMD_LAYOUT (pr, pg, pb), ABL>), dim3 (wide_grid_blocks (ncol, strips * n)), dim3 (64), ABL == GSTAMD_FAST_LUT ? 256 : 0, stream, \
          fp, batch, ncol, pairs, K, strips, strips * n); \
    else \
      hipLaunchKernelGGL ((k_convert_strip<CH, GSTAMD_LAYOUT (pr, pg, pb), ABL>), dim3 (ncol, strips, n), dim3 (64), ABL == GSTAMD_FAST_LUT ? 256 : 0, stream, fp, batch, pairs, K); \
    return true;
  switch (fast_layout (fp)) {
    GSTAMD_FOR_LAYOUTS (W)
  }
#undef W
  return false;
}

// wide variant (video_fast.h): 1-D grid of one-wave workgroups, XCD-aware block order, rows staged through LDS
template <int CH, int L, int ABL>
__global__ __launch_bounds__ (64) void k_convert_wide (FastParams fp, FrameBatch batch, int nxb, int pairs, int K, int strips, int total_strips,
    int vec)
{
  __shared__ WideLds lds;
  int xb, S;
  if (!wide_block_map (blockIdx.x, nxb, total_strips, &xb, &S))
    return;
  const int z = S / strips, p0 = (S - z * strips) * K, p1 = p0 + K < pairs ? p0 + K : pairs;
  const Planes pl = batch_planes (batch, z);
  const int xw = xb * GSTAMD_WIDE_PX, lane = threadIdx.x;
  const bool v = vec != 0;
  WideRegs r;
  wide_fetch_chroma<CH> (fp, pl, xw, p0 - 1 > fp.crow_lo ? p0 - 1 : fp.crow_lo, lane, v, r);
  wide_commit_chroma<CH> (fp, xw, lane, r, lds.c[0]);
  wide_fetch<CH> (fp, pl, xw, p0, lane, v, r);
  for (int p = p0; p < p1; p++) {
    const int k = p - p0;
    wide_commit<CH> (fp, xw, lane, r, &lds, (k + 1) & 1);
    __syncthreads ();                     /* one wave per workgroup: orders the LDS writes before the reads */
    if (p + 1 < p1)
      wide_fetch<CH> (fp, pl, xw, p + 1, lane, v, r);      /* in flight while this pair is converted */
    wide_emit<CH, L, ABL> (fp, batch.dst[z], batch.dstride, xw, p, lane, &lds, k & 1);
    __syncthreads ();
  }
}

static bool wide_vec_ok (const FrameBatch &batch, int n)
{
  bool ok = (batch.ystride % 16) == 0 && (batch.uvstride % 16) == 0;
  for (int i = 0; i < n; i++)
    ok = ok && aligned (batch.y[i], 16) && aligned (batch.uv[i], 16);
  return ok;
}

template <int CH, int ABL>
static bool launch_wide_variant (const FastParams &fp, const FrameBatch &batch, int n, int K, hipStream_t stream)
{
  const int pairs = fp.height / 2 + 1, nxb = (fp.width + GSTAMD_WIDE_PX - 1) / GSTAMD_WIDE_PX;
  const int strips = (pairs + K - 1) / K;
  const int blocks = wide_grid_blocks (nxb, strips * n);
  const int vec = wide_vec_ok (batch, n) ? 1 : 0;
#define W(pr, pg, pb) case GSTAMD_LAYOUT (pr, pg, pb): \
    hipLaunchKernelGGL ((k_convert_wide<CH, GSTAMD_LAYOUT (pr, pg, pb), ABL>), dim3 (blocks), dim3 (64), 0, stream, fp, batch, nxb, pairs, K, strips, strips * n, vec); \
    return true;
  switch (fast_layout (fp)) {
    GSTAMD_FOR_LAYOUTS (W)
  }
#undef W
  return false;
}

// Kernel-shape selector for profiling sessions and the variant parity tests: GSTAMD_FAST_VARIANT="shape,abl,K,order" with shape 0 = strip,
// 1 = wide; K = line pairs per wave; order 1 = XCD-aware block order (strip) - every combination computes the same bytes.  abl 1 =
// memory-only ablation (h2cs only) exists in -DGSTAMD_TUNING builds alone (python -m gstreamer_amd.build --tuning).  Unset = shipped.
struct FastVariant { int set, shape, abl, K, order; };
static const FastVariant &fast_variant ()
{
  static FastVariant v = {-1, 0, 0, 0, 0};
  if (v.set < 0) {
    const char *e = tuning_text ("GSTAMD_FAST_VARIANT");
    v.set = e && sscanf (e, "%d,%d,%d,%d", &v.shape, &v.abl, &v.K, &v.order) >= 1;
  }
  return v;
}

template <int CH>
static bool launch_fast (const FastParams &fp, const FrameBatch &batch, int n, hipStream_t stream)
{
  /* shipped configuration (MI355X sweeps, profiles/r01_c2_variants.txt): strip kernel, 4-pixel columns, 3 line pairs per
   * lane with the next pair's loads in flight during the current pair's math, one wave per workgroup */
  const FastVariant &v = fast_variant ();
  /* (one frame of less than 6 M pixels per launch: two pairs per lane - half again as many waves to fill the chip with; 2560 x 1440: 8.1 -> 6.7 us,
     scripts/strip_k_probe.py, round 6; at 4K and in lists three pairs stay) */
  const int k_default = n == 1 && (long) fp.width * fp.height < 6000000L ? 2 : 3;
  const int shape = v.set ? v.shape : 0, K = v.set && v.K > 0 ? v.K : k_default;
#ifdef GSTAMD_TUNING       /* memory-only ablation (wrong pixels on purpose): profiling builds only, never in the product library */
  if (v.set && v.abl && CH == CHROMA_H_H2_CS)
    return shape == 1 ? launch_wide_variant<CHROMA_H_H2_CS, 1> (fp, batch, n, K, stream) :
        launch_strip_variant<CHROMA_H_H2_CS, 1> (fp, batch, n, K, v.order != 0, stream);
#endif
  if (fp.lut)                   /* GammaPlan::lut_direct: the composed gamma table between the pack and the store */
    return launch_strip_variant<CH, GSTAMD_FAST_LUT> (fp, batch, n, K, v.set && v.order != 0, stream);
  if (shape == 1 && fp.width >= GSTAMD_WIDE_PX / 2 && fp.px_bytes == 4)
    return launch_wide_variant<CH, 0> (fp, batch, n, K, stream);
  return launch_strip_variant<CH, 0> (fp, batch, n, K, v.set && v.order != 0, stream);
}

hipError_t launch_convert_pair (const FastParams &fp_in, int chroma_h, int n_frames, const uint8_t *const *y, const uint8_t *const *uv,
    uint8_t *const *dst, int ystride, int uvstride, int dstride, hipStream_t stream)
{
  video_frame_list_touch (dst[0]);
  /* store policy (video_fast.h store16_policy): write-through for launches of few frames - what such a launch leaves dirty in the L2s is written back
     at its end, on the critical path of the next launch; a long list amortises that and takes the write-through streaming store (sc0 sc1 nt), the
     fastest of the five at 32 frames (profiles/r06/store_policy.md).  GSTAMD_STORE_POLICY pins it. */
  FastParams fpl = fp_in;
  {
    const int pin = tuning_int ("GSTAMD_STORE_POLICY", -1);
    const int few = tuning_int ("GSTAMD_STORE_WT_BELOW", 9);
    fpl.store_policy = pin >= 0 ? pin : (n_frames < few ? 1 : 3);
  }
  const FastParams &fp = fpl;
  for (int base = 0; base < n_frames; base += GSTAMD_MAX_BATCH) {
    const int n = n_frames - base < GSTAMD_MAX_BATCH ? n_frames - base : GSTAMD_MAX_BATCH;
    FrameBatch batch;
    memset (&batch, 0, sizeof (batch));
    for (int i = 0; i < n; i++) {
      batch.y[i] = y[base + i];
      batch.uv[i] = uv[base + i];
      batch.dst[i] = dst[base + i];
    }
    batch.ystride = ystride;
    batch.uvstride = uvstride;
    batch.dstride = dstride;
    bool ok;
    switch (chroma_h) {
      case CHROMA_H_H2_CS:
        ok = launch_fast<CHROMA_H_H2_CS> (fp, batch, n, stream);
        break;
      case CHROMA_H_H2:
        ok = launch_fast<CHROMA_H_H2> (fp, batch, n, stream);
        break;
      default:
        ok = launch_fast<CHROMA_H_NONE> (fp, batch, n, stream);
        break;
    }
    if (!ok)
      return hipErrorInvalidValue;      /* destination byte order outside the four the planner admits */
  }
  return hipGetLastError ();
}

static int front_vec_ok (const FrontParams &f, const Planes &pl)
{
  if (!kind_has_planes (f.kind) || f.w_sub != 1)
    return 0;
  int ok = aligned (pl.p[0], 8) && (pl.stride[0] % 8) == 0;
  if (f.kind == UNPACK_SEMI)
    ok = ok && aligned (pl.p[1], 8) && (pl.stride[1] % 8) == 0;
  else
    ok = ok && aligned (pl.p[1], 4) && aligned (pl.p[2], 4) && (pl.stride[1] % 4) == 0 && (pl.stride[2] % 4) == 0;
  return ok;
}

// word-load staging (front_span8_packed) applies: planes aligned, 2x horizontally subsampled chroma, no colour step
// before the scaler
static int front_packed_ok (const SrcFront &s)
{
  return s.vec_ok && s.f.w_sub == 1 && kind_has_planes (s.f.kind) && s.pre.matrix.kind == MATRIX_NONE && s.pre.alpha_kind == ALPHA_NONE;
}

// byte-dot-product N-tap pass: opaque source (alpha 0xff in, 0xff out) and no colour step before the scaler
static int dot4_source_ok (const SrcFront &s)
{
  return (kind_has_planes (s.f.kind) || (s.f.kind == UNPACK_PACKED422 && s.f.hi_depth == 0)) && s.pre.matrix.kind == MATRIX_NONE && s.pre.alpha_kind == ALPHA_NONE;
}

// 16-pixel staging of video_hscale420.h: planes with 2x horizontal subsampling, 16-byte luma / interleaved-chroma and 8-byte
// planar-chroma loads aligned, whole 16-pixel groups only
static int h420_source_ok (const SrcFront &s)
{
  const FrontParams &f = s.f;
  if (!kind_has_planes (f.kind) || f.w_sub != 1 || (f.width % 16) != 0 || f.chroma_v2 == 2)          /* (a field's pair table: the per-pixel front) */
    return 0;
  int ok = aligned (s.pl.p[0], 16) && (s.pl.stride[0] % 16) == 0;
  if (f.kind == UNPACK_SEMI)
    ok = ok && aligned (s.pl.p[1], 16) && (s.pl.stride[1] % 16) == 0;
  else
    ok = ok && aligned (s.pl.p[1], 8) && aligned (s.pl.p[2], 8) && (s.pl.stride[1] % 8) == 0 && (s.pl.stride[2] % 8) == 0;
  return ok;
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
    const int pack_pos[4], int out_w, int out_h, int max_span, TileGeom geom, const PostFast &pf, hipStream_t stream)
{
  video_frame_list_touch (dst);
  SrcFront src;
  src.f = f;
  src.pl = pl;
  src.vpair = vpair_dev;
  src.pre = pre;
  src.vec_ok = front_vec_ok (f, pl);
  Dst d = make_dst (dst, dstride, final, post, pack_pos);
  dim3 block (256), grid ((out_w + 255) / 256, out_h);
#ifdef GSTAMD_TUNING
  const int ablate = tuning_int ("GSTAMD_ABLATE", 0);
#else
  const int ablate = 0;         /* the stage-skipping switches of k_hscale420_dot4 are dead code in the product build */
#endif
  /* a packed 8-bit frame with nothing ahead of the scaler: the lean source (video_scale_fast.h SrcLean) */
  if (horizontal && geom.tile_w > 0 && geom.lds_px * 4 <= WAVE_TILE_LDS_BYTES && f.hi_depth == 0 && pre.matrix.kind == MATRIX_NONE &&
      pre.alpha_kind == ALPHA_NONE && (f.kind == UNPACK_PACKED4 || f.kind == UNPACK_PACKED422) && aligned (pl.p[0], 4) && (pl.stride[0] % 4) == 0 &&
      f.height - 1 <= f.luma_last) {
    SrcLean ls;
    memset ((void *) &ls, 0, sizeof (ls));
    ls.p = pl.p[0], ls.stride = pl.stride[0], ls.width = f.width;
    ls.p422 = f.kind == UNPACK_PACKED422;
    ls.sel = (uint32_t) f.pos[0] | ((uint32_t) f.pos[1] << 8) | ((uint32_t) f.pos[2] << 16) | ((uint32_t) f.pos[3] << 24);
    ls.pos1 = f.pos[1], ls.pos2 = f.pos[2], ls.pos3 = f.pos[3], ls.chroma_h = f.chroma_h, ls.swap_k = f.swap_k;
    dim3 wgrid ((out_w + geom.tile_w - 1) / geom.tile_w, out_h);
    hipLaunchKernelGGL (k_hscale_wave<SrcLean>, wgrid, dim3 (64), (size_t) geom.lds_px * 4, stream, ls, sd, d, pf, out_w, out_h, geom.tile_w, geom.lds_px, 0);
    return hipGetLastError ();
  }
  const int h420_rows_env = tuning_int ("GSTAMD_H420_ROWS", -1);      /* 0: kernel off */
  if (horizontal && geom.tile16_w > 0 && sd.tapw && dot4_source_ok (src) && h420_source_ok (src) && h420_rows_env != 0) {
    /* lines per wave: the waves of the launch should all be resident at once (one round, no tail of late waves), each walking
     * down as many lines as that takes - but at least 4, for the chroma rows and the taps a wave keeps between lines */
    const int tiles = (out_w + geom.tile16_w - 1) / geom.tile16_w;
    int h420_rows = h420_rows_env;
    if (h420_rows < 0) {
      static int slots = 0;
      if (!slots) {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        const void *fn = sd.nw == 5 ? (const void *) k_hscale420_dot4<5> : (const void *) k_hscale420_dot4<0>;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor (&per_cu, fn, 64, (size_t) GSTAMD_H420_PLANE_BYTES * 3) != hipSuccess || per_cu <= 0)
          per_cu = 16;
        if (hipGetDevice (&dev) != hipSuccess || hipGetDeviceProperties (&prop, dev) != hipSuccess)
          prop.multiProcessorCount = 256;
        slots = per_cu * prop.multiProcessorCount;
      }
      h420_rows = (int) (((long long) tiles * out_h + slots - 1) / slots);
      if (h420_rows < 4)
        h420_rows = 4;
    }
    dim3 wgrid (tiles, (out_h + h420_rows - 1) / h420_rows);
    if (sd.nw == 5)
      hipLaunchKernelGGL (k_hscale420_dot4<5>, wgrid, dim3 (64), (size_t) GSTAMD_H420_PLANE_BYTES * 3, stream, src, sd, d, pf, out_w, out_h,
          geom.tile16_w, h420_rows, ablate);
    else
      hipLaunchKernelGGL (k_hscale420_dot4<0>, wgrid, dim3 (64), (size_t) GSTAMD_H420_PLANE_BYTES * 3, stream, src, sd, d, pf, out_w, out_h,
          geom.tile16_w, h420_rows, ablate);
  } else if (horizontal && geom.tile_w > 0 && sd.tapw && dot4_source_ok (src)) {
    /* planes of lds_px bytes + 8 (the aligned filter window may run a few zero-tap bytes past the span) */
    const int plane_w = (geom.lds_px + 8) / 4;
    dim3 wgrid ((out_w + geom.tile_w - 1) / geom.tile_w, out_h);
    if (sd.nw == 5)
      hipLaunchKernelGGL (k_hscale_dot4_wave<5>, wgrid, dim3 (64), (size_t) plane_w * 12, stream, src, sd, d, pf, out_w, out_h, geom.tile_w,
          plane_w, front_packed_ok (src) | ablate);
    else
      hipLaunchKernelGGL (k_hscale_dot4_wave<0>, wgrid, dim3 (64), (size_t) plane_w * 12, stream, src, sd, d, pf, out_w, out_h, geom.tile_w,
          plane_w, front_packed_ok (src));
  } else if (horizontal && geom.tile_w > 0 && geom.lds_px * 4 <= WAVE_TILE_LDS_BYTES) {
    dim3 wgrid ((out_w + geom.tile_w - 1) / geom.tile_w, out_h);
    hipLaunchKernelGGL (k_hscale_wave<SrcFront>, wgrid, dim3 (64), (size_t) geom.lds_px * 4, stream, src, sd, d, pf, out_w, out_h,
        geom.tile_w, geom.lds_px, front_packed_ok (src));
  } else if (horizontal && max_span <= HSCALE_LDS_PX)
    hipLaunchKernelGGL (k_hscale_lds<SrcFront>, grid, block, (size_t) max_span * 4, stream, src, sd, d, out_w, out_h);
  else if (horizontal)
    hipLaunchKernelGGL (k_hscale<SrcFront>, grid, block, 0, stream, src, sd, d, out_w, out_h);
  else
    hipLaunchKernelGGL (k_vscale<SrcFront>, grid, block, 0, stream, src, sd, d, out_w, out_h);
  return hipGetLastError ();
}

template <int NW, int SEMI>
static hipError_t launch_h420_reg_nw (H420RegParams p, int chroma_h, int n_taps, hipStream_t stream)
{
  const int lpw_env = tuning_int ("GSTAMD_H420_ROWS", -1);
  const size_t lds = (size_t) GSTAMD_H420_LINE_WORDS * 4 * 2;
  const void *fn = chroma_h == CHROMA_H_H2_CS ? (const void *) k_hscale420_reg<NW, CHROMA_H_H2_CS, SEMI> :
      (chroma_h == CHROMA_H_H2 ? (const void *) k_hscale420_reg<NW, CHROMA_H_H2, SEMI> : (const void *) k_hscale420_reg<NW, CHROMA_H_NONE, SEMI>);
  const int tiles = (p.out_w + p.tile_w - 1) / p.tile_w;
  /* line pairs per wave: every wave of the launch resident at once (one round, no tail) */
  static int slots[3] = {0, 0, 0};
  int &sl = slots[chroma_h == CHROMA_H_H2_CS ? 0 : (chroma_h == CHROMA_H_H2 ? 1 : 2)];
  if (!sl) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor (&per_cu, fn, 64, lds) != hipSuccess || per_cu <= 0)
      per_cu = 16;
    if (hipGetDevice (&dev) != hipSuccess || hipGetDeviceProperties (&prop, dev) != hipSuccess)
      prop.multiProcessorCount = 256;
    sl = per_cu * prop.multiProcessorCount;
  }
  const int pairs = p.height / 2 + 1;
  int ppw = lpw_env > 0 ? (lpw_env + 1) / 2 : (int) (((long long) tiles * pairs + sl - 1) / sl);
  if (ppw < 2)
    ppw = 2;
  p.lines_per_wave = 2 * ppw;
  dim3 grid (tiles, (pairs + ppw - 1) / ppw);
  if (chroma_h == CHROMA_H_H2_CS)
    hipLaunchKernelGGL ((k_hscale420_reg<NW, CHROMA_H_H2_CS, SEMI>), grid, dim3 (64), lds, stream, p, n_taps);
  else if (chroma_h == CHROMA_H_H2)
    hipLaunchKernelGGL ((k_hscale420_reg<NW, CHROMA_H_H2, SEMI>), grid, dim3 (64), lds, stream, p, n_taps);
  else
    hipLaunchKernelGGL ((k_hscale420_reg<NW, CHROMA_H_NONE, SEMI>), grid, dim3 (64), lds, stream, p, n_taps);
  return hipGetLastError ();
}

// regular 4:2:0 horizontal pass into the AYUV intermediate; hipErrorNotSupported: the caller takes the general kernels
hipError_t launch_hscale420_reg (const H420RegParams &p, int chroma_h, int nw, int n_taps, hipStream_t stream)
{
  video_frame_list_touch (p.dst);
  if (tuning_on ("GSTAMD_NO_H420_REG"))
    return hipErrorNotSupported;
  int ok = (p.width % 16) == 0 && aligned (p.y, 16) && (p.ystride % 16) == 0 && aligned (p.dst, 4) && (p.dstride % 4) == 0;
  if (p.semi)
    ok = ok && aligned (p.c0, 16) && (p.cstride % 16) == 0;
  else
    ok = ok && aligned (p.c0, 8) && aligned (p.c1, 8) && (p.cstride % 8) == 0;
  if (!ok)
    return hipErrorNotSupported;
  switch (nw) {
    case 3: return p.semi ? launch_h420_reg_nw<3, 1> (p, chroma_h, n_taps, stream) : launch_h420_reg_nw<3, 0> (p, chroma_h, n_taps, stream);
    case 4: return p.semi ? launch_h420_reg_nw<4, 1> (p, chroma_h, n_taps, stream) : launch_h420_reg_nw<4, 0> (p, chroma_h, n_taps, stream);
    case 5: return p.semi ? launch_h420_reg_nw<5, 1> (p, chroma_h, n_taps, stream) : launch_h420_reg_nw<5, 0> (p, chroma_h, n_taps, stream);
    default: return hipErrorNotSupported;
  }
}

hipError_t launch_convert420p (const Fast420pParams &p, uint8_t *dst, int dstride, hipStream_t stream)
{
  int nz;
  const FrameDeltas &fl = frame_list_for (p.y, dst, &nz);
  dim3 grid ((p.fp.width / 8 + 255) / 256, (p.fp.height + 1) / 2, nz);
  hipLaunchKernelGGL (k_convert420p, grid, dim3 (256), 0, stream, p, dst, dstride, fl);
  return hipGetLastError ();
}

hipError_t launch_convert422 (const Fast422Params &p, const uint8_t *src, int sstride, uint8_t *dst, int dstride, hipStream_t stream, bool ayuv)
{
  int nz;
  const FrameDeltas &fl = frame_list_for (src, dst, &nz);
  dim3 grid ((p.fp.width / 8 + 255) / 256, p.fp.height, nz);
  if (ayuv)
    hipLaunchKernelGGL (k_convert422_ayuv, grid, dim3 (256), 0, stream, p, src, sstride, dst, dstride, fl);
  else
    hipLaunchKernelGGL (k_convert422, grid, dim3 (256), 0, stream, p, src, sstride, dst, dstride, fl);
  return hipGetLastError ();
}

hipError_t launch_scale_from_image (bool horizontal, const uint8_t *simg, int sstride, const ScaleDev &sd, uint8_t *dst,
    int dstride, bool final, const ColorParams &post, const int pack_pos[4], int out_w, int out_h, int max_span, int src_w,
    TileGeom geom, const PostFast &pf, hipStream_t stream)
{
  video_frame_list_touch (dst);
  SrcImage src;
  src.p = simg;
  src.stride = sstride;
  src.width = src_w;
  Dst d = make_dst (dst, dstride, final, post, pack_pos);
  dim3 block (256), grid ((out_w + 255) / 256, out_h);
  if (horizontal && geom.tile_w > 0 && geom.lds_px * 4 <= WAVE_TILE_LDS_BYTES) {
    dim3 wgrid ((out_w + geom.tile_w - 1) / geom.tile_w, out_h);
    hipLaunchKernelGGL (k_hscale_wave<SrcImage>, wgrid, dim3 (64), (size_t) geom.lds_px * 4, stream, src, sd, d, pf, out_w, out_h,
        geom.tile_w, geom.lds_px, 1);
  } else if (horizontal && max_span <= HSCALE_LDS_PX)
    hipLaunchKernelGGL (k_hscale_lds<SrcImage>, grid, block, (size_t) max_span * 4, stream, src, sd, d, out_w, out_h);
  else if (horizontal)
    hipLaunchKernelGGL (k_hscale<SrcImage>, grid, block, 0, stream, src, sd, d, out_w, out_h);
  else if (sd.kind == SCALE_NTAP) {
    /* the packed kernel runs the generic post stage too (post_px), so it serves every N-tap vertical pass */
    const int vrows = tuning_int ("GSTAMD_VSCALE_ROWS", 1);
    /* neighbouring output rows of an N-tap filter always share source rows (the window is 2a steps wide) */
    if (vrows == 4) {
      dim3 vgrid ((out_w + 255) / 256, (out_h + 3) / 4);
      hipLaunchKernelGGL (k_vscale_pk_rows<4>, vgrid, dim3 (64), 0, stream, src, sd, d, pf, out_w, out_h);
    } else if (vrows == 2) {
      dim3 vgrid ((out_w + 255) / 256, (out_h + 1) / 2);
      hipLaunchKernelGGL (k_vscale_pk_rows<2>, vgrid, dim3 (64), 0, stream, src, sd, d, pf, out_w, out_h);
    } else {
      dim3 vgrid ((out_w + 255) / 256, out_h);
      hipLaunchKernelGGL (k_vscale_pk, vgrid, dim3 (64), 0, stream, src, sd, d, pf, out_w, out_h);
    }
  } else
    hipLaunchKernelGGL (k_vscale<SrcImage>, grid, block, 0, stream, src, sd, d, out_w, out_h);
  return hipGetLastError ();
}

__global__ __launch_bounds__ (256) void k_bilinear4_rows (Bil4Params b, Dst dst, PostFast pf)
{
  bilinear4_rows_lane (b, dst, pf, (int) (blockIdx.x * blockDim.x + threadIdx.x) * 4, (int) blockIdx.y * b.rows);
}

// the same lane on a packed 4:2:2 frame: every fetch unpacks a pixel (luma byte, chroma of its macropixel and of the neighbour the upsampler asks for) -
// two dword loads where k_scale2x2_wave staged the lines through LDS one wave per row and tile (YUY2 4K -> NV12 1080p: 50.7 us in that kernel)
__global__ __launch_bounds__ (256) void k_bilinear422_rows (Bil4Params b, Dst dst, PostFast pf)
{
  bilinear4_rows_lane<1> (b, dst, pf, (int) (blockIdx.x * blockDim.x + threadIdx.x) * 4, (int) blockIdx.y * b.rows);
}

// bilinear4_up_lane: a wave = 256 outputs x a strip of b.rows output rows (<= 64: one lane per row holds the row's table entries)
template <int PLAIN>
__global__ __launch_bounds__ (256) void k_bilinear4_up (Bil4Params b, Dst dst, PostFast pf, uint32_t plain_sel)
{
  const int lane = (int) threadIdx.x & 63;
  const int y0 = (int) blockIdx.y * b.rows, y1 = y0 + b.rows < b.out_h ? y0 + b.rows : b.out_h;
  /* table reads inside the row loop would be vector loads followed by s_waitcnt vmcnt(0) - which also waits for the line just requested */
  const int yl = y0 + lane < b.out_h ? y0 + lane : b.out_h - 1;
  const int tab_ya = (int) b.sv.offset[yl], tab_p1 = (int) b.sv.taps[(size_t) yl * 2 + 1];
  bilinear4_up_lane<PLAIN> (b, dst, pf, plain_sel, (int) (blockIdx.x * blockDim.x + threadIdx.x) * 4, y0, y1,
      [&] (int y, int *ya, uint32_t *p1) {
        *ya = __builtin_amdgcn_readlane (tab_ya, y - y0);
        *p1 = (uint32_t) __builtin_amdgcn_readlane (tab_p1, y - y0);
      });
}

hipError_t launch_scale2x2_from_front (const FrontParams &f, const Planes &pl, const int *vpair_dev, const ColorParams &pre,
    const ScaleDev &sh, const ScaleDev &sv, bool h_first, uint8_t *dst, int dstride, const ColorParams &post,
    const int pack_pos[4], int out_w, int out_h, int h_span, TileGeom geom, const PostFast &pf, hipStream_t stream)
{
  video_frame_list_touch (dst);
  SrcFront src;
  src.f = f;
  src.pl = pl;
  src.vpair = vpair_dev;
  src.pre = pre;
  src.vec_ok = front_vec_ok (f, pl);
  Dst d = make_dst (dst, dstride, true, post, pack_pos);
  dim3 block (256), grid ((out_w + 255) / 256, out_h);
  if (f.kind == UNPACK_PACKED4 && f.hi_depth == 0 && pre.matrix.kind == MATRIX_NONE && pre.alpha_kind == ALPHA_NONE && aligned (pl.p[0], 4) &&
      (pl.stride[0] % 4) == 0 && aligned (dst, 4) && (dstride % 4) == 0 && !tuning_on ("GSTAMD_NO_BILINEAR4")) {
    /* 4-byte packed source, nothing in front of the scaler: four outputs per lane, straight from memory (video_scale_fast.h) */
    Bil4Params b;
    memset ((void *) &b, 0, sizeof (b));
    b.src = pl.p[0], b.sstride = pl.stride[0], b.src_w = f.width, b.src_h = f.height;
    b.sel_in = (uint32_t) f.pos[0] | ((uint32_t) f.pos[1] << 8) | ((uint32_t) f.pos[2] << 16) | ((uint32_t) f.pos[3] << 24);
    b.sh = sh, b.sv = sv, b.h_first = h_first ? 1 : 0;
    /* rows per lane, measured: two for a 4K destination (BGRA 1080p -> 4K 31.0 us at four rows, 27.2 at two, 29.1 at one), one for smaller ones (4K ->
       1080p: 130 000 lanes at four rows, two waves per SIMD - 29.2 us against 23.8): every row of a lane starts with two dependent table reads */
    b.out_w = out_w, b.out_h = out_h, b.rows = (long) out_w * out_h >= 6000000 ? 2 : 1;
    if (bilinear4_up_ok (b) && !tuning_on ("GSTAMD_NO_BILINEAR4_UP")) {
      /* horizontal first, two taps both ways: source lines carried down strips of rows */
      b.rows = 8;
#ifdef GSTAMD_TUNING
      if (tuning_on ("GSTAMD_BIL4_UP_ROWS"))
        b.rows = tuning_int ("GSTAMD_BIL4_UP_ROWS", b.rows);
#endif
      b.rows = b.rows < 1 ? 1 : (b.rows > 64 ? 64 : b.rows);
      uint32_t sel = 0;
      const dim3 ugrid (((out_w + 3) / 4 + 255) / 256, (out_h + b.rows - 1) / b.rows);
      if (bilinear4_plain_sel (d, pf, &sel))
        hipLaunchKernelGGL (k_bilinear4_up<1>, ugrid, dim3 (256), 0, stream, b, d, pf, sel);
      else
        hipLaunchKernelGGL (k_bilinear4_up<0>, ugrid, dim3 (256), 0, stream, b, d, pf, sel);
      return hipGetLastError ();
    }
    hipLaunchKernelGGL (k_bilinear4_rows, dim3 (((out_w + 3) / 4 + 255) / 256, (out_h + b.rows - 1) / b.rows), dim3 (256), 0, stream, b, d, pf);
    return hipGetLastError ();
  }
  if (f.kind == UNPACK_PACKED422 && f.hi_depth == 0 && pre.matrix.kind == MATRIX_NONE && pre.alpha_kind == ALPHA_NONE && aligned (dst, 4) && (dstride % 4) == 0 &&
      !tuning_on ("GSTAMD_NO_BILINEAR4")) {
    Bil4Params b;
    memset ((void *) &b, 0, sizeof (b));
    b.src = pl.p[0], b.sstride = pl.stride[0], b.src_w = f.width, b.src_h = f.height;
    b.pos1 = f.pos[1], b.pos2 = f.pos[2], b.pos3 = f.pos[3], b.chroma_h = f.chroma_h, b.swap_k = f.swap_k;
    b.sh = sh, b.sv = sv, b.h_first = h_first ? 1 : 0;
    b.out_w = out_w, b.out_h = out_h, b.rows = (long) out_w * out_h >= 6000000 ? 2 : 1;
    hipLaunchKernelGGL (k_bilinear422_rows, dim3 (((out_w + 3) / 4 + 255) / 256, (out_h + b.rows - 1) / b.rows), dim3 (256), 0, stream, b, d, pf);
    return hipGetLastError ();
  }
  if (geom.tile_w > 0 && geom.lds_px * 8 <= WAVE_TILE_LDS_BYTES) {
    dim3 wgrid ((out_w + geom.tile_w - 1) / geom.tile_w, out_h);
    hipLaunchKernelGGL (k_scale2x2_wave<SrcFront>, wgrid, dim3 (64), (size_t) geom.lds_px * 8, stream, src, sh, sv, h_first ? 1 : 0, d,
        pf, out_w, out_h, geom.tile_w, geom.lds_px, front_packed_ok (src));
  } else if (h_span <= SCALE2_LDS_PX)
    hipLaunchKernelGGL (k_scale2x2_lds<SrcFront>, grid, block, (size_t) h_span * 8, stream, src, sh, sv, h_first ? 1 : 0, d, out_w, out_h, h_span);
  else
    hipLaunchKernelGGL (k_scale2x2<SrcFront>, grid, block, 0, stream, src, sh, sv, h_first ? 1 : 0, d, out_w, out_h);
  return hipGetLastError ();
}

// waves of k_bilinear420_rows the device holds at once (one workgroup = one wave): the balanced strip count aims at exactly one
// resident round, every SIMD with the same number of rows to do
static int bilr_wave_slots ()
{
  static int slots = 0;
  if (slots == 0) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice (&dev) != hipSuccess || hipDeviceGetAttribute (&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    const void *fn = (const void *) k_bilinear420_rows<CHROMA_H_H2_CS, GSTAMD_LAYOUT (2, 1, 0), 3>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor (&per_cu, fn, 64, bilr_lds_bytes ()) != hipSuccess || per_cu <= 0)
      per_cu = 16;
    (void) hipGetLastError ();
    slots = cus * per_cu;
  }
  return slots;
}

static int bil_vec_ok (const BilParams &bp, const Planes &pl)
{
  int vec = aligned (pl.p[0], 16) && aligned (pl.p[1], 16) && (pl.stride[0] % 16) == 0 && (pl.stride[1] % 16) == 0;
  if (bp.planar)
    /* planar sources only through the straight-line fetch: whole 16-pixel pieces, 8-byte chroma loads */
    vec = aligned (pl.p[0], 16) && (pl.stride[0] % 16) == 0 && aligned (pl.p[1], 8) && aligned (pl.p[2], 8) && (pl.stride[1] % 8) == 0 &&
        (pl.stride[2] % 8) == 0 && (bp.fp.width % 16) == 0;
  return vec;
}

static hipError_t launch_bilinear420_rows (const BilParams &bp, int chroma_h, int n, const Planes *pl, uint8_t *const *dst, int dstride, hipStream_t stream)
{
  const int rtiles = (bp.out_w + bp.rows_tile_w - 1) / bp.rows_tile_w;
  BilParams rp = bp;
  int slots = bilr_wave_slots ();
#ifdef GSTAMD_TUNING
  if (tuning_on ("GSTAMD_BIL_SLOTS"))
    slots = tuning_int ("GSTAMD_BIL_SLOTS", slots);
#endif
  /* one frame: as many strips as make one resident round (a round that tips over into a second one costs 20 %: keep a margin);
   * several frames: the rounds follow each other anyway, strips of six rows */
  rp.strips = n > 1 && bp.rows < 0 ? bilr_strips (bp.out_h, 6, rtiles, 0) : bilr_strips (bp.out_h, bp.rows, rtiles, slots - slots / 16);
  int wg = 1;
#ifdef GSTAMD_TUNING
  if (tuning_on ("GSTAMD_BIL_WG"))
    wg = tuning_int ("GSTAMD_BIL_WG", wg);
  if (tuning_on ("GSTAMD_BIL_VERBOSE"))
    fprintf (stderr, "k_bilinear420_rows: %d frame(s), slots %d, tiles %d, strips %d\n", n, slots, rtiles, rp.strips);
#endif
  const int blocks_per_frame = wide_grid_blocks (rtiles, rp.strips) / wg;         /* a multiple of 256 / wg */
  const size_t rlds = (size_t) wg * bilr_lds_bytes ();
  for (int base = 0; base < n; base += GSTAMD_BIL_MAX_BATCH) {
    const int nb = n - base < GSTAMD_BIL_MAX_BATCH ? n - base : GSTAMD_BIL_MAX_BATCH;
    BilBatch fb;
    memset (&fb, 0, sizeof (fb));
    for (int f = 0; f < nb; f++) {
      for (int k = 0; k < 3; k++)
        fb.p[k][f] = pl[base + f].p[k];
      fb.dst[f] = dst[base + f];
    }
    for (int k = 0; k < 3; k++)
      fb.stride[k] = pl[0].stride[k];
    dim3 rgrid ((unsigned) blocks_per_frame * (unsigned) nb), rblock (64 * wg);
#define WR(CH, LAY) \
    if (nb == 1 && bp.rows_tile_w > 256) \
      hipLaunchKernelGGL ((k_bilinear420_rows<CH, LAY, 3>), rgrid, rblock, rlds, stream, rp, pl[base], dst[base], dstride, rtiles); \
    else if (nb == 1) \
      hipLaunchKernelGGL ((k_bilinear420_rows<CH, LAY, 2>), rgrid, rblock, rlds, stream, rp, pl[base], dst[base], dstride, rtiles); \
    else if (bp.rows_tile_w > 256) \
      hipLaunchKernelGGL ((k_bilinear420_rows_frames<CH, LAY, 3>), rgrid, rblock, rlds, stream, rp, fb, dstride, rtiles, blocks_per_frame); \
    else \
      hipLaunchKernelGGL ((k_bilinear420_rows_frames<CH, LAY, 2>), rgrid, rblock, rlds, stream, rp, fb, dstride, rtiles, blocks_per_frame);
#define W(pr, pg, pb) case GSTAMD_LAYOUT (pr, pg, pb): \
    if (chroma_h == CHROMA_H_H2_CS) { \
      WR (CHROMA_H_H2_CS, GSTAMD_LAYOUT (pr, pg, pb)) \
    } else if (chroma_h == CHROMA_H_H2) { \
      WR (CHROMA_H_H2, GSTAMD_LAYOUT (pr, pg, pb)) \
    } else { \
      WR (CHROMA_H_NONE, GSTAMD_LAYOUT (pr, pg, pb)) \
    } \
    break;
    switch (fast_layout (bp.fp)) {
      GSTAMD_FOR_LAYOUTS (W)
      W (0, 0, 4)      /* GSTAMD_LAYOUT_AYUV: no colour stage */
      default:
        return hipErrorInvalidValue;
    }
#undef W
#undef WR
  }
  return hipGetLastError ();
}

hipError_t launch_bilinear420_frames (const BilParams &bp, int chroma_h, int n, const Planes *pl, uint8_t *const *dst, int dstride, hipStream_t stream)
{
  video_frame_list_touch (dst[0]);
  if (bilinear420_half_usable (bp, n, pl, dst, dstride))
    return launch_bilinear420_half (bp, chroma_h, n, pl, dst, dstride, stream);
  bool rows_ok = bp.rows != 0 && (bp.fp.width % 16) == 0 && bp.regular_pairs;
  for (int f = 0; f < n && rows_ok; f++)
    rows_ok = bil_vec_ok (bp, pl[f]) && pl[f].stride[0] == pl[0].stride[0] && pl[f].stride[1] == pl[0].stride[1] && pl[f].stride[2] == pl[0].stride[2];
  if (rows_ok)
    return launch_bilinear420_rows (bp, chroma_h, n, pl, dst, dstride, stream);
  for (int f = 0; f < n; f++) {
    const hipError_t e = launch_bilinear420 (bp, chroma_h, pl[f], dst[f], dstride, stream);
    if (e != hipSuccess)
      return e;
  }
  return hipSuccess;
}

hipError_t launch_bilinear420 (const BilParams &bp, int chroma_h, const Planes &pl, uint8_t *dst, int dstride, hipStream_t stream)
{
  video_frame_list_touch (dst);
  if (bilinear420_half_usable (bp, 1, &pl, &dst, dstride))
    return launch_bilinear420_half (bp, chroma_h, 1, &pl, &dst, dstride, stream);
  const int vec = bil_vec_ok (bp, pl);
  if (bp.planar && !vec)
    return hipErrorNotSupported;
  if (bp.rows != 0 && vec && (bp.fp.width % 16) == 0 && bp.regular_pairs)
    return launch_bilinear420_rows (bp, chroma_h, 1, &pl, &dst, dstride, stream);
  const int tiles_x = (bp.out_w + bp.tile_w - 1) / bp.tile_w;
  dim3 grid (wide_grid_blocks (tiles_x, bp.out_h));
  const size_t lds_bytes = bil_lds_words (bp.ylen) * 4;
#define W(pr, pg, pb) case GSTAMD_LAYOUT (pr, pg, pb): \
    if (chroma_h == CHROMA_H_H2_CS) \
      hipLaunchKernelGGL ((k_bilinear420<CHROMA_H_H2_CS, GSTAMD_LAYOUT (pr, pg, pb)>), grid, dim3 (64), lds_bytes, stream, bp, pl, dst, dstride, vec, tiles_x); \
    else if (chroma_h == CHROMA_H_H2) \
      hipLaunchKernelGGL ((k_bilinear420<CHROMA_H_H2, GSTAMD_LAYOUT (pr, pg, pb)>), grid, dim3 (64), lds_bytes, stream, bp, pl, dst, dstride, vec, tiles_x); \
    else \
      hipLaunchKernelGGL ((k_bilinear420<CHROMA_H_NONE, GSTAMD_LAYOUT (pr, pg, pb)>), grid, dim3 (64), lds_bytes, stream, bp, pl, dst, dstride, vec, tiles_x); \
    return hipGetLastError ();
  switch (fast_layout (bp.fp)) {
    GSTAMD_FOR_LAYOUTS (W)
    W (0, 0, 4)      /* GSTAMD_LAYOUT_AYUV: no colour stage */
  }
#undef W
  return hipErrorInvalidValue;
}

hipError_t launch_plane_simple (int kind, const uint8_t *src, int sstride, uint8_t *dst, int dstride, int n_elems, int ow, int oh, hipStream_t stream)
{
  video_frame_list_touch (dst);
  const SrcPlane s = {src, sstride, n_elems, 0};
  const DstPlane d = {dst, dstride, n_elems};
  hipLaunchKernelGGL (k_plane_simple, dim3 ((ow + 255) / 256, oh), dim3 (256), 0, stream, kind, s, d, ow, oh);
  return hipGetLastError ();
}

hipError_t launch_plane_pass (bool horizontal, const ScaleDev &sd, const uint8_t *src, int sstride, uint8_t *dst, int dstride, int n_elems,
    int ow, int oh, hipStream_t stream)
{
  video_frame_list_touch (dst);
  const SrcPlane s = {src, sstride, n_elems, 0};
  const DstPlane d = {dst, dstride, n_elems};
  if (horizontal)
    hipLaunchKernelGGL (k_plane_hscale, dim3 ((ow + 255) / 256, oh), dim3 (256), 0, stream, s, sd, d, ow, oh);
  else
    hipLaunchKernelGGL (k_plane_vscale, dim3 ((ow + 255) / 256, oh), dim3 (256), 0, stream, s, sd, d, ow, oh);
  return hipGetLastError ();
}

hipError_t launch_fill_border (uint8_t *p, int stride, int es, uint32_t value, uint32_t value_hi, int maxw, int maxh, int x0, int y0, int w, int h,
    hipStream_t stream)
{
  video_frame_list_touch (p);
  hipLaunchKernelGGL (k_fill_border, dim3 ((maxw + 255) / 256, maxh), dim3 (256), 0, stream, p, stride, es, value, value_hi, maxw, maxh, x0, y0, w, h);
  return hipGetLastError ();
}

hipError_t launch_encode420 (const Enc420Params &ep, bool semi, const uint8_t *src, int sstride, uint8_t *const planes[3], const int strides[3],
    hipStream_t stream)
{
  DstPlanes d;
  for (int i = 0; i < 3; i++) {
    d.p[i] = planes[i];
    d.stride[i] = strides[i];
  }
  int nz;
  const FrameDeltas &fl = frame_list_for (src, planes[0], &nz);
  const dim3 grid ((ep.width / 4 + 63) / 64, (ep.height + 1) / 2, nz);
  if (semi)
    hipLaunchKernelGGL (k_encode420<1>, grid, dim3 (64), 0, stream, ep, src, sstride, d, fl);
  else
    hipLaunchKernelGGL (k_encode420<0>, grid, dim3 (64), 0, stream, ep, src, sstride, d, fl);
  return hipGetLastError ();
}

__global__ __launch_bounds__ (256) void k_pack_down_v (PackPlanarParams pk, uint8_t *__restrict__ img, int stride)
{
  pack_down_v_px (pk, img, stride, (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

__global__ __launch_bounds__ (256) void k_pack_down_h (PackPlanarParams pk, uint8_t *__restrict__ img, int stride)
{
  pack_down_h_px (pk, img, stride, (int) (blockIdx.x * blockDim.x + threadIdx.x), (int) blockIdx.y);
}

// error-diffusion dither ahead of a planar / 3-byte / packed 4:2:2 pack: the chroma downsamplers in place on the AYUV image, the dither
// pass over every pixel of it, then the pack kernel as a pure selection (video_pack.h pack_select_only)
hipError_t launch_pack_planar_ed (const PackPlanarParams &pk, uint8_t *img, int sstride, uint8_t *const planes[3], const int strides[3],
    hipStream_t stream, void *ed_carry)
{
  video_frame_list_touch (planes[0]);
  const int rows = (pk.height + (1 << pk.h_sub) - 1) >> pk.h_sub;
  const dim3 grid ((pk.width + 255) / 256, rows);
  if (pk.down_v)
    hipLaunchKernelGGL (k_pack_down_v, grid, dim3 (256), 0, stream, pk, img, sstride);
  if (pk.down_h && pk.w_sub >= 1)
    hipLaunchKernelGGL (k_pack_down_h, grid, dim3 (256), 0, stream, pk, img, sstride);
  hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return e;
  DitherParams d = pk.dither;
  if ((e = launch_dither4 (d, img, sstride, pk.width, pk.height, stream, ed_carry)) != hipSuccess)
    return e;
  return launch_pack_planar (pack_select_only (pk), img, sstride, planes, strides, stream);
}

__global__ __launch_bounds__ (256) void k_convert_pack_422 (PackPlanarParams pk, Src422Dup src, DstPlanes d, int wide, FrameDeltas fl)
{
  GSTAMD_FRAME_Z;
  src.p += fls_;
  const int x0 = (int) (blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (x0 >= pk.width)
    return;
  if (wide && (pk.kind == UNPACK_PLANAR || pk.kind == UNPACK_SEMI) && pack_422dup_block8 (pk, src, d, x0, (int) blockIdx.y, fld_))
    return;
  pack_planar_body (pk, src, d, x0, (int) blockIdx.y, fld_);
  pack_planar_body (pk, src, d, x0 + 4, (int) blockIdx.y, fld_);
}

// the same for packed 4:2:2 frames whose chain upsamples the chroma horizontally (YUY2 / UYVY -> NV12 & co): pack_planar_block4 on Src422Up's
// rows inside the picture, the general body on its pixels along the edges
__global__ __launch_bounds__ (64) void k_convert_pack_422up (PackPlanarParams pk, Src422Up src, DstPlanes d, int wide, FrameDeltas fl)
{
  GSTAMD_FRAME_Z;
  src.p += fls_;
  const int x0 = (int) (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (wide && pack_planar_block4 (pk, src, d, x0, (int) blockIdx.y, fld_))
    return;
  pack_planar_body (pk, src, d, x0, (int) blockIdx.y, fld_);
}

// which unscaled chains the fused form pays for: the pixel source must be cheap per pixel, a lane evaluates up to 20 of them for its 4 x 2 block
bool convert_pack_usable (const FrontParams &f, const Planes &pl, const ColorParams &color)
{
  if (f.kind == UNPACK_PACKED4)
    return ((uintptr_t) pl.p[0] % 4) == 0 && (pl.stride[0] % 4) == 0;
  if (f.kind == UNPACK_PACKED422 && f.chroma_h != CHROMA_H_NONE && tuning_on ("GSTAMD_NO_CONVERT_PACK_422UP"))
    return false;
  return f.kind == UNPACK_PACKED422 && !f.chroma_v2 && color.matrix.kind == MATRIX_NONE && color.alpha_kind == ALPHA_NONE &&
      ((uintptr_t) pl.p[0] % 4) == 0 && (pl.stride[0] % 4) == 0;
}

hipError_t launch_convert_pack (const PackPlanarParams &pk, const FrontParams &f, const Planes &pl, const int *vpair, const ColorParams &color,
    uint8_t *const planes[3], const int strides[3], hipStream_t stream)
{
  DstPlanes d;
  for (int i = 0; i < 3; i++) {
    d.p[i] = planes[i];
    d.stride[i] = strides[i];
  }
  const int lanes = (pk.width + 3) / 4, rows = (pk.height + (1 << pk.h_sub) - 1) >> pk.h_sub;
  int nz;
  const FrameDeltas &fl = frame_list_for (pl.p[0], planes[0], &nz);
  if (f.kind == UNPACK_PACKED422 && f.chroma_h != CHROMA_H_NONE) {
    const Src422Up src = {pl.p[0], pl.stride[0], 8 * f.pos[1], 8 * f.pos[2], 8 * f.pos[3], f.swap_k, f.chroma_h, f.width, f.luma_last};
    /* pack_planar_block4 on Src422Up's rows: macropixels on 4 bytes (any 16-byte phase), plane rows on 4 */
    int wide = !pk.dither.on && (pk.kind == UNPACK_PLANAR || pk.kind == UNPACK_SEMI);
    for (int i = 0; wide && i < (pk.kind == UNPACK_SEMI ? 2 : 3); i++)
      wide = ((uintptr_t) planes[i] % 4) == 0 && (strides[i] % 4) == 0;
    hipLaunchKernelGGL (k_convert_pack_422up, dim3 ((lanes + 63) / 64, rows, nz), dim3 (64), 0, stream, pk, src, d, wide, fl);
  } else if (f.kind == UNPACK_PACKED422) {
    const Src422Dup src = {pl.p[0], pl.stride[0], 8 * f.pos[1], 8 * f.pos[2], 8 * f.pos[3], f.swap_k};
    /* pack_422dup_block8: source rows on 16 bytes, plane rows on 8 */
    int wide = ((uintptr_t) pl.p[0] % 16) == 0 && (pl.stride[0] % 16) == 0;
    for (int i = 0; wide && i < (pk.kind == UNPACK_SEMI ? 2 : 3); i++)
      wide = ((uintptr_t) planes[i] % 8) == 0 && (strides[i] % 8) == 0;
    hipLaunchKernelGGL (k_convert_pack_422, dim3 (((pk.width + 7) / 8 + 255) / 256, rows, nz), dim3 (256), 0, stream, pk, src, d, wide, fl);
  } else {
    (void) vpair;
    int wide = !pk.dither.on && (pk.kind == UNPACK_PLANAR || pk.kind == UNPACK_SEMI) && ((uintptr_t) pl.p[0] % 16) == 0 && (pl.stride[0] % 16) == 0 &&
        !tuning_on ("GSTAMD_NO_CONVERT_PACK_WIDE");
    for (int i = 0; wide && i < (pk.kind == UNPACK_SEMI ? 2 : 3); i++)
      wide = ((uintptr_t) planes[i] % 4) == 0 && (strides[i] % 4) == 0;
    hipLaunchKernelGGL (k_convert_pack, dim3 ((lanes + 63) / 64, rows, nz), dim3 (64), 0, stream, pk, make_src_packed4 (f, pl, color), d, wide, fl);
  }
  return hipGetLastError ();
}

hipError_t launch_pack_planar (const PackPlanarParams &pk, const uint8_t *src, int sstride, uint8_t *const planes[3], const int strides[3],
    hipStream_t stream)
{
  video_frame_list_touch (planes[0]);
  DstPlanes d;
  for (int i = 0; i < 3; i++) {
    d.p[i] = planes[i];
    d.stride[i] = strides[i];
  }
  const int lanes = (pk.width + 3) / 4, rows = (pk.height + (1 << pk.h_sub) - 1) >> pk.h_sub;
  /* pack_planar_block4: planar / semi-planar YUV without a dither stage, image rows on 16 bytes, plane rows on 4 */
  int wide = !pk.dither.on && (pk.kind == UNPACK_PLANAR || pk.kind == UNPACK_SEMI) && ((uintptr_t) src % 16) == 0 && (sstride % 16) == 0;
  for (int i = 0; wide && i < (pk.kind == UNPACK_SEMI ? 2 : 3); i++)
    wide = ((uintptr_t) planes[i] % 4) == 0 && (strides[i] % 4) == 0;
  hipLaunchKernelGGL (k_pack_planar, dim3 ((lanes + 255) / 256, rows), dim3 (256), 0, stream, pk, src, sstride, d, wide);
  return hipGetLastError ();
}

}  // namespace gstamd
