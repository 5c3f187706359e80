// video_kernels.h - device-side parameter blocks and host launchers of video_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "planner.h"
#include "video_types.h"
#include "video_scale_fast.h"
#include "video_bilinear_fast.h"
#include "video_hscale420.h"
#include "video_bilinear_rows.h"
#include "video_bilinear_half.h"
#include "video_422_fast.h"
#include "video_gamma.h"
#include "video_planes.h"
#include "video_relayout.h"
#include "video_swizzle34.h"
#include "video_v210_fast.h"

namespace gstamd {

// frame lists of the single-kernel plans (video_kernels.hip "frame lists"): workgroup z of a list launch adds s[z] to every source pointer and d[z] to
// every destination pointer
#define GSTAMD_MAX_BATCH 32
struct FrameDeltas {
  long long s[GSTAMD_MAX_BATCH], d[GSTAMD_MAX_BATCH];
};
#define GSTAMD_FRAME_Z const long long fls_ = fl.s[blockIdx.z], fld_ = fl.d[blockIdx.z]
const FrameDeltas &video_frame_list_for (const void *sp, const void *dp, int *nz);

hipError_t launch_convert (const FrontParams &f, const Planes &pl, const int *vpair_dev, const ColorParams &color,
    const int pack_pos[4], uint8_t *dst, int dstride, hipStream_t stream, int extra_rows = 0);

// 4-byte packed -> 4-byte packed, no matrix, no alpha operation: the copy-shaped permutation kernel
bool swizzle4_usable (const FrontParams &f, const Planes &pl, const ColorParams &color, const uint8_t *dst, int dstride);
hipError_t launch_swizzle4 (const FrontParams &f, const Planes &pl, const int pack_pos[4], uint8_t *dst, int dstride, hipStream_t stream);

hipError_t launch_convert16 (const FrontParams &f, const Planes &pl, const int *vpair_dev, const Deep16Params &d, const PostParams &post, uint8_t *dst,
    int dstride, hipStream_t stream);

hipError_t launch_convert_gamma (const FrontParams &f, const Planes &pl, const int *vpair_dev, const ColorParams &color, const int pack_pos[4], uint8_t *dst,
    int dstride, const GammaDev &g, hipStream_t stream);
hipError_t launch_deep_planes (const DeepPlanesParams &d, const DeepPlanesPtrs &pp, hipStream_t stream);
hipError_t launch_encode16 (const Enc16Params &ep, const uint8_t *src, int sstride, uint8_t *const planes[3], const int strides[3], hipStream_t stream);
hipError_t launch_pack16 (const PackPlanarParams &pk, int hi_depth, const DitherParams &dt, const uint8_t *src, int sstride, uint8_t *const planes[3],
    const int strides[3], hipStream_t stream);
hipError_t launch_gamma_stage (const GammaDev &g, int mask, const uint8_t *src, int sstride, uint8_t *dst, int dstride, int w, int h, hipStream_t stream);
// method bayer / none: k_dither4; verterr: k_dither_verterr; floyd-steinberg / sierra-lite: k_dither_ed, which wants `ed_carry` (w x 8 bytes of
// device memory) for rectangles taller than 1024 lines
hipError_t launch_dither4 (const DitherParams &d, uint8_t *img, int stride, int w, int h, hipStream_t stream, void *ed_carry = nullptr);
hipError_t launch_pack_planar_ed (const PackPlanarParams &pk, uint8_t *img, int sstride, uint8_t *const planes[3], const int strides[3],
    hipStream_t stream, void *ed_carry);
hipError_t launch_dither16_any (const DitherParams &d, uint8_t *img, int stride, int w, int h, hipStream_t stream, void *ed_carry);
hipError_t launch_pack16_ed (const PackPlanarParams &pk, int hi_depth, const DitherParams &dt, uint8_t *img, int sstride, uint8_t *const planes[3],
    const int strides[3], hipStream_t stream, void *ed_carry);
hipError_t launch_dither16_image (const DitherParams &d, uint8_t *img, int stride, int w, int h, hipStream_t stream);
struct Deep16Image;
hipError_t launch_front16 (const FrontParams &f, const Planes &pl, const int *vpair_dev, uint8_t *img, int istride, hipStream_t stream);
bool front_hscale16_usable (const FrontParams &f);
hipError_t launch_front_hscale16 (const FrontParams &f, const Planes &pl, const int *vpair_dev, const ScaleDev &sd, uint8_t *dst, int dstride, int ow, hipStream_t stream);
// d / post non-NULL: the last pass, fused with matrix16 + narrowing + alpha + pack into the 4-byte destination
hipError_t launch_scale16 (const Deep16Image &im, const ScaleDev &sd, bool horizontal, uint8_t *dst, int dstride, int ow, int oh, const Deep16Params *d,
    const PostParams *post, hipStream_t stream);

struct FastParams;
hipError_t launch_convert_pair (const FastParams &fp, int chroma_h, int n_frames, const uint8_t *const *y, const uint8_t *const *uv,
    uint8_t *const *dst, int ystride, int uvstride, int dstride, hipStream_t stream);

hipError_t launch_scale_from_front (bool horizontal, const FrontParams &f, const Planes &pl, const int *vpair_dev,
    const ColorParams &pre, const ScaleDev &sd, uint8_t *dst, int dstride, bool final, const ColorParams &post,
    const int pack_pos[4], int out_w, int out_h, int max_span, TileGeom geom, const PostFast &pf, hipStream_t stream);

hipError_t launch_scale_from_image (bool horizontal, const uint8_t *simg, int sstride, const ScaleDev &sd, uint8_t *dst,
    int dstride, bool final, const ColorParams &post, const int pack_pos[4], int out_w, int out_h, int max_span, int src_w,
    TileGeom geom, const PostFast &pf, hipStream_t stream);

hipError_t launch_scale2x2_from_front (const FrontParams &f, const Planes &pl, const int *vpair_dev, const ColorParams &pre,
    const ScaleDev &sh, const ScaleDev &sv, bool h_first, uint8_t *dst, int dstride, const ColorParams &post,
    const int pack_pos[4], int out_w, int out_h, int h_span, TileGeom geom, const PostFast &pf, hipStream_t stream);

hipError_t launch_hscale420_reg (const H420RegParams &p, int chroma_h, int nw, int n_taps, hipStream_t stream);
hipError_t launch_convert422 (const Fast422Params &p, const uint8_t *src, int sstride, uint8_t *dst, int dstride, hipStream_t stream, bool ayuv = false);
hipError_t launch_convert420p (const Fast420pParams &p, uint8_t *dst, int dstride, hipStream_t stream);
hipError_t launch_bilinear420 (const BilParams &bp, int chroma_h, const Planes &pl, uint8_t *dst, int dstride, hipStream_t stream);
// n frames of the same geometry in as few launches as the kernel allows (one, when k_bilinear420_rows takes them)
// the exact halving (BilParams::half) on frames whose rows are 16-byte aligned: video_bilinear_half.hip
bool bilinear420_half_usable (const BilParams &bp, int n, const Planes *pl, uint8_t *const *dst, int dstride);
hipError_t launch_bilinear420_half (const BilParams &bp, int chroma_h, int n, const Planes *pl, uint8_t *const *dst, int dstride, hipStream_t stream);
hipError_t launch_bilinear420_frames (const BilParams &bp, int chroma_h, int n, const Planes *pl, uint8_t *const *dst, int dstride, hipStream_t stream);
hipError_t launch_plane_simple (int kind, const uint8_t *src, int sstride, uint8_t *dst, int dstride, int n_elems, int ow, int oh, hipStream_t stream);
hipError_t launch_plane_pass (bool horizontal, const ScaleDev &sd, const uint8_t *src, int sstride, uint8_t *dst, int dstride, int n_elems,
    int ow, int oh, hipStream_t stream);
hipError_t launch_fill_border (uint8_t *p, int stride, int es, uint32_t value, uint32_t value_hi, int maxw, int maxh, int x0, int y0, int w, int h,
    hipStream_t stream);
struct Enc420Params;
hipError_t launch_encode420 (const Enc420Params &ep, bool semi, const uint8_t *src, int sstride, uint8_t *const planes[3], const int strides[3],
    hipStream_t stream);
hipError_t launch_plane_frame (const PlaneJobs &jobs, size_t lds_bytes, hipStream_t stream);
bool swizzle34_setup (int src_bytes, const int *src_pos, int dst_bytes, const int *dst_pos, const uint8_t *src, int sstride, uint8_t *dst, int dstride,
    int width, Swz34Params *p);
hipError_t launch_swizzle34 (const Swz34Params &p, int src_bytes, int dst_bytes, int height, hipStream_t stream);
/* frame lists for single-kernel plans (video_kernels.hip) */
void video_frame_list_begin (int n, const void *const *src, void *const *dst, size_t src_size, size_t dst_size);
void video_frame_list_scratch (const void *p, size_t per_frame);
int video_frame_list_end ();
void video_frame_list_touch (const void *dp);
hipError_t launch_pack16_alpha_plane (const PackPlanarParams &pk, int hi_depth, const DitherParams &dt, const uint8_t *src, int sstride, uint8_t *plane, int stride,
    hipStream_t stream);
hipError_t launch_pack_alpha_plane (const PackPlanarParams &pk, const uint8_t *img, int istride, uint8_t *plane, int stride, hipStream_t stream);
hipError_t launch_v210_fast (const V210FastParams &p, hipStream_t stream);
hipError_t launch_lut3 (uint8_t *img, int stride, int w, int h, const uint8_t *comp_dev, int keep, hipStream_t stream);
bool relayout_usable (const RelayoutParams &p);
hipError_t launch_planes_relayout (const RelayoutParams &p, hipStream_t stream);
bool convert_pack_usable (const FrontParams &f, const Planes &pl, const ColorParams &color);
hipError_t launch_convert_pack (const PackPlanarParams &pk, const FrontParams &f, const Planes &pl, const int *vpair, const ColorParams &color,
    uint8_t *const planes[3], const int strides[3], hipStream_t stream);
hipError_t launch_pack_planar (const PackPlanarParams &pk, const uint8_t *src, int sstride, uint8_t *const planes[3], const int strides[3],
    hipStream_t stream);

}  // namespace gstamd
