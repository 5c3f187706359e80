// video_testsrc.hip - gstamd_video_test_pattern_*: GstVideoTestSrc's frames painted in HBM (video_testsrc.h).
//
// One launch paints the frame's A, c1, c2, c3 image (a lane per four pixels, 16-byte stores), one converter call takes it into the caps' format through
// this library's generic chain (AYUV / ARGB -> format: the chroma downsampler of the caps' chroma-site and the format's packer - what the reference's
// convert_hline_generic calls line by line).  A frame whose caps ARE the painted format (AYUV, ARGB) is painted straight into the destination.
#include <hip/hip_runtime.h>
#include <string.h>

#include <string>

#include "../../include/gstamd_video.h"
#include "planner.h"
#include "video_testsrc.h"

using namespace gstamd;

extern "C" void gstamd_internal_set_error (const char *msg);

__global__ __launch_bounds__ (256) void k_test_pattern (TestPatternParams p, uint8_t *__restrict__ img, int stride)
{
  const int x0 = ((int) blockIdx.x * 256 + (int) threadIdx.x) * 4, y = (int) blockIdx.y;
  if (x0 >= p.w)
    return;
  uint32_t v[4];
#pragma unroll
  for (int i = 0; i < 4; i++)
    v[i] = x0 + i < p.w ? test_pattern_px (p, x0 + i, y) : 0u;
  uint8_t *row = img + (size_t) y * stride + (size_t) x0 * 4;
  if (x0 + 4 <= p.w && (((uintptr_t) row) & 15) == 0) {
    *(uint4 *) row = make_uint4 (v[0], v[1], v[2], v[3]);
  } else {
    for (int i = 0; i < 4 && x0 + i < p.w; i++)
      ((uint32_t *) row)[i] = v[i];
  }
}

struct GstAmdVideoTestPattern {
  TestPatternParams params;
  GstAmdVideoInfo info, painted;
  GstAmdVideoConverter *conv = nullptr;          // painted image -> the caps' format; NULL: the caps' format is the painted one
  uint8_t *img = nullptr;
  std::string desc;
};

extern "C" GstAmdVideoTestPattern *gstamd_video_test_pattern_new (const GstAmdVideoInfo *info, int pattern, uint32_t foreground_argb, uint32_t background_argb, int *status)
{
  auto fail = [&](int code, const char *msg) -> GstAmdVideoTestPattern * {
    gstamd_internal_set_error (msg);
    if (status)
      *status = code;
    return nullptr;
  };
  if (status)
    *status = GSTAMD_OK;
  if (!info || info->width <= 0 || info->height <= 0)
    return fail (GSTAMD_ERR_INVALID, "test pattern: NULL info or empty frame");
  if (!test_pattern_built (pattern))
    return fail (GSTAMD_ERR_UNSUPPORTED, "test pattern: this pattern is not built on the GPU path (circular, zone-plate, chroma-zone-plate, gamut, pinwheel, spokes, smpte-rp-219)");
  if (info->interlace_mode != GSTAMD_INTERLACE_MODE_PROGRESSIVE)
    return fail (GSTAMD_ERR_UNSUPPORTED, "test pattern: progressive frames only");
  GstAmdVideoTestPattern *t = new GstAmdVideoTestPattern ();
  t->info = *info;
  test_pattern_setup (&t->params, info, pattern, foreground_argb, background_argb);
  GstAmdVideoConverterConfig cfg;
  test_pattern_conversion (info, &t->painted, &cfg);
  if (t->painted.format != info->format) {
    int st = GSTAMD_OK;
    t->conv = gstamd_video_converter_new (&t->painted, info, &cfg, &st);
    if (!t->conv) {
      if (status)
        *status = st;
      delete t;
      return nullptr;
    }
    const hipError_t e = hipMalloc ((void **) &t->img, (size_t) t->painted.size);
    if (e != hipSuccess) {
      const std::string msg = std::string ("test pattern: hipMalloc of the painted image (") + std::to_string ((size_t) t->painted.size) + " bytes) failed: " + hipGetErrorString (e);
      gstamd_video_converter_free (t->conv);
      delete t;
      return fail (GSTAMD_ERR_HIP, msg.c_str ());
    }
  }
  t->desc = std::string ("test_pattern[") + std::to_string (pattern) + "]" + (t->conv ? std::string (" -> ") + gstamd_video_converter_describe (t->conv) : std::string (""));
  return t;
}

extern "C" int gstamd_video_test_pattern_frame (GstAmdVideoTestPattern *t, uint64_t n_frames, void *dest, void *stream_)
{
  if (!t || !dest) {
    gstamd_internal_set_error ("test pattern: NULL argument");
    return GSTAMD_ERR_INVALID;
  }
  hipStream_t stream = (hipStream_t) stream_;
  TestPatternParams p = t->params;
  test_pattern_frame (&p, n_frames);
  uint8_t *img = t->conv ? t->img : (uint8_t *) dest + t->info.offset[0];
  const int stride = t->conv ? t->painted.stride[0] : t->info.stride[0];
  const dim3 grid ((unsigned) ((p.w + 1023) / 1024), (unsigned) p.h);
  hipLaunchKernelGGL (k_test_pattern, grid, dim3 (256), 0, stream, p, img, stride);
  const hipError_t e = hipGetLastError ();
  if (e != hipSuccess) {
    gstamd_internal_set_error ((std::string ("k_test_pattern: ") + hipGetErrorString (e)).c_str ());
    return GSTAMD_ERR_HIP;
  }
  return t->conv ? gstamd_video_converter_frame (t->conv, t->img, dest, stream_) : GSTAMD_OK;
}

extern "C" const char *gstamd_video_test_pattern_describe (const GstAmdVideoTestPattern *t) { return t ? t->desc.c_str () : ""; }

extern "C" void gstamd_video_test_pattern_free (GstAmdVideoTestPattern *t)
{
  if (!t)
    return;
  if (t->img) {
    (void) hipDeviceSynchronize ();
    (void) hipFree (t->img);
  }
  gstamd_video_converter_free (t->conv);
  delete t;
}
