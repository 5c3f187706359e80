// video_planes.h - convert_scale_planes of the reference (video-converter.c:7757, setup_scale :7958-8200) for planar and
// semi-planar formats: every destination plane is produced from one source plane on its own - copied, halved / doubled
// by the video_orc_planar_chroma_* helpers (video-orc.orc:1271-1334) or sent through gst_video_scaler_2d as 1-byte
// (GRAY8), 2-byte (the UV plane of the NV12 family) or 3-byte (RGB / BGR) pixels; a packed 4:2:2 line is scaled
// vertically as plain bytes.  The scaler arithmetic is the 4 x u8 one of video_device.h
// applied to the bytes the plane has (hscale_px / vscale_px on words whose upper bytes are zero).
#pragma once
#include "video_device.h"

namespace gstamd {

struct SrcPlane {
  const uint8_t *p;
  int stride;
  int n;                // bytes per pixel: 1, 2 or 3 (RGB / BGR)
  GSTAMD_HD uint32_t at (int x, int y) const
  {
    const uint8_t *q = p + (size_t) y * stride + (size_t) x * n;
    uint32_t v = q[0];
    if (n >= 2)
      v |= (uint32_t) q[1] << 8;
    if (n >= 3)
      v |= (uint32_t) q[2] << 16;
    return v;
  }
};

struct DstPlane {
  uint8_t *p;
  int stride;
  int n;
  GSTAMD_HD void put (int x, int y, uint32_t px) const
  {
    uint8_t *q = p + (size_t) y * stride + (size_t) x * n;
    q[0] = (uint8_t) px;
    if (n >= 2)
      q[1] = (uint8_t) (px >> 8);
    if (n >= 3)
      q[2] = (uint8_t) (px >> 16);
  }
};

GSTAMD_HD uint32_t avgub (uint32_t a, uint32_t b) { return (a + b + 1) >> 1; }

// the pass-free plane kinds (n == 1): one lane = one output byte
GSTAMD_HD void plane_simple_body (int kind, const SrcPlane &s, const DstPlane &d, int ow, int oh, int x, int y)
{
  if (x >= ow || y >= oh)
    return;
  uint32_t v;
  switch (kind) {
    case PLANE_H_HALVE:      /* video_orc_planar_chroma_444_422 */
      v = avgub (s.at (2 * x, y), s.at (2 * x + 1, y));
      break;
    case PLANE_H_DOUBLE:     /* video_orc_planar_chroma_422_444 */
      v = s.at (x >> 1, y);
      break;
    case PLANE_V_HALVE:      /* video_orc_planar_chroma_422_420 */
      v = avgub (s.at (x, 2 * y), s.at (x, 2 * y + 1));
      break;
    case PLANE_V_DOUBLE:     /* video_orc_planar_chroma_420_422 */
      v = s.at (x, y >> 1);
      break;
    case PLANE_HV_HALVE:     /* video_orc_planar_chroma_444_420: vertical averages first, then the pair */
      v = avgub (avgub (s.at (2 * x, 2 * y), s.at (2 * x, 2 * y + 1)), avgub (s.at (2 * x + 1, 2 * y), s.at (2 * x + 1, 2 * y + 1)));
      break;
    case PLANE_HV_DOUBLE:    /* video_orc_planar_chroma_420_444 */
      v = s.at (x >> 1, y >> 1);
      break;
    default:                 /* PLANE_COPY */
      v = s.at (x, y);
      break;
  }
  d.put (x, y, v);
}

GSTAMD_HD void plane_hscale_body (const SrcPlane &s, const ScaleDev &sd, const DstPlane &d, int ow, int rows, int x, int y)
{
  if (x >= ow || y >= rows)
    return;
  const RowOfSrc<SrcPlane> row = {s, y};
  d.put (x, y, hscale_px (row, sd, x));
}

GSTAMD_HD void plane_vscale_body (const SrcPlane &s, const ScaleDev &sd, const DstPlane &d, int width, int oh, int x, int y)
{
  if (x >= width || y >= oh)
    return;
  d.put (x, y, vscale_px (s, sd, x, y));
}

}  // namespace gstamd
