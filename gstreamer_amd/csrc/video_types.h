// video_types.h - parameter blocks shared by the host launchers and the device code
#pragma once
#include <stdint.h>

#include "planner.h"

namespace gstamd {

struct Planes {
  const uint8_t *p[4];
  int stride[4];
};

struct ColorParams {      // colour matrix + alpha stage (do_convert_lines + do_alpha_lines)
  MatrixParams matrix;
  int alpha_kind;
  int alpha_value;
};

struct ScaleDev {         // one scaler pass, tables resident in HBM (every member has a default: fields are filled one by one in places)
  int kind = 0;           // ScaleKind
  int n_taps = 0;
  int inc = 0;
  const uint32_t *offset = nullptr;
  const int16_t *taps = nullptr;
  const uint32_t *tapw = nullptr;   // ScalePass::tapw (byte-dot-product form), NULL when not applicable
  int nw = 0, nw4 = 0;
  int merged = 0;             // 0: plain.  1 / 2: the merged scaler of a packed 4:2:2 line (gst_video_scaler_combine_packed_YUV): outputs are
                          // BYTES, tap k of output byte x reads source byte offset[x] + k * 2 (luma: (x & 1) == merged - 1) or + k * 4 (chroma)
};

// an AYUV64 image in HBM (the 16-bit chain's lines, video_deep.h): 8 bytes per pixel, memory order A, c1, c2, c3
struct Deep16Image {
  const uint8_t *p;
  int stride;
  int width, height;
};

}  // namespace gstamd
