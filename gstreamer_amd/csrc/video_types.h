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

struct ScaleDev {         // one scaler pass, tables resident in HBM
  int kind;               // ScaleKind
  int n_taps;
  int inc;
  const uint32_t *offset;
  const int16_t *taps;
  const uint32_t *tapw;   // ScalePass::tapw (byte-dot-product form), NULL when not applicable
  int nw, nw4;
};

// an AYUV64 image in HBM (the 16-bit chain's lines, video_deep.h): 8 bytes per pixel, memory order A, c1, c2, c3
struct Deep16Image {
  const uint8_t *p;
  int stride;
  int width, height;
};

}  // namespace gstamd
