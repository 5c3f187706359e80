// tests/emu/emu_compositor.cpp - TEST INFRASTRUCTURE: host loop over compositor_device.h bodies.
#include <cstring>

#include "../../gstreamer_amd/csrc/compositor_device.h"

using namespace gstamd;

extern "C" void emu_compositor_run (const AggregateParams *p, uint8_t *dst, int dstride, int rx0, int ry0, int rw, int rh)
{
  for (int y = ry0; y < ry0 + rh; y++)
    for (int x = rx0; x < rx0 + rw; x++) {
      uint32_t *dp = (uint32_t *) (dst + (size_t) y * dstride + 4 * (size_t) x);
      *dp = aggregate_px (*p, p->bg_kind == 2 ? *dp : 0u, x, y);
    }
}

extern "C" int emu_sizeof_params (void) { return (int) sizeof (AggregateParams); }
