// tests/emu/emu_compositor.cpp - TEST INFRASTRUCTURE: host loop over compositor_device.h bodies.
#include <cstring>

#include "../../gstreamer_amd/csrc/compositor_device.h"

using namespace gstamd;

extern "C" void emu_compositor_run (const AggregateParams *p, uint8_t *dst, int dstride, int rx0, int ry0, int rw, int rh)
{
  /* same lane decomposition as k_aggregate: 4 pixels per lane, scalar tail */
  AggregateParams q = *p;
  q.fast = !p->overlay;
  for (int i = 0; i < p->n_pads; i++)
    if (p->pads[i].mode == GSTAMD_COMPOSITOR_BLEND_MODE_SOURCE || p->pads[i].width < 4)
      q.fast = 0;
  for (int y = ry0; y < ry0 + rh; y++)
    for (int gx = 0; gx < rw; gx += 4) {
      const int bx0 = rx0 + (gx / 256) * 256, bx1 = bx0 + 256 < rx0 + rw ? bx0 + 256 : rx0 + rw;    /* the wave's strip */
      PadHit hits[GSTAMD_MAX_FUSED_PADS];
      int nh = 0;
      for (int k = 0; k < q.n_pads; k++)
        if (pad_hit_test (q, k, bx0, bx1, y, &hits[nh]))
          nh++;
      const int x = rx0 + gx, n = rw - gx < 4 ? rw - gx : 4;
      uint32_t *dp = (uint32_t *) (dst + (size_t) y * dstride + 4 * (size_t) x);
      if (n == 4) {
        uint32_t d[4];
        for (int i = 0; i < 4; i++)
          d[i] = q.bg_kind == 2 ? dp[i] : 0u;
        if (q.ashift == 0)
          aggregate_span4<0, 0> (q, hits, nh, d, x, y);
        else
          aggregate_span4<0, 24> (q, hits, nh, d, x, y);
        memcpy (dp, d, 16);
      } else {
        for (int i = 0; i < n; i++)
          dp[i] = aggregate_px (q, q.bg_kind == 2 ? dp[i] : 0u, x + i, y);
      }
    }
}

extern "C" int emu_sizeof_params (void) { return (int) sizeof (AggregateParams); }
