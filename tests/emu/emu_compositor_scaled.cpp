// tests/emu/emu_compositor_scaled.cpp - TEST INFRASTRUCTURE: gstamd_compositor_aggregate_scaled on the host - the planner's pad
// scaler plans with their tables in host memory, compositor_scaled.h's body over k_aggregate_scaled's grid.
#include <cstring>
#include <string>
#include <vector>

#include "../../gstreamer_amd/csrc/compositor_scaled.h"

using namespace gstamd;

static int g_tile_stages = 0;
extern "C" int emu_scaled_tile_stages (void) { return g_tile_stages; }

struct EmuScaledPad {
  const uint8_t *data;
  int width, height, stride, xpos, ypos;
  double alpha;
  int mode;
  int out_w, out_h;     // 0: blended as it is
  int method;           // GstVideoResamplerMethod of the pad's converter
};

extern "C" int emu_compositor_aggregate_scaled (int format, int ashift, int background, const EmuScaledPad *pads, int n_pads, uint8_t *dst,
    int dw, int dh, int dstride, uint32_t black_word, uint32_t white_word, int tile_rows)
{
  int th = tile_rows;           /* 0: as the library picks it (the first scaled pad's plan) */
  std::vector<VideoPlan> plans ((size_t) n_pads);
  ScaledAggParams p;
  memset ((void *) &p, 0, sizeof (p));
  p.ashift = ashift;
  p.overlay = background == GSTAMD_COMPOSITOR_BACKGROUND_TRANSPARENT;
  p.bg_kind = background == GSTAMD_COMPOSITOR_BACKGROUND_CHECKER ? 0 : 1;
  p.checker_yuv = format == GSTAMD_VIDEO_FORMAT_AYUV || format == GSTAMD_VIDEO_FORMAT_VUYA;
  p.bg_word = background == GSTAMD_COMPOSITOR_BACKGROUND_BLACK ? black_word : (background == GSTAMD_COMPOSITOR_BACKGROUND_WHITE ? white_word : 0);
  int done = 0;
  bool first = true;
  while (first || done < n_pads) {
    p.n_pads = 0;
    while (done < n_pads && p.n_pads < GSTAMD_MAX_SCALED_PADS) {
      const int i = done++;
      const EmuScaledPad &in = pads[i];
      int s_alpha = (int) (in.alpha * 255);
      s_alpha = s_alpha < 0 ? 0 : (s_alpha > 255 ? 255 : s_alpha);
      if (s_alpha == 0)
        continue;
      ScaledPadDev &sp = p.pads[p.n_pads];
      sp = ScaledPadDev ();
      sp.pad.data = in.data;
      sp.pad.width = in.width;
      sp.pad.height = in.height;
      sp.pad.stride = in.stride;
      sp.pad.xpos = in.xpos;
      sp.pad.ypos = in.ypos;
      sp.pad.s_alpha = s_alpha;
      sp.pad.mode = in.mode;
      sp.src_w = in.width;
      if (in.out_w > 0) {
        GstAmdVideoInfo ii, oi;
        GstAmdVideoConverterConfig cfg;
        std::string err;
        int hi, vi;
        video_info_set_format (&ii, format, in.width, in.height);
        video_info_set_format (&oi, format, in.out_w, in.out_h);
        converter_config_init (&cfg);
        cfg.resampler_method = in.method;
        if (plan_video_converter (&ii, &oi, &cfg, &plans[i], &err) != GSTAMD_OK || !plan_is_pad_scaler (plans[i], &hi, &vi))
          return -1;
        ScaleDev *out[2] = {&sp.sh, &sp.sv};
        const int idx[2] = {hi, vi};
        for (int k = 0; k < 2; k++)
          if (idx[k] >= 0) {
            const ScalePass &ps = plans[i].passes[idx[k]];
            out[k]->kind = ps.kind;
            out[k]->n_taps = ps.n_taps;
            out[k]->inc = ps.inc;
            out[k]->offset = ps.offset.data ();
            out[k]->taps = ps.taps.data ();
          }
        sp.h_first = hi >= 0 && vi >= 0 && hi < vi;
        sp.n_pass = (hi >= 0) + (vi >= 0);
        sp.pad.width = in.out_w;
        sp.pad.height = in.out_h;
        if (th <= 0)
          th = scaled_tile_rows_for (plans[i]);
      }
      p.n_pads++;
    }
    /* k_aggregate_scaled: a workgroup per 64 x 16 tile, the barriers are the ends of the lane loops */
    std::vector<uint32_t> lds (SCALED_LDS_PX);
    if (th <= 0)
      th = SCALED_TILE_H;
    for (int ty0 = 0; ty0 < dh; ty0 += th)
      for (int tx0 = 0; tx0 < dw; tx0 += SCALED_TILE_W) {
        const int tx1 = tx0 + SCALED_TILE_W < dw ? tx0 + SCALED_TILE_W : dw, ty1 = ty0 + th < dh ? ty0 + th : dh;
        uint32_t d[256][SCALED_TILE_H / 4];
        for (int tid = 0; tid < 256; tid++)
          for (int k = 0; k < SCALED_TILE_H / 4; k++) {
            const int x = tx0 + (tid & 63), y = ty0 + (tid >> 6) + 4 * k;
            d[tid][k] = 0;
            if (x < dw && y < ty1)
              d[tid][k] = p.bg_kind == 0 ? checker_px (x, y, p.ashift, p.checker_yuv) : (p.bg_kind == 1 ? p.bg_word : *((const uint32_t *) (dst + (size_t) y * dstride) + x));
          }
        for (int i = 0; i < p.n_pads; i++) {
          const ScaledPadDev &sp = p.pads[i];
          const ScaledTileGeom g = scaled_tile_geom (sp, tx0, ty0, tx1, ty1);
          if (g.mode == 0)
            continue;
          if (g.mode >= 2) {
            g_tile_stages++;
            for (int tid = 0; tid < 256; tid++)
              scaled_tile_stage (sp, g, lds.data (), tid, 256, sp.src_w);
          }
          for (int tid = 0; tid < 256; tid++)
            for (int k = 0; k < SCALED_TILE_H / 4; k++) {
              const int sx = tx0 + (tid & 63) - sp.pad.xpos, sy = ty0 + (tid >> 6) + 4 * k - sp.pad.ypos;
              if (sx >= g.sx0 && sx < g.sx1 && sy >= g.sy0 && sy < g.sy1) {
                const uint32_t s = g.mode >= 2 ? scaled_tile_px (sp, g, lds.data (), sx, sy) : scaled_pad_px (sp, sx, sy);
                d[tid][k] = apply_pad (d[tid][k], s, sp.pad.s_alpha, sp.pad.mode, p.ashift, p.overlay);
              }
            }
        }
        for (int tid = 0; tid < 256; tid++)
          for (int k = 0; k < SCALED_TILE_H / 4; k++) {
            const int x = tx0 + (tid & 63), y = ty0 + (tid >> 6) + 4 * k;
            if (x < dw && y < ty1)
              *((uint32_t *) (dst + (size_t) y * dstride) + x) = d[tid][k];
          }
      }
    p.bg_kind = 2;
    first = false;
  }
  return 0;
}
