// tests/emu/emu_compositor.cpp - TEST INFRASTRUCTURE: host loop over compositor_device.h bodies.
#include <cstdlib>
#include <cstring>

#include "../../gstreamer_amd/csrc/compositor_device.h"

using namespace gstamd;

static int emu_rows_runs = 0, emu_strip_runs = 0, emu_direct_runs = 0;
extern "C" int emu_compositor_direct_runs (void) { return emu_direct_runs; }
extern "C" int emu_compositor_rows_runs (void) { return emu_rows_runs; }
extern "C" int emu_compositor_strip_runs (void) { return emu_strip_runs; }

static long emu_culled = 0;
extern "C" long emu_compositor_culled (void) { return emu_culled; }          /* pad-strip hits the culled form left out so far */
static void run_impl (const AggregateParams *p, const OpacityMaps *om, uint8_t *dst, int dstride, int rx0, int ry0, int rw, int rh);

extern "C" void emu_compositor_run (const AggregateParams *p, uint8_t *dst, int dstride, int rx0, int ry0, int rw, int rh)
{
  run_impl (p, nullptr, dst, dstride, rx0, ry0, rw, rh);
}

/* k_aggregate_direct_cull where launch () takes it (the direct form), the plain kernels elsewhere: maps[k] / all as OpacityMaps, indexed like p->pads */
extern "C" void emu_compositor_run_cull (const AggregateParams *p, const unsigned long long *const *maps, unsigned all, uint8_t *dst, int dstride, int rx0,
    int ry0, int rw, int rh)
{
  OpacityMaps om;
  memset (&om, 0, sizeof (om));
  for (int k = 0; k < p->n_pads; k++)
    om.map[k] = maps[k];
  om.all = all;
  run_impl (p, &om, dst, dstride, rx0, ry0, rw, rh);
}

/* gstamd_compositor_pad_opacity_map's result (k_opacity_map computes the same bits with a ballot) */
extern "C" void emu_compositor_opacity_map (const uint8_t *data, int w, int h, int stride, int ashift, unsigned long long *map)
{
  for (int y = 0; y < h; y++)
    map[y] = opacity_row_bits (data + (size_t) y * stride, w, ashift);
}

static void run_impl (const AggregateParams *p, const OpacityMaps *om, uint8_t *dst, int dstride, int rx0, int ry0, int rw, int rh)
{
  /* same lane decomposition as k_aggregate: 4 pixels per lane, scalar tail */
  AggregateParams q = *p;
  q.fast = !p->overlay;
  for (int i = 0; i < p->n_pads; i++)
    if (p->pads[i].mode == GSTAMD_COMPOSITOR_BLEND_MODE_SOURCE || p->pads[i].width < 4)
      q.fast = 0;
  if (q.fast && q.n_pads > 0 && q.bg_kind != 2 && getenv ("EMU_AGG_ROWS")) {
    /* k_aggregate_rows (tuning builds only): strips of 256 columns x `rows` rows, one entry list per pass */
    const int rows = atoi (getenv ("EMU_AGG_ROWS"));
    emu_rows_runs++;
    for (int y0 = ry0; y0 < ry0 + rh; y0 += rows)
      for (int wx0 = rx0; wx0 < rx0 + rw; wx0 += 256) {
        const int wx1 = wx0 + 256 < rx0 + rw ? wx0 + 256 : rx0 + rw, yend = y0 + rows < ry0 + rh ? y0 + rows : ry0 + rh;
        int nhx = 0;
        for (int k = 0; k < q.n_pads; k++)
          nhx += pad_xhit (q.pads[k], wx0, wx1) ? 1 : 0;
        const int per = agg_rows_per_pass (nhx, rows);
        for (int y = y0; y < yend; y += per) {
          RowHit list[AGG_LIST_MAX];
          const int ny = per < yend - y ? per : yend - y;
          const int n = agg_build_list_host (q, wx0, wx1, y, ny, list);
          for (int lane = 0; lane < 64; lane++) {
            const int x = wx0 + 4 * lane;
            int nv = rx0 + rw - x;
            nv = nv < 0 ? 0 : (nv > 4 ? 4 : nv);
            if (q.ashift == 0)
              aggregate_rows4<0, 4> (q, list, n, dst, dstride, x, y, nv);
            else
              aggregate_rows4<24, 4> (q, list, n, dst, dstride, x, y, nv);
          }
        }
      }
    return;
  }
  if (q.fast && q.n_pads > 0 && q.bg_kind != 2 && getenv ("EMU_AGG_STRIP_ROWS")) {
    /* k_aggregate_strip: strips of 64 * npx columns x `rows` rows, the pad walk of aggregate_strip */
    const int rows = atoi (getenv ("EMU_AGG_STRIP_ROWS"));
    int npx = getenv ("EMU_AGG_STRIP_PX") ? atoi (getenv ("EMU_AGG_STRIP_PX")) : 4;
    for (int k = 0; k < q.n_pads; k++)
      if (q.pads[k].width < 8)
        npx = 4;
    emu_strip_runs++;
    for (int y0 = ry0; y0 < ry0 + rh; y0 += rows)
      for (int wx0 = rx0; wx0 < rx0 + rw; wx0 += 64 * npx) {
        const int wx1 = wx0 + 64 * npx < rx0 + rw ? wx0 + 64 * npx : rx0 + rw, y1 = y0 + rows < ry0 + rh ? y0 + rows : ry0 + rh;
        uint32_t xmask = 0;
        for (int k = 0; k < q.n_pads; k++)
          xmask |= pad_xhit (q.pads[k], wx0, wx1) ? 1u << k : 0u;
        AggsLanePad lp = {0, 0, 0};
        for (int lane = 0; lane < 64; lane++) {
          const int x = wx0 + npx * lane;
          int nv = rx0 + rw - x;
          nv = nv < 0 ? 0 : (nv > npx ? npx : nv);
          if (q.ashift == 0 && npx == 4)
            aggregate_strip<0, 4, 4> (q, lp, xmask, dst, dstride, x, y0, y1, nv);
          else if (q.ashift == 0)
            aggregate_strip<0, 4, 8> (q, lp, xmask, dst, dstride, x, y0, y1, nv);
          else if (npx == 4)
            aggregate_strip<24, 4, 4> (q, lp, xmask, dst, dstride, x, y0, y1, nv);
          else
            aggregate_strip<24, 4, 8> (q, lp, xmask, dst, dstride, x, y0, y1, nv);
        }
      }
    return;
  }
  if (q.fast && q.n_pads > 0 && rw >= 4 && (q.bg_kind != 2 || (rw & 3) == 0) && !getenv ("EMU_AGG_NO_DIRECT")) {
    /* k_aggregate_direct: one wave = a strip of 256 columns of one row; the lanes past the rectangle leave, a lane with fewer than
     * four pixels left moves back onto the last four */
    emu_direct_runs++;
    const int last = rx0 + rw - 4;
    if (om) {
      /* k_aggregate_direct_cull: EMU_CULL_ROWS (default 2, as AGG_CULL_ROWS) canvas rows per wave */
      const int R = getenv ("EMU_CULL_ROWS") ? atoi (getenv ("EMU_CULL_ROWS")) : 2;
      for (int y0 = ry0; y0 < ry0 + rh; y0 += R)
        for (int s0 = rx0; s0 < rx0 + rw; s0 += 256) {
          const int wx1 = s0 + 256 < rx0 + rw ? s0 + 256 : rx0 + rw, wx0 = s0 < last ? s0 : last;
          unsigned long long masks[4] = {0, 0, 0, 0};
          for (int r = 0; r < R; r++) {
            if (y0 + r >= ry0 + rh)
              continue;
            const unsigned long long hits = direct_pads_host (q, wx0, wx1, y0 + r).mask;
            masks[r] = cull_mask (hits, direct_cover_mask_host (q, *om, wx0, wx1, y0 + r));
            emu_culled += __builtin_popcountll (hits) - __builtin_popcountll (masks[r]);
          }
          DirectPads dp = direct_pads_host (q, wx0, wx1, y0);
          uint32_t out[64][4][4];
          int xs[64], nl = 0;
          for (int lane = 0; lane < 64; lane++) {
            int x = s0 + 4 * lane;
            if (x >= rx0 + rw)
              break;
            x = x < last ? x : last;
            for (int r = 0; r < R; r++)
              for (int i = 0; i < 4; i++)
                out[nl][r][i] = q.bg_kind == 2 && y0 + r < ry0 + rh ? ((const uint32_t *) (dst + (size_t) (y0 + r) * dstride + 4 * (size_t) x))[i] : 0u;
#define CULL_CALL(S, K) do { if (R == 1) aggregate_direct4_rows<S, 0, K, 1> (q, dp, masks, out[nl], x, y0); else if (R == 2) aggregate_direct4_rows<S, 0, K, 2> (q, dp, masks, out[nl], x, y0); \
              else aggregate_direct4_rows<S, 0, K, 4> (q, dp, masks, out[nl], x, y0); } while (0)
            if (q.ashift == 0 && q.bg_kind == 2) CULL_CALL (0, 1);
            else if (q.ashift == 0) CULL_CALL (0, 0);
            else if (q.bg_kind == 2) CULL_CALL (24, 1);
            else CULL_CALL (24, 0);
#undef CULL_CALL
            xs[nl++] = x;
          }
          for (int l = 0; l < nl; l++)
            for (int r = 0; r < R && y0 + r < ry0 + rh; r++)
              memcpy (dst + (size_t) (y0 + r) * dstride + 4 * (size_t) xs[l], out[l][r], 16);
        }
      return;
    }
    for (int y = ry0; y < ry0 + rh; y++)
      for (int s0 = rx0; s0 < rx0 + rw; s0 += 256) {
        const int wx1 = s0 + 256 < rx0 + rw ? s0 + 256 : rx0 + rw, wx0 = s0 < last ? s0 : last;
        const DirectPads dp = direct_pads_host (q, wx0, wx1, y);
        uint32_t out[64][4];
        int xs[64], nl = 0;
        /* all lanes of the wave read (KEEP) before any of them stores, like the SIMD does */
        for (int lane = 0; lane < 64; lane++) {
          int x = s0 + 4 * lane;
          if (x >= rx0 + rw)
            break;
          x = x < last ? x : last;
          const uint32_t *dp32 = (const uint32_t *) (dst + (size_t) y * dstride + 4 * (size_t) x);
          for (int i = 0; i < 4; i++)
            out[nl][i] = q.bg_kind == 2 ? dp32[i] : 0u;
          if (q.ashift == 0 && q.bg_kind == 2)
            aggregate_direct4<0, 0, 1> (q, dp, out[nl], x, y);
          else if (q.ashift == 0)
            aggregate_direct4<0, 0, 0> (q, dp, out[nl], x, y);
          else if (q.bg_kind == 2)
            aggregate_direct4<24, 0, 1> (q, dp, out[nl], x, y);
          else
            aggregate_direct4<24, 0, 0> (q, dp, out[nl], x, y);
          xs[nl++] = x;
        }
        for (int l = 0; l < nl; l++)
          memcpy (dst + (size_t) y * dstride + 4 * (size_t) xs[l], out[l], 16);
      }
    return;
  }
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

// ---- plane-by-plane aggregation (compositor_planes.h): the host rectangle arithmetic + k_aggregate_plane's grid
#include "../../gstreamer_amd/csrc/compositor_planes.h"

struct EmuFramePad {
  const uint8_t *data[3];
  int stride[3];
  int width, height, xpos, ypos;
  double alpha;
  int mode;
};

extern "C" int emu_compositor_aggregate_frame (int format, int background, const int *black, const int *white, const EmuFramePad *pads, int n_pads,
    uint8_t *const *dest, const int *dstride, int dw, int dh)
{
  const FormatDesc *f = format_desc (format);
  PlaneGeom geom[3];
  const int n_planes = compositor_plane_geometry (f, geom);
  if (!n_planes)
    return -1;
  for (int pl = 0; pl < n_planes; pl++) {
    PlaneJob job;
    memset (&job, 0, sizeof (job));
    job.dst = dest[pl];
    job.dstride = dstride[pl];
    job.wbytes = compositor_plane_row_bytes (f, geom[pl], dw, background);
    job.rows = sub_scale (dh, geom[pl].h_sub);
    compositor_plane_background (f, geom[pl], pl, background, black, white, &job);
    int done = 0;
    bool first = true;
    while (first || done < n_pads) {
      job.n = 0;
      while (done < n_pads && job.n < GSTAMD_PLANE_MAX_PADS) {
        const EmuFramePad &in = pads[done++];
        FramePad fp;
        for (int k = 0; k < 3; k++) {
          fp.data[k] = in.data[k];
          fp.stride[k] = in.stride[k];
        }
        fp.width = in.width;
        fp.height = in.height;
        fp.xpos = in.xpos;
        fp.ypos = in.ypos;
        fp.alpha = in.alpha;
        fp.mode = in.mode;
        if (compositor_pad_rect (f, geom[pl], pl, fp, dw, dh, &job.r[job.n]))
          job.n++;
      }
      for (int y = 0; y < job.rows; y++)
        for (int x0 = 0; x0 < job.wbytes; x0 += 4)
          plane_word_body (job, x0, y);
      job.bg_kind = 2;
      first = false;
    }
  }
  return 0;
}

// ---- ARGB64 / AYUV64 canvases (compositor_wide.h): k_aggregate64's grid
#include "../../gstreamer_amd/csrc/compositor_wide.h"

extern "C" int emu_sizeof_wide64 (void) { return (int) sizeof (Wide64Params); }

extern "C" void emu_compositor_wide64 (const Wide64Params *p, uint8_t *dst, int dstride, int rx0, int ry0, int rw, int rh)
{
  for (int y = 0; y < rh; y++)
    for (int x = 0; x < rw; x++)
      wide64_px (*p, dst, dstride, rx0 + x, ry0 + y);
}
