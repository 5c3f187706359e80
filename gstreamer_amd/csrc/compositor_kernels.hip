// compositor_kernels.hip - gfx950 kernel + C ABI for the compositor blend path.
//
// ONE kernel serves every entry point: it evaluates, for each destination pixel of a rectangle,
// background -> pad 0 -> pad 1 -> ... in registers and stores the pixel once (16 bytes per lane).
// The reference read-modify-writes the canvas once per overlapping pad (compositor.c:1678-1697);
// here HBM traffic is the sum of the covered source pixels plus one canvas write.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gstamd_video.h"
#include "compositor_device.h"
#include "tuning.h"
#include "compositor_planes.h"
#include "compositor_wide.h"
#include "compositor_scaled.h"
#include "compositor_walk.h"

using namespace gstamd;

#define AGG_CULL_ROWS 2       /* canvas rows per wave of k_aggregate_direct_cull (1, 2 or 4) */
#define AGG_DIRECT_NT 1      /* cache policy of k_aggregate_direct's pad requests (1: nt); the tuning build reads GSTAMD_AGG_NT */

struct __attribute__ ((aligned (4))) px4 { uint32_t v[4]; };

// the canvas is written once and not read again by this kernel: streaming (nontemporal) 16-byte store at a
// 4-byte aligned address
typedef unsigned int u32x4_a4 __attribute__ ((ext_vector_type (4), aligned (4)));
static __device__ __forceinline__ void store_px4_stream (uint8_t *p, const px4 &d)
{
  const u32x4_a4 v = {d.v[0], d.v[1], d.v[2], d.v[3]};
  __builtin_nontemporal_store (v, (u32x4_a4 *) p);
}

// One wave = 256 consecutive destination pixels of one row (4 per lane).  Each wave first finds the pads that touch
// its strip - one lane per pad, ballot-compacted into the wave's own LDS slice in pad order, no block barrier -
// then every lane walks that list (aggregate_span4).
template <int ABL, int ASH>
__global__ __launch_bounds__ (256) void k_aggregate (AggregateParams p, uint8_t *__restrict__ dst, int dstride, int rx0,
    int ry0, int rw, int rh)
{
  __shared__ PadHit hits_all[4][GSTAMD_MAX_FUSED_PADS];
  const int wave = (int) threadIdx.x >> 6, lane = (int) threadIdx.x & 63;
  PadHit *hits = hits_all[wave];
  const int gx = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int y = ry0 + (int) blockIdx.y;
  const int wx0 = rx0 + (int) (blockIdx.x * blockDim.x + wave * 64) * 4;
  if (wx0 >= rx0 + rw)
    return;                                       // whole wave right of the rectangle
  const int wx1 = wx0 + 256 < rx0 + rw ? wx0 + 256 : rx0 + rw;
  PadHit h;
  const bool hit = pad_hit_test (p, lane < p.n_pads ? lane : 0, wx0, wx1, y, &h) && lane < p.n_pads;
  const unsigned long long m = __ballot (hit);
  if (hit)
    hits[__popcll (m & ((1ull << lane) - 1ull))] = h;
  const int nh = __popcll (m);
  __builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier ();
  __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "wavefront");
  if (gx >= rw)
    return;
  const int x = rx0 + gx;
  uint8_t *row = dst + (size_t) y * dstride;
  const int n = rw - gx < 4 ? rw - gx : 4;
  if (n == 4) {
    px4 d;
    if (p.bg_kind == 2)
      d = *(const px4 *) (row + 4 * (size_t) x);
    else
      d.v[0] = d.v[1] = d.v[2] = d.v[3] = 0;
    aggregate_span4<ABL, ASH> (p, hits, nh, d.v, x, y);
    store_px4_stream (row + 4 * (size_t) x, d);
  } else {
    for (int i = 0; i < n; i++) {
      uint32_t *dp = (uint32_t *) (row + 4 * (size_t) (x + i));
      *dp = aggregate_px (p, p.bg_kind == 2 ? *dp : 0u, x + i, y);
    }
  }
}

// The opaque-blend path for <= AGG_DIRECT_PADS pads (compositor_device.h: aggregate_direct4): one wave = 256 destination pixels of
// one row, pad descriptors and hit tests on the scalar unit, every pad row's 16 bytes requested before the first blend.
// KEEP: a continuation chunk (bg_kind 2) starts from the canvas - a compile-time variant, because a canvas load that is only
// *possibly* outstanding makes the compiler put a full s_waitcnt vmcnt(0) in front of every request of aggregate_direct4.
// No scalar tail: the lanes past the rectangle and a lane with fewer than four pixels left move back onto the rectangle's last
// four (it and its neighbour compute the same values for the pixels they share; with KEEP the launcher requires rw % 4 == 0) - any
// compiler-visible memory operation ahead of the requests would put counted waits between them.
template <int ASH, int KEEP, int NT>
__global__ __launch_bounds__ (64) void k_aggregate_direct (AggregateParams p, uint8_t *__restrict__ dst, int dstride, int rx0, int ry0, int rw)
{
  const int lane = (int) threadIdx.x;
  const int y = ry0 + (int) blockIdx.y;
  const int last = rx0 + rw - 4;
  int wx0 = rx0 + (int) blockIdx.x * 256;
  const int wx1 = wx0 + 256 < rx0 + rw ? wx0 + 256 : rx0 + rw;
  int x = wx0 + 4 * lane;
  /* every lane stays: lane k tests pad k below (a lane that left before the ballot took its pad with it - in a short last strip the pads
     from index ceil (rem / 4) up were never blended), and pad k's fields are read back from lane k.  Lanes past the rectangle move
     onto its last four pixels like the one lane that straddles the edge: the same values to the same address. */
  x = x < last ? x : last;
  wx0 = wx0 < last ? wx0 : last;
  // lane k tests pad k and keeps what a hit needs
  const PadDev pad = p.pads[lane < p.n_pads ? lane : 0];
  DirectPads dp;
  dp.mask = __ballot ((lane < p.n_pads) & pad_hits_strip (pad, wx0, wx1, y));
  const uint64_t prow = (uint64_t) (uintptr_t) (pad.data + (ptrdiff_t) (y - pad.ypos) * pad.stride);
  dp.row_lo = (uint32_t) prow;
  dp.row_hi = (uint32_t) (prow >> 32);
  dp.alpha8081 = (uint32_t) pad.s_alpha * 0x8081u;
  dp.xpos = pad.xpos;
  dp.width = pad.width;
  uint8_t *row = dst + (size_t) y * dstride;
  px4 d;
  if (KEEP)
    d = *(const px4 *) (row + 4 * (size_t) x);
  else
    d.v[0] = d.v[1] = d.v[2] = d.v[3] = 0;
  aggregate_direct4<ASH, NT, KEEP> (p, dp, d.v, x, y);
  store_px4_stream (row + 4 * (size_t) x, d);
}

#ifdef GSTAMD_TUNING
// Several rows per wave (compositor_device.h, "Several rows per wave"): workgroup = one wave = 256 columns x `rows` rows.  Measured
// on C4 and NOT used by the product library: 1 row 38.5 us, 2 rows 37.1, 4 rows 40.5, 8 rows 53 (k_aggregate: 36.1) - the
// waves get fewer and the ones under nine pads longer, which costs more than the descriptor fetch and the pipeline drain it saves
// (profiles/r02_c4_rows_variants.log).
template <int ASH, int DEPTH>
__global__ __launch_bounds__ (64) void k_aggregate_rows (AggregateParams p, uint8_t *__restrict__ dst, int dstride, int rx0, int ry0,
    int rw, int rh, int rows, int strips)
{
  __shared__ RowHit list[AGG_LIST_MAX];
  const int lane = (int) threadIdx.x;
  const int strip = (int) blockIdx.x % strips, rg = (int) blockIdx.x / strips;
  const int wx0 = rx0 + strip * 256, wx1 = wx0 + 256 < rx0 + rw ? wx0 + 256 : rx0 + rw;
  const int x = wx0 + 4 * lane;
  int nv = rx0 + rw - x;
  nv = nv < 0 ? 0 : (nv > 4 ? 4 : nv);
  int y = ry0 + rg * rows;
  const int yend = y + rows < ry0 + rh ? y + rows : ry0 + rh;
  const PadDev pad = p.pads[lane < p.n_pads ? lane : 0];
  const bool xhit = (lane < p.n_pads) & pad_xhit (pad, wx0, wx1);
  const int per = agg_rows_per_pass (__popcll (__ballot (xhit)), rows);
  const unsigned long long below = (1ull << lane) - 1ull;
  while (y < yend) {
    const int ny = per < yend - y ? per : yend - y;
    int n = 0;
    for (int r = 0; r < ny; r++) {
      const int sy = y + r - pad.ypos;
      const bool hit = xhit & (sy >= 0) & (sy < pad.height);
      const unsigned long long m = __ballot (hit);
      const int c = __popcll (m);
      if (hit) {
        const int at = n + __popcll (m & below);
        RowHit e;
        e.row = pad.data + (ptrdiff_t) sy * pad.stride;
        e.xpos = pad.xpos;
        e.width = pad.width;
        e.ctl = pad.s_alpha | (at == n + c - 1 ? AGG_ROW_END : 0);
        list[at] = e;
      }
      if (c == 0 && lane == 0) {
        RowHit e;
        e.row = pad.data;                         // lane 0 holds pad 0: any readable 16 bytes
        e.xpos = 0;
        e.width = 4;
        e.ctl = AGG_ROW_SKIP | AGG_ROW_END;
        list[n] = e;
      }
      n += c > 0 ? c : 1;
    }
    __builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier ();
    __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "wavefront");
    aggregate_rows4<ASH, DEPTH> (p, list, n, dst, dstride, x, y, nv);
    __builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier ();
    __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "wavefront");
    y += ny;
  }
}
#endif

#ifdef GSTAMD_TUNING
// Strip form with the pad walk on the scalar unit (compositor_device.h, "k_aggregate_strip"): workgroup = one wave = 64 NPX columns
// x `rows` rows.  Measured on C4 and NOT used by the product library: 17 % fewer vector instructions than k_aggregate (15.1 M against
// 18.1 M per frame) and the same 37 us - with the blend arithmetic taken out altogether k_aggregate still takes 33 us, the kernel
// is bound by its memory path (166 MB in 16-byte pieces from nine pad rows per wave), not by instruction issue; 8 pixels per lane:
// 49 us (profiles/r02_c4_strip_variants.log, r02_c4_ablation.log, r02_c4_sq_*.json).
template <int ASH, int DEPTH, int NPX>
__global__ __launch_bounds__ (64) void k_aggregate_strip (AggregateParams p, uint8_t *__restrict__ dst, int dstride, int rx0, int ry0, int rw,
    int rh, int rows, int strips)
{
  const int lane = (int) threadIdx.x;
  const int strip = (int) blockIdx.x % strips, rg = (int) blockIdx.x / strips;
  const int wx0 = rx0 + strip * 64 * NPX, wx1 = wx0 + 64 * NPX < rx0 + rw ? wx0 + 64 * NPX : rx0 + rw;
  const int x = wx0 + NPX * lane;
  int nv = rx0 + rw - x;
  nv = nv < 0 ? 0 : (nv > NPX ? NPX : nv);
  const int y0 = ry0 + rg * rows, y1 = y0 + rows < ry0 + rh ? y0 + rows : ry0 + rh;
  const PadDev pad = p.pads[lane < p.n_pads ? lane : 0];
  AggsLanePad lp;
  lp.xhit = (lane < p.n_pads) & pad_xhit (pad, wx0, wx1);
  lp.ypos = pad.ypos;
  lp.height = pad.height;
  const uint32_t xmask = (uint32_t) __ballot (lp.xhit);
  aggregate_strip<ASH, DEPTH, NPX> (p, lp, xmask, dst, dstride, x, y0, y1, nv);
}
#endif

/* ARGB64 / AYUV64 canvases: one lane per destination pixel (compositor_wide.h) */
__global__ __launch_bounds__ (64) void k_aggregate64 (Wide64Params p, uint8_t *dst, int dstride, int rx0, int ry0, int rw)
{
  const int x = (int) (blockIdx.x * blockDim.x + threadIdx.x);
  if (x < rw)
    wide64_px (p, dst, dstride, rx0 + x, ry0 + (int) blockIdx.y);
}

/* pads scaled inside the blend pass (compositor_scaled.h): a 64 x 16 tile per workgroup, four rows per lane, first passes through LDS */
__global__ __launch_bounds__ (256) void k_aggregate_scaled (ScaledAggParams p, uint8_t *dst, int dstride, int dw, int dh, int th)
{
  __shared__ uint32_t lds[SCALED_LDS_PX];
  const int tid = (int) threadIdx.x;
  const int tx0 = (int) blockIdx.x * SCALED_TILE_W, ty0 = (int) blockIdx.y * th;          /* th <= SCALED_TILE_H rows per tile */
  const int tx1 = tx0 + SCALED_TILE_W < dw ? tx0 + SCALED_TILE_W : dw, ty1 = ty0 + th < dh ? ty0 + th : dh;
  const int x = tx0 + (tid & 63), yb = ty0 + (tid >> 6);
  const bool xin = x < dw;
  uint32_t d[SCALED_TILE_H / 4];
#pragma unroll
  for (int k = 0; k < SCALED_TILE_H / 4; k++) {
    const int y = yb + 4 * k;
    d[k] = 0;
    if (xin && y < ty1)
      d[k] = p.bg_kind == 0 ? checker_px (x, y, p.ashift, p.checker_yuv) : (p.bg_kind == 1 ? p.bg_word : *((const uint32_t *) (dst + (size_t) y * dstride) + x));
  }
  for (int i = 0; i < p.n_pads; i++) {
    const ScaledPadDev &sp = p.pads[i];
    const ScaledTileGeom g = scaled_tile_geom (sp, tx0, ty0, tx1, ty1);
    if (g.mode == 0)
      continue;
    if (g.mode >= 2) {
      scaled_tile_stage (sp, g, lds, tid, 256, sp.src_w);
      __syncthreads ();
    }
    const int sx = x - sp.pad.xpos;
#pragma unroll
    for (int k = 0; k < SCALED_TILE_H / 4; k++) {
      const int sy = yb + 4 * k - sp.pad.ypos;
      if (sx >= g.sx0 && sx < g.sx1 && sy >= g.sy0 && sy < g.sy1) {
        const uint32_t s = g.mode >= 2 ? scaled_tile_px (sp, g, lds, sx, sy) : scaled_pad_px (sp, sx, sy);
        d[k] = apply_pad (d[k], s, sp.pad.s_alpha, sp.pad.mode, p.ashift, p.overlay);
      }
    }
    if (g.mode >= 2)
      __syncthreads ();
  }
#pragma unroll
  for (int k = 0; k < SCALED_TILE_H / 4; k++) {
    const int y = yb + 4 * k;
    if (xin && y < ty1)
      *((uint32_t *) (dst + (size_t) y * dstride) + x) = d[k];
  }
}

/* scaled pads as column walks (compositor_walk.h): one wave per workgroup; blocks [0, n_blocks) walk, the rest fill what no scaled pad covers */
/* eight waves per SIMD for the form a mosaic takes (opaque blends, no unscaled pads: 63 registers, nothing spilled; C4-A 45.9 us against 50.4
 * with six), six for the forms that carry the general blend */
template <int ASH, int FAST, int PLAIN>
__global__ __launch_bounds__ (64) __attribute__ ((amdgpu_waves_per_eu ((FAST && !PLAIN) ? 8 : 6, (FAST && !PLAIN) ? 8 : 6))) void k_aggregate_walk (WalkParams p, uint8_t *dst, int dstride, int dw, int dh)
{
  __shared__ uint4 lds4[64];
  int b = (int) blockIdx.x;
  if (p.xcd_span && b < 8 * p.xcd_span) {      /* workgroups go round the eight XCDs: give each a contiguous run of strips (neighbours share halo columns and canvas lines in its L2) */
    b = (b & 7) * p.xcd_span + (b >> 3);
    if (b >= p.n_blocks)
      return;
  }
  if (b >= p.n_blocks) {
    walk_fill<ASH> (p, b - (p.xcd_span ? 8 * p.xcd_span : p.n_blocks), dst, dstride, dw, dh);
    return;
  }
  int pi = 0;
  while (pi + 1 < p.n_walk && b >= p.walk[pi + 1].first_block)
    pi++;
  const WalkPad &wp = p.walk[pi];
  b -= wp.first_block;
  walk_wave<ASH, FAST, PLAIN> (p, wp, b % wp.tiles, b / wp.tiles, dst, dstride, dw, dh, (uint32_t *) lds4);
}

static thread_local std::string g_comp_error;
extern "C" const char *gstamd_last_error (void);

static int family_ashift (int format)
{
  switch (format) {
    case GSTAMD_VIDEO_FORMAT_ARGB: case GSTAMD_VIDEO_FORMAT_ABGR: case GSTAMD_VIDEO_FORMAT_AYUV:
      return 0;
    case GSTAMD_VIDEO_FORMAT_BGRA: case GSTAMD_VIDEO_FORMAT_RGBA: case GSTAMD_VIDEO_FORMAT_VUYA:     /* blend.h:58-65: VUYA takes BGRA's functions */
      return 24;
    default:
      return -1;
  }
}

// k_aggregate_direct with the pads under an opaque strip left out of the mask (compositor_device.h "opaque culling"): lane k also asks whether pad k
// covers the strip with opaque pixels - one 8-byte load of the pad row's opacity word - and the walk starts at the topmost such pad.  A kernel of its
// own: the plain form's arguments and code stay as they were measured.
template <int ASH, int KEEP, int NT, int R>
__global__ __launch_bounds__ (64) void k_aggregate_direct_cull (AggregateParams p, OpacityMaps om, uint8_t *__restrict__ dst, int dstride, int rx0, int ry0,
    int rw, int rh)
{
  const int lane = (int) threadIdx.x;
  const int y0 = ry0 + (int) blockIdx.y * R;
  const int last = rx0 + rw - 4;
  int wx0 = rx0 + (int) blockIdx.x * 256;
  const int wx1 = wx0 + 256 < rx0 + rw ? wx0 + 256 : rx0 + rw;
  int x = wx0 + 4 * lane;
  x = x < last ? x : last;
  wx0 = wx0 < last ? wx0 : last;
  const int k = lane < p.n_pads ? lane : 0;
  const PadDev pad = p.pads[k];
  const unsigned long long *map = om.map[k];
  const bool all = (om.all >> k) & 1u;
  unsigned long long masks[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int y = y0 + r;
    const bool hit = (lane < p.n_pads) & (y < ry0 + rh) & pad_hits_strip (pad, wx0, wx1, y);
    const unsigned long long hits = __ballot (hit);
    const unsigned long long covers = __ballot (hit && pad_covers_strip (pad, map, all, wx0, wx1, y));
    masks[r] = cull_mask (hits, covers);
  }
  DirectPads dp;
  dp.mask = 0;
  const uint64_t prow = (uint64_t) (uintptr_t) (pad.data + (ptrdiff_t) (y0 - pad.ypos) * pad.stride);
  dp.row_lo = (uint32_t) prow;
  dp.row_hi = (uint32_t) (prow >> 32);
  dp.alpha8081 = (uint32_t) pad.s_alpha * 0x8081u;
  dp.xpos = pad.xpos;
  dp.width = pad.width;
  dp.stride = pad.stride;
  px4 d[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    if (KEEP && y0 + r < ry0 + rh)
      d[r] = *(const px4 *) (dst + (size_t) (y0 + r) * dstride + 4 * (size_t) x);
    else
      d[r].v[0] = d[r].v[1] = d[r].v[2] = d[r].v[3] = 0;
  }
  uint32_t dv[R][4];
#pragma unroll
  for (int r = 0; r < R; r++)
#pragma unroll
    for (int i = 0; i < 4; i++)
      dv[r][i] = d[r].v[i];
  aggregate_direct4_rows<ASH, NT, KEEP, R> (p, dp, masks, dv, x, y0);
#pragma unroll
  for (int r = 0; r < R; r++)
    if (y0 + r < ry0 + rh) {
#pragma unroll
      for (int i = 0; i < 4; i++)
        d[r].v[i] = dv[r][i];
      store_px4_stream (dst + (size_t) (y0 + r) * dstride + 4 * (size_t) x, d[r]);
    }
}

// gstamd_compositor_pad_opacity_map: one wave = 256 pixels of one pad row, four per lane; 16 lanes = one bit of the row's word
__global__ __launch_bounds__ (64) void k_opacity_map (const uint8_t *__restrict__ src, int w, int stride, int ashift, unsigned long long *__restrict__ map)
{
  const int lane = (int) threadIdx.x, y = (int) blockIdx.y;
  const int x = ((int) blockIdx.x * 64 + lane) * 4;
  const uint8_t *row = src + (size_t) y * stride;
  bool ok = true;
  for (int i = 0; i < 4; i++)
    if (x + i < w)
      ok &= ((load_px1 (row + 4 * (size_t) (x + i)) >> ashift) & 0xffu) == 0xffu;
  const unsigned long long b = __ballot (ok);
  if (lane == 0) {
    unsigned long long bits = 0;
    for (int j = 0; j < 4; j++)
      if ((int) blockIdx.x * 256 + 64 * j < w && ((b >> (16 * j)) & 0xffffull) == 0xffffull)
        bits |= 1ull << ((int) blockIdx.x * 4 + j);
    if (bits)
      atomicOr (&map[y], bits);
  }
}

static int launch (const AggregateParams &p, void *dest, int dstride, int rx0, int ry0, int rw, int rh, void *stream, const OpacityMaps *om = nullptr)
{
  if (rw <= 0 || rh <= 0)
    return GSTAMD_OK;
  AggregateParams q = p;
  q.fast = !p.overlay;
  for (int i = 0; i < p.n_pads; i++)
    if (p.pads[i].mode == GSTAMD_COMPOSITOR_BLEND_MODE_SOURCE || p.pads[i].width < 4)
      q.fast = 0;
  /* one wave per workgroup: waves of a row carry very different pad counts, and single-wave groups let the
   * dispatcher fill wave slots as they free up (C4: 37.9 us per frame against 47.2 with 4-wave groups) */
  const int lanes = (rw + 3) / 4;
  int bx = 64;
  /* development knobs (not a product interface): ablation of the blend (1), the pad loads (2), the hit loop (3),
   * both (4); workgroup width */
#ifdef GSTAMD_TUNING
  const int abl = tuning_int ("GSTAMD_AGG_ABLATE", 0);
  if (tuning_on ("GSTAMD_AGG_BX"))
    bx = tuning_int ("GSTAMD_AGG_BX", bx);
#else
  const int abl = 0;
#endif
#ifdef GSTAMD_TUNING
  const int rows = tuning_int ("GSTAMD_AGG_ROWS", 0);
  const int depth = tuning_int ("GSTAMD_AGG_DEPTH", 4);
  if (rows > 0 && q.fast && q.n_pads > 0 && q.bg_kind != 2 && abl == 0) {
    const int strips = (rw + 255) / 256;
    dim3 rgrid (strips * ((rh + rows - 1) / rows)), rblock (64);
#define AGG_ROWS_LAUNCH(S, D) hipLaunchKernelGGL ((k_aggregate_rows<S, D>), rgrid, rblock, 0, (hipStream_t) stream, q, (uint8_t *) dest, dstride, rx0, ry0, rw, rh, rows, strips)
    if (q.ashift == 0) {
      if (depth == 8) AGG_ROWS_LAUNCH (0, 8); else if (depth == 6) AGG_ROWS_LAUNCH (0, 6); else AGG_ROWS_LAUNCH (0, 4);
    } else {
      if (depth == 8) AGG_ROWS_LAUNCH (24, 8); else if (depth == 6) AGG_ROWS_LAUNCH (24, 6); else AGG_ROWS_LAUNCH (24, 4);
    }
#undef AGG_ROWS_LAUNCH
    return hipGetLastError () == hipSuccess ? GSTAMD_OK : GSTAMD_ERR_HIP;
  }
#endif
#ifdef GSTAMD_TUNING
  {
    int srows = tuning_int ("GSTAMD_AGG_STRIP_ROWS", 0);
    int spx = tuning_int ("GSTAMD_AGG_STRIP_PX", 4);
    for (int i = 0; i < q.n_pads; i++)
      if (q.pads[i].width < 8)
        spx = 4;
    if (srows > 0 && q.fast && q.n_pads > 0 && q.bg_kind != 2 && abl == 0) {
      const int strips = (rw + 64 * spx - 1) / (64 * spx);
      dim3 sgrid (strips * ((rh + srows - 1) / srows)), sblock (64);
#define STRIP_LAUNCH(S, N) hipLaunchKernelGGL ((k_aggregate_strip<S, 4, N>), sgrid, sblock, 0, (hipStream_t) stream, q, (uint8_t *) dest, dstride, rx0, ry0, rw, rh, srows, strips)
      if (q.ashift == 0) {
        if (spx == 8) STRIP_LAUNCH (0, 8); else STRIP_LAUNCH (0, 4);
      } else {
        if (spx == 8) STRIP_LAUNCH (24, 8); else STRIP_LAUNCH (24, 4);
      }
#undef STRIP_LAUNCH
      return hipGetLastError () == hipSuccess ? GSTAMD_OK : GSTAMD_ERR_HIP;
    }
  }
#endif
  if (q.fast && q.n_pads > 0 && abl == 0 && bx == 64 && rw >= 4 && (q.bg_kind != 2 || (rw & 3) == 0)) {
    dim3 dgrid ((rw + 255) / 256, rh);
#ifdef GSTAMD_TUNING
    const int nt = tuning_int ("GSTAMD_AGG_NT", AGG_DIRECT_NT);
#else
    const int nt = AGG_DIRECT_NT;
#endif
    if (om) {
#ifdef GSTAMD_TUNING
      const int cr = tuning_int ("GSTAMD_AGG_CULL_ROWS", AGG_CULL_ROWS);
#else
      const int cr = AGG_CULL_ROWS;     /* measured on C4's layout with opaque pads: 1 row 16.8 / 19.1 us (all_opaque / maps), 2 rows 16.5 / 19.05, 4 rows 17.25 / 18.7 */
#endif
      dim3 cgrid ((rw + 255) / 256, (rh + cr - 1) / cr);
#define AGG_CULL_LAUNCH_R(S, K, R) hipLaunchKernelGGL ((k_aggregate_direct_cull<S, K, AGG_DIRECT_NT, R>), cgrid, dim3 (64), 0, (hipStream_t) stream, q, *om, (uint8_t *) dest, dstride, rx0, ry0, rw, rh)
#define AGG_CULL_LAUNCH(S, K) do { if (cr == 1) AGG_CULL_LAUNCH_R (S, K, 1); else if (cr == 2) AGG_CULL_LAUNCH_R (S, K, 2); else AGG_CULL_LAUNCH_R (S, K, 4); } while (0)
      if (q.ashift == 0) {
        if (q.bg_kind == 2) AGG_CULL_LAUNCH (0, 1); else AGG_CULL_LAUNCH (0, 0);
      } else {
        if (q.bg_kind == 2) AGG_CULL_LAUNCH (24, 1); else AGG_CULL_LAUNCH (24, 0);
      }
#undef AGG_CULL_LAUNCH
#undef AGG_CULL_LAUNCH_R
      return hipGetLastError () == hipSuccess ? GSTAMD_OK : GSTAMD_ERR_HIP;
    }
#define AGG_DIRECT_LAUNCH(S, K) do { if (nt) hipLaunchKernelGGL ((k_aggregate_direct<S, K, 1>), dgrid, dim3 (64), 0, (hipStream_t) stream, q, (uint8_t *) dest, dstride, rx0, ry0, rw); \
      else hipLaunchKernelGGL ((k_aggregate_direct<S, K, 0>), dgrid, dim3 (64), 0, (hipStream_t) stream, q, (uint8_t *) dest, dstride, rx0, ry0, rw); } while (0)
    if (q.ashift == 0) {
      if (q.bg_kind == 2) AGG_DIRECT_LAUNCH (0, 1); else AGG_DIRECT_LAUNCH (0, 0);
    } else {
      if (q.bg_kind == 2) AGG_DIRECT_LAUNCH (24, 1); else AGG_DIRECT_LAUNCH (24, 0);
    }
#undef AGG_DIRECT_LAUNCH
    return hipGetLastError () == hipSuccess ? GSTAMD_OK : GSTAMD_ERR_HIP;
  }
  /* k_aggregate's packed span (aggregate_span4) forces the alpha byte of every pixel of the span once after the last pad.  On a continuation chunk
     (bg_kind 2: more pads than one launch takes) the canvas may hold alpha that a SOURCE pad of an earlier chunk lowered, on pixels no pad of this
     chunk touches - k_aggregate_direct<KEEP> keeps track of the touched pixels, this kernel does not, so a continuation that ends up here (a width
     that is not a multiple of four) takes the per-pixel form.  (Compositor fuzz, fresh seeds 7018 / 7022 / 7045 / 7085: 40 pads, only the alpha byte wrong.) */
  if (q.bg_kind == 2)
    q.fast = 0;
  dim3 grid ((lanes + bx - 1) / bx, rh), block (bx);
#define AGG_LAUNCH(A, S) hipLaunchKernelGGL ((k_aggregate<A, S>), grid, block, 0, (hipStream_t) stream, q, (uint8_t *) dest, dstride, rx0, ry0, rw, rh)
  if (q.ashift == 0) {
    switch (abl) {
#ifdef GSTAMD_TUNING
      case 1: AGG_LAUNCH (1, 0); break;
      case 2: AGG_LAUNCH (2, 0); break;
      case 3: AGG_LAUNCH (3, 0); break;
      case 4: AGG_LAUNCH (4, 0); break;
#endif
      default: AGG_LAUNCH (0, 0); break;
    }
  } else {
    switch (abl) {
#ifdef GSTAMD_TUNING
      case 1: AGG_LAUNCH (1, 24); break;
      case 2: AGG_LAUNCH (2, 24); break;
      case 3: AGG_LAUNCH (3, 24); break;
      case 4: AGG_LAUNCH (4, 24); break;
#endif
      default: AGG_LAUNCH (0, 24); break;
    }
  }
#undef AGG_LAUNCH
  return hipGetLastError () == hipSuccess ? GSTAMD_OK : GSTAMD_ERR_HIP;
}

static bool is_wide64 (int format)
{
  return format == GSTAMD_VIDEO_FORMAT_ARGB64 || format == GSTAMD_VIDEO_FORMAT_AYUV64;
}

static int wide64_alpha (double a)
{
  /* s_alpha = CLAMP ((gint) (src_alpha * G_MAXUINT16), 0, G_MAXUINT16) (blend.c:1202) */
  const int s = (int) (a * 65535);
  return s < 0 ? 0 : (s > 65535 ? 65535 : s);
}

static int launch64 (const Wide64Params &p, void *dest, int dstride, int rx0, int ry0, int rw, int rh, void *stream)
{
  if (rw <= 0 || rh <= 0)
    return GSTAMD_OK;
  hipLaunchKernelGGL (k_aggregate64, dim3 ((rw + 63) / 64, rh), dim3 (64), 0, (hipStream_t) stream, p, (uint8_t *) dest, dstride, rx0, ry0, rw);
  return hipGetLastError () == hipSuccess ? GSTAMD_OK : GSTAMD_ERR_HIP;
}

static uint64_t wide64_color (int c1, int c2, int c3)
{
  return 0xffffull | ((uint64_t) (c1 & 0xffff) << 16) | ((uint64_t) (c2 & 0xffff) << 32) | ((uint64_t) (c3 & 0xffff) << 48);
}

/* fill_color_* word (blend.c:218-240): GUINT32_FROM_BE ((0xff << A) | (c1 << C1) | (c2 << C2) | (c3 << C3)) */
static bool color_word (int format, int c1, int c2, int c3, uint32_t *out)
{
  int A, C1, C2, C3;
  switch (format) {
    case GSTAMD_VIDEO_FORMAT_ARGB: case GSTAMD_VIDEO_FORMAT_AYUV: A = 24; C1 = 16; C2 = 8; C3 = 0; break;
    case GSTAMD_VIDEO_FORMAT_BGRA: case GSTAMD_VIDEO_FORMAT_VUYA: A = 0; C1 = 8; C2 = 16; C3 = 24; break;     /* A32_COLOR (vuya, 0, 8, 16, 24) */
    case GSTAMD_VIDEO_FORMAT_ABGR: A = 24; C1 = 0; C2 = 8; C3 = 16; break;
    case GSTAMD_VIDEO_FORMAT_RGBA: A = 0; C1 = 24; C2 = 16; C3 = 8; break;
    default: return false;
  }
  const uint32_t be = (0xffu << A) | ((uint32_t) c1 << C1) | ((uint32_t) c2 << C2) | ((uint32_t) c3 << C3);
  *out = __builtin_bswap32 (be);
  return true;
}

__global__ __launch_bounds__ (256) void k_aggregate_plane (PlaneJob job)
{
  plane_word_body (job, (int) (blockIdx.x * blockDim.x + threadIdx.x) * 4, (int) blockIdx.y);
}

extern "C" {

int gstamd_compositor_blend (int format, int overlay, const void *src, int sw, int sh, int sstride, int xpos, int ypos,
    double src_alpha, void *dest, int dw, int dh, int dstride, int dst_y_start, int dst_y_end, int mode, void *stream)
{
  if (is_wide64 (format)) {
    if (!src || !dest)
      return GSTAMD_ERR_INVALID;
    const int a64 = wide64_alpha (src_alpha);
    if (a64 == 0)
      return GSTAMD_OK;
    if (dst_y_end > dh)
      dst_y_end = dh;
    const int wx0 = xpos < 0 ? 0 : xpos, wy0 = ypos < dst_y_start ? dst_y_start : ypos;
    const int wx1 = xpos + sw > dw ? dw : xpos + sw, wy1 = ypos + sh > dst_y_end ? dst_y_end : ypos + sh;
    Wide64Params w;
    memset (&w, 0, sizeof (w));
    w.overlay = overlay ? 1 : 0;
    w.bg_kind = 2;
    w.n_pads = 1;
    w.pads[0].data = (const uint8_t *) src;
    w.pads[0].width = sw;
    w.pads[0].height = sh;
    w.pads[0].stride = sstride;
    w.pads[0].xpos = xpos;
    w.pads[0].ypos = ypos;
    w.pads[0].s_alpha = a64;
    w.pads[0].mode = mode;
    return launch64 (w, dest, dstride, wx0, wy0, wx1 - wx0, wy1 - wy0, stream);
  }
  const int ashift = family_ashift (format);
  if (ashift < 0 || !src || !dest)
    return GSTAMD_ERR_INVALID;
  int s_alpha = (int) (src_alpha * 255);
  s_alpha = s_alpha < 0 ? 0 : (s_alpha > 255 ? 255 : s_alpha);
  if (s_alpha == 0)
    return GSTAMD_OK;                 /* "completely transparent... we just return" (blend.c:66-68) */
  if (dst_y_end > dh)
    dst_y_end = dh;
  /* clipped rectangle exactly as BLEND_A32 (blend.c:70-91) */
  int x0 = xpos < 0 ? 0 : xpos, y0 = ypos < dst_y_start ? dst_y_start : ypos;
  int x1 = xpos + sw > dw ? dw : xpos + sw, y1 = ypos + sh > dst_y_end ? dst_y_end : ypos + sh;
  AggregateParams p;
  memset (&p, 0, sizeof (p));
  p.ashift = ashift;
  p.overlay = overlay ? 1 : 0;
  p.bg_kind = 2;
  p.n_pads = 1;
  p.pads[0].data = (const uint8_t *) src;
  p.pads[0].width = sw;
  p.pads[0].height = sh;
  p.pads[0].stride = sstride;
  p.pads[0].xpos = xpos;
  p.pads[0].ypos = ypos;
  p.pads[0].s_alpha = s_alpha;
  p.pads[0].mode = mode;
  return launch (p, dest, dstride, x0, y0, x1 - x0, y1 - y0, stream);
}

int gstamd_compositor_fill_checker (int format, void *dest, int dw, int dh, int dstride, int y_start, int y_end, void *stream)
{
  if (is_wide64 (format) && dest) {
    Wide64Params w;
    memset (&w, 0, sizeof (w));
    w.bg_kind = 0;
    w.checker_yuv = format == GSTAMD_VIDEO_FORMAT_AYUV64;
    return launch64 (w, dest, dstride, 0, y_start, dw, y_end - y_start, stream);
  }
  const int ashift = family_ashift (format);
  if (ashift < 0 || !dest)
    return GSTAMD_ERR_INVALID;
  AggregateParams p;
  memset (&p, 0, sizeof (p));
  p.ashift = ashift;
  p.bg_kind = 0;
  p.checker_yuv = format == GSTAMD_VIDEO_FORMAT_AYUV || format == GSTAMD_VIDEO_FORMAT_VUYA;
  return launch (p, dest, dstride, 0, y_start, dw, y_end - y_start, stream);
}

int gstamd_compositor_fill_color (int format, void *dest, int dw, int dh, int dstride, int y_start, int y_end, int c1,
    int c2, int c3, void *stream)
{
  if (is_wide64 (format) && dest) {
    /* fill_color_argb64 (blend.c:1354-1384): the picture's pixels, 16-bit components */
    Wide64Params w;
    memset (&w, 0, sizeof (w));
    w.bg_kind = 1;
    w.bg_px = wide64_color (c1, c2, c3);
    return launch64 (w, dest, dstride, 0, y_start, dw, y_end - y_start, stream);
  }
  AggregateParams p;
  memset (&p, 0, sizeof (p));
  if (!dest || !color_word (format, c1, c2, c3, &p.bg_word))
    return GSTAMD_ERR_INVALID;
  p.ashift = family_ashift (format);
  p.bg_kind = 1;
  /* compositor_orc_splat_u32 over (y_end - y_start) * (stride / 4) words: row padding included */
  return launch (p, dest, dstride, 0, y_start, dstride / 4, y_end - y_start, stream);
}

static int aggregate_impl (int format, int background, const GstAmdCompositorPad *pads, const GstAmdCompositorPadOpacity *opacity, int n_pads, void *dest,
    int dw, int dh, int dstride, void *stream);

int gstamd_compositor_aggregate (int format, int background, const GstAmdCompositorPad *pads, int n_pads, void *dest,
    int dw, int dh, int dstride, void *stream)
{
  return aggregate_impl (format, background, pads, nullptr, n_pads, dest, dw, dh, dstride, stream);
}

int gstamd_compositor_aggregate_opaque (int format, int background, const GstAmdCompositorPad *pads, const GstAmdCompositorPadOpacity *opacity, int n_pads,
    void *dest, int dw, int dh, int dstride, void *stream)
{
  return aggregate_impl (format, background, pads, opacity, n_pads, dest, dw, dh, dstride, stream);
}

int gstamd_compositor_pad_opacity_map (int format, const void *data, int width, int height, int stride, uint64_t *map, void *stream)
{
  const int ashift = family_ashift (format);
  if (ashift < 0 || !data || !map || width <= 0 || height <= 0 || width > 4096)
    return GSTAMD_ERR_INVALID;
  if (hipMemsetAsync (map, 0, (size_t) height * sizeof (uint64_t), (hipStream_t) stream) != hipSuccess)
    return GSTAMD_ERR_HIP;
  hipLaunchKernelGGL (k_opacity_map, dim3 ((width + 255) / 256, height), dim3 (64), 0, (hipStream_t) stream, (const uint8_t *) data, width, stride, ashift,
      (unsigned long long *) map);
  return hipGetLastError () == hipSuccess ? GSTAMD_OK : GSTAMD_ERR_HIP;
}

static int aggregate_impl (int format, int background, const GstAmdCompositorPad *pads, const GstAmdCompositorPadOpacity *opacity, int n_pads, void *dest,
    int dw, int dh, int dstride, void *stream)
{
  if (is_wide64 (format)) {
    if (!dest || (n_pads > 0 && !pads))
      return GSTAMD_ERR_INVALID;
    Wide64Params w;
    memset (&w, 0, sizeof (w));
    w.overlay = background == GSTAMD_COMPOSITOR_BACKGROUND_TRANSPARENT;
    const bool yuv64 = format == GSTAMD_VIDEO_FORMAT_AYUV64;
    switch (background) {
      case GSTAMD_COMPOSITOR_BACKGROUND_CHECKER:
        w.bg_kind = 0;
        w.checker_yuv = yuv64;
        break;
      case GSTAMD_COMPOSITOR_BACKGROUND_BLACK:    /* gst_video_color_range_offsets on 16 bits (compositor.c:1131-1149) */
        w.bg_kind = 1;
        w.bg_px = yuv64 ? wide64_color (16 << 8, 128 << 8, 128 << 8) : wide64_color (0, 0, 0);
        break;
      case GSTAMD_COMPOSITOR_BACKGROUND_WHITE:
        w.bg_kind = 1;
        w.bg_px = yuv64 ? wide64_color (235 << 8, 128 << 8, 128 << 8) : wide64_color (65535, 65535, 65535);
        break;
      case GSTAMD_COMPOSITOR_BACKGROUND_TRANSPARENT:
        w.bg_kind = 1;
        w.bg_px = 0;
        break;
      default:
        return GSTAMD_ERR_INVALID;
    }
    int wdone = 0;
    bool wfirst = true;
    while (wfirst || wdone < n_pads) {
      w.n_pads = 0;
      while (wdone < n_pads && w.n_pads < GSTAMD_MAX_FUSED_PADS) {
        const GstAmdCompositorPad &in = pads[wdone++];
        const int a64 = wide64_alpha (in.alpha);
        if (a64 == 0 || !in.data)
          continue;
        PadDev &pd = w.pads[w.n_pads++];
        pd.data = (const uint8_t *) in.data;
        pd.width = in.width;
        pd.height = in.height;
        pd.stride = in.stride;
        pd.xpos = in.xpos;
        pd.ypos = in.ypos;
        pd.s_alpha = a64;
        pd.mode = in.blend_mode;
      }
      const int r = launch64 (w, dest, dstride, 0, 0, dw, dh, stream);
      if (r != GSTAMD_OK)
        return r;
      w.bg_kind = 2;
      wfirst = false;
    }
    return GSTAMD_OK;
  }
  const int ashift = family_ashift (format);
  if (ashift < 0 || !dest || (n_pads > 0 && !pads))
    return GSTAMD_ERR_INVALID;
  AggregateParams p;
  memset (&p, 0, sizeof (p));
  p.ashift = ashift;
  p.overlay = background == GSTAMD_COMPOSITOR_BACKGROUND_TRANSPARENT;
  const bool yuv = format == GSTAMD_VIDEO_FORMAT_AYUV || format == GSTAMD_VIDEO_FORMAT_VUYA;
  switch (background) {
    case GSTAMD_COMPOSITOR_BACKGROUND_CHECKER:
      p.bg_kind = 0;
      p.checker_yuv = yuv;
      break;
    case GSTAMD_COMPOSITOR_BACKGROUND_BLACK:      /* black/white_color: compositor.c:1133-1149 */
      p.bg_kind = 1;
      color_word (format, yuv ? 16 : 0, yuv ? 128 : 0, yuv ? 128 : 0, &p.bg_word);
      break;
    case GSTAMD_COMPOSITOR_BACKGROUND_WHITE:
      p.bg_kind = 1;
      color_word (format, yuv ? 235 : 255, yuv ? 128 : 255, yuv ? 128 : 255, &p.bg_word);
      break;
    case GSTAMD_COMPOSITOR_BACKGROUND_TRANSPARENT:
      p.bg_kind = 1;
      p.bg_word = 0;                              /* memset 0 (compositor.c:1641-1668) */
      break;
    default:
      return GSTAMD_ERR_INVALID;
  }
  int done = 0;
  bool first = true;
  OpacityMaps om;
  while (first || done < n_pads) {
    p.n_pads = 0;
    memset (&om, 0, sizeof (om));
    bool any_opaque = false;
    while (done < n_pads && p.n_pads < GSTAMD_MAX_FUSED_PADS) {
      const GstAmdCompositorPad &in = pads[done];
      const GstAmdCompositorPadOpacity *op = opacity ? &opacity[done] : nullptr;
      done++;
      int s_alpha = (int) (in.alpha * 255);
      s_alpha = s_alpha < 0 ? 0 : (s_alpha > 255 ? 255 : s_alpha);
      if (s_alpha == 0 || !in.data)
        continue;
      if (op && s_alpha == 255 && (op->all_opaque || (op->map && in.width <= 4096))) {
        om.map[p.n_pads] = (const unsigned long long *) op->map;
        if (op->all_opaque)
          om.all |= 1u << p.n_pads;
        any_opaque = true;
      }
      PadDev &pd = p.pads[p.n_pads++];
      pd.data = (const uint8_t *) in.data;
      pd.width = in.width;
      pd.height = in.height;
      pd.stride = in.stride;
      pd.xpos = in.xpos;
      pd.ypos = in.ypos;
      pd.s_alpha = s_alpha;
      pd.mode = in.blend_mode;
    }
    int r = launch (p, dest, dstride, 0, 0, dw, dh, stream, any_opaque ? &om : nullptr);
    if (r != GSTAMD_OK)
      return r;
    p.bg_kind = 2;                                /* further chunks continue on the canvas */
    first = false;
  }
  return GSTAMD_OK;
}

int gstamd_internal_pad_scaler (GstAmdVideoConverter *c, gstamd::ScaleDev *sh, gstamd::ScaleDev *sv, int *h_first, int *in_w, int *in_h,
    int *out_w, int *out_h, int *format);

int gstamd_internal_pad_scaler_tile_rows (GstAmdVideoConverter *c);

/* rows per tile: the first scaled pad decides (the pads of one canvas usually share a ratio) - see scaled_tile_rows_for */
static int scaled_tile_rows (GstAmdVideoConverter *c)
{
  const int e = tuning_int ("GSTAMD_SCALED_TILE_ROWS", 0);
  if (e >= 4 && e <= SCALED_TILE_H)
    return e;
  return gstamd_internal_pad_scaler_tile_rows (c);
}

int gstamd_compositor_pad_scaler_usable (GstAmdVideoConverter *convert)
{
  ScaleDev sh, sv;
  int hf, iw, ih, ow, oh, fmt;
  return gstamd_internal_pad_scaler (convert, &sh, &sv, &hf, &iw, &ih, &ow, &oh, &fmt) && family_ashift (fmt) >= 0;
}

int gstamd_internal_pad_walk (GstAmdVideoConverter *c, const uint32_t **vt, const uint32_t **ht, int *vbase, int *hbase);

static thread_local int g_last_scaled_kernel = 0;       /* 1: k_aggregate_walk, 2: k_aggregate_scaled, 3: no scaled pad (k_aggregate) */
extern "C" int gstamd_internal_last_scaled_kernel (void) { return g_last_scaled_kernel; }

/* The column walk (compositor_walk.h) when the pad set fits it: every scaled pad an exact halving with 8-tap passes, vertical first, no two
 * scaled pads overlapping on the canvas, at most 16 scaled and 16 unscaled pads.  1: launched, 0: not eligible, < 0: error */
static int try_walk (int format, int ashift, int background, const GstAmdCompositorScaledPad *pads, int n_pads, void *dest, int dw, int dh,
    int dstride, void *stream)
{
  if (tuning_on ("GSTAMD_NO_AGG_WALK"))
    return 0;
  if ((unsigned long long) dstride * (unsigned long long) dh >= 0x7fffffffull)           /* the canvas is addressed through a buffer resource */
    return 0;
  WalkParams p;
  memset ((void *) &p, 0, sizeof (p));
  p.ashift = ashift;
  p.overlay = background == GSTAMD_COMPOSITOR_BACKGROUND_TRANSPARENT;
  const bool yuv = format == GSTAMD_VIDEO_FORMAT_AYUV || format == GSTAMD_VIDEO_FORMAT_VUYA;
  switch (background) {
    case GSTAMD_COMPOSITOR_BACKGROUND_CHECKER: p.bg_kind = 0; p.checker_yuv = yuv; break;
    case GSTAMD_COMPOSITOR_BACKGROUND_BLACK: p.bg_kind = 1; color_word (format, yuv ? 16 : 0, yuv ? 128 : 0, yuv ? 128 : 0, &p.bg_word); break;
    case GSTAMD_COMPOSITOR_BACKGROUND_WHITE: p.bg_kind = 1; color_word (format, yuv ? 235 : 255, yuv ? 128 : 255, yuv ? 128 : 255, &p.bg_word); break;
    case GSTAMD_COMPOSITOR_BACKGROUND_TRANSPARENT: p.bg_kind = 1; p.bg_word = 0; break;
    default: return GSTAMD_ERR_INVALID;
  }
  bool fast = !p.overlay;
  int rows_pref = tuning_int ("GSTAMD_WALK_ROWS", 0);
  /* rows a wave walks.  One resident round is the aim (8 waves per SIMD x 1024 SIMDs when every blend is the opaque one, else 6): the strips of
     all scaled pads x the chunks of their rows should come to that many waves; a walk shorter than ~12 rows spends itself on priming its ring */
  int strips = 0, total_rows = 0, n_scaled = 0;
  for (int i = 0; i < n_pads; i++)
    if (pads[i].scaler && pads[i].data && pads[i].alpha * 255 >= 1) {
      ScaleDev sh, sv;
      int hf, iw, ih, ow = 0, oh = 0, fmt;
      if (gstamd_internal_pad_scaler (pads[i].scaler, &sh, &sv, &hf, &iw, &ih, &ow, &oh, &fmt)) {
        strips += (ow + GSTAMD_WALK_TILE - 1) / GSTAMD_WALK_TILE;
        total_rows += oh;
        n_scaled++;
      }
    }
  int walk_rows = 17;
  if (n_scaled > 0 && strips > 0) {
    const long long slots = 8192;
    const long long strip_rows = (long long) strips * (total_rows / n_scaled);          /* rows x strips to hand out */
    walk_rows = (int) ((strip_rows + slots - 1) / slots);
    walk_rows = walk_rows < 12 ? 12 : (walk_rows > 64 ? 64 : walk_rows);
  }
  long long covered = 0;
  for (int i = 0; i < n_pads; i++) {
    const GstAmdCompositorScaledPad &in = pads[i];
    int s_alpha = (int) (in.alpha * 255);
    s_alpha = s_alpha < 0 ? 0 : (s_alpha > 255 ? 255 : s_alpha);
    if (s_alpha == 0 || !in.data)
      continue;
    if (((uintptr_t) in.data % 4) != 0 || (in.stride % 4) != 0)
      return 0;
    if (!in.scaler) {
      if (p.n_plain >= GSTAMD_WALK_MAX_PADS)
        return 0;
      PadDev &q = p.plain[p.n_plain++];
      q.data = (const uint8_t *) in.data;
      q.width = in.width, q.height = in.height, q.stride = in.stride;
      q.xpos = in.xpos, q.ypos = in.ypos, q.s_alpha = s_alpha, q.mode = in.blend_mode;
      continue;
    }
    if (p.n_walk >= GSTAMD_WALK_MAX_PADS)
      return 0;
    ScaleDev sh, sv;
    int hf, iw, ih, ow, oh, fmt;
    if (!gstamd_internal_pad_scaler (in.scaler, &sh, &sv, &hf, &iw, &ih, &ow, &oh, &fmt) || fmt != format || iw != in.width || ih != in.height)
      return 0;
    WalkPad &w = p.walk[p.n_walk];
    if (!gstamd_internal_pad_walk (in.scaler, &w.vt, &w.ht, &w.vbase, &w.hbase))
      return 0;
    w.src = (const uint8_t *) in.data;
    w.sstride = in.stride, w.src_w = iw, w.src_h = ih;
    if ((unsigned long long) w.sstride * (unsigned long long) ih >= 0x7fffffffull)
      return 0;
    w.xpos = in.xpos, w.ypos = in.ypos, w.ow = ow, w.oh = oh;
    w.s_alpha = s_alpha, w.mode = in.blend_mode;
    w.n_below = p.n_plain;
    if (in.blend_mode == GSTAMD_COMPOSITOR_BLEND_MODE_SOURCE)
      fast = false;
    for (int k = 0; k < p.n_walk; k++) {          /* no two scaled pads on one canvas pixel */
      const WalkPad &o = p.walk[k];
      if (o.xpos < w.xpos + w.ow && w.xpos < o.xpos + o.ow && o.ypos < w.ypos + w.oh && w.ypos < o.ypos + o.oh)
        return 0;
    }
    w.tiles = (ow + GSTAMD_WALK_TILE - 1) / GSTAMD_WALK_TILE;
    w.tile_w = (ow + w.tiles - 1) / w.tiles;
    w.chunk_rows = rows_pref > 0 ? rows_pref : walk_rows;
    w.chunks = (oh + w.chunk_rows - 1) / w.chunk_rows;
    w.chunk_rows = (oh + w.chunks - 1) / w.chunks;
    w.first_block = p.n_blocks;
    p.n_blocks += w.tiles * w.chunks;
    const int cx0 = w.xpos > 0 ? w.xpos : 0, cy0 = w.ypos > 0 ? w.ypos : 0;
    const int cx1 = w.xpos + ow < dw ? w.xpos + ow : dw, cy1 = w.ypos + oh < dh ? w.ypos + oh : dh;
    if (cx1 > cx0 && cy1 > cy0)
      covered += (long long) (cx1 - cx0) * (cy1 - cy0);
    p.n_walk++;
  }
  if (!p.n_walk)
    return 0;
  if (covered < (long long) dw * dh) {
    p.fill_tiles_x = (dw + 63) / 64;
    p.fill_tiles_y = (dh + 15) / 16;
  }
  int walk_grid = p.n_blocks;
  /* Workgroups go round the eight XCDs; handing each XCD a contiguous run of strips lets neighbours meet in ONE L2: the 128-byte lines two strips straddle
     and part of the vertical halo are fetched once.  C4-A, round 6 (profiles/r06/walk_variants.txt): L2 -> fabric reads 198 -> 149 MB per frame (1.49 ->
     1.12 x the pads' bytes), the time unchanged (45.6 / 45.8 us: the walk is not bound by those bytes) - on, because the bytes it leaves on the fabric are
     another stream's */
  if (tuning_int ("GSTAMD_WALK_XCD", 1) > 0) {
    p.xcd_span = (p.n_blocks + 7) / 8;
    walk_grid = 8 * p.xcd_span;
  }
  const dim3 grid ((unsigned) (walk_grid + p.fill_tiles_x * p.fill_tiles_y)), block (64);
  uint8_t *d8 = (uint8_t *) dest;
#define WALK_LAUNCH(A, F, P) hipLaunchKernelGGL ((k_aggregate_walk<A, F, P>), grid, block, 0, (hipStream_t) stream, p, d8, dstride, dw, dh)
  const int variant = (ashift ? 4 : 0) | (fast ? 2 : 0) | (p.n_plain ? 1 : 0);
  switch (variant) {
    case 0: WALK_LAUNCH (0, 0, 0); break;
    case 1: WALK_LAUNCH (0, 0, 1); break;
    case 2: WALK_LAUNCH (0, 1, 0); break;
    case 3: WALK_LAUNCH (0, 1, 1); break;
    case 4: WALK_LAUNCH (24, 0, 0); break;
    case 5: WALK_LAUNCH (24, 0, 1); break;
    case 6: WALK_LAUNCH (24, 1, 0); break;
    default: WALK_LAUNCH (24, 1, 1); break;
  }
#undef WALK_LAUNCH
  return hipGetLastError () == hipSuccess ? 1 : GSTAMD_ERR_HIP;
}

int gstamd_compositor_aggregate_scaled (int format, int background, const GstAmdCompositorScaledPad *pads, int n_pads, void *dest,
    int dw, int dh, int dstride, void *stream)
{
  const int ashift = family_ashift (format);
  if (ashift < 0 || !dest || (n_pads > 0 && !pads) || dw <= 0 || dh <= 0)
    return GSTAMD_ERR_INVALID;
  bool any = false;
  for (int i = 0; i < n_pads; i++)
    any = any || pads[i].scaler != nullptr;
  g_last_scaled_kernel = any ? 2 : 3;
  if (any) {
    const int wr = try_walk (format, ashift, background, pads, n_pads, dest, dw, dh, dstride, stream);
    if (wr < 0)
      return wr;
    if (wr == 1) {
      g_last_scaled_kernel = 1;
      return GSTAMD_OK;
    }
  }
  if (!any) {                                     /* nothing to scale: the packed 4-pixel kernel */
    std::vector<GstAmdCompositorPad> plain ((size_t) n_pads);
    for (int i = 0; i < n_pads; i++) {
      plain[i].data = pads[i].data;
      plain[i].width = pads[i].width;
      plain[i].height = pads[i].height;
      plain[i].stride = pads[i].stride;
      plain[i].xpos = pads[i].xpos;
      plain[i].ypos = pads[i].ypos;
      plain[i].alpha = pads[i].alpha;
      plain[i].blend_mode = pads[i].blend_mode;
    }
    return gstamd_compositor_aggregate (format, background, plain.data (), n_pads, dest, dw, dh, dstride, stream);
  }
  ScaledAggParams p;
  memset ((void *) &p, 0, sizeof (p));
  p.ashift = ashift;
  p.overlay = background == GSTAMD_COMPOSITOR_BACKGROUND_TRANSPARENT;
  const bool yuv = format == GSTAMD_VIDEO_FORMAT_AYUV || format == GSTAMD_VIDEO_FORMAT_VUYA;
  switch (background) {
    case GSTAMD_COMPOSITOR_BACKGROUND_CHECKER:
      p.bg_kind = 0;
      p.checker_yuv = yuv;
      break;
    case GSTAMD_COMPOSITOR_BACKGROUND_BLACK:
      p.bg_kind = 1;
      color_word (format, yuv ? 16 : 0, yuv ? 128 : 0, yuv ? 128 : 0, &p.bg_word);
      break;
    case GSTAMD_COMPOSITOR_BACKGROUND_WHITE:
      p.bg_kind = 1;
      color_word (format, yuv ? 235 : 255, yuv ? 128 : 255, yuv ? 128 : 255, &p.bg_word);
      break;
    case GSTAMD_COMPOSITOR_BACKGROUND_TRANSPARENT:
      p.bg_kind = 1;
      p.bg_word = 0;
      break;
    default:
      return GSTAMD_ERR_INVALID;
  }
  int done = 0, th = SCALED_TILE_H;
  bool first = true, th_set = false;
  while (first || done < n_pads) {
    p.n_pads = 0;
    while (done < n_pads && p.n_pads < GSTAMD_MAX_SCALED_PADS) {
      const GstAmdCompositorScaledPad &in = pads[done++];
      int s_alpha = (int) (in.alpha * 255);
      s_alpha = s_alpha < 0 ? 0 : (s_alpha > 255 ? 255 : s_alpha);
      if (s_alpha == 0 || !in.data)
        continue;
      ScaledPadDev &sp = p.pads[p.n_pads];
      sp = ScaledPadDev ();
      sp.pad.data = (const uint8_t *) in.data;
      sp.pad.width = in.width;
      sp.pad.height = in.height;
      sp.pad.stride = in.stride;
      sp.pad.xpos = in.xpos;
      sp.pad.ypos = in.ypos;
      sp.pad.s_alpha = s_alpha;
      sp.pad.mode = in.blend_mode;
      sp.src_w = in.width;
      if (in.scaler) {
        int hf, iw, ih, ow, oh, fmt;
        if (!gstamd_internal_pad_scaler (in.scaler, &sp.sh, &sp.sv, &hf, &iw, &ih, &ow, &oh, &fmt) || fmt != format || iw != in.width ||
            ih != in.height || ((uintptr_t) in.data % 4) != 0 || (in.stride % 4) != 0)
          return GSTAMD_ERR_UNSUPPORTED;
        sp.h_first = hf;
        sp.n_pass = (sp.sh.kind != SCALE_NONE) + (sp.sv.kind != SCALE_NONE);
        sp.pad.width = ow;
        sp.pad.height = oh;
        if (!th_set) {
          th = scaled_tile_rows (in.scaler);
          th_set = true;
        }
      }
      p.n_pads++;
    }
    hipLaunchKernelGGL (k_aggregate_scaled, dim3 ((dw + SCALED_TILE_W - 1) / SCALED_TILE_W, (dh + th - 1) / th), dim3 (256), 0,
        (hipStream_t) stream, p, (uint8_t *) dest, dstride, dw, dh, th);
    if (hipGetLastError () != hipSuccess)
      return GSTAMD_ERR_HIP;
    p.bg_kind = 2;
    first = false;
  }
  return GSTAMD_OK;
}

/* ---- outputs without per-pixel alpha: plane-by-plane aggregation (compositor_planes.h) ------------------- */
static hipError_t launch_plane_job (const PlaneJob &job, hipStream_t stream)
{
  const int lanes = (job.wbytes + 3) / 4;
  hipLaunchKernelGGL (k_aggregate_plane, dim3 ((lanes + 255) / 256, job.rows), dim3 (256), 0, stream, job);
  return hipGetLastError ();
}

int gstamd_compositor_aggregate_frame (int format, int background, const int32_t black[3], const int32_t white[3],
    const GstAmdCompositorFramePad *pads, int n_pads, void *const dest[3], const int32_t dstride[3], int dw, int dh, void *stream)
{
  const FormatDesc *f = format_desc (format);
  PlaneGeom geom[3];
  const int n_planes = compositor_plane_geometry (f, geom);
  if (!n_planes || !dest || !dstride || dw <= 0 || dh <= 0 || (n_pads > 0 && !pads) || background < 0 || background > 3)
    return GSTAMD_ERR_INVALID;
  const int yuv = f->yuv ? 1 : 0;
  const int up = f->hi_depth ? hi_depth_bits (f->hi_depth) - 8 : 0;          /* deeper samples: the 8-bit defaults shifted up (limited range) */
  const int dblack[3] = {(yuv ? 16 : 0) << up, (yuv ? 128 : 0) << up, (yuv ? 128 : 0) << up};
  const int dwhite[3] = {(yuv ? 235 : 255) << up, (yuv ? 128 : 255) << up, (yuv ? 128 : 255) << up};
  const int bk[3] = {black ? black[0] : dblack[0], black ? black[1] : dblack[1], black ? black[2] : dblack[2]};
  const int wh[3] = {white ? white[0] : dwhite[0], white ? white[1] : dwhite[1], white ? white[2] : dwhite[2]};
  for (int pl = 0; pl < n_planes; pl++) {
    if (!dest[pl])
      return GSTAMD_ERR_INVALID;
    PlaneJob job;
    memset (&job, 0, sizeof (job));
    job.dst = (uint8_t *) dest[pl];
    job.dstride = dstride[pl];
    job.wbytes = compositor_plane_row_bytes (f, geom[pl], dw, background);
    job.rows = sub_scale (dh, geom[pl].h_sub);
    compositor_plane_background (f, geom[pl], pl, background, bk, wh, &job);
    int done = 0;
    bool first = true;
    while (first || done < n_pads) {
      job.n = 0;
      while (done < n_pads && job.n < GSTAMD_PLANE_MAX_PADS) {
        const GstAmdCompositorFramePad &in = pads[done++];
        FramePad fp;
        for (int k = 0; k < 3; k++) {
          fp.data[k] = (const uint8_t *) in.data[k];
          fp.stride[k] = in.stride[k];
        }
        fp.width = in.width;
        fp.height = in.height;
        fp.xpos = in.xpos;
        fp.ypos = in.ypos;
        fp.alpha = in.alpha;
        fp.mode = in.blend_mode;
        if (compositor_pad_rect (f, geom[pl], pl, fp, dw, dh, &job.r[job.n]))
          job.n++;
      }
      if (launch_plane_job (job, (hipStream_t) stream) != hipSuccess)
        return GSTAMD_ERR_HIP;
      job.bg_kind = 2;                            /* further chunks continue on the canvas */
      first = false;
    }
  }
  return GSTAMD_OK;
}

}  // extern "C"
