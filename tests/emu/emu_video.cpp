// tests/emu/emu_video.cpp - TEST INFRASTRUCTURE: runs the product's kernel BODIES
// (gstreamer_amd/csrc/video_device.h) over the launch grid on the host CPU, so the kernel logic and
// the planner can be checked against the oracle in this GPU-less container.  Not linked into the
// product, never used as a fallback.
#define GSTAMD_EMU_BOUNDS 1
#include <stdio.h>
#include <cstring>
#include <string>
#include <vector>
#include <memory>

#include "../../gstreamer_amd/csrc/planner.h"
#include "../../gstreamer_amd/csrc/video_device.h"
#include "../../gstreamer_amd/csrc/video_fast.h"
#include "../../gstreamer_amd/csrc/video_scale_fast.h"
#include "../../gstreamer_amd/csrc/video_hscale420.h"
#include "../../gstreamer_amd/csrc/video_scale420_fused.h"
#include "../../gstreamer_amd/csrc/video_scale_col.h"
#include "../../gstreamer_amd/csrc/video_422_fast.h"
#include <cstdlib>
#include <algorithm>
#include "../../gstreamer_amd/csrc/video_pack.h"
#include "../../gstreamer_amd/csrc/video_bilinear_fast.h"
#include "../../gstreamer_amd/csrc/video_bilinear_rows.h"
#include "../../gstreamer_amd/csrc/video_bilinear_half.h"
#include "../../gstreamer_amd/csrc/video_planes.h"
#include "../../gstreamer_amd/csrc/video_encode_fast.h"
#include "../../gstreamer_amd/csrc/video_deep.h"
#include "../../gstreamer_amd/csrc/video_deep_pack.h"
#include "../../gstreamer_amd/csrc/video_gamma.h"
#include "../../gstreamer_amd/csrc/video_dither.h"
#include "../../gstreamer_amd/csrc/video_dither_ed.h"
#include "../../gstreamer_amd/csrc/video_relayout.h"
#include "../../gstreamer_amd/csrc/video_swizzle34.h"
#include "../../gstreamer_amd/csrc/video_v210_fast.h"

using namespace gstamd;

template <class SRC>
static void run_hscale_lds (const SRC &src, const ScaleDev &sd, const Dst &d, int out_w, int out_h)
{
  std::vector<uint32_t> lds (12288);
  for (int y = 0; y < out_h; y++)
    for (int t0 = 0; t0 < out_w; t0 += 256) {
      const int t1 = t0 + 256 < out_w ? t0 + 256 : out_w;
      int lo, hi;
      hscale_span (sd, t0, t1, &lo, &hi);
      for (int tid = 0; tid < 256; tid++)
        hscale_stage<SRC> (src, lds.data (), lo, hi, y, tid, 256);
      for (int x = t0; x < t1; x++)
        hscale_from_lds (lds.data (), lo, sd, d, x, y);
    }
}

static FastParams emu_fast_params (const VideoPlan &p, bool rgb24 = false)      /* make_fast_params of capi_video.cpp */
{
  FastParams fp;
  fp.width = p.front.width;
  fp.height = p.front.height;
  fast_params_finish (fp, p.matrix.p, p.post.pack_pos, p.front.u_plane);
  /* with a source crop the chroma upsampler still sees the frame's rows above / below the crop (do_unpack_lines :2966) */
  fp.crow_lo = -(p.rect.in_y >> 1);
  fp.crow_hi = ((p.rect.in_maxh + 1) >> 1) - 1 - (p.rect.in_y >> 1);
  if (rgb24)
    fast_params_rgb24 (fp, p.matrix.p, p.fout->pos, p.front.u_plane);
  return fp;
}

static int emu_packed_ok (const SrcFront &s)           /* front_packed_ok of video_kernels.hip */
{
  return s.vec_ok && s.f.w_sub == 1 && kind_has_planes (s.f.kind) && s.pre.matrix.kind == MATRIX_NONE && s.pre.alpha_kind == ALPHA_NONE;
}
static int emu_packed_ok (const SrcImage &) { return 1; }
static int emu_packed_ok (const SrcLean &) { return 0; }

// k_hscale_wave: one wave per tile, staging phase for all 64 lanes, then the filter phase
template <class SRC>
static void run_hscale_wave (const SRC &src, const ScaleDev &sd, const Dst &d, const PostFast &pf, int out_w, int out_h, TileGeom g)
{
  std::vector<uint32_t> lds (g.lds_px);
  for (int y = 0; y < out_h; y++)
    for (int t0 = 0; t0 < out_w; t0 += g.tile_w) {
      const int t1 = t0 + g.tile_w < out_w ? t0 + g.tile_w : out_w;
      int lo, hi;
      hscale_span (sd, t0, t1, &lo, &hi);
      const int xa = lo & ~7;
      for (int lane = 0; lane < 64; lane++)
        tile_stage_row (src, lds.data (), xa, hi, y, lane, emu_packed_ok (src));
      for (int lane = 0; lane < 64; lane++)
        hscale_tile_lane (lds.data (), xa, sd, d, pf, t0, t1, y, lane);
    }
}

// k_hscale_dot4_wave
static void run_hscale_dot4 (const SrcFront &src, const ScaleDev &sd, const Dst &d, const PostFast &pf, int out_w, int out_h, TileGeom g)
{
  const int plane_w = (g.lds_px + 8) / 4;
  std::vector<uint32_t> lds (3 * plane_w);
  uint32_t *py = lds.data (), *pu = py + plane_w, *pv = pu + plane_w;
  for (int y = 0; y < out_h; y++)
    for (int t0 = 0; t0 < out_w; t0 += g.tile_w) {
      const int t1 = t0 + g.tile_w < out_w ? t0 + g.tile_w : out_w;
      int lo, hi;
      hscale_span (sd, t0, t1, &lo, &hi);
      const int xa = lo & ~7;
      for (int lane = 0; lane < 64; lane++)
        tile_stage_row_planes (src, py, pu, pv, xa, hi, y, lane, emu_packed_ok (src));
      for (int lane = 0; lane < 64; lane++) {
        if (sd.nw == 5) {
          Dot4Taps<5> ft;
          hscale_dot4_fetch<5> (sd, xa, t0, t1, lane, ft);
          hscale_dot4_lane<5> (py, pu, pv, ft, sd, sd.nw, d, pf, t0, t1, y, lane);
        } else {
          Dot4Taps<0> ft;
          hscale_dot4_fetch<0> (sd, xa, t0, t1, lane, ft);
          hscale_dot4_lane<0> (py, pu, pv, ft, sd, sd.nw, d, pf, t0, t1, y, lane);
        }
      }
    }
}

// k_hscale420_dot4: per-lane chroma cache carried down the lines of a tile
static int g_h420_runs = 0;
static int emu_h420_rows ()
{
  const char *e = getenv ("GSTAMD_H420_ROWS");
  return e ? atoi (e) : 4;
}
static bool emu_h420_ok (const SrcFront &s)
{
  const FrontParams &f = s.f;
  auto al = [](const void *p, int a) { return ((uintptr_t) p % a) == 0; };
  if (!kind_has_planes (f.kind) || f.w_sub != 1 || (f.width % 16) != 0 || f.chroma_v2 == 2 || emu_h420_rows () <= 0)
    return false;
  bool ok = al (s.pl.p[0], 16) && (s.pl.stride[0] % 16) == 0;
  if (f.kind == UNPACK_SEMI)
    ok = ok && al (s.pl.p[1], 16) && (s.pl.stride[1] % 16) == 0;
  else
    ok = ok && al (s.pl.p[1], 8) && al (s.pl.p[2], 8) && (s.pl.stride[1] % 8) == 0 && (s.pl.stride[2] % 8) == 0;
  return ok;
}
static bool emu_h420_ok (const SrcImage &) { return false; }
static void run_hscale420 (const SrcImage &, const ScaleDev &, const Dst &, const PostFast &, int, int, TileGeom) {}
static void run_hscale420 (const SrcFront &src, const ScaleDev &sd, const Dst &d, const PostFast &pf, int out_w, int out_h, TileGeom g)
{
  g_h420_runs++;
  const int plane_w = GSTAMD_H420_PLANE_BYTES / 4, rpw = emu_h420_rows ();
  std::vector<uint32_t> lds (3 * plane_w + 4);
  /* 16-byte aligned planes like the LDS allocation */
  uint32_t *py = (uint32_t *) (((uintptr_t) lds.data () + 15) & ~(uintptr_t) 15), *pu = py + plane_w, *pv = pu + plane_w;
  for (int y0 = 0; y0 < out_h; y0 += rpw)
    for (int t0 = 0; t0 < out_w; t0 += g.tile16_w) {
      const int t1 = t0 + g.tile16_w < out_w ? t0 + g.tile16_w : out_w;
      int lo, hi;
      hscale_span (sd, t0, t1, &lo, &hi);
      const int xa = lo & ~15;
      std::vector<H420State> c (64);
      const int y1 = std::min (y0 + rpw, out_h);
      for (int lane = 0; lane < 64; lane++)
        h420_begin (src, c[lane], xa, hi, y0, lane);
      for (int y = y0; y < y1; y++) {
        for (int lane = 0; lane < 64; lane++)
          h420_stage_line_any (src, c[lane], py, pu, pv, xa, hi, y, y + 1 < y1 ? y + 1 : -1, lane);
        for (int lane = 0; lane < 64; lane++) {
          if (sd.nw == 5) {
            Dot4Taps<5> ft;
            hscale_dot4_fetch<5> (sd, xa, t0, t1, lane, ft);
            hscale_dot4_lane<5> (py, pu, pv, ft, sd, sd.nw, d, pf, t0, t1, y, lane);
          } else {
            Dot4Taps<0> ft;
            hscale_dot4_fetch<0> (sd, xa, t0, t1, lane, ft);
            hscale_dot4_lane<0> (py, pu, pv, ft, sd, sd.nw, d, pf, t0, t1, y, lane);
          }
        }
      }
    }
}

// k_hscale420_reg: the decision of capi_video.cpp (closed-form pairing) + launch_hscale420_reg (alignment, window words), then the
// kernel's loop with per-lane register state
static int g_h420_reg_runs = 0;
template <int NW, int CH, int SEMI>
static void run_h420_reg (H420RegParams p, int n_taps)
{
  std::vector<uint32_t> ldsv (2 * GSTAMD_H420_LINE_WORDS + 4);
  uint32_t *lds = (uint32_t *) (((uintptr_t) ldsv.data () + 15) & ~(uintptr_t) 15);
  const int pairs = p.height / 2 + 1, ppw = p.lines_per_wave / 2;
  struct LaneState { uint32_t P[8], Q[8]; H420Pair cur, nxt; Dot4Taps<NW> ft; int x0; };
  for (int b = 0; b * ppw < pairs; b++)
    for (int t0 = 0; t0 < p.out_w; t0 += p.tile_w) {
      const int t1 = std::min (t0 + p.tile_w, p.out_w);
      const int u0 = b * ppw, u1 = std::min (u0 + ppw, pairs);
      int x_lo, x_hi;
      h420r_span (p, n_taps, t0, t1, &x_lo, &x_hi);
      const int xa = x_lo & ~15;
      std::vector<LaneState> L (64);
      for (int lane = 0; lane < 64; lane++) {
        LaneState &s = L[lane];
        s.x0 = xa + 16 * lane;
        if (s.x0 + 16 > p.width)
          s.x0 = p.width - 16;
        h420r_fetch_taps<NW> (p, xa, t0, t1, lane, s.ft);
        H420Raw r;
        h420r_load_raw<SEMI> (p, h420r_crow (p, u0 - 1), s.x0 >> 1, r);
        h420_filter_raw2<CH> (SEMI != 0, p.u_first != 0, r, s.P);
        h420r_request<SEMI> (p, u0, s.x0, s.cur);
      }
      for (int u = u0; u < u1; u += 2) {
        for (int lane = 0; lane < 64; lane++) {
          h420r_stage_pair<CH, SEMI> (p, L[lane].cur, L[lane].P, L[lane].Q, lds, 4 * lane);
          h420r_request<SEMI> (p, u + 1 < u1 ? u + 1 : u1 - 1, L[lane].x0, L[lane].nxt);
        }
        for (int lane = 0; lane < 64; lane++)
          h420r_filter_pair<NW> (p, lds, L[lane].ft, u, t0, t1, lane);
        if (u + 1 >= u1)
          break;
        for (int lane = 0; lane < 64; lane++) {
          h420r_stage_pair<CH, SEMI> (p, L[lane].nxt, L[lane].Q, L[lane].P, lds, 4 * lane);
          h420r_request<SEMI> (p, u + 2 < u1 ? u + 2 : u1 - 1, L[lane].x0, L[lane].cur);
        }
        for (int lane = 0; lane < 64; lane++)
          h420r_filter_pair<NW> (p, lds, L[lane].ft, u + 1, t0, t1, lane);
      }
    }
}

template <int NW, int SEMI>
static void run_h420_reg_ch2 (const H420RegParams &p, int chroma_h, int n_taps)
{
  if (chroma_h == CHROMA_H_H2_CS)
    run_h420_reg<NW, CHROMA_H_H2_CS, SEMI> (p, n_taps);
  else if (chroma_h == CHROMA_H_H2)
    run_h420_reg<NW, CHROMA_H_H2, SEMI> (p, n_taps);
  else
    run_h420_reg<NW, CHROMA_H_NONE, SEMI> (p, n_taps);
}
template <int NW>
static void run_h420_reg_ch (const H420RegParams &p, int chroma_h, int n_taps)
{
  if (p.semi)
    run_h420_reg_ch2<NW, 1> (p, chroma_h, n_taps);
  else
    run_h420_reg_ch2<NW, 0> (p, chroma_h, n_taps);
}

// ---- k_scale_col (video_col_kernels.hip): video_scale_col.h's col_wave with the 64 lanes of a wave run one after the other, the waves of a
// workgroup from the bottom one up (a wave's hand-over copy exists before the wave above asks for it - on the device a flag says so) ----
static int g_col_runs = 0, g_col_regwin = 0;
extern "C" int emu_col_runs (void) { return g_col_runs; }
extern "C" int emu_col_regwin_groups (void) { return g_col_regwin; }          /* line groups filtered from register windows (col_hfilter_regs) */
// quad_grid_find over a whole grid: 1 when every (job, workgroup) pair comes up exactly once
extern "C" int emu_quad_grid_check (int n0, int n1, int n2)
{
  QuadGrid g;
  memset (&g, 0, sizeof (g));
  g.n[0] = n0, g.n[1] = n1, g.n[2] = n2;
  std::vector<int> seen[3] = {std::vector<int> (n0, 0), std::vector<int> (n1, 0), std::vector<int> (n2, 0)};
  for (int b = 0; b < n0 + n1 + n2; b++) {
    int k = -1, local = -1;
    quad_grid_find (g, b, &k, &local);
    if (k < 0 || k > 2 || local < 0 || local >= g.n[k] || seen[k][local]++)
      return 0;
  }
  return 1;
}
static int g_emu_deep16_runs = 0;
static int g_emu_pack4_runs = 0;
extern "C" int emu_pack4_runs (void) { return g_emu_pack4_runs; }
static int g_emu_enc16_runs = 0;
extern "C" int emu_enc16_runs (void) { return g_emu_enc16_runs; }
static int g_emu_quad_runs = 0, g_emu_quad_modes = 0;
extern "C" int emu_quad_modes (void) { const int m = g_emu_quad_modes; g_emu_quad_modes = 0; return m; }
extern "C" int emu_quad_runs (void) { return g_emu_quad_runs; }
extern "C" int emu_deep16_runs (void) { return g_emu_deep16_runs; }

template <int OPL, int NW, int NGV>
struct ColExecEmu {
  ColLane<OPL, NW> L[64];
  ColRaw<OPL> ra[64];
  ColRingRegs<OPL, NGV> rg[64];
  template <class F> void each (F f) { for (int lane = 0; lane < 64; lane++) f (lane, L[lane], ra[lane], rg[lane]); }
  // register windows: the word [k][c][j] of the lane `dist` places up (the device shifts `v` across the lanes: v_mov_b32_dpp wave_shl:1, zero into the last lane)
  struct Nb {
    ColExecEmu *x;
    int lane;
    uint32_t operator() (uint32_t, int k, int c, int j, int dist) const { return lane + dist < 64 ? x->L[lane + dist].win[k][c][j] : 0u; }
  };
  template <class F> void each_nb (F f) { g_col_regwin++; for (int lane = 0; lane < 64; lane++) { Nb nb = {this, lane}; f (lane, L[lane], ra[lane], rg[lane], nb); } }
  template <class F> bool all (F pred) { bool r = true; for (int lane = 0; lane < 64; lane++) r = pred (lane, L[lane]) && r; return r; }
  void sync () {}
  void publish (uint32_t *flags, int wave) { flags[wave] = 1u; }
  void wait_flag (uint32_t *flags, int wave)
  {
    if (!flags[wave]) {
      fprintf (stderr, "emu k_scale_col: wave %d asks for a hand-over copy that was never published\n", wave);
      abort ();
    }
  }
};

template <int OPL, int NW, int NGV, int CH, int SEMI, int WSTEP, int A8, int POST>
static void run_scale_col (const ColParams &p, const ColSrc &src, const Dst &dst, const PostFast &pf, int nwaves)
{
  const size_t wave_bytes = col_wave_bytes (OPL, NGV, p.pubn, OPL == 2 && WSTEP == 1 && A8 == 2);
  std::vector<uint32_t> lds_words ((GSTAMD_COL_FLAG_BYTES + (size_t) (nwaves + 1) * wave_bytes) / 4 + 16);
  uint8_t *lds = (uint8_t *) lds_words.data ();
  for (int chunk = 0; chunk < p.n_chunks; chunk++)
    for (int ti = 0; ti < p.n_tiles; ti++) {
      memset (lds, 0xa5, lds_words.size () * 4);          /* LDS holds whatever the last workgroup left: no result may depend on it */
      uint32_t *flags = (uint32_t *) lds;
      for (int w = 0; w < nwaves; w++)
        flags[w] = 0;
      for (int wave = nwaves - 1; wave >= 0; wave--) {
        ColWavePlan wp;
        if (!col_wave_plan (p, chunk, wave, &wp))
          continue;
        uint8_t *mine = lds + GSTAMD_COL_FLAG_BYTES + (size_t) wave * wave_bytes;
        auto x = std::make_unique<ColExecEmu<OPL, NW, NGV>> ();
        memset ((void *) x->rg, 0x5a, sizeof (x->rg));
        col_wave<OPL, NW, NGV, CH, SEMI, WSTEP, A8, POST> (*x, p, src, p.tiles + 4 * ti, wp, mine, mine + wave_bytes, flags, wave, dst, pf);
      }
    }
}

template <int OPL, int NW, int NGV, int WSTEP, int A8, int POST>
static void run_scale_col_post (const ColParams &p, int chroma_h, int semi, const ColSrc &src, const Dst &dst, const PostFast &pf, int nwaves)
{
  if (chroma_h == CHROMA_H_H2) {
    if (semi) run_scale_col<OPL, NW, NGV, CHROMA_H_H2, 1, WSTEP, A8, POST> (p, src, dst, pf, nwaves);
    else run_scale_col<OPL, NW, NGV, CHROMA_H_H2, 0, WSTEP, A8, POST> (p, src, dst, pf, nwaves);
  } else {
    if (semi) run_scale_col<OPL, NW, NGV, CHROMA_H_H2_CS, 1, WSTEP, A8, POST> (p, src, dst, pf, nwaves);
    else run_scale_col<OPL, NW, NGV, CHROMA_H_H2_CS, 0, WSTEP, A8, POST> (p, src, dst, pf, nwaves);
  }
}

template <int OPL, int NW, int NGV, int WSTEP, int A8>
static void run_scale_col_src (const ColParams &p, int chroma_h, int semi, const ColSrc &src, const Dst &dst, const PostFast &pf, int nwaves)
{
  if (pf.use)
    run_scale_col_post<OPL, NW, NGV, WSTEP, A8, 1> (p, chroma_h, semi, src, dst, pf, nwaves);
  else
    run_scale_col_post<OPL, NW, NGV, WSTEP, A8, 0> (p, chroma_h, semi, src, dst, pf, nwaves);
}

// the decision of capi_video.cpp (build_tables + convert_to_packed) for k_scale_col
static bool emu_scale_col (const VideoPlan &p, const SrcFront &sf, const Dst &dst, const PostFast &pf)
{
  int lo, hi;
  if (getenv ("GSTAMD_NO_COL") || !col_plan_regular (p, &lo, &hi) || sf.pre.matrix.kind != MATRIX_NONE || sf.pre.alpha_kind != ALPHA_NONE)
    return false;
  const Planes &pl = sf.pl;
  const bool semi = p.front.kind == UNPACK_SEMI;
  if (!semi && pl.stride[p.front.u_plane] != pl.stride[p.front.v_plane])
    return false;
  ColTables t;
  ColForm f;
  const char *eo = getenv ("GSTAMD_COL_OPL"), *es = getenv ("GSTAMD_COL_SHARE");
  if (!col_choose (p.passes[0], p.passes[1], p.front.width, p.front.height, eo ? atoi (eo) : 0, !(es && atoi (es) == 0), &t, &f, !getenv ("GSTAMD_COL_NO_REGWIN")))
    return false;
  if (((uintptr_t) dst.p % (4 * f.opl)) != 0 || (dst.stride % (4 * f.opl)) != 0)
    return false;
  const char *ew = getenv ("GSTAMD_COL_WAVES"), *er = getenv ("GSTAMD_COL_ROWS");
  const int nwaves = ew && atoi (ew) > 0 ? std::min (atoi (ew), GSTAMD_COL_MAX_WAVES) : 3;
  ColParams q;
  memset ((void *) &q, 0, sizeof (q));
  q.ystride = pl.stride[0];
  q.cstride = semi ? pl.stride[1] : pl.stride[p.front.u_plane];
  q.width = p.front.width;
  q.height = p.front.height;
  q.u_first = p.front.u_plane != 0;
  q.crow_lo = lo;
  q.crow_hi = hi;
  q.tiles = t.tiles.data ();
  q.hout = t.hout.data ();
  q.vrow = t.vrow.data ();
  q.out_w = p.passes[0].out_size;
  q.out_h = p.passes[1].out_size;
  q.n_tiles = (int) t.tiles.size () / 4;
  q.rows_per_wave = std::max (t.min_rows_per_wave, er && atoi (er) > 0 ? atoi (er) : 5);
  q.nwaves = nwaves;
  q.rows_last = col_rows_last (t, q.rows_per_wave, q.out_h);
  q.rows_per_wg = q.rows_per_wave * (nwaves - 1) + q.rows_last;
  q.n_chunks = (q.out_h + q.rows_per_wg - 1) / q.rows_per_wg;
  q.pubn = t.pubn;
  q.dstride = dst.stride;
  const long long crow_bytes = (long long) q.cstride * (hi - lo) + (semi ? q.width : q.width / 2);
  ColSrc src;
  src.y = col_plane (pl.p[0], 0, (uint32_t) (q.ystride * (q.height - 1) + q.width));
  src.c0 = col_plane (semi ? pl.p[1] : pl.p[p.front.u_plane], (long long) lo * q.cstride, (uint32_t) crow_bytes);
  src.c1 = col_plane (semi ? pl.p[1] : pl.p[p.front.v_plane], (long long) lo * q.cstride, (uint32_t) crow_bytes);
  src.out = col_plane (dst.p, 0, (uint32_t) (dst.stride * (q.out_h - 1) + 4 * q.out_w));
  g_col_runs++;
#define V(o, n, g, w, a) if (f.opl == o && f.nw == n && f.ngv == g && f.wstep == w && f.a8 == a) { run_scale_col_src<o, n, g, w, a> (q, p.front.chroma_h, semi, src, dst, pf, nwaves); return true; }
  GSTAMD_COL_FORMS (V)
#undef V
  g_col_runs--;
  return false;
}

static bool emu_scale420_fused (const VideoPlan &p, H420RegParams hp, int nw, int n_taps_h, const Dst &dst, const PostFast &pf);
// returns 1: the horizontal pass went to tmp, 2: the fused kernel rendered the final image, 0: not applicable
static int emu_hscale420_reg (const VideoPlan &p, const SrcFront &sf, const ScaleDev &sd0, uint8_t *tmp, int tmp_w, const Dst *final_dst, const PostFast *pf)
{
  if (getenv ("GSTAMD_NO_H420_REG"))
    return false;
  const TileGeom g = p.passes[0].horizontal ? pass_tile_geom (p.passes[0]) : TileGeom {0, 0, 0};
  if (!(p.passes.size () == 2 && p.passes[0].horizontal && p.passes[0].kind == SCALE_NTAP && p.passes[0].dot4_ok && g.tile16_w > 0 &&
      p.front.chroma_v2 == 1 && kind_has_planes (p.front.kind) && p.front.w_sub == 1 && p.front.h_sub == 1 && !p.matrix_before_scale &&
      (int) p.vpair.size () >= 2 * p.front.height))
    return false;
  const int lo = -(p.rect.in_y >> 1), hi = ((p.rect.in_maxh + 1) >> 1) - 1 - (p.rect.in_y >> 1);
  for (int y = 0; y < p.front.height; y++) {
    int heavy, light;
    h420r_rows (lo, hi, y, &heavy, &light);
    const int e0 = p.vpair[2 * y], ta = vpair_row (e0), tb = p.vpair[2 * y + 1];
    const int th = vpair_role (e0) == 0 ? ta : tb, tl = vpair_role (e0) == 0 ? tb : ta;
    if (th != heavy || tl != light)
      return false;
  }
  const Planes &pl = sf.pl;
  const bool semi = p.front.kind == UNPACK_SEMI;
  if (!semi && pl.stride[p.front.u_plane] != pl.stride[p.front.v_plane])
    return false;
  H420RegParams hp;
  memset (&hp, 0, sizeof (hp));
  hp.y = pl.p[0];
  hp.ystride = pl.stride[0];
  hp.semi = semi;
  hp.u_first = p.front.u_plane != 0;
  hp.c0 = semi ? pl.p[1] : pl.p[p.front.u_plane];
  hp.c1 = semi ? pl.p[1] : pl.p[p.front.v_plane];
  hp.cstride = semi ? pl.stride[1] : pl.stride[p.front.u_plane];
  hp.width = p.front.width;
  hp.height = p.front.height;
  hp.crow_lo = lo;
  hp.crow_hi = hi;
  hp.offset = sd0.offset;
  hp.tapw = sd0.tapw;
  hp.nw4 = sd0.nw4;
  hp.dst = tmp;
  hp.dstride = tmp_w * 4;
  hp.out_w = tmp_w;
  hp.tile_w = g.tile16_w;
  auto al = [](const void *q, int a) { return ((uintptr_t) q % a) == 0; };
  bool ok = (hp.width % 16) == 0 && al (hp.y, 16) && (hp.ystride % 16) == 0;
  ok = ok && (semi ? (al (hp.c0, 16) && (hp.cstride % 16) == 0) : (al (hp.c0, 8) && al (hp.c1, 8) && (hp.cstride % 8) == 0));
  if (!ok || sd0.nw < 3 || sd0.nw > 5)
    return false;
  if (final_dst && emu_scale420_fused (p, hp, sd0.nw, sd0.n_taps, *final_dst, *pf))
    return 2;
  const char *e = getenv ("GSTAMD_H420_ROWS");
  int lpw = e && atoi (e) > 0 ? atoi (e) : 12;
  hp.lines_per_wave = std::max (4, (lpw + 1) & ~1);
  g_h420_reg_runs++;
  if (sd0.nw == 3)
    run_h420_reg_ch<3> (hp, p.front.chroma_h, sd0.n_taps);
  else if (sd0.nw == 4)
    run_h420_reg_ch<4> (hp, p.front.chroma_h, sd0.n_taps);
  else
    run_h420_reg_ch<5> (hp, p.front.chroma_h, sd0.n_taps);
  return 1;
}

// k_scale420_fused (video_fused_kernels.hip): the same phases in the same order, waves one after the other between barriers
static int g_fused_runs = 0;
extern "C" int emu_fused_runs (void) { return g_fused_runs; }

template <int NW, int CH, int SEMI>
static void run_fused420 (const Fused420Params &p, const Dst &dst, const PostFast &pf, int nwaves)
{
  const int tiles = (p.h.out_w + p.h.tile_w - 1) / p.h.tile_w, chunks = (p.out_h + p.rows_per_chunk - 1) / p.rows_per_chunk;
  const int stage_lines = p.sched == 2 ? 1 : 2;
  std::vector<uint32_t> lds ((size_t) p.ring * GSTAMD_FUSED_GROUP_WORDS + (size_t) nwaves * stage_lines * GSTAMD_H420_LINE_WORDS + 4);
  uint32_t *base = (uint32_t *) (((uintptr_t) lds.data () + 15) & ~(uintptr_t) 15);
  std::vector<Fused420Lane<NW>> L ((size_t) nwaves * 64);
  std::vector<int> gnext (nwaves);
  for (int by = 0; by < chunks; by++)
    for (int bx = 0; bx < tiles; bx++) {
      memset (base, 0xAB, ((size_t) p.ring * GSTAMD_FUSED_GROUP_WORDS + (size_t) nwaves * stage_lines * GSTAMD_H420_LINE_WORDS) * 4);    /* LDS starts as garbage */
      uint32_t *ring = base;
      const int t0 = bx * p.h.tile_w, t1 = std::min (t0 + p.h.tile_w, p.h.out_w);
      const int j0 = by * p.rows_per_chunk, j1 = std::min (j0 + p.rows_per_chunk, p.out_h);
      int x_lo, x_hi;
      h420r_span (p.h, p.n_taps_h, t0, t1, &x_lo, &x_hi);
      const int xa = x_lo & ~15;
      int gl, g_last;
      fused_round_groups (p, j0, j1 - 1, &gl, &g_last);
      for (int w = 0; w < nwaves; w++) {
        gnext[w] = gl + w;
        for (int lane = 0; lane < 64; lane++) {
          Fused420Lane<NW> &s = L[(size_t) w * 64 + lane];
          s.x0 = xa + 16 * lane;
          if (s.x0 + 16 > p.h.width)
            s.x0 = p.h.width - 16;
          h420r_fetch_taps<NW> (p.h, xa, t0, t1, lane, s.ft);
          fused_request_group<NW, SEMI> (p.h, std::min (gnext[w], g_last), s);
        }
      }
      if (p.sched == 2) {
        /* k_scale420_fused2: between two barriers every wave filters its row of round k and produces its groups of round k + 1 (odd
           waves the other way round); the waves run one after the other here, which is the worst order for a ring that is too short */
        auto produce = [&](int w, int gh) {
          uint32_t *stage = base + (size_t) p.ring * GSTAMD_FUSED_GROUP_WORDS + (size_t) w * GSTAMD_H420_LINE_WORDS;
          Fused420Lane<NW> *Lw = &L[(size_t) w * 64];
          int &g = gnext[w];
          while (g <= gh) {
            const int gn = std::min (g + nwaves, g_last);
            uint32_t *slot = ring + (size_t) (g % p.ring) * GSTAMD_FUSED_GROUP_WORDS;
            for (int lane = 0; lane < 64; lane++) fused2_stage<NW, CH, SEMI, 0> (p.h, Lw[lane], stage, g, gn, lane);
            for (int lane = 0; lane < 64; lane++) fused2_filter<NW, 0> (Lw[lane], stage, slot, lane);
            for (int lane = 0; lane < 64; lane++) fused2_stage<NW, CH, SEMI, 1> (p.h, Lw[lane], stage, g, gn, lane);
            for (int lane = 0; lane < 64; lane++) fused2_filter<NW, 1> (Lw[lane], stage, slot, lane);
            for (int lane = 0; lane < 64; lane++) fused2_stage<NW, CH, SEMI, 2> (p.h, Lw[lane], stage, g, gn, lane);
            for (int lane = 0; lane < 64; lane++) fused2_filter<NW, 2> (Lw[lane], stage, slot, lane);
            for (int lane = 0; lane < 64; lane++) fused2_stage<NW, CH, SEMI, 3> (p.h, Lw[lane], stage, g, gn, lane);
            for (int lane = 0; lane < 64; lane++) fused2_filter<NW, 3> (Lw[lane], stage, slot, lane);
            g += nwaves;
          }
        };
        auto vrow = [&](int j) {
          for (int lane = 0; lane < 64; lane++) {
            if (p.ngv == 5)
              fused_vrow<5> (p, ring, dst, pf, j, t0, t1, lane);
            else
              fused_vrow<0> (p, ring, dst, pf, j, t0, t1, lane);
          }
        };
        int jr = j0, rows = p.first_rows;
        int jl = std::min (jr + rows, j1) - 1;
        {
          int gl_r, gh;
          fused_round_groups (p, jr, jl, &gl_r, &gh);
          for (int w = 0; w < nwaves; w++)
            produce (w, gh);
        }
        for (;;) {
          const int jr2 = jr + rows;
          const bool more = jr2 < j1;
          const int jl2 = std::min (jr2 + nwaves, j1) - 1;
          int gh2 = -1;
          if (more) {
            int gl_r;
            fused_round_groups (p, jr2, jl2, &gl_r, &gh2);
          }
          /* odd waves first: they produce before the even waves have read the round's window */
          for (int pass = 0; pass < 2; pass++)
            for (int w = 0; w < nwaves; w++) {
              if ((w & 1) != (pass == 0 ? 1 : 0))
                continue;
              const int j = jr + w;
              if (w & 1) {
                produce (w, gh2);
                if (j <= jl)
                  vrow (j);
              } else {
                if (j <= jl)
                  vrow (j);
                produce (w, gh2);
              }
            }
          if (!more)
            break;
          jr = jr2;
          rows = nwaves;
          jl = jl2;
        }
        continue;
      }
      int rows = p.first_rows;
      for (int jr = j0; jr < j1; jr += rows, rows = nwaves) {
        const int jl = std::min (jr + rows, j1) - 1;
        int gl_r, gh;
        fused_round_groups (p, jr, jl, &gl_r, &gh);
        for (int w = 0; w < nwaves; w++) {
          uint32_t *stage = base + (size_t) p.ring * GSTAMD_FUSED_GROUP_WORDS + (size_t) w * 2 * GSTAMD_H420_LINE_WORDS;
          Fused420Lane<NW> *Lw = &L[(size_t) w * 64];
          int &g = gnext[w];
          while (g <= gh) {
            const int gn = std::min (g + nwaves, g_last);
            for (int lane = 0; lane < 64; lane++)
              fused_phase_a<NW, CH, SEMI> (p.h, Lw[lane], stage, g, lane);
            for (int lane = 0; lane < 64; lane++)
              fused_phase_b<NW> (Lw[lane], stage);
            for (int lane = 0; lane < 64; lane++)
              fused_phase_c<NW, CH, SEMI> (p.h, Lw[lane], stage, gn, lane);
            for (int lane = 0; lane < 64; lane++)
              fused_phase_d<NW> (Lw[lane], stage, ring + (size_t) (g % p.ring) * GSTAMD_FUSED_GROUP_WORDS, lane);
            g += nwaves;
          }
        }
        for (int w = 0; w < nwaves; w++) {
          const int j = jr + w;
          if (j <= jl)
            for (int lane = 0; lane < 64; lane++) {
              if (p.ngv == 5)
                fused_vrow<5> (p, ring, dst, pf, j, t0, t1, lane);
              else
                fused_vrow<0> (p, ring, dst, pf, j, t0, t1, lane);
            }
        }
      }
    }
}

template <int NW, int SEMI>
static void run_fused420_ch (const Fused420Params &p, int chroma_h, const Dst &dst, const PostFast &pf, int nwaves)
{
  if (chroma_h == CHROMA_H_H2_CS)
    run_fused420<NW, CHROMA_H_H2_CS, SEMI> (p, dst, pf, nwaves);
  else if (chroma_h == CHROMA_H_H2)
    run_fused420<NW, CHROMA_H_H2, SEMI> (p, dst, pf, nwaves);
  else
    run_fused420<NW, CHROMA_H_NONE, SEMI> (p, dst, pf, nwaves);
}

// the decision of capi_video.cpp (ensure_tables + convert_to_packed) for k_scale420_fused; hp = the k_hscale420_reg parameters
static bool emu_scale420_fused (const VideoPlan &p, H420RegParams hp, int nw, int n_taps_h, const Dst &dst, const PostFast &pf)
{
  if (getenv ("GSTAMD_NO_FUSED420") || p.passes[1].horizontal || p.passes[1].kind != SCALE_NTAP || nw < 3 || nw > 5)
    return false;
  Fused420Tables t;
  if (!make_fused420_tables (p.passes[1], p.front.height, &t))
    return false;
  const char *ew = getenv ("GSTAMD_FUSED_WAVES"), *er = getenv ("GSTAMD_FUSED_ROWS");
  const int nwaves = ew && atoi (ew) > 0 ? atoi (ew) : 8;
  const int rpc = std::max (nwaves, er && atoi (er) > 0 ? atoi (er) : 17);
  Fused420Params q;
  memset (&q, 0, sizeof (q));
  q.h = hp;
  q.n_taps_h = n_taps_h;
  q.vgroup = t.vgroup.data ();
  q.vtapw = t.vtapw.data ();
  q.ngv = t.ngv;
  q.out_h = p.out_info.height;
  q.rows_per_chunk = rpc;
  q.first_rows = std::min (nwaves, getenv ("GSTAMD_FUSED_FIRST") ? std::max (1, atoi (getenv ("GSTAMD_FUSED_FIRST"))) : fused420_first_rows (t, nwaves));
  q.sched = getenv ("GSTAMD_FUSED_SCHED") && atoi (getenv ("GSTAMD_FUSED_SCHED")) == 1 ? 1 : 2;
  q.ring = q.sched == 2 ? fused420_ring_groups2 (t, rpc, nwaves, q.first_rows) : fused420_ring_groups (t, rpc, nwaves, q.first_rows);
  if (getenv ("GSTAMD_FUSED_RING_SHORT"))        /* tests: a ring one slot too short must break the picture (the emulator's wave order shows it) */
    q.ring -= 1;
  q.n_groups = t.n_groups;
  if (((uintptr_t) dst.p % 4) != 0 || (dst.stride % 4) != 0)
    return false;
  g_fused_runs++;
  if (nw == 3)
    hp.semi ? run_fused420_ch<3, 1> (q, p.front.chroma_h, dst, pf, nwaves) : run_fused420_ch<3, 0> (q, p.front.chroma_h, dst, pf, nwaves);
  else if (nw == 4)
    hp.semi ? run_fused420_ch<4, 1> (q, p.front.chroma_h, dst, pf, nwaves) : run_fused420_ch<4, 0> (q, p.front.chroma_h, dst, pf, nwaves);
  else
    hp.semi ? run_fused420_ch<5, 1> (q, p.front.chroma_h, dst, pf, nwaves) : run_fused420_ch<5, 0> (q, p.front.chroma_h, dst, pf, nwaves);
  return true;
}

static bool emu_dot4_ok (const SrcFront &s, const ScaleDev &sd)
{
  return sd.tapw && (kind_has_planes (s.f.kind) || (s.f.kind == UNPACK_PACKED422 && s.f.hi_depth == 0)) && s.pre.matrix.kind == MATRIX_NONE && s.pre.alpha_kind == ALPHA_NONE;
}
static bool emu_dot4_ok (const SrcImage &, const ScaleDev &) { return false; }
static void run_hscale_dot4 (const SrcImage &, const ScaleDev &, const Dst &, const PostFast &, int, int, TileGeom) {}

static bool emu_is_image (const SrcImage &) { return true; }
static bool emu_is_image (const SrcFront &) { return false; }
static void emu_vscale_pk (const SrcImage &src, const ScaleDev &sd, const Dst &d, const PostFast &pf, int w, int h, int x0, int y)
{
  vscale_pk_lane (src, sd, d, pf, w, h, x0, y);
}
static void emu_vscale_pk (const SrcFront &, const ScaleDev &, const Dst &, const PostFast &, int, int, int, int) {}
static int emu_vscale_rows ()
{
  const char *e = getenv ("GSTAMD_VSCALE_ROWS");
  return e ? atoi (e) : 1;
}
static void emu_vscale_pk_rows (const SrcImage &src, const ScaleDev &sd, const Dst &d, const PostFast &pf, int w, int h, int rows)
{
  for (int y0 = 0; y0 < h; y0 += rows)
    for (int x0 = 0; x0 < w; x0 += 4) {
      if (rows == 4)
        vscale_pk_rows_lane<4> (src, sd, d, pf, w, h, x0, y0);
      else
        vscale_pk_rows_lane<2> (src, sd, d, pf, w, h, x0, y0);
    }
}
static void emu_vscale_pk_rows (const SrcFront &, const ScaleDev &, const Dst &, const PostFast &, int, int, int) {}

/* launch_scale_from_front's first branch: k_hscale_wave<SrcLean> */
static bool emu_lean_source (const SrcFront &s, SrcLean *ls)
{
  const FrontParams &f = s.f;
  if (!(f.hi_depth == 0 && s.pre.matrix.kind == MATRIX_NONE && s.pre.alpha_kind == ALPHA_NONE && (f.kind == UNPACK_PACKED4 || f.kind == UNPACK_PACKED422) &&
          ((uintptr_t) s.pl.p[0] % 4) == 0 && (s.pl.stride[0] % 4) == 0 && f.height - 1 <= f.luma_last))
    return false;
  memset ((void *) ls, 0, sizeof (*ls));
  ls->p = s.pl.p[0], ls->stride = s.pl.stride[0], ls->width = f.width;
  ls->p422 = f.kind == UNPACK_PACKED422;
  ls->sel = (uint32_t) f.pos[0] | ((uint32_t) f.pos[1] << 8) | ((uint32_t) f.pos[2] << 16) | ((uint32_t) f.pos[3] << 24);
  ls->pos1 = f.pos[1], ls->pos2 = f.pos[2], ls->pos3 = f.pos[3], ls->chroma_h = f.chroma_h, ls->swap_k = f.swap_k;
  return true;
}
static bool emu_lean_source (const SrcImage &, SrcLean *) { return false; }

template <class SRC>
static void run_scale (bool horizontal, const SRC &src, const ScaleDev &sd, const Dst &d, int out_w, int out_h, int max_span, TileGeom g,
    const PostFast &pf)
{
  SrcLean lean;
  if (horizontal && g.tile_w > 0 && g.lds_px * 4 <= 16384 && emu_lean_source (src, &lean)) {
    run_hscale_wave (lean, sd, d, pf, out_w, out_h, g);
    return;
  }
  if (horizontal && g.tile16_w > 0 && emu_dot4_ok (src, sd) && emu_h420_ok (src)) {
    run_hscale420 (src, sd, d, pf, out_w, out_h, g);
    return;
  }
  if (horizontal && g.tile_w > 0 && emu_dot4_ok (src, sd)) {
    run_hscale_dot4 (src, sd, d, pf, out_w, out_h, g);
    return;
  }
  if (horizontal && g.tile_w > 0 && g.lds_px * 4 <= 16384) {
    run_hscale_wave (src, sd, d, pf, out_w, out_h, g);
    return;
  }
  if (horizontal && max_span <= 12288) {
    run_hscale_lds (src, sd, d, out_w, out_h);
    return;
  }
  if (!horizontal && sd.kind == SCALE_NTAP && emu_is_image (src)) {      /* k_vscale_pk */
    if (emu_vscale_rows () == 2 || emu_vscale_rows () == 4) {
      emu_vscale_pk_rows (src, sd, d, pf, out_w, out_h, emu_vscale_rows ());
      return;
    }
    for (int y = 0; y < out_h; y++)
      for (int x0 = 0; x0 < out_w; x0 += 4)
        emu_vscale_pk (src, sd, d, pf, out_w, out_h, x0, y);
    return;
  }
  for (int y = 0; y < out_h; y++)
    for (int x = 0; x < out_w; x++) {
      if (horizontal)
        hscale_body<SRC> (src, sd, d, out_w, out_h, x, y);
      else
        vscale_body<SRC> (src, sd, d, out_w, out_h, x, y);
    }
}

static int g_fast422_runs = 0;

static int g_bil4_up_runs = 0, g_emu_pack422up_runs = 0;
extern "C" int emu_pack422up_runs (void) { return g_emu_pack422up_runs; }
extern "C" int emu_bil4_up_runs (void) { return g_bil4_up_runs; }
// k_bilinear4_up's grid: a lane per four outputs, strips of EMU_BIL4_UP_ROWS rows (launch_scale2x2_from_front's gate)
static bool emu_bilinear4_up (Bil4Params b, const Dst &d, const PostFast &pf)
{
  if (!bilinear4_up_ok (b) || getenv ("EMU_NO_BILINEAR4_UP"))
    return false;
  g_bil4_up_runs++;
  g_bil4_lo = b.src, g_bil4_hi = b.src + (size_t) b.sstride * b.src_h;
  b.rows = getenv ("EMU_BIL4_UP_ROWS") ? atoi (getenv ("EMU_BIL4_UP_ROWS")) : 5;
  uint32_t sel = 0;
  const bool plain = bilinear4_plain_sel (d, pf, &sel);
  const auto tab = [&] (int y, int *ya, uint32_t *p1) {
    *ya = (int) b.sv.offset[y];
    *p1 = (uint32_t) (int) b.sv.taps[(size_t) y * 2 + 1];
  };
  for (int y0 = 0; y0 < b.out_h; y0 += b.rows)
    for (int x0 = 0; x0 < b.out_w; x0 += 4) {
      const int y1 = y0 + b.rows < b.out_h ? y0 + b.rows : b.out_h;
      if (plain)
        bilinear4_up_lane<1> (b, d, pf, sel, x0, y0, y1, tab);
      else
        bilinear4_up_lane<0> (b, d, pf, sel, x0, y0, y1, tab);
    }
  return true;
}
static int g_bil_runs = 0, g_bilr_runs = 0, g_bilh_runs = 0;
extern "C" int emu_bilh_runs (void) { return g_bilh_runs; }
extern "C" int emu_bil_runs (void) { return g_bil_runs; }
extern "C" int emu_bilr_runs (void) { return g_bilr_runs; }
static int g_fast420p_runs = 0;
extern "C" int emu_fast420p_runs (void) { return g_fast420p_runs; }
extern "C" int emu_fast422_runs (void) { return g_fast422_runs; }
static int emu_convert_packed (const VideoPlan &p, const GstAmdVideoInfo *in, const Planes &pl, uint8_t *d0, int dstride, int vec_ok, bool rgb24);
static const DeepPackParams *g_deep_hook = nullptr;    /* set while the sub-conversion of a shrinking 10-bit plan runs with the source itself as its pixels (k_deep_scale_pack) */
int g_emu_deep_pack_runs = 0, g_emu_deep_pack_wide = 0;
extern "C" int emu_deep_pack_runs (void) { return g_emu_deep_pack_runs; }
extern "C" int emu_deep_pack_wide (void) { return g_emu_deep_pack_wide; }
int g_bil_ayuv_runs = 0;
extern "C" int emu_bil_ayuv_runs (void) { return g_bil_ayuv_runs; }
static const GammaDev *g_gamma_hook = nullptr;          /* set while the direct conversion of a fused gamma plan runs (k_convert_gamma) */
static int g_gamma_fused_runs = 0;
extern "C" int emu_gamma_fused_runs (void) { return g_gamma_fused_runs; }

extern "C" int emu_h420_runs (void) { return g_h420_runs; }
extern "C" int emu_h420_reg_runs (void) { return g_h420_reg_runs; }

// k_swizzle34 (video_kernels.hip swizzle34_setup + the kernel's grid)
static bool emu_swizzle34 (int sb, const int *src_pos, int db, const int *dst_pos, const uint8_t *src, int sstride, uint8_t *dst, int dstride, int width, int height)
{
  if (((uintptr_t) src % 4) != 0 || (sstride % 4) != 0 || ((uintptr_t) dst % 4) != 0 || (dstride % 4) != 0 || getenv ("EMU_NO_SWIZZLE34"))
    return false;
  uint8_t map[4] = {0, 0, 0, 0};
  for (int c = db == 4 ? 0 : 1; c < 4; c++)
    map[dst_pos[c]] = c == 0 && sb == 3 ? 0xff : (uint8_t) src_pos[c];
  Swz34Params sp;
  memset ((void *) &sp, 0, sizeof (sp));
  if (sb == 3 && db == 4)
    swz34_selectors<3, 4> (map, &sp);
  else if (sb == 4 && db == 3)
    swz34_selectors<4, 3> (map, &sp);
  else
    swz34_selectors<3, 3> (map, &sp);
  sp.src = src, sp.sstride = sstride, sp.dst = dst, sp.dstride = dstride, sp.width = width;
  for (int y = 0; y < height; y++)
    for (int lane = 0; lane < ((width + 3) / 4 + 255) / 256 * 256; lane++) {
      if (sb == 3 && db == 4)
        swizzle34_body<3, 4> (sp, lane, y);
      else if (sb == 4 && db == 3)
        swizzle34_body<4, 3> (sp, lane, y);
      else
        swizzle34_body<3, 3> (sp, lane, y);
    }
  return true;
}

static const uint8_t *g_post_lut = nullptr;      /* GammaPlan::lut_direct: set around the direct conversion (GstAmdVideoConverter::post_lut) */
static int g_post_lut_keep = 0;
static bool g_post_lut_done = false;
static int g_swizzle4_runs = 0;
static int g_extra_rows = 0;          /* the AYUV image of a planar destination is being rendered with the line past the picture */
extern "C" int emu_swizzle4_runs (void) { return g_swizzle4_runs; }

/* gstamd_video_converter_divergence of the last top-level emu_video_convert (the plan's note + those of its sub-conversions) */
static std::string g_emu_divergence;
extern "C" const char *emu_video_last_divergence (void) { return g_emu_divergence.c_str (); }

// fill_borders of capi_video.cpp: k_fill_border's rule plane by plane, values from border_plane_value
static void emu_fill_borders (const VideoPlan &p, const GstAmdVideoInfo *out, uint8_t *dst)
{
  const FormatDesc *f = p.fout;
  auto up = [](int v, int sub) { return -((-v) >> sub); };
  const int n_planes = f->kind == UNPACK_PLANAR_A ? 4 : f->kind == UNPACK_SEMI_LE32 || f->kind == UNPACK_SEMI_LE40 || f->kind == UNPACK_SEMI_TILED || f->kind == UNPACK_SEMI_LE40_TILED ? 2 : f->kind == UNPACK_SEMI_A || f->kind == UNPACK_PLANAR_H4 ? 3 : kind_has_planes (f->kind) ? (f->kind == UNPACK_SEMI ? 2 : 3) : 1;
  for (int i = 0; i < n_planes; i++) {
    int es;
    uint32_t lo, hi;
    border_plane_value (f, p.rect.border, i, &es, &lo, &hi);
    const bool nv61_fastpath = i == 1 && f->format == GSTAMD_VIDEO_FORMAT_NV61 && !p.ref_fastpath.empty ();
    if (nv61_fastpath)
      lo = (lo >> 8) | ((lo & 0xffu) << 8);
    uint8_t v[8];
    memcpy (v, &lo, 4);
    memcpy (v + 4, &hi, 4);
    const bool pairs = f->kind == UNPACK_PACKED422 || f->kind == UNPACK_P422_16;
    const bool full = i == 0 || i == GSTAMD_KIND_ALPHA_PLANE (f->kind);
    const int ws = !full || pairs ? f->w_sub : 0, hs = !full ? f->h_sub : 0;
    const int mw = up (p.rect.out_maxw, ws), mh = up (p.rect.out_maxh, hs), x0 = p.rect.out_x >> ws, y0 = p.rect.out_y >> hs;
    const int w = border_picture_positions (f, p.rect, p.out_info.width, ws), h = up (p.out_info.height, hs);
    for (int y = 0; y < mh; y++)
      for (int x = 0; x < mw; x++)
        if (!(x >= x0 && x < x0 + w && y >= y0 && y < y0 + h))
          memcpy (dst + out->offset[i] + (size_t) y * out->stride[i] + (size_t) x * es, v, es);
    if (i == 1 && f->format == GSTAMD_VIDEO_FORMAT_NV61 && (p.rect.out_maxw & 1) && !nv61_fastpath) {          /* the second k_fill_border launch of fill_borders */
      const uint8_t sw[2] = {v[1], v[0]};
      const bool reaches = p.rect.out_x + p.out_info.width == p.rect.out_maxw;
      for (int y = 0; y < mh; y++)
        if (!(reaches && y >= y0 && y < y0 + h))
          memcpy (dst + out->offset[i] + (size_t) y * out->stride[i] + (size_t) (mw - 1) * 2, sw, 2);
    }
    if (f->format == GSTAMD_VIDEO_FORMAT_VYUY && (p.rect.out_maxw & 1) && p.ref_fastpath.empty () && !p.plane_mode) {          /* the VYUY launch of fill_borders */
      const uint8_t sw[4] = {v[2], v[3], v[0], v[1]};
      const bool reaches = p.rect.out_x + p.out_info.width == p.rect.out_maxw;
      for (int y = 0; y < mh; y++)
        if (!(reaches && y >= y0 && y < y0 + h))
          memcpy (dst + out->offset[i] + (size_t) y * out->stride[i] + (size_t) (mw - 1) * 4, sw, 4);
    }
  }
}

extern "C" int emu_video_convert (const GstAmdVideoInfo *in, const GstAmdVideoInfo *out,
    const GstAmdVideoConverterConfig *cfg, const uint8_t *src, uint8_t *dst, int vec_ok, char *desc, int desc_len)
{
  VideoPlan p;
  std::string err;
  if (desc)
    g_emu_divergence.clear ();          /* a top-level call (the sub-conversions of a composite plan pass no buffer) */
  int r = plan_video_converter (in, out, cfg, &p, &err);
  if (r != GSTAMD_OK) {
    if (desc)
      strncpy (desc, err.c_str (), desc_len - 1);
    return r;
  }
  g_emu_divergence += p.divergence;
  if (desc)
    strncpy (desc, p.description.c_str (), desc_len - 1);
  if (p.interlaced) {
    /* frame_planes_plan_order of capi_video.cpp: the two field conversions, field f over lines f, f + 2, ... of every plane - the source chroma planes of
       a plan whose pair table names rows of the frame's chroma planes stay the frame's */
    const std::string keep = g_emu_divergence;
    GstAmdVideoConverterConfig fcfg;
    (void) plan_field_config (in, out, cfg, &fcfg);
    cfg = &fcfg;
    for (int f = 0; f < 2; f++) {
      GstAmdVideoInfo fin, fout;
      plan_field_infos (in, out, f, &fin, &fout);
      VideoPlan fp;
      if ((r = plan_video_converter (&fin, &fout, cfg, &fp, &err)) != GSTAMD_OK)
        return r;
      if (fp.field_src_chroma_frame) {
        int perm[4];
        format_plane_perm (in->format, perm);
        const int alpha_plane = GSTAMD_KIND_ALPHA_PLANE (fp.fin->kind);
        for (int i = 1; i < in->n_planes; i++)
          if (i != alpha_plane)
            fin.offset[i] = in->offset[i], fin.stride[i] = in->stride[i];
      }
      if ((r = emu_video_convert (&fin, &fout, cfg, src, dst, vec_ok, nullptr, 0)) != GSTAMD_OK)
        return r;
    }
    g_emu_divergence = keep;
    return GSTAMD_OK;
  }
  /* plane pointers come from the plan's view of the two frames (GBR: planes R, G, B - format_plan_planes), as in capi_video.cpp */
  GstAmdVideoInfo in_planes = *in, out_planes = *out;
  format_plan_planes (p.fin, &in_planes);
  format_plan_planes (p.fout, &out_planes);
  in = &in_planes, out = &out_planes;
  if (p.v210_fast) {            /* k_v210_fast */
    V210FastParams vp;
    memset ((void *) &vp, 0, sizeof (vp));
    const FormatDesc *f8 = p.fin->kind == UNPACK_V210 ? p.fout : p.fin;
    vp.to_v210 = p.fout->kind == UNPACK_V210;
    vp.kind = f8->kind, vp.h_sub = f8->h_sub, vp.u_plane = f8->u_plane, vp.v_plane = f8->v_plane;
    vp.bps = f8->hi_depth ? 2 : 1;
    memcpy (vp.pos, f8->pos, sizeof (vp.pos));
    vp.width = in->width, vp.height = in->height;
    for (int i = 0; i < in->n_planes && i < 3; i++)
      vp.s[i] = src + in->offset[i], vp.sstride[i] = in->stride[i];
    for (int i = 0; i < out->n_planes && i < 3; i++)
      vp.d[i] = dst + out->offset[i], vp.dstride[i] = out->stride[i];
    if (v210_fast_vec_ok (vp) && vec_ok) {          /* k_v210_fast_vec */
      for (int r0 = 0; r0 < v210_fast_rows (vp); r0++)
        for (int b0 = 0; b0 < (v210_fast_blocks (vp) + 63) / 64 * 64; b0++)
          v210_fast_block (vp, b0, r0);
      return GSTAMD_OK;
    }
    for (int r0 = 0; r0 < v210_fast_rows (vp); r0++)
      for (int g0 = 0; g0 < (v210_fast_groups (vp) + 255) / 256 * 256; g0++)
        v210_fast_body (vp, g0, r0);
    return GSTAMD_OK;
  }
  if (p.gamma.on) {
    /* convert_gamma of capi_video.cpp: sub-conversion / 16-bit front, the stage kernels' bodies over their grids, the u16 scalers,
       encode + sub-conversion or the 16-bit packer */
    const GammaPlan &g = p.gamma;
    if (g.planes_fast && getenv ("GSTAMD_NO_DEEP_PLANES") == nullptr) {        /* k_deep_planes over its grid */
      DeepPlanesPtrs pp;
      memset (&pp, 0, sizeof (pp));
      for (int i = 0; i < in->n_planes && i < 3; i++) {
        pp.in[i] = src + in->offset[i];
        pp.in_stride[i] = in->stride[i];
      }
      for (int i = 0; i < out->n_planes && i < 3; i++) {
        pp.out[i] = dst + out->offset[i];
        pp.out_stride[i] = out->stride[i];
      }
      pp.vec = vec_ok ? 1 : 0;
      for (int i = 0; i < 3; i++)
        pp.vec = pp.vec && ((uintptr_t) pp.in[i] % 16) == 0 && (pp.in_stride[i] % 16) == 0 && ((uintptr_t) pp.out[i] % 16) == 0 && (pp.out_stride[i] % 16) == 0;
      if (pp.vec && deep_planes16_ok (g.planes) && getenv ("GSTAMD_NO_DEEP_PLANES16") == nullptr) {   /* k_deep_planes16 over its grid */
        g_emu_deep16_runs++;
        for (int row = 0; row < deep_planes16_rows (g.planes); row++)
          for (int lx = 0; lx < (g.planes.width / 16 + 63) / 64 * 64; lx++) {
            if (g.planes.in_hi && g.planes.out_hi)
              deep_planes16_body<2> (g.planes, pp, lx, row);
            else if (g.planes.out_hi)
              deep_planes16_body<1> (g.planes, pp, lx, row);
            else
              deep_planes16_body<0> (g.planes, pp, lx, row);
          }
        return GSTAMD_OK;
      }
      const int chh = (g.planes.height + (1 << g.planes.h_sub) - 1) >> g.planes.h_sub;
      for (int row = 0; row < g.planes.height + chh; row++)
        for (int lx = 0; lx < ((g.planes.width + 7) / 8 + 255) / 256 * 256; lx++)
          deep_planes_body (g.planes, pp, lx, row);
      return GSTAMD_OK;
    }
    const int in_w = g.mid_in.width, in_h = g.mid_in.height, out_w = g.mid_out.width, out_h = g.mid_out.height;
    std::vector<uint8_t> mid_a, mid_b, a, b;
    GammaDev gd;
    gd.to_rgb = g.to_rgb;
    gd.to_yuv = g.to_yuv;
    gd.prim = g.prim;
    gd.alpha_kind = g.alpha_kind;
    gd.alpha_value = g.alpha_value;
    gd.dec = g.dec.data ();
    gd.enc = g.enc.data ();
    gd.comp = g.comp.empty () || getenv ("EMU_NO_GAMMA_COMP") ? nullptr : g.comp.data ();
    gd.to_rgb16 = g.to_rgb16;
    gd.to_yuv16 = g.to_yuv16;
    gd.dec16 = g.dec16.data ();
    gd.enc16 = g.enc16.data ();
    const bool dec16 = !g.dec16.empty (), enc16 = !g.enc16.empty ();
    auto stage = [&](int mask, const uint8_t *s8, int ss, uint8_t *d8, int ds, int w, int h) {
      for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
          gamma_stage_px (gd, mask, s8, ss, d8, ds, x, y);
    };
    if (g.fused && g.lut_direct) {          /* the direct conversion (planned with the chain's to_RGB matrix), then k_lut3 over the converted rectangle */
      plan_set_matrix_override (&g.to_rgb);
      g_post_lut = g.comp.data ();
      g_post_lut_keep = p.fout->pos[0];
      g_post_lut_done = false;
      plan_set_border_override (p.rect.border);
      r = emu_video_convert (&g.sub_in_info, &g.mid_in, &g.cfg_in, src, dst, vec_ok, nullptr, 0);
      plan_set_border_override (nullptr);
      g_post_lut = nullptr;
      plan_set_matrix_override (nullptr);
      if (r != GSTAMD_OK || g_post_lut_done)
        return r;
      const int ds = p.orig_out.stride[0];
      uint8_t *rect = dst + p.orig_out.offset[0] + plane_origin (p.fout, 0, p.rect.out_x, p.rect.out_y, ds);
      for (int y = 0; y < p.out_info.height; y++)
        for (int x = 0; x < p.out_info.width; x++) {
          uint32_t *q = (uint32_t *) (rect + (size_t) y * ds) + x;
          *q = gamma_lut3_px (g.comp.data (), *q, p.fout->pos[0]);
        }
      return GSTAMD_OK;
    }
    if (g.fused && getenv ("GSTAMD_NO_GAMMA_FUSED") == nullptr) {
      g_gamma_hook = &gd;
      plan_set_border_override (p.rect.border);
      r = emu_video_convert (&g.sub_in_info, &g.mid_in, &g.cfg_in, src, dst, vec_ok, nullptr, 0);
      plan_set_border_override (nullptr);
      g_gamma_hook = nullptr;
      return r;
    }
    Enc16Params ep16;
    if (enc16_params (p, &ep16) && getenv ("GSTAMD_NO_ENCODE16") == nullptr) {          /* k_encode16 over its grid */
      const uint8_t *sp = src + in->offset[0] + plane_origin (p.fin, 0, p.rect.in_x, p.rect.in_y, in->stride[0]);
      DstPlanes16 d;
      memset (&d, 0, sizeof (d));
      bool ok = vec_ok && ((uintptr_t) sp % 16) == 0 && (in->stride[0] % 16) == 0;
      for (int i = 0; i < out->n_planes && i < 3; i++) {
        d.p[i] = dst + out->offset[i] + plane_origin (p.fout, i, p.rect.out_x, p.rect.out_y, out->stride[i]);
        d.stride[i] = out->stride[i];
        ok = ok && ((uintptr_t) d.p[i] % 8) == 0 && (d.stride[i] % 8) == 0;
      }
      if (ok) {
        g_emu_enc16_runs++;
        if (p.rect.fill)
          emu_fill_borders (p, out, dst);
        const int rows = (ep16.height + (1 << ep16.pk.h_sub) - 1) >> ep16.pk.h_sub;
        bool wide = (ep16.width % 8) == 0 && getenv ("GSTAMD_ENCODE16_NARROW") == nullptr;
        for (int i = 0; i < out->n_planes && i < 3; i++)
          wide = wide && ((uintptr_t) d.p[i] % 16) == 0 && (d.stride[i] % 16) == 0;
        const int npx = wide ? 8 : 4;
        for (int yb = 0; yb < rows; yb++)
          for (int x0 = 0; x0 < (ep16.width / npx + 63) / 64 * 64 * npx; x0 += npx) {
            if (ep16.pk.kind == UNPACK_SEMI) {
              if (wide)
                enc16_block<1, 2> (ep16, sp, in->stride[0], d, x0, yb);
              else
                enc16_block<1, 1> (ep16, sp, in->stride[0], d, x0, yb);
            } else {
              if (wide)
                enc16_block<0, 2> (ep16, sp, in->stride[0], d, x0, yb);
              else
                enc16_block<0, 1> (ep16, sp, in->stride[0], d, x0, yb);
            }
          }
        return GSTAMD_OK;
      }
    }
    const bool has_mid = g.prim.has_matrix || g.alpha_kind != ALPHA_NONE;
    const size_t n = p.passes.size ();
    bool mid_done = !has_mid;
    Deep16Image cur = {nullptr, 0, 0, 0};
    bool cur_is_source = false;
    if (g.src64) {
      cur.p = src + in->offset[0] + plane_origin (p.fin, 0, p.rect.in_x, p.rect.in_y, in->stride[0]), cur.stride = in->stride[0], cur.width = in_w, cur.height = in_h;
      cur_is_source = true;
    } else if (g.src16) {
      Planes pl;
      memset (&pl, 0, sizeof (pl));
      for (int i = 0; i < in->n_planes; i++) {
        pl.p[i] = src + in->offset[i] + plane_origin (p.fin, i, p.rect.in_x, p.rect.in_y, in->stride[i]);
        pl.stride[i] = in->stride[i];
      }
      DeepPackParams ds16;
      if (!g_gamma_hook && getenv ("GSTAMD_NO_DEEP_SCALE_PACK") == nullptr && deep_scale_pack16_plan_ok (p, &ds16)) {       /* k_deep_scale_pack16 over its grid */
        ds16.pl = pl;
        ds16.vpair = p.vpair.data ();
        ds16.sh.offset = p.passes[0].offset.data (), ds16.sh.taps = p.passes[0].taps.data ();
        ds16.sv.offset = p.passes[1].offset.data (), ds16.sv.taps = p.passes[1].taps.data ();
        DstPlanes16 d16;
        memset (&d16, 0, sizeof (d16));
        for (int i = 0; i < out->n_planes && i < 3; i++) {
          d16.p[i] = dst + out->offset[i] + plane_origin (p.fout, i, p.rect.out_x, p.rect.out_y, out->stride[i]);
          d16.stride[i] = out->stride[i];
        }
        if (p.rect.fill)
          emu_fill_borders (p, out, dst);
        g_emu_deep_pack_runs++;
        const int rows16 = (g.pack.height + (1 << g.pack.h_sub) - 1) >> g.pack.h_sub;
        for (int yb = 0; yb < rows16; yb++)
          for (int l = 0; l < g.pack.width / 4; l++)
            deep_scale_pack16_any (deep_pack_variant (p.front), g.pack, g.pack_hi_depth, g.dither16, ds16, d16, 4 * l, yb);
        return GSTAMD_OK;
      }
      {
        /* convert_gamma's k_deep_scale_pack branch: the sub-conversion with the 10-bit source as its pixels */
        VideoPlan subp;
        std::string serr;
        DeepPackParams dsp;
        plan_set_border_override (p.rect.border);
        const int sr = plan_video_converter (&g.mid_out, &g.sub_out_info, &g.cfg_out, &subp, &serr);
        plan_set_border_override (nullptr);
        if (sr == GSTAMD_OK && !g_gamma_hook && getenv ("GSTAMD_NO_DEEP_SCALE_PACK") == nullptr && deep_scale_pack_plan_ok (p, subp, &dsp)) {
          dsp.pl = pl;
          dsp.vpair = p.vpair.data ();
          dsp.sh.offset = p.passes[0].offset.data (), dsp.sh.taps = p.passes[0].taps.data ();
          dsp.sv.offset = p.passes[1].offset.data (), dsp.sv.taps = p.passes[1].taps.data ();
          mid_b.assign ((size_t) out_w * out_h * 4, 0);
          g_deep_hook = &dsp;
          plan_set_border_override (p.rect.border);
          r = emu_video_convert (&g.mid_out, &g.sub_out_info, &g.cfg_out, mid_b.data (), dst, vec_ok, nullptr, 0);
          plan_set_border_override (nullptr);
          g_deep_hook = nullptr;
          return r;
        }
      }
      a.resize ((size_t) in_w * in_h * 8);
      for (int y = 0; y < in_h; y++)
        for (int x0 = 0; x0 < (in_w / 4 + 256) / 256 * 1024; x0 += 4)
          front16_lane4 (p.front, pl, p.vpair.data (), a.data (), in_w * 8, x0, y);
      cur.p = a.data (), cur.stride = in_w * 8, cur.width = in_w, cur.height = in_h;
    } else {
      mid_a.resize ((size_t) in_w * in_h * 4);
      if ((r = emu_video_convert (&g.sub_in_info, &g.mid_in, &g.cfg_in, src, mid_a.data (), vec_ok, nullptr, 0)) != GSTAMD_OK)
        return r;
      if (n == 0 && !g.pack16 && !g.store64) {
        mid_b.resize ((size_t) out_w * out_h * 4);
        stage (GAMMA_STAGE_DEC | GAMMA_STAGE_MID | GAMMA_STAGE_ENC, mid_a.data (), in_w * 4, mid_b.data (), out_w * 4, out_w, out_h);
      } else {
        const bool mid_now = !mid_done && (n == 0 || !g.shrink);
        a.resize ((size_t) in_w * in_h * 8);
        stage (GAMMA_STAGE_DEC | (mid_now ? GAMMA_STAGE_MID : 0), mid_a.data (), in_w * 4, a.data (), in_w * 8, in_w, in_h);
        mid_done = mid_done || mid_now;
        cur.p = a.data (), cur.stride = in_w * 8, cur.width = in_w, cur.height = in_h;
      }
    }
    if (cur.p && dec16) {           /* convert_gamma: the 16-bit decode (+ the convert stage where it comes before the scalers) on the first image */
      const bool mid_now = !mid_done && (n == 0 || !g.shrink);
      if (cur_is_source) {
        a.resize ((size_t) cur.width * cur.height * 8);
        stage (GAMMA_STAGE_DEC16 | (mid_now ? GAMMA_STAGE_MID : 0), cur.p, cur.stride, a.data (), cur.width * 8, cur.width, cur.height);
        cur.p = a.data (), cur.stride = cur.width * 8;
        cur_is_source = false;
      } else {
        stage (GAMMA_STAGE_DEC16 | (mid_now ? GAMMA_STAGE_MID : 0), cur.p, cur.stride, (uint8_t *) cur.p, cur.stride, cur.width, cur.height);
      }
      mid_done = mid_done || mid_now;
    }
    if (cur.p) {
      if (!mid_done && (n == 0 || !g.shrink)) {
        if (cur_is_source) {
          a.resize ((size_t) cur.width * cur.height * 8);
          stage (GAMMA_STAGE_MID, cur.p, cur.stride, a.data (), cur.width * 8, cur.width, cur.height);
          cur.p = a.data (), cur.stride = cur.width * 8;
          cur_is_source = false;
        } else {
          stage (GAMMA_STAGE_MID, cur.p, cur.stride, (uint8_t *) cur.p, cur.stride, cur.width, cur.height);
        }
        mid_done = true;
      }
      for (size_t i = 0; i < n; i++) {
        ScaleDev sd;
        memset (&sd, 0, sizeof (sd));
        sd.kind = p.passes[i].kind;
        sd.n_taps = p.passes[i].n_taps;
        sd.inc = p.passes[i].inc;
        sd.offset = p.passes[i].offset.data ();
        sd.taps = p.passes[i].taps.data ();
        const bool hz = p.passes[i].horizontal;
        const int ow = hz ? p.passes[i].out_size : cur.width, oh = hz ? cur.height : p.passes[i].out_size;
        std::vector<uint8_t> next ((size_t) ow * oh * 8);
        for (int y = 0; y < oh; y++)
          for (int x = 0; x < (ow + 255) / 256 * 256; x++)
            scale16_lane (cur, sd, hz, next.data (), ow * 8, ow, oh, x, y);
        b.swap (next);
        a.swap (b);
        cur.p = a.data (), cur.stride = ow * 8, cur.width = ow, cur.height = oh;
        cur_is_source = false;
      }
      if (g.pack16 || g.store64) {
        const int m16 = (mid_done ? 0 : GAMMA_STAGE_MID) | (enc16 ? GAMMA_STAGE_ENC16 : 0);
        if (m16 && cur_is_source) {
          a.resize ((size_t) cur.width * cur.height * 8);
          stage (m16, cur.p, cur.stride, a.data (), cur.width * 8, cur.width, cur.height);
          cur.p = a.data (), cur.stride = cur.width * 8;
        } else if (m16) {
          stage (m16, cur.p, cur.stride, (uint8_t *) cur.p, cur.stride, cur.width, cur.height);
        }
      } else {
        mid_b.resize ((size_t) out_w * out_h * 4);
        stage ((mid_done ? 0 : GAMMA_STAGE_MID) | GAMMA_STAGE_ENC, cur.p, cur.stride, mid_b.data (), out_w * 4, out_w, out_h);
      }
    }
    const bool frame_pack = g.pack16 && g.pack.kind == UNPACK_V210 && g.pack.frame_on == 2;          /* as convert_gamma of capi_video.cpp */
    if ((g.store64 || g.pack16) && p.rect.fill && !frame_pack)
      emu_fill_borders (p, out, dst);
    if (g.store64) {
      uint8_t *rect = dst + out->offset[0] + plane_origin (p.fout, 0, p.rect.out_x, p.rect.out_y, out->stride[0]);
      for (int y = 0; y < out_h; y++)
        memcpy (rect + (size_t) y * out->stride[0], cur.p + (size_t) y * cur.stride, (size_t) out_w * 8);
      if (dither_is_diffusion (g.dither16))          /* launch_dither16_any: k_dither16_verterr / k_dither16_ed */
        ed16_image_host (g.dither16, rect, out->stride[0], out_w, out_h);
      else if (g.dither16.on)           /* k_dither16_image over the finished picture */
        for (int y = 0; y < out_h; y++)
          for (int x = 0; x < out_w; x++)
            dither16_image_px (g.dither16, rect, out->stride[0], out_w, out_h, x, y);
      return GSTAMD_OK;
    }
    if (g.pack16) {
      DstPlanes16 d;
      memset (&d, 0, sizeof (d));
      for (int i = 0; i < out->n_planes && i < 3; i++) {
        d.p[i] = dst + out->offset[i] + (frame_pack ? 0 : plane_origin (p.fout, i, p.rect.out_x, p.rect.out_y, out->stride[i]));
        d.stride[i] = out->stride[i];
      }
      PackPlanarParams pk16 = g.pack;
      DitherParams dt16 = g.dither16;
      std::vector<uint8_t> ed_img;
      if (dither_is_diffusion (g.dither16)) {
        /* launch_pack16_ed: the downsamplers in place on (a copy of) the image, the dither pass over all of it, then selection only */
        ed_img.resize ((size_t) out_w * 8 * out_h);
        for (int y = 0; y < out_h; y++)
          memcpy (ed_img.data () + (size_t) y * out_w * 8, cur.p + (size_t) y * cur.stride, (size_t) out_w * 8);
        const int prow = (pk16.height + (1 << pk16.h_sub) - 1) >> pk16.h_sub;
        for (int pass = 0; pass < 2; pass++)
          for (int yb = 0; yb < prow; yb++)
            for (int x = 0; x < pk16.width; x++)
              if (pass == 0)
                pack16_down_v_px (pk16, ed_img.data (), out_w * 8, x, yb);
              else
                pack16_down_h_px (pk16, ed_img.data (), out_w * 8, x, yb);
        ed16_image_host (g.dither16, ed_img.data (), out_w * 8, out_w, out_h);
        cur.p = ed_img.data (), cur.stride = out_w * 8;
        pk16 = pack_select_only (pk16);
        memset (&dt16, 0, sizeof (dt16));
      }
      if (g.pack.kind == UNPACK_P422_16 || g.pack.kind == UNPACK_P422_UYVP || GSTAMD_KIND_PX16 (g.pack.kind) || g.pack.kind == UNPACK_V210) {          /* k_pack16_packed */
        for (int y = 0; y < pack16_rows (g.pack); y++)
          for (int un = 0; un < (pack16_units (g.pack) + 255) / 256 * 256; un++)
            pack16_packed_body (pk16, g.pack_hi_depth, dt16, cur.p, cur.stride, d.p[0], d.stride[0], un, y);
        return GSTAMD_OK;
      }
      const int rows = (g.pack.height + (1 << g.pack.h_sub) - 1) >> g.pack.h_sub;
      for (int yb = 0; yb < rows; yb++)
        for (int x0 = 0; x0 < (g.pack.width / 4 + 256) / 256 * 1024; x0 += 4)
          pack16_body (pk16, g.pack_hi_depth, dt16, cur.p, cur.stride, d, x0, yb);
      if (p.fout->kind == UNPACK_PLANAR_A)          /* k_pack16_alpha_plane */
        for (int y = 0; y < g.pack.height; y++)
          for (int x0 = 0; x0 < g.pack.width; x0 += 4)
            pack16_alpha_plane_body (pk16, g.pack_hi_depth, dt16, cur.p, cur.stride,
                dst + out->offset[3] + plane_origin (p.fout, 3, p.rect.out_x, p.rect.out_y, out->stride[3]), out->stride[3], x0, y);
      return GSTAMD_OK;
    }
    plan_set_border_override (p.rect.border);
    r = emu_video_convert (&g.mid_out, &g.sub_out_info, &g.cfg_out, mid_b.data (), dst, vec_ok, nullptr, 0);
    plan_set_border_override (nullptr);
    return r;
  }
  Planes pl;
  memset (&pl, 0, sizeof (pl));
  for (int i = 0; i < in->n_planes; i++) {
    pl.p[i] = src + in->offset[i] + plane_origin (p.fin, i, p.rect.in_x, p.rect.in_y, in->stride[i]);
    pl.stride[i] = in->stride[i];
  }
  /* destination rectangle + borders (k_fill_border's rule, plane by plane) */
  GstAmdVideoInfo orect = *out;
  if (p.rect.fill)
    emu_fill_borders (p, out, dst);
  for (int i = 0; i < out->n_planes; i++)
    orect.offset[i] = out->offset[i] + plane_origin (p.fout, i, p.rect.out_x, p.rect.out_y, out->stride[i]);
  out = &orect;
  bool plane_frame = p.plane_mode && p.planes.size () <= PLN_MAX_JOBS && !getenv ("EMU_NO_PLANE_FRAME");
  for (const PlanePlan &pp : p.planes) {
    for (const ScalePass &sp : pp.passes)
      plane_frame = plane_frame && sp.merged == 0;
    plane_frame = plane_frame && plane_job_lds_bytes (pp) <= PLN_LDS_BYTES;
  }
  if (plane_frame) {                      /* k_plane_frame: tile by tile, the two phases either side of the barrier */
    std::vector<uint32_t> lds (PLN_LDS_BYTES / 4);
    for (const PlanePlan &pp : p.planes) {
      PlaneJob J;
      memset ((void *) &J, 0, sizeof (J));
      J.kind = pp.kind;
      J.s = {pl.p[pp.src_plane], pl.stride[pp.src_plane], pp.n_elems, pp.n_elems == 2 && ((uintptr_t) pl.p[pp.src_plane] % 2) == 0 && (pl.stride[pp.src_plane] % 2) == 0};
      J.d = {dst + out->offset[pp.dst_plane], out->stride[pp.dst_plane], pp.n_elems};
      J.iw = pp.iw, J.ih = pp.ih, J.ow = pp.ow, J.oh = pp.oh;
      J.n_pass = (int) pp.passes.size ();
      J.h_first = J.n_pass ? pp.passes[0].horizontal : 0;
      for (size_t k = 0; k < pp.passes.size (); k++) {
        J.pass[k].kind = pp.passes[k].kind;
        J.pass[k].n_taps = pp.passes[k].n_taps;
        J.pass[k].inc = pp.passes[k].inc;
        J.pass[k].offset = pp.passes[k].offset.data ();
        J.pass[k].taps = pp.passes[k].taps.data ();
      }
      const int unit = 4 * pp.n_elems;
      J.wide = pp.n_elems <= 2 && ((uintptr_t) J.d.p % unit) == 0 && (J.d.stride % unit) == 0;
      J.wide_src = ((uintptr_t) J.s.p % 8) == 0 && (J.s.stride % 8) == 0;
      J.tiles_x = (pp.ow + PLN_TW - 1) / PLN_TW;
      J.dstep = getenv ("GSTAMD_PLANE_QUAD_NO_DSTEP") ? 0 : plane_quad_dstep (pp);
      J.quad = getenv ("GSTAMD_NO_PLANE_QUAD") ? 0 : plane_job_quad (J, plane_quad_ok (pp), plane_quad_ok (pp, 8), getenv ("GSTAMD_PLANE_QUAD_MODE") ? atoi (getenv ("GSTAMD_PLANE_QUAD_MODE")) : 2);
      if (J.quad && getenv ("EMU_QUAD_DEBUG"))
        fprintf (stderr, "plane %d: n %d quad %d dstep %d h_first %d kinds %d %d mode %d\n", (int) (&pp - &p.planes[0]), J.s.n, J.quad, J.dstep, J.h_first, J.pass[0].kind, J.pass[1].kind,
            J.kind == PLANE_SCALE ? quad_mode_of (J) : -1);
      if (J.quad) {                           /* k_plane_quad over its grid: 64-lane waves, `rows` rows each */
        g_emu_quad_runs++;
        g_emu_quad_modes |= 1 << (J.quad - 1);
        const int bytes = quad_mode_bytes (J.quad - 1), rows = getenv ("GSTAMD_PLANE_QUAD_ROWS") ? atoi (getenv ("GSTAMD_PLANE_QUAD_ROWS")) : 3;
        const int lanes = (((J.ow * J.s.n + bytes - 1) / bytes + 63) / 64) * 64;
        for (int y0 = 0; y0 < J.oh; y0 += rows)
          for (int lane = 0; lane < lanes; lane++)
            plane_rows_body (J, lane, y0, rows);
        continue;
      }
      const int tiles = J.tiles_x * ((pp.oh + PLN_TH - 1) / PLN_TH);
      for (int t = 0; t < tiles; t++) {
        if (plane_job_is_direct (J)) {        /* k_plane_direct */
          for (int tid = 0; tid < PLN_THREADS; tid++)
            plane_direct_body (J, t, tid);
          continue;
        }
        for (int phase = 0; phase < PLN_PHASES; phase++)          /* k_plane_tiles */
          for (int tid = 0; tid < PLN_THREADS; tid++)
            plane_tile_body (J, (uint8_t *) lds.data (), t, tid, phase);
      }
    }
    return GSTAMD_OK;
  }
  if (p.plane_mode) {                     /* convert_scale_planes: the plane kernels' bodies over their grids */
    for (const PlanePlan &pp : p.planes) {
      const SrcPlane sp = {pl.p[pp.src_plane], pl.stride[pp.src_plane], pp.n_elems, 0};
      const DstPlane dp = {dst + out->offset[pp.dst_plane], out->stride[pp.dst_plane], pp.n_elems};
      if (pp.kind != PLANE_SCALE) {
        for (int y = 0; y < pp.oh; y++)
          for (int x = 0; x < pp.ow; x++)
            plane_simple_body (pp.kind, sp, dp, pp.ow, pp.oh, x, y);
        continue;
      }
      ScaleDev sd[2];
      for (size_t k = 0; k < pp.passes.size (); k++) {
        memset (&sd[k], 0, sizeof (sd[k]));
        sd[k].kind = pp.passes[k].kind;
        sd[k].n_taps = pp.passes[k].n_taps;
        sd[k].inc = pp.passes[k].inc;
        sd[k].offset = pp.passes[k].offset.data ();
        sd[k].taps = pp.passes[k].taps.data ();
        sd[k].merged = pp.passes[k].merged;
      }
      auto run = [&](bool horizontal, const ScaleDev &s1, const SrcPlane &a, const DstPlane &b, int w, int h) {
        for (int y = 0; y < h; y++)
          for (int x = 0; x < w; x++) {
            if (horizontal)
              plane_hscale_body (a, s1, b, w, h, x, y);
            else
              plane_vscale_body (a, s1, b, w, h, x, y);
          }
      };
      if (pp.passes.size () == 1) {
        run (pp.passes[0].horizontal, sd[0], sp, dp, pp.ow, pp.oh);
      } else {
        const int tw = pp.passes[0].horizontal ? pp.ow : pp.iw, th = pp.passes[0].horizontal ? pp.ih : pp.oh;
        std::vector<uint8_t> tmp ((size_t) tw * th * pp.n_elems);
        const DstPlane td = {tmp.data (), tw * pp.n_elems, pp.n_elems};
        const SrcPlane ts = {tmp.data (), tw * pp.n_elems, pp.n_elems, 0};
        run (pp.passes[0].horizontal, sd[0], sp, td, tw, th);
        run (pp.passes[1].horizontal, sd[1], ts, dp, pp.ow, pp.oh);
      }
    }
    return GSTAMD_OK;
  }
  if (p.out_planar && g_deep_hook) {       /* k_deep_scale_pack over its grid (launch_deep_scale_pack) */
    DeepPackParams dp = *g_deep_hook;
    DstPlanes d;
    memset (&d, 0, sizeof (d));
    for (int i = 0; i < out->n_planes && i < 3; i++) {
      d.p[i] = dst + out->offset[i];
      d.stride[i] = out->stride[i];
    }
    const int variant = deep_pack_variant (dp.f);
    const int nblk = p.pack.width / 4, rows = (p.pack.height + (1 << p.pack.h_sub) - 1) >> p.pack.h_sub;
    g_emu_deep_pack_runs++;
    g_emu_deep_pack_wide++;
    for (int yb = 0; yb < rows; yb++)
      for (int l = 0; l < nblk; l++)          /* (the device's lanes 0 and 63 of a workgroup repeat a neighbour's block and store nothing) */
        deep_scale_pack_any (variant, p.pack, dp, d, 4 * l, yb);
    return GSTAMD_OK;
  }
  if (p.out_planar && p.fout->kind == UNPACK_PACKED3 && p.passes.empty () && !p.deep16 && !p.pack.dither.on && p.matrix.kind == MATRIX_NONE &&
      p.post.alpha_kind == ALPHA_NONE && p.front.hi_depth == 0 && (p.front.kind == UNPACK_PACKED4 || p.front.kind == UNPACK_PACKED3) &&
      p.post.pack_pos[0] == 0 && p.post.pack_pos[1] == 1 && p.post.pack_pos[2] == 2 && p.post.pack_pos[3] == 3 &&
      emu_swizzle34 (p.front.kind == UNPACK_PACKED4 ? 4 : 3, p.front.pos, 3, p.pack.pos, pl.p[0], pl.stride[0], dst + out->offset[0], out->stride[0],
          p.out_info.width, p.out_info.height))
    return GSTAMD_OK;
  if (p.out_planar && p.fout->kind == UNPACK_PACKED3 && p.passes.empty () && p.fast_pair && vec_ok)
    return emu_convert_packed (p, in, pl, dst + out->offset[0], out->stride[0], vec_ok >= 200 ? 1 : vec_ok, true);     /* the line-pair kernel stores 3-byte pixels itself */
  if (p.out_planar && p.fast_enc420 && vec_ok) {      /* k_encode420's grid: 4 x 2 pixel blocks */
    const Enc420Params ep = make_enc420_params (p);
    DstPlanes d;
    memset (&d, 0, sizeof (d));
    for (int i = 0; i < out->n_planes && i < 3; i++) {
      d.p[i] = dst + out->offset[i];
      d.stride[i] = out->stride[i];
    }
    for (int r2 = 0; r2 < (ep.height + 1) / 2; r2++)
      for (int x0 = 0; x0 < (ep.width / 4 + 63) / 64 * 256; x0 += 4) {
        if (p.fout->kind == UNPACK_SEMI)
          enc420_block<1> (ep, pl.p[0], pl.stride[0], d, x0, r2);
        else
          enc420_block<0> (ep, pl.p[0], pl.stride[0], d, x0, r2);
      }
    return GSTAMD_OK;
  }
  if (p.out_planar && p.relayout && !getenv ("EMU_NO_RELAYOUT")) {         /* k_planes_relayout when relayout_usable says so */
    RelayoutParams rp;
    memset ((void *) &rp, 0, sizeof (rp));
    rp.width = p.out_info.width, rp.height = p.out_info.height;
    rp.cw = (rp.width + (1 << p.fout->w_sub) - 1) >> p.fout->w_sub, rp.ch = (rp.height + (1 << p.fout->h_sub) - 1) >> p.fout->h_sub;
    rp.in_semi = p.fin->kind == UNPACK_SEMI, rp.out_semi = p.fout->kind == UNPACK_SEMI;
    rp.in_u = p.fin->u_plane, rp.in_v = p.fin->v_plane, rp.out_u = p.fout->u_plane, rp.out_v = p.fout->v_plane;
    bool ok = true;
    for (int i = 0; i < p.in_info.n_planes && i < 3; i++) {
      rp.in[i] = pl.p[i], rp.in_stride[i] = pl.stride[i];
      ok = ok && ((uintptr_t) rp.in[i] % 16) == 0 && (rp.in_stride[i] % 16) == 0;
    }
    for (int i = 0; i < out->n_planes && i < 3; i++) {
      rp.out[i] = dst + out->offset[i], rp.out_stride[i] = out->stride[i];
      ok = ok && ((uintptr_t) rp.out[i] % 16) == 0 && (rp.out_stride[i] % 16) == 0;
    }
    if (ok) {
      for (int row = 0; row < relayout_rows (rp); row++)
        for (int lane = 0; lane < (relayout_lanes (rp) + 255) / 256 * 256; lane++)
          relayout_body (rp, lane, row);
      return GSTAMD_OK;
    }
  }
  if (p.out_planar && GSTAMD_KIND_ALPHA_PLANE (p.fout->kind) < 0 && p.passes.empty () && !p.deep16 && !(p.pack.dither.on && p.pack.dither.method != GSTAMD_DITHER_NONE && p.pack.dither.method != GSTAMD_DITHER_BAYER) &&
      p.post.pack_pos[0] == 0 && p.post.pack_pos[1] == 1 && p.post.pack_pos[2] == 2 && p.post.pack_pos[3] == 3 && !getenv ("EMU_NO_CONVERT_PACK") &&
      (p.front.kind == UNPACK_PACKED4 || (p.front.kind == UNPACK_PACKED422 && (p.front.chroma_h == CHROMA_H_NONE || !getenv ("GSTAMD_NO_CONVERT_PACK_422UP")) &&
              !p.front.chroma_v2 && p.matrix.kind == MATRIX_NONE && p.post.alpha_kind == ALPHA_NONE))) {          /* convert_pack_usable, alignment aside */
    /* k_convert_pack: the pack body with the chain itself as its pixel source */
    DstPlanes d;
    memset (&d, 0, sizeof (d));
    for (int i = 0; i < out->n_planes && i < 3; i++) {
      d.p[i] = dst + out->offset[i];
      d.stride[i] = out->stride[i];
    }
    SrcFront sf;
    sf.f = p.front;
    sf.pl = pl;
    sf.vpair = p.vpair.empty () ? nullptr : p.vpair.data ();
    memset (&sf.pre, 0, sizeof (sf.pre));
    sf.pre.matrix = p.matrix;
    sf.pre.alpha_kind = p.post.alpha_kind;
    sf.pre.alpha_value = p.post.alpha_value;
    sf.vec_ok = 0;
    const int rows = (p.out_info.height + (1 << p.pack.h_sub) - 1) >> p.pack.h_sub;
    const Src422Dup s422 = {pl.p[0], pl.stride[0], 8 * p.front.pos[1], 8 * p.front.pos[2], 8 * p.front.pos[3], p.front.swap_k};
    for (int yb = 0; yb < rows; yb++)
      for (int x0 = 0; x0 < p.out_info.width; x0 += 4)
        if (p.front.kind == UNPACK_PACKED422 && p.front.chroma_h != CHROMA_H_NONE) {          /* k_convert_pack_422up */
          const Src422Up sup = {pl.p[0], pl.stride[0], 8 * p.front.pos[1], 8 * p.front.pos[2], 8 * p.front.pos[3], p.front.swap_k, p.front.chroma_h,
            p.front.width, p.front.luma_last};
          bool wide = vec_ok && !p.pack.dither.on && (p.pack.kind == UNPACK_PLANAR || p.pack.kind == UNPACK_SEMI);
          for (int i = 0; wide && i < (p.pack.kind == UNPACK_SEMI ? 2 : 3); i++)
            wide = ((uintptr_t) d.p[i] % 4) == 0 && (d.stride[i] % 4) == 0;
          if (wide && pack_planar_block4 (p.pack, sup, d, x0, yb)) {
            g_emu_pack422up_runs++;
            continue;
          }
          pack_planar_body (p.pack, sup, d, x0, yb);
        } else if (p.front.kind == UNPACK_PACKED422) {               /* k_convert_pack_422: 8 pixels per lane */
          if (x0 & 4)
            continue;
          bool wide = ((uintptr_t) pl.p[0] % 16) == 0 && (pl.stride[0] % 16) == 0 && (p.pack.kind == UNPACK_PLANAR || p.pack.kind == UNPACK_SEMI);
          for (int i = 0; wide && i < (p.pack.kind == UNPACK_SEMI ? 2 : 3); i++)
            wide = ((uintptr_t) d.p[i] % 8) == 0 && (d.stride[i] % 8) == 0;
          if (wide && pack_422dup_block8 (p.pack, s422, d, x0, yb))
            continue;
          pack_planar_body (p.pack, s422, d, x0, yb);
          pack_planar_body (p.pack, s422, d, x0 + 4, yb);
        } else {
          const SrcPacked4 s4 = make_src_packed4 (p.front, pl, sf.pre);
          bool wide = vec_ok && !p.pack.dither.on && (p.pack.kind == UNPACK_PLANAR || p.pack.kind == UNPACK_SEMI) && ((uintptr_t) pl.p[0] % 16) == 0 &&
              (pl.stride[0] % 16) == 0 && getenv ("GSTAMD_NO_CONVERT_PACK_WIDE") == nullptr;
          for (int i = 0; wide && i < (p.pack.kind == UNPACK_SEMI ? 2 : 3); i++)
            wide = ((uintptr_t) d.p[i] % 4) == 0 && (d.stride[i] % 4) == 0;
          if (wide && pack_planar_block4 (p.pack, s4, d, x0, yb)) {
            g_emu_pack4_runs++;
            continue;
          }
          pack_planar_body (p.pack, s4, d, x0, yb);
        }
    return GSTAMD_OK;
  }
  {
    PlanePlan raw4;
    bool enc420 = false;
    if (GSTAMD_KIND_ALPHA_PLANE (p.fout->kind) < 0 && plane_raw4_pack_plan (p, &raw4, &enc420) && getenv ("GSTAMD_NO_PLANE_QUAD") == nullptr && vec_ok) {
      /* k_plane_quad on the raw 4-byte pixels into the image, then k_encode420 / k_convert_pack from it (capi_video.cpp: raw4_pack) */
      const int ow = p.out_info.width, oh = p.out_info.height;
      std::vector<uint32_t> img ((size_t) ow * (oh + 1) + 4);
      PlaneJob J;
      memset ((void *) &J, 0, sizeof (J));
      J.kind = PLANE_SCALE;
      J.s = {pl.p[0], pl.stride[0], 4, 0};
      J.d = {(uint8_t *) img.data (), ow * 4, 4};
      J.iw = raw4.iw, J.ih = raw4.ih, J.ow = ow, J.oh = oh;
      J.n_pass = 2;
      J.h_first = p.passes[0].horizontal ? 1 : 0;
      for (int q = 0; q < 2; q++) {
        J.pass[q].kind = p.passes[q].kind;
        J.pass[q].n_taps = p.passes[q].n_taps;
        J.pass[q].inc = p.passes[q].inc;
        J.pass[q].offset = p.passes[q].offset.data ();
        J.pass[q].taps = p.passes[q].taps.data ();
      }
      J.dstep = getenv ("GSTAMD_PLANE_QUAD_NO_DSTEP") ? 0 : plane_quad_dstep (raw4);
      J.quad = 1 + QUAD_8;
      g_emu_quad_runs++;
      const int qrows = getenv ("GSTAMD_PLANE_QUAD_ROWS") ? atoi (getenv ("GSTAMD_PLANE_QUAD_ROWS")) : 3;
      const int lanes = (((ow * 4 + 7) / 8 + 63) / 64) * 64;
      for (int y0 = 0; y0 < oh; y0 += qrows)
        for (int lane = 0; lane < lanes; lane++)
          plane_rows_body (J, lane, y0, qrows);
      DstPlanes d;
      memset (&d, 0, sizeof (d));
      bool enc = enc420;
      for (int i = 0; i < out->n_planes && i < 3; i++) {
        d.p[i] = dst + out->offset[i];
        d.stride[i] = out->stride[i];
        enc = enc && ((uintptr_t) d.p[i] % 4) == 0 && (d.stride[i] % 4) == 0;
      }
      if (enc) {
        const Enc420Params ep = make_enc420_params (p);
        for (int r2 = 0; r2 < (ep.height + 1) / 2; r2++)
          for (int x0 = 0; x0 < (ep.width / 4 + 63) / 64 * 256; x0 += 4) {
            if (p.fout->kind == UNPACK_SEMI)
              enc420_block<1> (ep, (const uint8_t *) img.data (), ow * 4, d, x0, r2);
            else
              enc420_block<0> (ep, (const uint8_t *) img.data (), ow * 4, d, x0, r2);
          }
        return GSTAMD_OK;
      }
      FrontParams f2 = p.front;
      f2.width = ow, f2.height = oh;
      Planes pl2;
      memset ((void *) &pl2, 0, sizeof (pl2));
      pl2.p[0] = (const uint8_t *) img.data (), pl2.stride[0] = ow * 4;
      ColorParams color;
      memset (&color, 0, sizeof (color));
      color.matrix = p.matrix;
      color.alpha_kind = p.post.alpha_kind;
      color.alpha_value = p.post.alpha_value;
      const SrcPacked4 s4 = make_src_packed4 (f2, pl2, color);
      bool wide = !p.pack.dither.on && (p.pack.kind == UNPACK_PLANAR || p.pack.kind == UNPACK_SEMI) && getenv ("GSTAMD_NO_CONVERT_PACK_WIDE") == nullptr;
      for (int i = 0; wide && i < (p.pack.kind == UNPACK_SEMI ? 2 : 3); i++)
        wide = ((uintptr_t) d.p[i] % 4) == 0 && (d.stride[i] % 4) == 0;
      wide = wide && ((uintptr_t) img.data () % 16) == 0 && ((ow * 4) % 16) == 0;
      const int prow = (oh + (1 << p.pack.h_sub) - 1) >> p.pack.h_sub;
      for (int yb = 0; yb < prow; yb++)
        for (int x0 = 0; x0 < ow; x0 += 4) {
          if (wide && pack_planar_block4 (p.pack, s4, d, x0, yb))
            continue;
          pack_planar_body (p.pack, s4, d, x0, yb);
        }
      return GSTAMD_OK;
    }
  }
  if (p.out_planar) {                     /* chain -> AYUV image, then the pack kernel body over its grid */
    std::vector<uint8_t> img ((size_t) p.out_info.width * 4 * (p.out_info.height + 1));          /* + the line past the picture */
    g_extra_rows = p.pack.virtual_line;
    r = emu_convert_packed (p, in, pl, img.data (), p.out_info.width * 4, vec_ok, false);
    g_extra_rows = 0;
    if (r != GSTAMD_OK)
      return r;
    DstPlanes d;
    memset (&d, 0, sizeof (d));
    for (int i = 0; i < out->n_planes && i < 3; i++) {
      d.p[i] = dst + out->offset[i];
      d.stride[i] = out->stride[i];
    }
    const int rows = (p.out_info.height + (1 << p.pack.h_sub) - 1) >> p.pack.h_sub;
    PackPlanarParams pk = p.pack;
    if (pk.dither.on && pk.dither.method != GSTAMD_DITHER_NONE && pk.dither.method != GSTAMD_DITHER_BAYER) {
      /* launch_pack_planar_ed: k_pack_down_v, k_pack_down_h, the dither pass, then selection only */
      for (int pass = 0; pass < 2; pass++)
        for (int yb = 0; yb < rows; yb++)
          for (int x = 0; x < p.out_info.width; x++)
            if (pass == 0)
              pack_down_v_px (pk, img.data (), p.out_info.width * 4, x, yb);
            else
              pack_down_h_px (pk, img.data (), p.out_info.width * 4, x, yb);
      ed_image_host (pk.dither, img.data (), p.out_info.width * 4, p.out_info.width, p.out_info.height);
      pk = pack_select_only (pk);
    }
    for (int yb = 0; yb < rows; yb++)
      for (int x0 = 0; x0 < p.out_info.width; x0 += 4)
      {
        /* k_pack_planar: the wide block form where launch_pack_planar allows it (the emulator's buffers: image rows on 16 bytes when the
           width is a multiple of 4; plane rows judged like the launcher does) */
        bool wide = !pk.dither.on && (pk.kind == UNPACK_PLANAR || pk.kind == UNPACK_SEMI) && ((uintptr_t) img.data () % 16) == 0 && ((p.out_info.width * 4) % 16) == 0;
        for (int i = 0; wide && i < (pk.kind == UNPACK_SEMI ? 2 : 3); i++)
          wide = ((uintptr_t) d.p[i] % 4) == 0 && (d.stride[i] % 4) == 0;
        if (wide && pack_planar_block4 (pk, img.data (), p.out_info.width * 4, d, x0, yb))
          continue;
        pack_planar_body (pk, SrcImage {img.data (), p.out_info.width * 4, p.out_info.width}, d, x0, yb);
      }
    if (GSTAMD_KIND_ALPHA_PLANE (p.fout->kind) >= 0) {          /* k_pack_alpha_plane (pk: the dither-free form after an error-diffusion pass) */
      const int ai = GSTAMD_KIND_ALPHA_PLANE (p.fout->kind);
      for (int y = 0; y < p.out_info.height; y++)
        for (int x0 = 0; x0 < p.out_info.width; x0 += 4)
          pack_alpha_plane_body (pk, img.data (), p.out_info.width * 4, dst + out->offset[ai], out->stride[ai], x0, y);
    }
    return GSTAMD_OK;
  }
  r = emu_convert_packed (p, in, pl, dst + out->offset[0], out->stride[0], vec_ok, false);
  if (r == GSTAMD_OK && p.dither.on && p.dither.method != GSTAMD_DITHER_NONE && p.dither.method != GSTAMD_DITHER_BAYER)
    ed_image_host (p.dither, dst + out->offset[0], out->stride[0], p.out_info.width, p.out_info.height);          /* k_dither_verterr / k_dither_ed */
  else if (r == GSTAMD_OK && p.dither.on)               /* k_dither4 over the converted rectangle */
    for (int y = 0; y < p.out_info.height; y++)
      for (int x0 = 0; x0 < p.out_info.width; x0 += 4)
        dither_lane4 (p.dither, dst + out->offset[0], out->stride[0], p.out_info.width, p.out_info.height, x0, y);
  return r;
}

static int emu_convert_packed (const VideoPlan &p, const GstAmdVideoInfo *in, const Planes &pl, uint8_t *d0, int dstride, int vec_ok, bool rgb24)
{
  ColorParams color, none;
  memset (&none, 0, sizeof (none));
  color.matrix = p.matrix;
  color.alpha_kind = p.post.alpha_kind;
  color.alpha_value = p.post.alpha_value;
  const int *vpair = p.vpair.data ();
  if (p.passes.empty () && p.fast_pair && vec_ok) {
    /* vec_ok: 1 = shipped configuration (strip kernel, 3 line pairs per lane); 100 + K = strip kernel with K pairs;
     * 200 + K = wide kernel (LDS-staged 1024-px runs) with K pairs per wave */
    FastParams fp = emu_fast_params (p, rgb24);
    const int pairs = fp.height / 2 + 1;
    const int lay = GSTAMD_LAYOUT (fp.pack_pos[1], fp.pack_pos[2], fp.pack_pos[3]);
#define FOR_LAYOUT(CH, CALL) \
    if (lay == GSTAMD_LAYOUT (2, 1, 0)) { CALL (CH, GSTAMD_LAYOUT (2, 1, 0)) } \
    else if (lay == GSTAMD_LAYOUT (0, 1, 2)) { CALL (CH, GSTAMD_LAYOUT (0, 1, 2)) } \
    else if (lay == GSTAMD_LAYOUT (1, 2, 3)) { CALL (CH, GSTAMD_LAYOUT (1, 2, 3)) } \
    else if (lay == GSTAMD_LAYOUT (3, 2, 1)) { CALL (CH, GSTAMD_LAYOUT (3, 2, 1)) } \
    else return GSTAMD_ERR_UNSUPPORTED;
#define FOR_CH(CALL) \
    switch (p.front.chroma_h) { \
      case CHROMA_H_H2_CS: FOR_LAYOUT (CHROMA_H_H2_CS, CALL) break; \
      case CHROMA_H_H2: FOR_LAYOUT (CHROMA_H_H2, CALL) break; \
      default: FOR_LAYOUT (CHROMA_H_NONE, CALL) break; \
    }
    if (vec_ok >= 200 && fp.width >= GSTAMD_WIDE_PX / 2) {
      const int nxb = (fp.width + GSTAMD_WIDE_PX - 1) / GSTAMD_WIDE_PX;
      const int K = vec_ok - 200 > 0 ? vec_ok - 200 : 1, strips = (pairs + K - 1) / K;
      const int blocks = wide_grid_blocks (nxb, strips);
      const bool vec = (((uintptr_t) pl.p[0] | (uintptr_t) pl.p[1]) % 16) == 0 && pl.stride[0] % 16 == 0 && pl.stride[1] % 16 == 0;
      static WideLds lds;
      static WideRegs regs[64];
#define WAVE(CH, LAY) \
      for (int b = 0; b < blocks; b++) { \
        int xb, S; \
        if (!wide_block_map (b, nxb, strips, &xb, &S)) \
          continue; \
        const int xw = xb * GSTAMD_WIDE_PX, p0 = S * K, p1 = p0 + K < pairs ? p0 + K : pairs; \
        for (int lane = 0; lane < 64; lane++) { \
          wide_fetch_chroma<CH> (fp, pl, xw, p0 - 1 > fp.crow_lo ? p0 - 1 : fp.crow_lo, lane, vec, regs[lane]); \
          wide_commit_chroma<CH> (fp, xw, lane, regs[lane], lds.c[0]); \
          wide_fetch<CH> (fp, pl, xw, p0, lane, vec, regs[lane]); \
        } \
        for (int pp = p0; pp < p1; pp++) { \
          const int k = pp - p0; \
          for (int lane = 0; lane < 64; lane++) \
            wide_commit<CH> (fp, xw, lane, regs[lane], &lds, (k + 1) & 1); \
          for (int lane = 0; lane < 64; lane++) { \
            if (pp + 1 < p1) \
              wide_fetch<CH> (fp, pl, xw, pp + 1, lane, vec, regs[lane]); \
            wide_emit<CH, LAY, 0> (fp, d0, dstride, xw, pp, lane, &lds, k & 1); \
          } \
        } \
      }
      FOR_CH (WAVE)
#undef WAVE
      return GSTAMD_OK;
    }
    {
      const int K = vec_ok >= 100 && vec_ok < 200 && vec_ok - 100 > 0 ? vec_ok - 100 : 3;
      if (g_post_lut && !rgb24) {           /* launch_strip_variant<CH, GSTAMD_FAST_LUT>: the composed gamma table between pack and store */
        memcpy (fast_lut_lds, g_post_lut, 256);
        fp.lut = g_post_lut;
        fp.lut_keep = g_post_lut_keep;
        g_post_lut_done = true;
#define STRIP(CH, LAY) \
      for (int p0 = 0; p0 < pairs; p0 += K) \
        for (int x0 = 0; x0 + 4 <= fp.width; x0 += 4) \
          fast_strip<CH, LAY, GSTAMD_FAST_LUT> (fp, pl, d0, dstride, x0, p0, p0 + K < pairs ? p0 + K : pairs);
        FOR_CH (STRIP)
#undef STRIP
      } else {
#define STRIP(CH, LAY) \
      for (int p0 = 0; p0 < pairs; p0 += K) \
        for (int x0 = 0; x0 + 4 <= fp.width; x0 += 4) \
          fast_strip<CH, LAY, 0> (fp, pl, d0, dstride, x0, p0, p0 + K < pairs ? p0 + K : pairs);
      FOR_CH (STRIP)
#undef STRIP
      }
    }
#undef FOR_CH
#undef FOR_LAYOUT
    return GSTAMD_OK;
  }
  if (p.fast_420p && vec_ok && ((uintptr_t) pl.p[0] % 8) == 0 && (pl.stride[0] % 8) == 0 && ((uintptr_t) pl.p[1] % 4) == 0 && ((uintptr_t) pl.p[2] % 4) == 0 &&
      (pl.stride[1] % 4) == 0 && pl.stride[1] == pl.stride[2] && ((uintptr_t) d0 % 16) == 0 && (dstride % 16) == 0 &&
      getenv ("GSTAMD_NO_FAST420P") == nullptr) {        /* k_convert420p */
    Fast420pParams q;
    q.fp = emu_fast_params (p);
    q.y = pl.p[0];
    q.u = pl.p[p.front.u_plane];
    q.v = pl.p[p.front.v_plane];
    q.ystride = pl.stride[0];
    q.cstride = pl.stride[1];
    g_fast420p_runs++;
    for (int r = 0; r < (p.front.height + 1) / 2; r++)
      for (int x0 = 0; x0 < p.front.width; x0 += 8)
        convert420p_lane8x2 (q, d0, dstride, x0, r);
    return GSTAMD_OK;
  }
  if (p.fast_422 && vec_ok && ((uintptr_t) pl.p[0] % 16) == 0 && (pl.stride[0] % 16) == 0 && ((uintptr_t) d0 % 16) == 0 && (dstride % 16) == 0 &&
      getenv ("GSTAMD_NO_FAST422") == nullptr) {        /* k_convert422 */
    Fast422Params q;
    q.fp = emu_fast_params (p);
    q.chroma_h = p.front.chroma_h;
    fast422_selectors (p.front.pos[1], p.front.pos[2], p.front.pos[3], &q);
    g_fast422_runs++;
    for (int y = 0; y < p.front.height; y++)
      for (int x0 = 0; x0 < p.front.width; x0 += 8)
        convert422_lane8_any (q, pl.p[0] + (size_t) y * pl.stride[0], d0 + (size_t) y * dstride, x0);
    return GSTAMD_OK;
  }
  if (p.fast_422_ayuv && !g_gamma_hook && !rgb24 && ((uintptr_t) pl.p[0] % 16) == 0 && (pl.stride[0] % 16) == 0 && ((uintptr_t) d0 % 16) == 0 && (dstride % 16) == 0 &&
      getenv ("GSTAMD_NO_FAST422") == nullptr) {        /* k_convert422_ayuv */
    Fast422Params q;
    memset ((void *) &q, 0, sizeof (q));
    q.fp.width = p.front.width;
    q.fp.height = p.front.height;
    q.chroma_h = p.front.chroma_h;
    fast422_selectors (p.front.pos[1], p.front.pos[2], p.front.pos[3], &q);
    for (int y = 0; y < p.front.height; y++)
      for (int x0 = 0; x0 < p.front.width; x0 += 8)
        convert422_lane8_ayuv (q, pl.p[0] + (size_t) y * pl.stride[0], d0 + (size_t) y * dstride, x0);
    return GSTAMD_OK;
  }
  if (p.deep16 && p.passes.empty ()) {     /* k_convert16 */
    for (int y = 0; y < p.front.height; y++)
      for (int x0 = 0; x0 < p.front.width; x0 += 4)
        convert16_lane4 (p.front, pl, vpair, p.deep, p.post, d0, dstride, x0, y);
    return GSTAMD_OK;
  }
  DeepPackParams ds4;
  if (p.deep16 && !p.matrix_before_scale && !g_gamma_hook && getenv ("GSTAMD_NO_DEEP_SCALE_PACK") == nullptr && deep_scale4_plan_ok (p, &ds4) && ((uintptr_t) d0 % 4) == 0 &&
      (dstride % 4) == 0) {         /* k_deep_scale4 over its grid */
    ds4.pl = pl;
    ds4.vpair = vpair;
    ds4.sh.offset = p.passes[0].offset.data (), ds4.sh.taps = p.passes[0].taps.data ();
    ds4.sv.offset = p.passes[1].offset.data (), ds4.sv.taps = p.passes[1].taps.data ();
    g_emu_deep_pack_runs++;
    for (int y = 0; y < ds4.out_h; y++)
      for (int l = 0; l < (ds4.out_w / 4 + 63) / 64 * 64; l++)
        deep_scale4_any (deep_pack_variant (p.front), ds4, p.deep, p.post, d0, dstride, 4 * l, y);
    return GSTAMD_OK;
  }
  if (p.deep16 && !p.matrix_before_scale) {        /* convert_deep_scaled: k_front16, k_scale16 ..., k_scale16_final */
    const int in_w = p.front.width, in_h = p.front.height;
    std::vector<uint8_t> a ((size_t) in_w * in_h * 8), b;
    for (int y = 0; y < in_h; y++)
      for (int x0 = 0; x0 < in_w; x0 += 4)
        front16_lane4 (p.front, pl, vpair, a.data (), in_w * 8, x0, y);
    Deep16Image cur = {a.data (), in_w * 8, in_w, in_h};
    size_t first = 0;
    if (p.passes.size () == 2 && p.passes[0].horizontal && deep_front4_variant (p.front) >= 0 && !getenv ("EMU_NO_CONVERT16_FAST")) {
      /* k_front_hscale16: the front inside the first, horizontal pass */
      ScaleDev sd16;
      memset (&sd16, 0, sizeof (sd16));
      sd16.kind = p.passes[0].kind;
      sd16.n_taps = p.passes[0].n_taps;
      sd16.offset = p.passes[0].offset.data ();
      sd16.taps = p.passes[0].taps.data ();
      const int ow = p.passes[0].out_size;
      b.assign ((size_t) ow * in_h * 8, 0);
      for (int y = 0; y < in_h; y++)
        for (int x = 0; x < (ow + 255) / 256 * 256; x++)
          front_hscale16_any (deep_front4_variant (p.front), p.front, pl, vpair, sd16, b.data (), ow * 8, ow, x, y);
      a.swap (b);
      cur.p = a.data (), cur.stride = ow * 8, cur.width = ow, cur.height = in_h;
      first = 1;
    }
    for (size_t i = first; i < p.passes.size (); i++) {
      const bool hz = p.passes[i].horizontal, last = i + 1 == p.passes.size ();
      const int ow = hz ? p.passes[i].out_size : cur.width, oh = hz ? cur.height : p.passes[i].out_size;
      ScaleDev sd16;
      memset (&sd16, 0, sizeof (sd16));
      sd16.kind = p.passes[i].kind;
      sd16.n_taps = p.passes[i].n_taps;
      sd16.offset = p.passes[i].offset.data ();
      sd16.taps = p.passes[i].taps.data ();
      if (!last)
        b.assign ((size_t) ow * oh * 8, 0);
      for (int y = 0; y < oh; y++)
        for (int x = 0; x < ow; x++) {
          if (last)
            scale16_final_lane (cur, sd16, hz, p.deep, p.post, d0, dstride, ow, oh, x, y);
          else
            scale16_lane (cur, sd16, hz, b.data (), ow * 8, ow, oh, x, y);
        }
      cur.p = b.data (), cur.stride = ow * 8, cur.width = ow, cur.height = oh;
    }
    return GSTAMD_OK;
  }
  /* launch_convert's 16-byte path for 4-byte packed sources */
  if (p.passes.empty () && vec_ok && p.front.kind == UNPACK_PACKED4 && ((uintptr_t) d0 % 16) == 0 && (dstride % 16) == 0 && ((uintptr_t) pl.p[0] % 16) == 0 &&
      (pl.stride[0] % 16) == 0)
    vec_ok = 2;
  else if (p.passes.empty () && p.front.kind == UNPACK_PACKED4 && vec_ok == 2)
    vec_ok = 1;
  if (p.passes.empty () && g_gamma_hook) {                /* k_convert_gamma: the same body with the gamma chain as its per-pixel step */
    const int spans = (p.front.width + K1_PX - 1) / K1_PX;
    GammaChainFn fn;
    fn.g = *g_gamma_hook;
    g_gamma_fused_runs++;
    for (int y = 0; y < p.front.height; y++)
      for (int s = 0; s < spans; s++) {
        const int *pp = p.post.pack_pos;
        switch (p.front.chroma_h) {
          case CHROMA_H_H2_CS:
            convert_body<CHROMA_H_H2_CS, GammaChainFn> (p.front, pl, vpair, color, pp[0], pp[1], pp[2], pp[3], d0, dstride, spans, vec_ok, s, y, fn);
            break;
          case CHROMA_H_H2:
            convert_body<CHROMA_H_H2, GammaChainFn> (p.front, pl, vpair, color, pp[0], pp[1], pp[2], pp[3], d0, dstride, spans, vec_ok, s, y, fn);
            break;
          default:
            convert_body<CHROMA_H_NONE, GammaChainFn> (p.front, pl, vpair, color, pp[0], pp[1], pp[2], pp[3], d0, dstride, spans, vec_ok, s, y, fn);
            break;
        }
      }
    return GSTAMD_OK;
  }
  if (p.passes.empty () && p.front.kind == UNPACK_PACKED3 && p.front.hi_depth == 0 && color.matrix.kind == MATRIX_NONE && color.alpha_kind == ALPHA_NONE &&
      !rgb24 && emu_swizzle34 (3, p.front.pos, 4, p.post.pack_pos, pl.p[0], pl.stride[0], d0, dstride, p.front.width, p.front.height))
    return GSTAMD_OK;
  if (p.passes.empty () && vec_ok == 2 && p.front.hi_depth == 0 && color.matrix.kind == MATRIX_NONE && color.alpha_kind == ALPHA_NONE) {
    /* k_swizzle4: a byte permutation per pixel, four pixels per lane */
    const uint32_t sel = swizzle4_selector (p.front.pos, p.post.pack_pos);
    g_swizzle4_runs++;
    for (int y = 0; y < p.front.height; y++) {
      const uint32_t *sp = (const uint32_t *) (pl.p[0] + (size_t) y * pl.stride[0]);
      uint32_t *dp = (uint32_t *) (d0 + (size_t) y * dstride);
      for (int x = 0; x < p.front.width; x++)
        dp[x] = swizzle4_px (sp[x], sel);
    }
    return GSTAMD_OK;
  }
  if (p.passes.empty ()) {
    const int spans = (p.front.width + K1_PX - 1) / K1_PX;
    for (int y = 0; y < p.front.height + g_extra_rows; y++)         /* launch_convert's extra_rows */
      for (int s = 0; s < spans; s++) {
        const int *pp = p.post.pack_pos;
        switch (p.front.chroma_h) {
          case CHROMA_H_H2_CS:
            convert_body<CHROMA_H_H2_CS> (p.front, pl, vpair, color, pp[0], pp[1], pp[2], pp[3], d0, dstride, spans, vec_ok, s, y);
            break;
          case CHROMA_H_H2:
            convert_body<CHROMA_H_H2> (p.front, pl, vpair, color, pp[0], pp[1], pp[2], pp[3], d0, dstride, spans, vec_ok, s, y);
            break;
          default:
            convert_body<CHROMA_H_NONE> (p.front, pl, vpair, color, pp[0], pp[1], pp[2], pp[3], d0, dstride, spans, vec_ok, s, y);
            break;
        }
      }
    return GSTAMD_OK;
  }
  const ColorParams &pre = p.matrix_before_scale ? color : none;
  const ColorParams &post = p.matrix_before_scale ? none : color;
  ScaleDev sd[2];
  for (size_t i = 0; i < p.passes.size (); i++) {
    sd[i].kind = p.passes[i].kind;
    sd[i].n_taps = p.passes[i].n_taps;
    sd[i].inc = p.passes[i].inc;
    sd[i].offset = p.passes[i].offset.data ();
    sd[i].taps = p.passes[i].taps.data ();
    sd[i].tapw = p.passes[i].dot4_ok ? p.passes[i].tapw.data () : nullptr;
    sd[i].nw = p.passes[i].nw;
    sd[i].nw4 = p.passes[i].nw4;
  }
  SrcFront sf;
  sf.f = p.front;
  sf.pl = pl;
  sf.vpair = vpair;
  sf.pre = pre;
  sf.vec_ok = vec_ok ? 1 : 0;
  auto mk = [&](uint8_t *ptr, int stride, bool fin) {
    Dst d;
    d.p = ptr;
    d.stride = stride;
    d.final = fin;
    d.post = fin ? post : none;
    memcpy (d.pack_pos, p.post.pack_pos, sizeof (d.pack_pos));
    return d;
  };
  PostFast pf, pf_none;
  memset (&pf_none, 0, sizeof (pf_none));
  pf.use = p.fast_post ? 1 : 0;
  pf.fp = emu_fast_params (p);
  if (p.deep16) {
    /* convert_deep_scaled, the picture grows: the convert stage first (k_convert16 into an 8-bit unpack-order image), then the
     * 8-bit scalers from that image */
    const int in_w = p.front.width, in_h = p.front.height;
    std::vector<uint8_t> a ((size_t) in_w * in_h * 4), b;
    PostParams mid = p.post;
    for (int i = 0; i < 4; i++)
      mid.pack_pos[i] = i;
    for (int y = 0; y < in_h; y++)
      for (int x0 = 0; x0 < in_w; x0 += 4)
        convert16_lane4 (p.front, pl, vpair, p.deep, mid, a.data (), in_w * 4, x0, y);
    SrcImage si;
    si.p = a.data ();
    si.stride = in_w * 4;
    si.width = in_w;
    int sh = in_h;
    Dst fin = mk (d0, dstride, true);
    fin.post = none;
    for (size_t i = 0; i < p.passes.size (); i++) {
      const bool hz = p.passes[i].horizontal, last = i + 1 == p.passes.size ();
      const int ow = hz ? p.passes[i].out_size : si.width, oh = hz ? sh : p.passes[i].out_size;
      if (!last)
        b.assign ((size_t) ow * oh * 4, 0);
      run_scale (hz, si, sd[i], last ? fin : mk (b.data (), ow * 4, false), ow, oh, p.passes[i].max_span, hz ? pass_tile_geom (p.passes[i]) : TileGeom {0, 0},
          pf_none);
      si.p = b.data (), si.stride = ow * 4, si.width = ow, sh = oh;
    }
    return GSTAMD_OK;
  }
  const auto small_kind = [](int k) { return k == SCALE_NEAREST || k == SCALE_2TAP; };
  if (p.passes.size () == 2 && small_kind (p.passes[0].kind) && small_kind (p.passes[1].kind)) {
    const bool h_first = p.passes[0].horizontal;
    const ScaleDev &sh = h_first ? sd[0] : sd[1], &sv = h_first ? sd[1] : sd[0];
    const Dst d = mk (d0, dstride, true);
    const int span = p.passes[h_first ? 0 : 1].max_span;
    const TileGeom g = pass_tile_geom (p.passes[h_first ? 0 : 1]);
    int bil_yl = 0;
    const int bil_tw = vec_ok >= 400 ? vec_ok - 400 : bil_pick_tile (p.out_info.width, p.passes[0].inc, &bil_yl);     /* 400 + w: tiles of w outputs */
    if (vec_ok >= 400)
      bil_yl = bil_ylen (p.out_info.width, p.passes[0].inc, bil_tw);
    const bool bil_ayuv = bilinear420_ayuv_plan (p) && getenv ("GSTAMD_NO_BILINEAR_AYUV") == nullptr;          /* bilinear420_params of capi_video.cpp */
    const bool bil_planar = p.front.kind == UNPACK_PLANAR;
    const bool bil_planar_ok = bil_planar && ((uintptr_t) pl.p[0] % 16) == 0 && pl.stride[0] % 16 == 0 && ((uintptr_t) pl.p[1] % 8) == 0 &&
        ((uintptr_t) pl.p[2] % 8) == 0 && pl.stride[1] % 8 == 0 && pl.stride[2] % 8 == 0 && (p.front.width % 16) == 0;
    if (h_first && p.passes[0].kind == SCALE_2TAP && p.passes[1].kind == SCALE_2TAP && (p.front.kind == UNPACK_SEMI || bil_planar_ok) && p.front.w_sub == 1 &&
        p.front.h_sub == 1 && !p.matrix_before_scale && (p.fast_post || bil_ayuv) && (!p.out_planar || bil_ayuv) && p.front.chroma_v2 != 2 && bil_tw > 0 && bil_yl > 0 && vec_ok != 300) {
      /* k_bilinear420 (video_bilinear_fast.h); vec_ok == 300 selects the generic tile kernel below instead */
      BilParams bp;
      bp.fp = pf.fp;
      bp.fp.ayuv = bil_ayuv ? (p.matrix.kind == MATRIX_NONE ? 1 : 2) : 0;
      bp.fp.m8 = p.matrix;
      if (bil_ayuv)
        g_bil_ayuv_runs++;
      bp.out_w = p.out_info.width;
      bp.out_h = p.out_info.height;
      bp.inc = p.passes[0].inc;
      bp.tile_w = bil_tw;
      bp.ylen = bil_yl;
      bp.voffset = sd[1].offset;
      bp.vtaps = sd[1].taps;
      bp.vpair = p.front.chroma_v2 ? vpair : nullptr;
      const bool vec = bil_planar ? true : (((uintptr_t) pl.p[0] | (uintptr_t) pl.p[1]) % 16) == 0 && pl.stride[0] % 16 == 0 && pl.stride[1] % 16 == 0;
      bp.planar = bil_planar ? 1 : 0;
      bp.u_plane = p.front.u_plane;
      bp.v_plane = p.front.v_plane;
      g_bil_runs++;
      const int lay = bp.fp.ayuv ? GSTAMD_LAYOUT_AYUV : GSTAMD_LAYOUT (bp.fp.pack_pos[1], bp.fp.pack_pos[2], bp.fp.pack_pos[3]);
      std::vector<uint32_t> lds_w (bil_lds_words (bp.ylen));
      const BilLds lds = bil_lds (lds_w.data (), bp.ylen);
#define BIL_L(CH, pr, pg, pb) if (lay == GSTAMD_LAYOUT (pr, pg, pb)) bil_emit<CH, GSTAMD_LAYOUT (pr, pg, pb)> (bp, d0, dstride, t0, t1, y, r0, lane, &lds);
#define BIL(CH) { BIL_L (CH, 2, 1, 0) BIL_L (CH, 0, 1, 2) BIL_L (CH, 1, 2, 3) BIL_L (CH, 3, 2, 1) BIL_L (CH, 0, 0, 4) }
      bp.regular_pairs = 0;
      bp.rows = 0;
      {
        /* k_bilinear420_rows (video_bilinear_rows.h): the gate of capi_video.cpp, here against the pairing table itself */
        bool fits = p.front.chroma_v2 && vec && (p.front.width % 16) == 0 && getenv ("EMU_NO_BILINEAR_ROWS") == nullptr;
        if (getenv ("EMU_BILR_DEBUG"))
          fprintf (stderr, "bilr gate: v2 %d vec %d w %d\n", (int) p.front.chroma_v2, (int) vec, p.front.width);
        for (int y = 0; y < bp.out_h && fits; y++) {
          fits = bilr_window_matches (bp, (int) bp.voffset[y]);
          if (!fits && getenv ("EMU_BILR_DEBUG")) {
            int ra, rb, role, wa, wb, wc;
            bilr_window (bp, (int) bp.voffset[y], &wa, &wb, &wc);
            fprintf (stderr, "bilr gate: y %d r0 %d window %d %d %d\n", y, (int) bp.voffset[y], wa, wb, wc);
            for (int l = 0; l < 2; l++) {
              bil_rows (bp, (int) bp.voffset[y] + l, &ra, &rb, &role);
              fprintf (stderr, "   line %d: ra %d rb %d role %d\n", (int) bp.voffset[y] + l, ra, rb, role);
            }
          }
        }
        int rows_ylen = 0;
        bp.rows_tile_w = getenv ("EMU_BIL_ROWS_TILE") ? atoi (getenv ("EMU_BIL_ROWS_TILE")) : bilr_pick_tile (bp.out_w, bp.inc, &rows_ylen);
        if (getenv ("EMU_BIL_ROWS_TILE"))
          rows_ylen = bil_ylen (bp.out_w, bp.inc, bp.rows_tile_w);
        fits = fits && bp.rows_tile_w > 0 && rows_ylen > 0;
        if (fits)
          bp.rows = getenv ("EMU_BIL_ROWS") ? atoi (getenv ("EMU_BIL_ROWS")) : 4;
      }
      if (bp.rows != 0 && p.front.chroma_v2 && getenv ("EMU_NO_BILINEAR_HALF") == nullptr) {
        /* k_bilinear420_half (video_bilinear_half.h): capi_video.cpp's gate - the pairing table is the closed form, the halving exact - and
         * bilinear420_half_usable's alignment rules */
        BilParams hp = bp;
        hp.regular_pairs = 1;
        bool regular = true;
        for (int y = 0; y < bp.out_h && regular; y++)
          for (int l = 0; l < 2 && regular; l++) {
            const int line = (int) bp.voffset[y] + l;
            int ra, rb, role;
            bil_rows (hp, line, &ra, &rb, &role);
            const int e0 = vpair[2 * line], ta = vpair_row (e0), trole = vpair_role (e0), tb = vpair[2 * line + 1];
            regular = ta == ra && tb == rb && (ra == rb || trole == role);
          }
        const bool al = ((uintptr_t) pl.p[0] % 16) == 0 && pl.stride[0] % 16 == 0 && ((uintptr_t) d0 % 16) == 0 && dstride % 16 == 0 &&
            (bil_planar ? bil_planar_ok : (((uintptr_t) pl.p[1] % 16) == 0 && pl.stride[1] % 16 == 0));
        if (regular && al && bilh_plan_ok (hp, bp.voffset, bp.vtaps)) {
          g_bilh_runs++;
          const int tiles = (hp.fp.width + BILH_TILE_SRC - 1) / BILH_TILE_SRC;
          const int rows = getenv ("EMU_BIL_HALF_ROWS") ? atoi (getenv ("EMU_BIL_HALF_ROWS")) : 5;
          hp.strips = bilh_strips (hp.out_h, rows, tiles, rows < 0 ? -rows : 0);
          const auto p1_of = [&] (int y) { return (int) hp.vtaps[(size_t) y * 2 + 1]; };
          const auto put = [&] (uint8_t *d, bool active, int half, uint32_t a, uint32_t b, uint32_t e, uint32_t f) {
            if (active)
              store16_stream (d + 16 * half, a, b, e, f);
          };
#define BILH_L(CH, pr, pg, pb) if (lay == GSTAMD_LAYOUT (pr, pg, pb)) bilh_strip<CH, GSTAMD_LAYOUT (pr, pg, pb)> (hp, pl, d0, dstride, x0, y0, y1, p1_of, put);
#define BILH(CH) { BILH_L (CH, 2, 1, 0) BILH_L (CH, 0, 1, 2) BILH_L (CH, 1, 2, 3) BILH_L (CH, 3, 2, 1) BILH_L (CH, 0, 0, 4) }
          for (int g = 0; g < hp.strips; g++) {
            const int y0 = (int) ((unsigned) g * (unsigned) hp.out_h / (unsigned) hp.strips);
            const int y1 = (int) ((unsigned) (g + 1) * (unsigned) hp.out_h / (unsigned) hp.strips);
            for (int tile = 0; tile < tiles; tile++)
              for (int lane = 0; lane < 64; lane++) {
                const int x0 = tile * BILH_TILE_SRC + BILH_SRC_PER_LANE * lane;
                if (p.front.chroma_h == CHROMA_H_H2_CS) BILH (CHROMA_H_H2_CS)
                else if (p.front.chroma_h == CHROMA_H_H2) BILH (CHROMA_H_H2)
                else BILH (CHROMA_H_NONE)
              }
          }
#undef BILH
#undef BILH_L
          return GSTAMD_OK;
        }
      }
      if (bp.rows != 0) {
        g_bilr_runs++;
        static BilrState st[64];
        static BilrLane lc[64];
        static uint32_t q[64][4][2];
        std::vector<uint8_t> lds (bilr_lds_bytes ());
#define BILR_L(CH, pr, pg, pb) if (lay == GSTAMD_LAYOUT (pr, pg, pb)) { \
          if (phase == 0) layout_init<GSTAMD_LAYOUT (pr, pg, pb)> (q[lane]); \
          else if (bp.rows_tile_w > 256) bilr_emit_row<GSTAMD_LAYOUT (pr, pg, pb), 3> (bp, lc[lane], lds.data (), d0, dstride, y, (int) bp.vtaps[(size_t) y * 2 + 1], q[lane][0]); \
          else bilr_emit_row<GSTAMD_LAYOUT (pr, pg, pb), 2> (bp, lc[lane], lds.data (), d0, dstride, y, (int) bp.vtaps[(size_t) y * 2 + 1], q[lane][0]); }
#define BILR_EMIT() { BILR_L (0, 2, 1, 0) BILR_L (0, 0, 1, 2) BILR_L (0, 1, 2, 3) BILR_L (0, 3, 2, 1) BILR_L (0, 0, 0, 4) }
        /* EMU_BIL_ROWS < 0: the balanced strips of the launcher for a device with -EMU_BIL_ROWS wave slots */
        bp.strips = bilr_strips (bp.out_h, bp.rows, (bp.out_w + bp.rows_tile_w - 1) / bp.rows_tile_w, bp.rows < 0 ? -bp.rows : 0);
        for (int g = 0; g < bp.strips; g++)
          for (int t0 = 0; t0 < bp.out_w; t0 += bp.rows_tile_w) {
            const int t1 = t0 + bp.rows_tile_w < bp.out_w ? t0 + bp.rows_tile_w : bp.out_w;
            const int y0 = (int) ((unsigned) g * (unsigned) bp.out_h / (unsigned) bp.strips);
            const int y1 = (int) ((unsigned) (g + 1) * (unsigned) bp.out_h / (unsigned) bp.strips);
            int x_lo, x_hi, k_lo, k_hi, y = 0, phase = 0;
            bil_span (bp, t0, t1, &x_lo, &x_hi, &k_lo, &k_hi);
            const int xa = x_lo & ~15;
            for (int lane = 0; lane < 64; lane++) {
              if (bp.rows_tile_w > 256)
                bilr_lane_setup<3> (bp, t0, t1, xa, lane, lc[lane]);
              else
                bilr_lane_setup<2> (bp, t0, t1, xa, lane, lc[lane]);
              bilr_state_init (st[lane]);
              BILR_EMIT ()
            }
            phase = 1;
            static BilrReq rq[64];
            for (int lane = 0; lane < 64; lane++)
              bilr_request (bp, pl, st[lane], (int) bp.voffset[y0], xa, x_hi, lane, rq[lane]);
            for (y = y0; y < y1; y++) {
              for (int lane = 0; lane < 64; lane++) {
                if (p.front.chroma_h == CHROMA_H_H2_CS)
                  bilr_install<CHROMA_H_H2_CS> (bp, pl, st[lane], rq[lane], (int) bp.voffset[y], xa, x_hi, lane, lds.data ());
                else if (p.front.chroma_h == CHROMA_H_H2)
                  bilr_install<CHROMA_H_H2> (bp, pl, st[lane], rq[lane], (int) bp.voffset[y], xa, x_hi, lane, lds.data ());
                else
                  bilr_install<CHROMA_H_NONE> (bp, pl, st[lane], rq[lane], (int) bp.voffset[y], xa, x_hi, lane, lds.data ());
                if (y + 1 < y1)
                  bilr_request (bp, pl, st[lane], (int) bp.voffset[y + 1], xa, x_hi, lane, rq[lane]);
              }
              for (int lane = 0; lane < 64; lane++)
                BILR_EMIT ()
            }
          }
#undef BILR_EMIT
#undef BILR_L
        return GSTAMD_OK;
      }
      static BilRegs regs[64];
      for (int y = 0; y < bp.out_h; y++)
        for (int t0 = 0; t0 < bp.out_w; t0 += bp.tile_w) {
          const int t1 = t0 + bp.tile_w < bp.out_w ? t0 + bp.tile_w : bp.out_w, r0 = (int) bp.voffset[y];
          for (int lane = 0; lane < 64; lane++) {
            bil_fetch (bp, pl, t0, t1, r0, lane, vec, regs[lane]);
            bil_commit (bp, t0, t1, lane, regs[lane], &lds);
          }
          for (int lane = 0; lane < 64; lane++) {
            if (p.front.chroma_h == CHROMA_H_H2_CS) BIL (CHROMA_H_H2_CS)
            else if (p.front.chroma_h == CHROMA_H_H2) BIL (CHROMA_H_H2)
            else BIL (CHROMA_H_NONE)
          }
        }
#undef BIL
#undef BIL_L
      return GSTAMD_OK;
    }
    if (p.matrix_before_scale && p.front.kind != UNPACK_PACKED4 && p.front.hi_depth == 0 && ((uintptr_t) d0 % 4) == 0 && (dstride % 4) == 0 &&
        !getenv ("EMU_NO_BILINEAR4")) {
      /* enlarging from planes / packed 4:2:2: k_convert at the source's size into an A, c1, c2, c3 image, then k_bilinear4_rows from it */
      const int in_w = p.front.width, in_h = p.front.height;
      std::vector<uint32_t> img ((size_t) in_w * in_h);
      if (p.fast_pre && vec_ok && ((uintptr_t) pl.p[0] % 4) == 0 && (pl.stride[0] % 4) == 0 && ((uintptr_t) pl.p[1] % 4) == 0 && (pl.stride[1] % 4) == 0 &&
          !getenv ("GSTAMD_NO_FAST_PRE")) {
        /* capi_video.cpp: the line-pair kernel (k_convert_strip, byte order A, R, G, B) makes the source-size image */
        FastParams fp;
        fp.width = in_w, fp.height = in_h;
        const int ident[4] = {0, 1, 2, 3};
        fast_params_finish (fp, p.matrix.p, ident, p.front.u_plane);
        fp.crow_lo = -(p.rect.in_y >> 1);
        fp.crow_hi = ((p.rect.in_maxh + 1) >> 1) - 1 - (p.rect.in_y >> 1);
        fp.store_policy = 1;
        const int pairs = fp.height / 2 + 1, K = 3;
        uint8_t *im = (uint8_t *) img.data ();
#define STRIP_PRE(CH) \
        for (int p0 = 0; p0 < pairs; p0 += K) \
          for (int x0 = 0; x0 + 4 <= fp.width; x0 += 4) \
            fast_strip<CH, GSTAMD_LAYOUT (1, 2, 3), 0> (fp, pl, im, in_w * 4, x0, p0, p0 + K < pairs ? p0 + K : pairs);
        if (p.front.chroma_h == CHROMA_H_H2_CS) { STRIP_PRE (CHROMA_H_H2_CS) }
        else if (p.front.chroma_h == CHROMA_H_H2) { STRIP_PRE (CHROMA_H_H2) }
        else { STRIP_PRE (CHROMA_H_NONE) }
#undef STRIP_PRE
      } else
      for (int y = 0; y < in_h; y++)
        for (int x = 0; x < in_w; x++)
          img[(size_t) y * in_w + x] = sf.at (x, y);
      Bil4Params b;
      memset ((void *) &b, 0, sizeof (b));
      b.src = (const uint8_t *) img.data (), b.sstride = in_w * 4, b.src_w = in_w, b.src_h = in_h;
      b.sel_in = 0x03020100u;
      b.sh = sh, b.sv = sv, b.h_first = h_first ? 1 : 0;
      b.out_w = p.out_info.width, b.out_h = p.out_info.height, b.rows = 4;
      PostFast pf_none;
      memset ((void *) &pf_none, 0, sizeof (pf_none));
      if (emu_bilinear4_up (b, d, pf_none))
        return GSTAMD_OK;
      for (int y0 = 0; y0 < b.out_h; y0 += b.rows)
        for (int x0 = 0; x0 < b.out_w; x0 += 4)
          bilinear4_rows_lane (b, d, pf_none, x0, y0);
      return GSTAMD_OK;
    }
    PlanePlan raw4;
    if (plane_raw4_plan (p, &raw4) && getenv ("GSTAMD_NO_PLANE_QUAD") == nullptr) {          /* k_plane_quad on 4-byte pixels */
      PlaneJob J;
      memset ((void *) &J, 0, sizeof (J));
      J.kind = PLANE_SCALE;
      J.s = {pl.p[0], pl.stride[0], 4, 0};
      J.d = {d0, dstride, 4};
      J.iw = raw4.iw, J.ih = raw4.ih, J.ow = raw4.ow, J.oh = raw4.oh;
      J.n_pass = 2;
      J.h_first = h_first ? 1 : 0;
      J.pass[h_first ? 0 : 1] = sh, J.pass[h_first ? 1 : 0] = sv;
      J.dstep = getenv ("GSTAMD_PLANE_QUAD_NO_DSTEP") ? 0 : plane_quad_dstep (raw4);
      J.quad = 1 + QUAD_8;
      g_emu_quad_runs++;
      const int rows = getenv ("GSTAMD_PLANE_QUAD_ROWS") ? atoi (getenv ("GSTAMD_PLANE_QUAD_ROWS")) : 3;
      const int lanes = (((J.ow * 4 + 7) / 8 + 63) / 64) * 64;
      for (int y0 = 0; y0 < J.oh; y0 += rows)
        for (int lane = 0; lane < lanes; lane++)
          plane_rows_body (J, lane, y0, rows);
      return GSTAMD_OK;
    }
    if (p.front.kind == UNPACK_PACKED4 && p.front.hi_depth == 0 && sf.pre.matrix.kind == MATRIX_NONE && sf.pre.alpha_kind == ALPHA_NONE &&
        ((uintptr_t) pl.p[0] % 4) == 0 && (pl.stride[0] % 4) == 0 && ((uintptr_t) d0 % 4) == 0 && (dstride % 4) == 0 && !getenv ("EMU_NO_BILINEAR4")) {
      /* k_bilinear4_rows: four outputs per lane, four rows per lane */
      Bil4Params b;
      memset ((void *) &b, 0, sizeof (b));
      b.src = pl.p[0], b.sstride = pl.stride[0], b.src_w = p.front.width, b.src_h = p.front.height;
      b.sel_in = (uint32_t) p.front.pos[0] | ((uint32_t) p.front.pos[1] << 8) | ((uint32_t) p.front.pos[2] << 16) | ((uint32_t) p.front.pos[3] << 24);
      b.sh = sh, b.sv = sv, b.h_first = h_first ? 1 : 0;
      b.out_w = p.out_info.width, b.out_h = p.out_info.height, b.rows = 4;
      if (emu_bilinear4_up (b, d, pf))
        return GSTAMD_OK;
      for (int y0 = 0; y0 < b.out_h; y0 += b.rows)
        for (int x0 = 0; x0 < b.out_w; x0 += 4)
          bilinear4_rows_lane (b, d, pf, x0, y0);
      return GSTAMD_OK;
    }
    if (p.front.kind == UNPACK_PACKED422 && p.front.hi_depth == 0 && sf.pre.matrix.kind == MATRIX_NONE && sf.pre.alpha_kind == ALPHA_NONE &&
        ((uintptr_t) d0 % 4) == 0 && (dstride % 4) == 0 && !getenv ("EMU_NO_BILINEAR4")) {
      /* k_bilinear422_rows */
      Bil4Params b;
      memset ((void *) &b, 0, sizeof (b));
      b.src = pl.p[0], b.sstride = pl.stride[0], b.src_w = p.front.width, b.src_h = p.front.height;
      b.pos1 = p.front.pos[1], b.pos2 = p.front.pos[2], b.pos3 = p.front.pos[3], b.chroma_h = p.front.chroma_h, b.swap_k = p.front.swap_k;
      b.sh = sh, b.sv = sv, b.h_first = h_first ? 1 : 0;
      b.out_w = p.out_info.width, b.out_h = p.out_info.height, b.rows = 2;
      for (int y0 = 0; y0 < b.out_h; y0 += b.rows)
        for (int x0 = 0; x0 < b.out_w; x0 += 4)
          bilinear4_rows_lane<1> (b, d, pf, x0, y0);
      return GSTAMD_OK;
    }
    if (g.tile_w > 0 && g.lds_px * 8 <= 16384) {      /* k_scale2x2_wave */
      std::vector<uint32_t> la (g.lds_px), lb (g.lds_px);
      for (int y = 0; y < p.out_info.height; y++)
        for (int t0 = 0; t0 < p.out_info.width; t0 += g.tile_w) {
          const int t1 = t0 + g.tile_w < p.out_info.width ? t0 + g.tile_w : p.out_info.width;
          int lo, hi;
          hscale_span (sh, t0, t1, &lo, &hi);
          const int xa = lo & ~7, ya = (int) sv.offset[y];
          for (int lane = 0; lane < 64; lane++) {
            tile_stage_row (sf, la.data (), xa, hi, ya, lane, emu_packed_ok (sf));
            if (sv.kind == SCALE_2TAP)
              tile_stage_row (sf, lb.data (), xa, hi, ya + 1, lane, emu_packed_ok (sf));
          }
          for (int lane = 0; lane < 64; lane++)
            scale2x2_tile_lane (la.data (), lb.data (), xa, sh, sv, h_first ? 1 : 0, d, pf, t0, t1, y, lane);
        }
      return GSTAMD_OK;
    }
    if (span <= 6144) {                    /* k_scale2x2_lds */
      std::vector<uint32_t> la (6144), lb (6144);
      for (int y = 0; y < p.out_info.height; y++)
        for (int t0 = 0; t0 < p.out_info.width; t0 += 256) {
          const int t1 = t0 + 256 < p.out_info.width ? t0 + 256 : p.out_info.width;
          int lo, hi;
          hscale_span (sh, t0, t1, &lo, &hi);
          const int ya = (int) sv.offset[y];
          for (int tid = 0; tid < 256; tid++) {
            sf.stage (la.data (), lo, hi, ya, tid, 256);
            if (sv.kind == SCALE_2TAP)
              sf.stage (lb.data (), lo, hi, ya + 1, tid, 256);
          }
          for (int x = t0; x < t1; x++)
            d.put (x, y, scale2x2_from_lds (la.data (), lb.data (), lo, sh, sv, h_first ? 1 : 0, x, y));
        }
      return GSTAMD_OK;
    }
    for (int y = 0; y < p.out_info.height; y++)
      for (int x = 0; x < p.out_info.width; x++)
        scale2x2_body<SrcFront> (sf, sh, sv, h_first ? 1 : 0, d, p.out_info.width, p.out_info.height, x, y);
    return GSTAMD_OK;
  }
  /* convert_to_packed's raw4: an identity-unpack 4-byte source without a colour step before the scaler goes to the image kernels */
  const bool raw4 = p.front.kind == UNPACK_PACKED4 && p.front.pos[0] == 0 && p.front.pos[1] == 1 && p.front.pos[2] == 2 && p.front.pos[3] == 3 &&
      sf.pre.matrix.kind == MATRIX_NONE && sf.pre.alpha_kind == ALPHA_NONE && ((uintptr_t) pl.p[0] % 4) == 0 && (pl.stride[0] % 4) == 0;
  SrcImage raw_img;
  raw_img.p = pl.p[0];
  raw_img.stride = pl.stride[0];
  raw_img.width = p.front.width;
  if (p.passes.size () == 1) {
    if (raw4)
      run_scale (p.passes[0].horizontal, raw_img, sd[0], mk (d0, dstride, true), p.out_info.width, p.out_info.height, p.passes[0].max_span,
          p.passes[0].horizontal ? pass_tile_geom (p.passes[0]) : TileGeom {0, 0}, pf);
    else
    run_scale (p.passes[0].horizontal, sf, sd[0], mk (d0, dstride, true), p.out_info.width, p.out_info.height, p.passes[0].max_span,
        p.passes[0].horizontal ? pass_tile_geom (p.passes[0]) : TileGeom {0, 0}, pf);
    return GSTAMD_OK;
  }
  const ScalePass &s0 = p.passes[0];
  const int tw = s0.horizontal ? s0.out_size : p.in_info.width, th = s0.horizontal ? p.in_info.height : s0.out_size;
  std::vector<uint8_t> tmp ((size_t) tw * 4 * (th + 1));
  const Dst final_dst = mk (d0, dstride, true);
  if (emu_scale_col (p, sf, final_dst, pf))
    return GSTAMD_OK;
  const int reg = emu_hscale420_reg (p, sf, sd[0], tmp.data (), tw, &final_dst, &pf);
  if (reg == 2)
    return GSTAMD_OK;
  if (!reg && raw4)
    run_scale (s0.horizontal, raw_img, sd[0], mk (tmp.data (), tw * 4, false), tw, th, s0.max_span, s0.horizontal ? pass_tile_geom (s0) : TileGeom {0, 0}, pf_none);
  else if (!reg)
  run_scale (s0.horizontal, sf, sd[0], mk (tmp.data (), tw * 4, false), tw, th, s0.max_span,
      s0.horizontal ? pass_tile_geom (s0) : TileGeom {0, 0}, pf_none);
  SrcImage si;
  si.p = tmp.data ();
  si.stride = tw * 4;
  si.width = tw;
  run_scale (p.passes[1].horizontal, si, sd[1], mk (d0, dstride, true), p.out_info.width, p.out_info.height, p.passes[1].max_span,
      p.passes[1].horizontal ? pass_tile_geom (p.passes[1]) : TileGeom {0, 0}, pf);
  return GSTAMD_OK;
}


// gstamd_video_test_pattern_frame of video_testsrc.hip: k_test_pattern's body over the frame, then the conversion of the painted image
#include "../../gstreamer_amd/csrc/video_testsrc.h"
extern "C" int emu_video_test_pattern (const GstAmdVideoInfo *info, int pattern, uint32_t fg, uint32_t bg, uint64_t n_frames, uint8_t *dst, char *desc, int desc_len)
{
  if (!test_pattern_built (pattern))
    return GSTAMD_ERR_UNSUPPORTED;
  TestPatternParams p;
  test_pattern_setup (&p, info, pattern, fg, bg);
  test_pattern_frame (&p, n_frames);
  GstAmdVideoInfo painted;
  GstAmdVideoConverterConfig cfg;
  test_pattern_conversion (info, &painted, &cfg);
  const bool direct = painted.format == info->format;
  std::vector<uint8_t> img ((size_t) painted.size);
  uint8_t *base = direct ? dst + info->offset[0] : img.data ();
  const int stride = direct ? info->stride[0] : painted.stride[0];
  for (int y = 0; y < p.h; y++)
    for (int x = 0; x < p.w; x++) {
      const uint32_t v = test_pattern_px (p, x, y);
      memcpy (base + (size_t) y * stride + (size_t) x * 4, &v, 4);
    }
  if (direct)
    return GSTAMD_OK;
  return emu_video_convert (&painted, info, &cfg, img.data (), dst, 1, desc, desc_len);
}
