// video_fused_kernels.hip - k_scale420_fused (video_scale420_fused.h): horizontal + vertical N-tap pass of a regular 4:2:0 source in
// one kernel, the horizontally filtered lines in an LDS ring (BASELINE C3: no AYUV intermediate in HBM).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "planner.h"
#include "video_kernels.h"
#include "video_device.h"
#include "video_fast.h"
#include "video_scale_fast.h"
#include "video_hscale420.h"
#include "video_scale420_fused.h"

namespace gstamd {

static __device__ __forceinline__ void fused_wave_sync ()
{
  __builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier ();
  __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "wavefront");
}

#ifdef GSTAMD_TUNING
#define FUSED_STAMP(k) do { if (p.trace && lane == 0 && (k) < 32) \
    p.trace[((size_t) (blockIdx.y * gridDim.x + blockIdx.x) * 16 + wave) * 32 + (k)] = __builtin_readcyclecounter (); } while (0)
#else
#define FUSED_STAMP(k) do { } while (0)
#endif

// LDS: [ring slots][12][64] words, then per wave two staged lines of three byte planes
template <int NW, int CH, int SEMI, int NGV>
__global__ __launch_bounds__ (1024) void k_scale420_fused (Fused420Params p, Dst dst, PostFast pf)
{
  extern __shared__ uint32_t lds_w[];
  const int nwaves = (int) (blockDim.x >> 6);
  const int wave = __builtin_amdgcn_readfirstlane ((int) (threadIdx.x >> 6)), lane = (int) (threadIdx.x & 63);
  uint32_t *ring = lds_w;
  uint32_t *stage = lds_w + (size_t) p.ring * GSTAMD_FUSED_GROUP_WORDS + (size_t) wave * 2 * GSTAMD_H420_LINE_WORDS;
  const int t0 = (int) blockIdx.x * p.h.tile_w;
  const int t1 = t0 + p.h.tile_w < p.h.out_w ? t0 + p.h.tile_w : p.h.out_w;
  const int j0 = (int) blockIdx.y * p.rows_per_chunk;
  const int j1 = j0 + p.rows_per_chunk < p.out_h ? j0 + p.rows_per_chunk : p.out_h;
  int x_lo, x_hi;
  h420r_span (p.h, p.n_taps_h, t0, t1, &x_lo, &x_hi);
  const int xa = x_lo & ~15;
  FUSED_STAMP (0);
  Fused420Lane<NW> s;
  s.x0 = xa + 16 * lane;
  if (s.x0 + 16 > p.h.width)
    s.x0 = p.h.width - 16;              // lanes past the span: harmless loads, their LDS bytes meet zero taps only
  h420r_fetch_taps<NW> (p.h, xa, t0, t1, lane, s.ft);
  int gl, g_last;
  fused_round_groups (p, j0, j1 - 1, &gl, &g_last);
  int g = gl + wave;                    // this wave's next group: the residue class (g - gl) % nwaves == wave
  fused_request_group<NW, SEMI> (p.h, g < g_last ? g : g_last, s);
  FUSED_STAMP (1);
  int stamp = 2;
  int rows = p.first_rows;
  for (int jr = j0; jr < j1; jr += rows, rows = nwaves) {
    const int jl = (jr + rows < j1 ? jr + rows : j1) - 1;
    int gl_r, gh;
    fused_round_groups (p, jr, jl, &gl_r, &gh);
    while (g <= gh) {
      const int gn = g + nwaves < g_last ? g + nwaves : g_last;
      fused_phase_a<NW, CH, SEMI> (p.h, s, stage, g, lane);
      fused_wave_sync ();
      FUSED_STAMP (stamp); stamp++;         /* first pair of the group staged: its loads have arrived */
      fused_phase_b<NW> (s, stage);
      fused_wave_sync ();
      fused_phase_c<NW, CH, SEMI> (p.h, s, stage, gn, lane);
      fused_wave_sync ();
      fused_phase_d<NW> (s, stage, ring + (size_t) (g % p.ring) * GSTAMD_FUSED_GROUP_WORDS, lane);
      fused_wave_sync ();
      g += nwaves;
      FUSED_STAMP (stamp); stamp++;         /* group done */
    }
    stamp = (stamp + 3) & ~3;
    FUSED_STAMP (stamp); stamp++;           /* horizontal part of the round done */
    __syncthreads ();
    FUSED_STAMP (stamp); stamp++;
    const int j = jr + wave;
    if (j <= jl)
      fused_vrow<NGV> (p, ring, dst, pf, j, t0, t1, lane);
    FUSED_STAMP (stamp); stamp++;
    __syncthreads ();
    FUSED_STAMP (stamp); stamp++;
  }
}

// the row's entries of the two vertical tables through the scalar unit: the row is wave-uniform, but the compiler keeps such reads on the
// vector memory path (the kernel also stores, so it cannot prove the tables unchanged), where they cost a full memory latency at the start
// of every vertical row; s_load_* takes ~1/4 of that and no vector registers.  Waits for its own results (and the wave's LDS traffic).
template <int NGV>
static __device__ __forceinline__ void fused_row_tables (const Fused420Params &p, int j, int *group, uint32_t *tw)
{
  static_assert (NGV == 5, "window of five groups");
  const int32_t *vg = p.vgroup;
  const uint32_t *vt = p.vtapw;
  typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
  const int og = j * 4, ot = j * 4 * NGV, ot4 = ot + 16;
  int g;
  u32x4 t;
  uint32_t t4;
  asm volatile ("s_load_dword %0, %3, %5\n\ts_load_dwordx4 %1, %4, %6\n\ts_load_dword %2, %4, %7\n\ts_waitcnt lgkmcnt(0)"
      : "=&s" (g), "=&s" (t), "=&s" (t4)
      : "s" (vg), "s" (vt), "s" (og), "s" (ot), "s" (ot4) : "memory");
  *group = g;
  tw[0] = t.x, tw[1] = t.y, tw[2] = t.z, tw[3] = t.w, tw[4] = t4;
}

#ifndef GSTAMD_FUSED2_PARITY
#define GSTAMD_FUSED2_PARITY 1
#endif
#if defined(GSTAMD_FUSED_ABL) && GSTAMD_FUSED_ABL == 5
template <int NW, int CH, int SEMI, int B>
static __device__ __forceinline__ void fused2_stage_skip (const H420RegParams &, Fused420Lane<NW> &, uint32_t *, int, int, int) {}
#endif
// schedule 2 (Fused420Params::sched): see there.  LDS: [ring slots][12][64] words, then per wave ONE staged line of three byte planes.
template <int NW, int CH, int SEMI, int NGV>
__global__ __launch_bounds__ (1024) void k_scale420_fused2 (Fused420Params p, Dst dst, PostFast pf)
{
  extern __shared__ uint32_t lds_w[];
  const int nwaves = (int) (blockDim.x >> 6);
  const int wave = __builtin_amdgcn_readfirstlane ((int) (threadIdx.x >> 6)), lane = (int) (threadIdx.x & 63);
  uint32_t *ring = lds_w;
  uint32_t *stage = lds_w + (size_t) p.ring * GSTAMD_FUSED_GROUP_WORDS + (size_t) wave * GSTAMD_H420_LINE_WORDS;
  const int t0 = (int) blockIdx.x * p.h.tile_w;
  const int t1 = t0 + p.h.tile_w < p.h.out_w ? t0 + p.h.tile_w : p.h.out_w;
  const int j0 = (int) blockIdx.y * p.rows_per_chunk;
  const int j1 = j0 + p.rows_per_chunk < p.out_h ? j0 + p.rows_per_chunk : p.out_h;
  int x_lo, x_hi;
  h420r_span (p.h, p.n_taps_h, t0, t1, &x_lo, &x_hi);
  const int xa = x_lo & ~15;
  Fused420Lane<NW> s;
  s.x0 = xa + 16 * lane;
  if (s.x0 + 16 > p.h.width)
    s.x0 = p.h.width - 16;
  h420r_fetch_taps<NW> (p.h, xa, t0, t1, lane, s.ft);
  int gl, g_last;
  fused_round_groups (p, j0, j1 - 1, &gl, &g_last);
  int g = gl + wave;
  fused_request_group<NW, SEMI> (p.h, g < g_last ? g : g_last, s);
  /* this wave's groups up to gh */
#if defined(GSTAMD_FUSED_ABL) && GSTAMD_FUSED_ABL == 1
  const int g_shared = j1 < p.out_h ? p.vgroup[j1] : (1 << 30);        /* profiling build: groups the next chunk makes as well are not made */
#endif
#if defined(GSTAMD_FUSED_ABL) && GSTAMD_FUSED_ABL == 5
#define fused2_stage fused2_stage_skip                               /* profiling build: no source loads, no chroma upsampling */
#endif
  auto produce = [&](int gh) {
    while (g <= gh) {
      const int gn = g + nwaves < g_last ? g + nwaves : g_last;
      uint32_t *slot = ring + (size_t) (g % p.ring) * GSTAMD_FUSED_GROUP_WORDS;
#if defined(GSTAMD_FUSED_ABL) && GSTAMD_FUSED_ABL == 1
      if (g >= g_shared) {
        g += nwaves;
        continue;
      }
#endif
      fused2_stage<NW, CH, SEMI, 0> (p.h, s, stage, g, gn, lane);
      fused_wave_sync ();
      fused2_filter<NW, 0> (s, stage, slot, lane);
      fused_wave_sync ();
      fused2_stage<NW, CH, SEMI, 1> (p.h, s, stage, g, gn, lane);
      fused_wave_sync ();
      fused2_filter<NW, 1> (s, stage, slot, lane);
      fused_wave_sync ();
      fused2_stage<NW, CH, SEMI, 2> (p.h, s, stage, g, gn, lane);
      fused_wave_sync ();
      fused2_filter<NW, 2> (s, stage, slot, lane);
      fused_wave_sync ();
      fused2_stage<NW, CH, SEMI, 3> (p.h, s, stage, g, gn, lane);
      fused_wave_sync ();
      fused2_filter<NW, 3> (s, stage, slot, lane);
      fused_wave_sync ();
      g += nwaves;
    }
  };
  /* round m of the chunk: rows [j0 + (m ? first_rows + (m - 1) nwaves : 0), ...) */
  const int n_rounds = j1 - j0 <= p.first_rows ? 1 : 1 + (j1 - j0 - p.first_rows + nwaves - 1) / nwaves;
  auto round_rows = [&](int m, int *jr, int *jl) {
    const int a = m == 0 ? j0 : j0 + p.first_rows + (m - 1) * nwaves;
    const int e = m == 0 ? j0 + p.first_rows : a + nwaves;
    *jr = a;
    *jl = (e < j1 ? e : j1) - 1;
  };
  /* One loop for every wave, one copy of each body in it.  An even wave's interval is [row of round k, groups of round k + 1, barrier];
     an odd wave runs its horizontal work half an interval ahead: [row of round k, barrier, groups of round k + 2] - so between two
     barriers half of the waves are in the (vector-ALU-bound) horizontal part while the other half sit in the (LDS / memory-latency-bound)
     vertical part.  Both kinds pass the same number of barriers; the ring holds a round's window and the next round's groups, and an
     odd wave's groups of round k + 2 are only written after the barrier that ends every read of round k. */
#if GSTAMD_FUSED2_PARITY
  const bool ahead = (wave & 1) != 0;
#else
  const bool ahead = false;
#endif
  for (int k = ahead ? -2 : -1; k < n_rounds; k++) {
    if (k >= 0) {
      int jr, jl;
      round_rows (k, &jr, &jl);
      const int j = __builtin_amdgcn_readfirstlane (jr + wave);
#if defined(GSTAMD_FUSED_ABL) && GSTAMD_FUSED_ABL == 4
      if (false) {                      /* profiling build: no vertical pass, nothing stored */
#else
      if (j <= jl) {
#endif
        if (NGV == 5) {
          int grp;
          uint32_t tw[5];
          fused_row_tables<5> (p, j, &grp, tw);
          fused_vrow_words<5> (p, ring, dst, pf, j, t0, t1, lane, grp % p.ring, tw);
        } else {
          fused_vrow<NGV> (p, ring, dst, pf, j, t0, t1, lane);
        }
      }
    }
    if (ahead && k >= -1 && k < n_rounds - 1)
      __syncthreads ();
    const int m = k + (ahead ? 2 : 1);
    if (m < n_rounds) {
      int jr, jl, gl_r, gh;
      round_rows (m, &jr, &jl);
      fused_round_groups (p, jr, jl, &gl_r, &gh);
      produce (gh);
    }
    if (!ahead && k < n_rounds - 1)
      __syncthreads ();
  }
}

static inline bool aligned (const void *p, size_t a) { return ((uintptr_t) p & (a - 1)) == 0; }

template <int NW, int CH, int SEMI>
static hipError_t launch_fused_ngv (const Fused420Params &p, const Dst &d, const PostFast &pf, int nwaves, size_t lds, dim3 grid, hipStream_t stream)
{
#define GO(NGV) do { \
    const void *fn = p.sched == 2 ? (const void *) k_scale420_fused2<NW, CH, SEMI, NGV> : (const void *) k_scale420_fused<NW, CH, SEMI, NGV>; \
    if (lds > 65536) { \
      hipError_t e = hipFuncSetAttribute (fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); \
      if (e != hipSuccess) return e; \
    } \
    if (p.sched == 2) \
      hipLaunchKernelGGL ((k_scale420_fused2<NW, CH, SEMI, NGV>), grid, dim3 (64 * nwaves), lds, stream, p, d, pf); \
    else \
      hipLaunchKernelGGL ((k_scale420_fused<NW, CH, SEMI, NGV>), grid, dim3 (64 * nwaves), lds, stream, p, d, pf); \
  } while (0)
  if (p.ngv == 5)
    GO (5);
  else
    GO (0);
#undef GO
  return hipGetLastError ();
}

template <int NW, int SEMI>
static hipError_t launch_fused_ch (const Fused420Params &p, int chroma_h, const Dst &d, const PostFast &pf, int nwaves, size_t lds, dim3 grid,
    hipStream_t stream)
{
  if (chroma_h == CHROMA_H_H2_CS)
    return launch_fused_ngv<NW, CHROMA_H_H2_CS, SEMI> (p, d, pf, nwaves, lds, grid, stream);
  if (chroma_h == CHROMA_H_H2)
    return launch_fused_ngv<NW, CHROMA_H_H2, SEMI> (p, d, pf, nwaves, lds, grid, stream);
  return launch_fused_ngv<NW, CHROMA_H_NONE, SEMI> (p, d, pf, nwaves, lds, grid, stream);
}

size_t fused420_lds_bytes (int ring, int nwaves, int sched)
{
  return ((size_t) ring * GSTAMD_FUSED_GROUP_WORDS + (size_t) nwaves * (sched == 2 ? 1 : 2) * GSTAMD_H420_LINE_WORDS) * 4;
}

// hipErrorNotSupported: the caller takes the two-pass form
hipError_t launch_scale420_fused (const Fused420Params &p, int chroma_h, int nw, int nwaves, uint8_t *dst, int dstride, const ColorParams &post,
    const int pack_pos[4], const PostFast &pf, hipStream_t stream)
{
  video_frame_list_touch (dst);
  int ok = (p.h.width % 16) == 0 && aligned (p.h.y, 16) && (p.h.ystride % 16) == 0 && aligned (dst, 4) && (dstride % 4) == 0;
  if (p.h.semi)
    ok = ok && aligned (p.h.c0, 16) && (p.h.cstride % 16) == 0;
  else
    ok = ok && aligned (p.h.c0, 8) && aligned (p.h.c1, 8) && (p.h.cstride % 8) == 0;
  const size_t lds = fused420_lds_bytes (p.ring, nwaves, p.sched);
  if (!ok || nw < 3 || nw > 5 || nwaves < 1 || nwaves > 16 || lds > 160 * 1024)
    return hipErrorNotSupported;
  Dst d;
  d.p = dst;
  d.stride = dstride;
  d.final = 1;
  d.post = post;
  for (int i = 0; i < 4; i++)
    d.pack_pos[i] = pack_pos[i];
  const int tiles = (p.h.out_w + p.h.tile_w - 1) / p.h.tile_w;
  dim3 grid (tiles, (p.out_h + p.rows_per_chunk - 1) / p.rows_per_chunk);
  switch (nw) {
    case 3: return p.h.semi ? launch_fused_ch<3, 1> (p, chroma_h, d, pf, nwaves, lds, grid, stream) : launch_fused_ch<3, 0> (p, chroma_h, d, pf, nwaves, lds, grid, stream);
    case 4: return p.h.semi ? launch_fused_ch<4, 1> (p, chroma_h, d, pf, nwaves, lds, grid, stream) : launch_fused_ch<4, 0> (p, chroma_h, d, pf, nwaves, lds, grid, stream);
    default: return p.h.semi ? launch_fused_ch<5, 1> (p, chroma_h, d, pf, nwaves, lds, grid, stream) : launch_fused_ch<5, 0> (p, chroma_h, d, pf, nwaves, lds, grid, stream);
  }
}

// workgroups of this kernel one CU holds (occupancy API on a representative instantiation), 0 on failure
int fused420_blocks_per_cu (int nwaves, size_t lds, int sched)
{
  int per_cu = 0;
  const void *fn = sched == 2 ? (const void *) k_scale420_fused2<5, CHROMA_H_H2_CS, 0, 5> : (const void *) k_scale420_fused<5, CHROMA_H_H2_CS, 0, 5>;
  if (lds > 65536 && hipFuncSetAttribute (fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) != hipSuccess)
    return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor (&per_cu, fn, 64 * nwaves, lds) != hipSuccess)
    return 0;
  return per_cu;
}

}  // namespace gstamd
