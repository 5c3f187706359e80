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
#include "video_scale420_mfma.h"

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

// ------------------------------------------------------------------------------------------------
// k_scale420_mfma (video_scale420_mfma.h): the horizontal pass on v_mfma_i32_16x16x64_i8, the vertical pass and the ring as above
// ------------------------------------------------------------------------------------------------
typedef int mfma_v4i __attribute__ ((ext_vector_type (4)));

static __device__ __forceinline__ mfma_v4i mfma_i8 (const uint4 &a, const uint4 &b, mfma_v4i c)
{
  const mfma_v4i va = {(int) a.x, (int) a.y, (int) a.z, (int) a.w}, vb = {(int) b.x, (int) b.y, (int) b.z, (int) b.w};
  return __builtin_amdgcn_mfma_i32_16x16x64_i8 (va, vb, c, 0, 0, 0);
}

// one block of 16 outputs x 16 lines: three chunks against the block's taps, results into the ring
static __device__ __forceinline__ void mfma_block (const Mfma420A &a0, const Mfma420A &a1, const Mfma420A &a2, const uint4 *b, uint32_t *ring_word)
{
  const mfma_v4i init = {128 * 64 + 32, 128 * 64 + 32, 128 * 64 + 32, 128 * 64 + 32};
  mfma_v4i cy = mfma_i8 (a0.y, b[0], init), cu = mfma_i8 (a0.u, b[0], init), cv = mfma_i8 (a0.v, b[0], init);
  cy = mfma_i8 (a1.y, b[1], cy), cu = mfma_i8 (a1.u, b[1], cu), cv = mfma_i8 (a1.v, b[1], cv);
  cy = mfma_i8 (a2.y, b[2], cy), cu = mfma_i8 (a2.u, b[2], cu), cv = mfma_i8 (a2.v, b[2], cv);
  ring_word[0] = mfma_group_word (cy.x, cy.y, cy.z, cy.w);
  ring_word[256] = mfma_group_word (cu.x, cu.y, cu.z, cu.w);
  ring_word[512] = mfma_group_word (cv.x, cv.y, cv.z, cv.w);
}

template <int CH, int SEMI, int NGV>
__global__ __launch_bounds__ (256) void k_scale420_mfma (Mfma420Params p, Dst dst, PostFast pf)
{
  extern __shared__ uint32_t lds_w[];
  uint32_t *ring = lds_w;
  const int nwaves = (int) (blockDim.x >> 6);
  const int wave = __builtin_amdgcn_readfirstlane ((int) (threadIdx.x >> 6)), lane = (int) (threadIdx.x & 63);
  const int col = lane & 15, kg = lane >> 4;
  const int tile_blocks = p.f.h.tile_w >> 4;
  const int B0 = (int) blockIdx.x * tile_blocks, B1 = B0 + tile_blocks < p.n_blocks ? B0 + tile_blocks : p.n_blocks;
  const int t0 = 16 * B0, t1 = t0 + p.f.h.tile_w < p.f.h.out_w ? t0 + p.f.h.tile_w : p.f.h.out_w;
  const int j0 = (int) blockIdx.y * p.f.rows_per_chunk;
  const int j1 = j0 + p.f.rows_per_chunk < p.f.out_h ? j0 + p.f.rows_per_chunk : p.f.out_h;
  const int rows_per_round = 4 * nwaves;
  int gl, g_last;
  fused_round_groups (p.f, j0, j1 - 1, &gl, &g_last);
  int kb = (gl >> 2) + wave;              // this wave's next line block (lines 16 kb - 1 .. 16 kb + 14 = groups 4 kb .. 4 kb + 3)
  for (int jr = j0; jr < j1; jr += rows_per_round) {
    const int jl = (jr + rows_per_round < j1 ? jr + rows_per_round : j1) - 1;
    int gl_r, gh;
    fused_round_groups (p.f, jr, jl, &gl_r, &gh);
    while (kb <= (gh >> 2)) {
      Mfma420Rows rows;
      mfma_rows (p.f.h, 16 * kb - 1 + col, rows);
      int slot = (4 * kb) % p.f.ring + kg;
      slot = slot >= p.f.ring ? slot - p.f.ring : slot;
      uint32_t *rw = ring + (size_t) slot * GSTAMD_FUSED_GROUP_WORDS + col;
      const uint4 *bt = p.btab + (size_t) B0 * (GSTAMD_MFMA_CHUNKS * 64) + lane;
      Mfma420Loads qa, qb, q0, q1, q2;
      Mfma420A a0, a1, a2;
      uint4 bc[3], bn[3];
      int c = B0 + p.d0;
      // chunks c .. c + 4 on their way; the window slides one chunk per block, loads run three blocks ahead
      mfma_request<SEMI> (p.f.h, rows, c, kg, qa);
      mfma_request<SEMI> (p.f.h, rows, c + 1, kg, qb);
      mfma_request<SEMI> (p.f.h, rows, c + 2, kg, q0);
      mfma_request<SEMI> (p.f.h, rows, c + 3, kg, q1);
      mfma_request<SEMI> (p.f.h, rows, c + 4, kg, q2);
      bc[0] = bt[0], bc[1] = bt[64], bc[2] = bt[128];
      mfma_make_a<CH, SEMI> (p.f.h, qa, a0);
      mfma_make_a<CH, SEMI> (p.f.h, qb, a1);
      // three blocks per turn: chunk registers, load buffers and tap registers keep fixed roles while the window slides
#define MFMA_PIN_LOADS() do { asm volatile ("" ::: "memory"); __builtin_amdgcn_sched_barrier (0); } while (0)
      MFMA_PIN_LOADS ();
#define MFMA_STEP(k, A0, A1, A2, Q, BC, BN) \
      { \
        const int nb = bg + (k) + 1 < B1 ? (k) + 1 : (k);                /* next block's taps (the last block reloads its own) */ \
        BN[0] = bt[(size_t) nb * (GSTAMD_MFMA_CHUNKS * 64)], BN[1] = bt[(size_t) nb * (GSTAMD_MFMA_CHUNKS * 64) + 64], \
            BN[2] = bt[(size_t) nb * (GSTAMD_MFMA_CHUNKS * 64) + 128]; \
        MFMA_PIN_LOADS ();              /* issued HERE: the scheduler otherwise sinks them to their use, a block later, and waits there */ \
        mfma_make_a<CH, SEMI> (p.f.h, Q, A2); \
        mfma_request<SEMI> (p.f.h, rows, c + 5 + (k), kg, Q); \
        MFMA_PIN_LOADS (); \
        mfma_block (A0, A1, A2, BC, rw + 16 * (k)); \
      }
      for (int bg = B0; bg < B1; bg += 3) {
        MFMA_STEP (0, a0, a1, a2, q0, bc, bn)
        if (bg + 1 >= B1)
          break;
        MFMA_STEP (1, a1, a2, a0, q1, bn, bc)
        if (bg + 2 >= B1)
          break;
        MFMA_STEP (2, a2, a0, a1, q2, bc, bn)
        // an odd number of steps per turn: the tap registers swap roles, put the next block's back into bc
        bc[0] = bn[0], bc[1] = bn[1], bc[2] = bn[2];
        c += 3;
        bt += 3 * GSTAMD_MFMA_CHUNKS * 64;
        rw += 48;
      }
#undef MFMA_STEP
#undef MFMA_PIN_LOADS
      kb += nwaves;
    }
    __syncthreads ();
    for (int j = jr + wave; j <= jl; j += nwaves)
      fused_vrow<NGV> (p.f, ring, dst, pf, j, t0, t1, lane);
    __syncthreads ();
  }
}

size_t mfma420_lds_bytes (int ring)
{
  return (size_t) ring * GSTAMD_FUSED_GROUP_WORDS * 4;
}

template <int CH, int SEMI>
static hipError_t launch_mfma_ngv (const Mfma420Params &p, const Dst &d, const PostFast &pf, int nwaves, size_t lds, dim3 grid, hipStream_t stream)
{
#define GO(NGV) do { \
    if (lds > 65536) { \
      hipError_t e = hipFuncSetAttribute ((const void *) k_scale420_mfma<CH, SEMI, NGV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); \
      if (e != hipSuccess) return e; \
    } \
    hipLaunchKernelGGL ((k_scale420_mfma<CH, SEMI, NGV>), grid, dim3 (64 * nwaves), lds, stream, p, d, pf); \
  } while (0)
  if (p.f.ngv == 5)
    GO (5);
  else
    GO (0);
#undef GO
  return hipGetLastError ();
}

int mfma420_blocks_per_cu (int nwaves, size_t lds)
{
  int per_cu = 0;
  const void *fn = (const void *) k_scale420_mfma<CHROMA_H_H2_CS, 0, 5>;
  if (lds > 65536 && hipFuncSetAttribute (fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) != hipSuccess)
    return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor (&per_cu, fn, 64 * nwaves, lds) != hipSuccess)
    return 0;
  return per_cu;
}

// hipErrorNotSupported: the caller takes k_scale420_fused
hipError_t launch_scale420_mfma (const Mfma420Params &p, int chroma_h, int nwaves, uint8_t *dst, int dstride, const ColorParams &post,
    const int pack_pos[4], const PostFast &pf, hipStream_t stream)
{
  const H420RegParams &h = p.f.h;
  int ok = (h.width % 16) == 0 && ((uintptr_t) h.y % 16) == 0 && (h.ystride % 16) == 0 && ((uintptr_t) dst % 4) == 0 && (dstride % 4) == 0 &&
      (h.tile_w % 16) == 0 && h.tile_w <= 256;
  if (h.semi)
    ok = ok && ((uintptr_t) h.c0 % 16) == 0 && (h.cstride % 16) == 0;
  else
    ok = ok && ((uintptr_t) h.c0 % 8) == 0 && ((uintptr_t) h.c1 % 8) == 0 && (h.cstride % 8) == 0;
  const size_t lds = mfma420_lds_bytes (p.f.ring);
  if (!ok || nwaves < 1 || nwaves > 8 || lds > 160 * 1024)
    return hipErrorNotSupported;
  Dst d;
  d.p = dst;
  d.stride = dstride;
  d.final = 1;
  d.post = post;
  for (int i = 0; i < 4; i++)
    d.pack_pos[i] = pack_pos[i];
  const int tile_blocks = h.tile_w / 16, tiles = (p.n_blocks + tile_blocks - 1) / tile_blocks;
  dim3 grid (tiles, (p.f.out_h + p.f.rows_per_chunk - 1) / p.f.rows_per_chunk);
  if (h.semi) {
    if (chroma_h == CHROMA_H_H2_CS)
      return launch_mfma_ngv<CHROMA_H_H2_CS, 1> (p, d, pf, nwaves, lds, grid, stream);
    if (chroma_h == CHROMA_H_H2)
      return launch_mfma_ngv<CHROMA_H_H2, 1> (p, d, pf, nwaves, lds, grid, stream);
    return launch_mfma_ngv<CHROMA_H_NONE, 1> (p, d, pf, nwaves, lds, grid, stream);
  }
  if (chroma_h == CHROMA_H_H2_CS)
    return launch_mfma_ngv<CHROMA_H_H2_CS, 0> (p, d, pf, nwaves, lds, grid, stream);
  if (chroma_h == CHROMA_H_H2)
    return launch_mfma_ngv<CHROMA_H_H2, 0> (p, d, pf, nwaves, lds, grid, stream);
  return launch_mfma_ngv<CHROMA_H_NONE, 0> (p, d, pf, nwaves, lds, grid, stream);
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
