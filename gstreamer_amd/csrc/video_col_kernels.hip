// video_col_kernels.hip - k_scale_col (video_scale_col.h): the column-walk scaler, both N-tap passes of a 4:2:0 planar / semi-planar source
// in one kernel, one wave per column tile and row run, no barrier inside the walk.
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
#include "video_scale_col.h"

namespace gstamd {

// the lanes of a wave on the device: the thread's own registers
template <int OPL, int NW, int NGV>
struct ColExecDev {
  int lane;
  ColLane<OPL, NW> L;
  ColRaw<OPL> ra;
  ColRingRegs<OPL, NGV> rg;
  template <class F> __device__ __forceinline__ void each (F f) { f (lane, L, ra, rg); }
  // word of the lane `dist` places up: a DPP wave shift of `v` (the lane's own word for dist 1, the dist-1 result for dist 2); the last lane reads 0
  struct Nb {
    __device__ __forceinline__ uint32_t operator() (uint32_t v, int, int, int, int) const { return (uint32_t) __builtin_amdgcn_update_dpp (0, (int) v, 0x130, 0xf, 0xf, true); }
  };
  template <class F> __device__ __forceinline__ void each_nb (F f) { Nb nb; f (lane, L, ra, rg, nb); }
  template <class F> __device__ __forceinline__ bool all (F pred) { return __ballot (pred (lane, L)) == ~0ull; }
  __device__ __forceinline__ void sync ()
  {
    __builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier ();
    __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "wavefront");
  }
  __device__ __forceinline__ void publish (uint32_t *flags, int wave)
  {
    __hip_atomic_store (flags + wave, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
#ifdef GSTAMD_COL_TRACE
  __device__ __forceinline__ unsigned long long now () { return __builtin_readcyclecounter (); }
  __device__ __forceinline__ void wait_loads () { asm volatile ("s_waitcnt vmcnt(0)" ::: "memory"); }
  __device__ __forceinline__ void trace_out (unsigned long long *trace, const unsigned long long *acc, int groups)
  {
    if (!trace || lane != 0)
      return;
    const size_t wg = ((size_t) blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    unsigned long long *d = trace + (wg * GSTAMD_COL_MAX_WAVES + (threadIdx.x >> 6)) * 8;
    for (int k = 0; k < 6; k++)
      d[k] = acc[k];
    d[6] = (unsigned long long) groups;
    d[7] = 1;
  }
#endif
  __device__ __forceinline__ void wait_flag (uint32_t *flags, int wave)
  {
    while (__hip_atomic_load (flags + wave, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u)
      __builtin_amdgcn_s_sleep (2);
  }
};

// LDS: [flags: one word per wave][per wave: staged group | ring | hand-over slots]
#ifndef GSTAMD_COL_RW_WAVES
#define GSTAMD_COL_RW_WAVES 4
#endif
template <int OPL, int NW, int NGV, int CH, int SEMI, int WSTEP, int A8, int POST>
__global__ __launch_bounds__ (64 * GSTAMD_COL_MAX_WAVES) __attribute__ ((amdgpu_waves_per_eu ((OPL == 2 && A8 == 2) ? GSTAMD_COL_RW_WAVES : 1, (OPL == 2 && A8 == 2) ? GSTAMD_COL_RW_WAVES : 8)))
void k_scale_col (ColParams p, ColFrames fr, Dst dst, PostFast pf)
{
  extern __shared__ __attribute__ ((aligned (16))) uint8_t col_lds[];
  const int wave = __builtin_amdgcn_readfirstlane ((int) (threadIdx.x >> 6)), lane = (int) (threadIdx.x & 63);
  uint32_t *flags = (uint32_t *) col_lds;
  if (lane == 0)
    flags[wave] = 0u;
  __syncthreads ();
  /* workgroups go round robin over the 8 XCDs (their own L2 each): give an XCD a contiguous run of (tile, chunk) pairs, tiles fastest -
     neighbouring column tiles share the source pixels under their windows' overlap, neighbouring chunks the lines */
  int id = (int) (blockIdx.y * gridDim.x + blockIdx.x);
  const int total = (int) (gridDim.x * gridDim.y);
  if ((total & 7) == 0)
    id = (id & 7) * (total >> 3) + (id >> 3);
  const int ti = id % p.n_tiles, chunk = id / p.n_tiles;
  ColWavePlan wp;
  if (!col_wave_plan (p, chunk, wave, &wp))
    return;
  uint32_t te[8];
  col_entry8 ((const uint32_t *) p.tiles, ti >> 1, te);          /* two 4-word tile entries per 8-word read */
  const int32_t tile[4] = {(int32_t) te[4 * (ti & 1)], (int32_t) te[4 * (ti & 1) + 1], (int32_t) te[4 * (ti & 1) + 2], (int32_t) te[4 * (ti & 1) + 3]};
  const int frame = (int) blockIdx.z;
  const uint32_t crow_bytes = (uint32_t) (p.cstride * (p.crow_hi - p.crow_lo) + (SEMI ? p.width : p.width / 2));
  ColSrc s;
  s.y = col_plane (fr.y[frame], 0, (uint32_t) (p.ystride * (p.height - 1) + p.width));
  s.c0 = col_plane (fr.c0[frame], (long long) p.crow_lo * p.cstride, crow_bytes);
  s.c1 = col_plane (fr.c1[frame], (long long) p.crow_lo * p.cstride, crow_bytes);
  s.out = col_plane (fr.dst[frame], 0, (uint32_t) (p.dstride * (p.out_h - 1) + 4 * p.out_w));
  const Dst &d = dst;
  const size_t wave_bytes = col_wave_bytes (OPL, NGV, p.pubn, OPL == 2 && WSTEP == 1 && A8 == 2);
  uint8_t *mine = col_lds + GSTAMD_COL_FLAG_BYTES + (size_t) wave * wave_bytes;
  ColExecDev<OPL, NW, NGV> x;
  x.lane = lane;
  col_wave<OPL, NW, NGV, CH, SEMI, WSTEP, A8, POST> (x, p, s, tile, wp, mine, mine + wave_bytes, flags, wave, d, pf);
}

struct ColVariant {
  int opl, nw, ngv, ch, semi, wstep, a8, post;
  const void *fn;
  void (*launch) (const ColParams &, const ColFrames &, const Dst &, const PostFast &, dim3, int, size_t, hipStream_t);
};

template <int OPL, int NW, int NGV, int CH, int SEMI, int WSTEP, int A8, int POST>
static void col_launch_one (const ColParams &p, const ColFrames &fr, const Dst &d, const PostFast &pf, dim3 grid, int nwaves, size_t lds, hipStream_t stream)
{
  hipLaunchKernelGGL ((k_scale_col<OPL, NW, NGV, CH, SEMI, WSTEP, A8, POST>), grid, dim3 (64 * nwaves), lds, stream, p, fr, d, pf);
}

#define COL_V1(OPL, NW, NGV, CH, SEMI, WSTEP, A8, POST) \
  {OPL, NW, NGV, CH, SEMI, WSTEP, A8, POST, (const void *) k_scale_col<OPL, NW, NGV, CH, SEMI, WSTEP, A8, POST>, col_launch_one<OPL, NW, NGV, CH, SEMI, WSTEP, A8, POST>},
#define COL_V(OPL, NW, NGV, CH, SEMI, WSTEP, A8) COL_V1 (OPL, NW, NGV, CH, SEMI, WSTEP, A8, 1) COL_V1 (OPL, NW, NGV, CH, SEMI, WSTEP, A8, 0)
#define COL_F(o, n, g, w, a) COL_V (o, n, g, CHROMA_H_H2_CS, 0, w, a) COL_V (o, n, g, CHROMA_H_H2_CS, 1, w, a) COL_V (o, n, g, CHROMA_H_H2, 0, w, a) COL_V (o, n, g, CHROMA_H_H2, 1, w, a)
static const ColVariant g_col_variants[] = {
  GSTAMD_COL_FORMS (COL_F)
};

// the kernel of a form (col_form_for); CHROMA_H_NONE runs as the co-sited filter with both selectors equal
static const ColVariant *col_find (const ColForm &f, int chroma_h, int semi, int post = 1)
{
  const int ch = chroma_h == CHROMA_H_H2 ? CHROMA_H_H2 : CHROMA_H_H2_CS;
  for (const ColVariant &v : g_col_variants)
    if (v.opl == f.opl && v.nw == f.nw && v.ngv == f.ngv && v.wstep == f.wstep && v.a8 == f.a8 && v.ch == ch && v.semi == semi && v.post == post)
      return &v;
  return nullptr;
}

size_t col_lds_bytes (const ColForm &f, int pubn, int nwaves) { return GSTAMD_COL_FLAG_BYTES + (size_t) nwaves * col_wave_bytes (f.opl, f.ngv, pubn, f.a8 == 2); }

// workgroups of `nwaves` waves one CU holds, 0 on failure
int col_blocks_per_cu (const ColForm &f, int chroma_h, int semi, int pubn, int nwaves)
{
  const ColVariant *v = col_find (f, chroma_h, semi);
  const size_t lds = col_lds_bytes (f, pubn, nwaves);
  int per_cu = 0;
  if (!v || lds > 160 * 1024 || (lds > 65536 && hipFuncSetAttribute (v->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) != hipSuccess))
    return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor (&per_cu, v->fn, 64 * nwaves, lds) != hipSuccess)
    return 0;
  return per_cu;
}

// n_frames <= GSTAMD_COL_MAX_FRAMES frames in one grid.  hipErrorNotSupported: the caller takes another path
hipError_t launch_scale_col (const ColParams &p, const ColForm &f, int chroma_h, int semi, int nwaves, const ColFrames &fr, int n_frames, int dstride,
    const ColorParams &post, const int pack_pos[4], const PostFast &pf, hipStream_t stream)
{
  video_frame_list_touch (fr.dst[0]);
  const ColVariant *v = col_find (f, chroma_h, semi, pf.use ? 1 : 0);
  if (!v || nwaves < 1 || nwaves > GSTAMD_COL_MAX_WAVES || n_frames < 1 || n_frames > GSTAMD_COL_MAX_FRAMES || (dstride % (4 * f.opl)) != 0)
    return hipErrorNotSupported;
  for (int k = 0; k < n_frames; k++)
    if (((uintptr_t) fr.dst[k] % (4 * f.opl)) != 0)
      return hipErrorNotSupported;
  const size_t lds = col_lds_bytes (f, p.pubn, nwaves);
  if (lds > 160 * 1024)
    return hipErrorNotSupported;
  if (lds > 65536) {
    const hipError_t e = hipFuncSetAttribute (v->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    if (e != hipSuccess)
      return e;
  }
  Dst d;
  d.p = fr.dst[0];
  d.stride = dstride;
  d.final = 1;
  d.post = post;
  for (int i = 0; i < 4; i++)
    d.pack_pos[i] = pack_pos[i];
  v->launch (p, fr, d, pf, dim3 (p.n_tiles, p.n_chunks, n_frames), nwaves, lds, stream);
  return hipGetLastError ();
}

}  // namespace gstamd
