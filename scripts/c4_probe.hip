// scripts/c4_probe.hip - what the memory system gives BASELINE C4's access pattern, measured with hand-written kernels (round 3;
// replaces the torch copy_ of scripts/bw_probe.py as the yardstick).
//   hipcc -O3 --offload-arch=gfx950 -o scripts/c4_probe scripts/c4_probe.hip && scripts/c4_probe > gpurun_out/c4_probe.jsonl
// Part 1 - linear yardsticks: float4 copy (the guide's 6.29 TB/s figure), read-only, write-only, and a 4:1 read:write mix of
//   linear streams moving exactly C4's bytes (132.7 MB read + 33.2 MB written per "frame").
// Part 2 - C4 skeleton: 16 pads of 1920x1080 BGRA at (640 (i % 4), 360 (i / 4)) on a 3840x2160 canvas, every covered source
//   pixel read once (16 bytes per lane and pad row, clamped like k_aggregate's span4_fetch), XOR instead of blend arithmetic,
//   one 16-byte store per lane.  Swept: waves per workgroup, rows per wave, block -> tile order (row-major, XCD bands, column-major),
//   loads in flight per lane, nontemporal loads / stores.
// Every timing: 60 ms pre-heat, then REPS launches over SETS rotating buffer sets (> the 256 MiB Infinity Cache) between HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf ("HIP error %s at line %d\n", hipGetErrorString (e_), __LINE__); exit (1); } } while (0)
typedef unsigned int u32x4 __attribute__ ((ext_vector_type (4)));

#define DW 3840
#define DH 2160
#define PW 1920
#define PH 1080
#define NP 16
#define SETS 3

struct Set {
  const unsigned char *pad[NP];
  unsigned char *dst;
};

template <int NT>
__device__ __forceinline__ u32x4 ld16 (const unsigned char *p)
{
  const __attribute__ ((address_space (1))) u32x4 *g = (const __attribute__ ((address_space (1))) u32x4 *) (unsigned long long) p;
  return NT ? __builtin_nontemporal_load (g) : *g;
}

template <int NT>
__device__ __forceinline__ void st16 (unsigned char *p, u32x4 v)
{
  if (NT)
    __builtin_nontemporal_store (v, (u32x4 *) p);
  else
    *(u32x4 *) p = v;
}

// ---- part 1: linear yardsticks ----------------------------------------------------------------------------------------------
template <int NTL, int NTS>
__global__ __launch_bounds__ (256) void k_copy (const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n)
{
  const size_t stride = (size_t) gridDim.x * blockDim.x;
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    st16<NTS> ((unsigned char *) (dst + i), ld16<NTL> ((const unsigned char *) (src + i)));
}

__global__ __launch_bounds__ (256) void k_read (const u32x4 *__restrict__ src, u32x4 *__restrict__ sink, size_t n)
{
  const size_t stride = (size_t) gridDim.x * blockDim.x;
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    acc ^= ld16<0> ((const unsigned char *) (src + i));
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u)
    sink[threadIdx.x] = acc;
}

template <int NTS>
__global__ __launch_bounds__ (256) void k_fill (u32x4 *__restrict__ dst, size_t n)
{
  const size_t stride = (size_t) gridDim.x * blockDim.x;
  const u32x4 v = {1, 2, 3, 4};
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    st16<NTS> ((unsigned char *) (dst + i), v);
}

// four linear read streams + one linear write stream of the canvas size each: C4's bytes, ideal addresses
template <int NTL, int NTS>
__global__ __launch_bounds__ (256) void k_mix41 (const u32x4 *__restrict__ a, const u32x4 *__restrict__ b, const u32x4 *__restrict__ c,
    const u32x4 *__restrict__ d, u32x4 *__restrict__ dst, size_t n)
{
  const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const u32x4 va = ld16<NTL> ((const unsigned char *) (a + i)), vb = ld16<NTL> ((const unsigned char *) (b + i));
  const u32x4 vc = ld16<NTL> ((const unsigned char *) (c + i)), vd = ld16<NTL> ((const unsigned char *) (d + i));
  st16<NTS> ((unsigned char *) (dst + i), va ^ vb ^ vc ^ vd);
}

// ---- part 2: the C4 skeleton ---------------------------------------------------------------------------------------------------
// A wave owns 256 canvas columns x ROWS rows.  ORDER: 0 row-major (x fastest, what k_aggregate does), 1 XCD bands (workgroup id % 8
// picks one of 8 horizontal bands of the canvas, so the workgroups resident on one XCD work on neighbouring rows), 2 column-major.
// DEPTH: 0 = every load of a row issued before the first use; n = at most n in flight (the real kernel's ring is 4).
template <int ROWS, int ORDER, int DEPTH, int NTL, int NTS>
__global__ void k_c4_skel (Set s, int tiles_x, int tiles_y)
{
  const int waves = (int) blockDim.x >> 6, wave = (int) threadIdx.x >> 6, lane = (int) threadIdx.x & 63;
  // workgroup -> (column group of `waves` strips, tile row)
  const int gx = (tiles_x + waves - 1) / waves;
  int wg = (int) blockIdx.x, bx, by;
  if (ORDER == 0) {
    bx = wg % gx;
    by = wg / gx;
  } else if (ORDER == 1) {
    const int xcd = wg & 7, k = wg >> 3;
    const int band = (tiles_y + 7) / 8;
    bx = k % gx;
    by = xcd * band + k / gx;
  } else {
    by = wg % tiles_y;
    bx = wg / tiles_y;
  }
  const int tx = bx * waves + wave;
  if (tx >= tiles_x || by >= tiles_y)
    return;
  const int x = tx * 256 + lane * 4;
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
    const int y = by * ROWS + r;
    if (y >= DH)
      break;
    u32x4 acc = {(unsigned) x, (unsigned) y, 0, 0};
    // pads under this strip and row: columns (x0 / 640 - 2 .. x0 / 640) & 0..3, rows likewise - resolved with uniform arithmetic
    const int wx0 = tx * 256, wx1 = wx0 + 256;
    if (DEPTH == 0) {
      u32x4 v[16];
#pragma unroll
      for (int py = 0; py < 4; py++)
#pragma unroll
        for (int px = 0; px < 4; px++) {
          const int xpos = 640 * px, ypos = 360 * py, sy = y - ypos;
          const bool hit = (sy >= 0) & (sy < PH) & (xpos < wx1) & (xpos + PW > wx0);
          v[py * 4 + px] = (u32x4) {0, 0, 0, 0};
          if (hit) {
            int sx = x - xpos;
            sx = sx < 0 ? 0 : (sx > PW - 4 ? PW - 4 : sx);
            v[py * 4 + px] = ld16<NTL> (s.pad[py * 4 + px] + (size_t) sy * (PW * 4) + 4 * (size_t) sx);
          }
        }
#pragma unroll
      for (int k = 0; k < 16; k++)
        acc ^= v[k];
    } else {
#pragma unroll
      for (int py = 0; py < 4; py++)
#pragma unroll
        for (int px = 0; px < 4; px++) {
          const int xpos = 640 * px, ypos = 360 * py, sy = y - ypos;
          const bool hit = (sy >= 0) & (sy < PH) & (xpos < wx1) & (xpos + PW > wx0);
          if (hit) {
            int sx = x - xpos;
            sx = sx < 0 ? 0 : (sx > PW - 4 ? PW - 4 : sx);
            acc ^= ld16<NTL> (s.pad[py * 4 + px] + (size_t) sy * (PW * 4) + 4 * (size_t) sx);
          }
        }
    }
    if (x < DW)
      st16<NTS> (s.dst + (size_t) y * (DW * 4) + 4 * (size_t) x, acc);
  }
}

static hipStream_t g_stream;
static hipEvent_t g_e0, g_e1;

template <typename F>
static double time_us (F launch, int reps)
{
  // 60 ms pre-heat (the device needs 20-30 ms of load to reach its steady clocks), then `reps` timed launches
  CK (hipEventRecord (g_e0, g_stream));
  float ms = 0;
  int it = 0;
  do {
    for (int k = 0; k < 8; k++)
      launch (it++);
    CK (hipEventRecord (g_e1, g_stream));
    CK (hipEventSynchronize (g_e1));
    CK (hipEventElapsedTime (&ms, g_e0, g_e1));
  } while (ms < 60.f);
  CK (hipEventRecord (g_e0, g_stream));
  for (int k = 0; k < reps; k++)
    launch (it++);
  CK (hipEventRecord (g_e1, g_stream));
  CK (hipEventSynchronize (g_e1));
  CK (hipEventElapsedTime (&ms, g_e0, g_e1));
  return 1000.0 * ms / reps;
}

static void report (const char *name, const char *variant, double us, double bytes)
{
  printf ("{\"probe\": \"%s\", \"variant\": \"%s\", \"us\": %.2f, \"TBps\": %.3f, \"frac_of_8\": %.3f}\n", name, variant, us, bytes / us * 1e-6,
      bytes / us * 1e-6 / 8.0);
  fflush (stdout);
}

template <int ROWS, int ORDER, int DEPTH, int NTL, int NTS>
static void run_skel (const Set *sets, int waves, int reps)
{
  const int tiles_x = DW / 256, tiles_y = (DH + ROWS - 1) / ROWS;
  const int gx = (tiles_x + waves - 1) / waves;
  int blocks = gx * tiles_y;
  if (ORDER == 1)
    blocks = gx * ((tiles_y + 7) / 8) * 8;
  char v[128];
  snprintf (v, sizeof v, "waves=%d rows=%d order=%d depth=%d ntl=%d nts=%d", waves, ROWS, ORDER, DEPTH, NTL, NTS);
  const double us = time_us ([&](int it) {
    hipLaunchKernelGGL ((k_c4_skel<ROWS, ORDER, DEPTH, NTL, NTS>), dim3 (blocks), dim3 (64 * waves), 0, g_stream, sets[it % SETS], tiles_x, tiles_y);
  }, reps);
  report ("c4_skeleton", v, us, 16.0 * PW * PH * 4 + 4.0 * DW * DH);
}

int main (int argc, char **argv)
{
  const int reps = argc > 1 ? atoi (argv[1]) : 200;
  CK (hipSetDevice (0));
  CK (hipStreamCreate (&g_stream));
  CK (hipEventCreate (&g_e0));
  CK (hipEventCreate (&g_e1));
  // ---- part 1 ----
  {
    const size_t bytes = (size_t) 1 << 30, n = bytes / 16;
    u32x4 *a, *b;
    CK (hipMalloc ((void **) &a, bytes));
    CK (hipMalloc ((void **) &b, bytes));
    CK (hipMemset (a, 1, bytes));
    CK (hipMemset (b, 2, bytes));
    const int grids[] = {256 * 8, 256 * 16, 256 * 32, 0};
    for (int gi = 0; gi < 4; gi++) {
      const int g = grids[gi] ? grids[gi] : (int) ((n + 255) / 256);
      char v[64];
      snprintf (v, sizeof v, "1GiB grid=%d", g);
      report ("copy_f4", v, time_us ([&](int) { hipLaunchKernelGGL ((k_copy<0, 0>), dim3 (g), dim3 (256), 0, g_stream, a, b, n); }, 20), 2.0 * bytes);
      snprintf (v, sizeof v, "1GiB grid=%d nt_store", g);
      report ("copy_f4", v, time_us ([&](int) { hipLaunchKernelGGL ((k_copy<0, 1>), dim3 (g), dim3 (256), 0, g_stream, a, b, n); }, 20), 2.0 * bytes);
      snprintf (v, sizeof v, "1GiB grid=%d nt_load nt_store", g);
      report ("copy_f4", v, time_us ([&](int) { hipLaunchKernelGGL ((k_copy<1, 1>), dim3 (g), dim3 (256), 0, g_stream, a, b, n); }, 20), 2.0 * bytes);
    }
    report ("read_f4", "1GiB grid=8192", time_us ([&](int) { hipLaunchKernelGGL (k_read, dim3 (8192), dim3 (256), 0, g_stream, a, b, n); }, 20), 1.0 * bytes);
    report ("fill_f4", "1GiB grid=8192", time_us ([&](int) { hipLaunchKernelGGL ((k_fill<0>), dim3 (8192), dim3 (256), 0, g_stream, b, n); }, 20), 1.0 * bytes);
    report ("fill_f4", "1GiB grid=8192 nt", time_us ([&](int) { hipLaunchKernelGGL ((k_fill<1>), dim3 (8192), dim3 (256), 0, g_stream, b, n); }, 20), 1.0 * bytes);
    // 4 reads : 1 write with C4's byte counts, from rotating regions of the 1 GiB buffers (5 x 33 MB per launch, 6 launches per lap)
    const size_t cn = (size_t) DW * DH * 4 / 16;
    for (int nt = 0; nt < 2; nt++) {
      char v[64];
      snprintf (v, sizeof v, "4x33MB read + 33MB write, linear, nt=%d", nt);
      auto fn = [&](int it) {
        const size_t base = (size_t) (it % 6) * 5 * cn;
        if (nt)
          hipLaunchKernelGGL ((k_mix41<0, 1>), dim3 ((cn + 255) / 256), dim3 (256), 0, g_stream, a + base, a + base + cn, a + base + 2 * cn, a + base + 3 * cn,
              b + base, cn);
        else
          hipLaunchKernelGGL ((k_mix41<0, 0>), dim3 ((cn + 255) / 256), dim3 (256), 0, g_stream, a + base, a + base + cn, a + base + 2 * cn, a + base + 3 * cn,
              b + base, cn);
      };
      report ("mix_4r1w", v, time_us (fn, reps), 5.0 * cn * 16);
    }
    CK (hipFree (a));
    CK (hipFree (b));
  }
  // ---- part 2 ----
  Set sets[SETS];
  for (int s = 0; s < SETS; s++) {
    for (int i = 0; i < NP; i++) {
      unsigned char *p;
      CK (hipMalloc ((void **) &p, (size_t) PW * PH * 4));
      CK (hipMemset (p, 17 * i + s, (size_t) PW * PH * 4));
      sets[s].pad[i] = p;
    }
    CK (hipMalloc ((void **) &sets[s].dst, (size_t) DW * DH * 4));
  }
  // block shape and order, everything in flight, nt stores (k_aggregate stores nontemporal)
  for (int waves : {1, 2, 4, 8, 15}) {
    run_skel<1, 0, 0, 0, 1> (sets, waves, reps);
    run_skel<1, 1, 0, 0, 1> (sets, waves, reps);
  }
  run_skel<1, 2, 0, 0, 1> (sets, 1, reps);
  run_skel<1, 2, 0, 0, 1> (sets, 4, reps);
  // rows per wave
  for (int waves : {1, 4}) {
    run_skel<2, 0, 0, 0, 1> (sets, waves, reps);
    run_skel<4, 0, 0, 0, 1> (sets, waves, reps);
    run_skel<2, 1, 0, 0, 1> (sets, waves, reps);
    run_skel<4, 1, 0, 0, 1> (sets, waves, reps);
    run_skel<8, 1, 0, 0, 1> (sets, waves, reps);
  }
  // load policy / depth
  run_skel<1, 0, 0, 1, 1> (sets, 4, reps);
  run_skel<1, 0, 0, 0, 0> (sets, 4, reps);
  run_skel<1, 0, 1, 0, 1> (sets, 4, reps);
  run_skel<1, 1, 0, 1, 1> (sets, 4, reps);
  run_skel<2, 1, 0, 1, 1> (sets, 4, reps);
  return 0;
}
