// scripts/membench3.hip - memory-pattern skeletons of the C2 strip kernel (no colour math), 8 frames per launch
// from pools larger than the Infinity Cache, exactly like bench.py.  Explores: pixels per wave-row (XG x 256),
// line pairs per lane (K), wide luma loads, XCD-aware block order, persistent grids, waves per workgroup.
//   hipcc -O3 --offload-arch=gfx950 -o membench3 membench3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store(void *p, unsigned a, unsigned b, unsigned c, unsigned d) { u32x4 v = {a, b, c, d}; __builtin_nontemporal_store(v, (u32x4 *)p); }
__device__ __forceinline__ void pl_store(void *p, unsigned a, unsigned b, unsigned c, unsigned d) { u32x4 v = {a, b, c, d}; *(u32x4 *)p = v; }
#define W 3840
#define H 2160
#define PAIRS (H / 2 + 1)
#define NB 8
struct Batch { const unsigned char *src[NB]; unsigned char *dst[NB]; };
struct __attribute__((aligned(4))) u2u { unsigned a, b; };

// one wave: column block xb (XG*256 px wide), strip s (K pairs) of frame z.  LW: 0 = 4-byte luma loads per 256-px
// group, 1 = one 16-byte luma load per lane per line (XG must be 4).  NTL: nontemporal luma loads.  NTS: nt stores.
template <int XG, int K, int LW, int NTL, int NTS>
__device__ __forceinline__ void tile(const Batch &bt, int z, int xb, int s, int lane) {
  const unsigned char *src = bt.src[z];
  unsigned char *dst = bt.dst[z];
  const unsigned char *cb = src + (size_t)W * H;
  const int p0 = s * K, p1 = p0 + K < PAIRS ? p0 + K : PAIRS;
  const int xw = xb * XG * 256;
  unsigned cprev[XG];
  {
    const int r = p0 > 0 ? p0 - 1 : 0;
#pragma unroll
    for (int g = 0; g < XG; g++) {
      const int x0 = xw + g * 256 + lane * 4;
      if (x0 < W) { u2u m = *(const u2u *)(cb + (size_t)r * W + x0); cprev[g] = m.a + m.b; } else cprev[g] = 0;
    }
  }
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int p = p0 + k;
    if (p >= p1) break;
    const int l0 = 2 * p - 1, l1 = 2 * p;
    const int cr = p < H / 2 ? p : H / 2 - 1;
    unsigned y0[XG] = {}, y1[XG] = {}, cc[XG] = {};
    const int r0 = l0 >= 0 ? l0 : 0, r1 = l1 < H ? l1 : H - 1;
    if (LW == 1) {
      const int x0 = xw + lane * 16;
      if (x0 < W) {
        u32x4 a = NTL ? __builtin_nontemporal_load((const u32x4 *)(src + (size_t)r0 * W + x0)) : *(const u32x4 *)(src + (size_t)r0 * W + x0);
        u32x4 b = NTL ? __builtin_nontemporal_load((const u32x4 *)(src + (size_t)r1 * W + x0)) : *(const u32x4 *)(src + (size_t)r1 * W + x0);
        u32x4 c = *(const u32x4 *)(cb + (size_t)cr * W + x0);
#pragma unroll
        for (int g = 0; g < XG; g++) { y0[g] = a[g & 3]; y1[g] = b[g & 3]; cc[g] = c[g & 3]; }
      }
    } else {
#pragma unroll
      for (int g = 0; g < XG; g++) {
        const int x0 = xw + g * 256 + lane * 4;
        if (x0 < W) {
          y0[g] = NTL ? __builtin_nontemporal_load((const unsigned *)(src + (size_t)r0 * W + x0)) : *(const unsigned *)(src + (size_t)r0 * W + x0);
          y1[g] = NTL ? __builtin_nontemporal_load((const unsigned *)(src + (size_t)r1 * W + x0)) : *(const unsigned *)(src + (size_t)r1 * W + x0);
          u2u m = *(const u2u *)(cb + (size_t)cr * W + x0);
          cc[g] = m.a + m.b;
        }
      }
    }
#pragma unroll
    for (int g = 0; g < XG; g++) {
      const int x0 = xw + g * 256 + lane * 4;
      if (x0 < W) {
        if (l0 >= 0) {
          void *d = dst + (size_t)l0 * W * 4 + (size_t)x0 * 4;
          if (NTS) nt_store(d, y0[g], y0[g] ^ cprev[g], y0[g] + cc[g], y0[g] - cc[g]); else pl_store(d, y0[g], y0[g] ^ cprev[g], y0[g] + cc[g], y0[g] - cc[g]);
        }
        if (l1 < H) {
          void *d = dst + (size_t)l1 * W * 4 + (size_t)x0 * 4;
          if (NTS) nt_store(d, y1[g], y1[g] ^ cc[g], y1[g] + cprev[g], y1[g] - cprev[g]); else pl_store(d, y1[g], y1[g] ^ cc[g], y1[g] + cprev[g], y1[g] - cprev[g]);
        }
        cprev[g] = cc[g];
      }
    }
  }
}

// ORDER: 0 = 3D grid as shipped (x fastest, strip, frame); 1 = 1D grid, XCD-contiguous logical order (block b on
// XCD b % 8 works through logical range [xcd * n/8, ...)); 2 = 1D grid plain linear order; 3 = persistent, XCD-contiguous
template <int XG, int K, int LW, int NTL, int NTS, int ORDER, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_skel(Batch bt, int nxb, int nstrips, int total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (ORDER == 0) {
    const int s = blockIdx.y * WPB + wv;
    if (s < nstrips) tile<XG, K, LW, NTL, NTS>(bt, blockIdx.z, blockIdx.x, s, lane);
    return;
  }
  const int nblk = gridDim.x;
  if (ORDER == 3) {
    // persistent: block b (XCD b % 8, slot b / 8) walks its XCD's contiguous range with stride = blocks per XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = nblk >> 3;
    const int chunk = (total + 7) / 8;
    const int lo = xcd * chunk, hi = lo + chunk < total ? lo + chunk : total;
    for (int l = lo + slot * WPB + wv; l < hi; l += per * WPB) {
      const int x = l % nxb, r = l / nxb;
      tile<XG, K, LW, NTL, NTS>(bt, r / nstrips, x, r % nstrips, lane);
    }
    return;
  }
  int l;
  if (ORDER == 1) {
    const int per = (nblk + 7) >> 3;
    l = ((blockIdx.x & 7) * per + (blockIdx.x >> 3)) * WPB + wv;
  } else {
    l = blockIdx.x * WPB + wv;
  }
  if (l >= total) return;
  const int x = l % nxb, r = l / nxb;
  tile<XG, K, LW, NTL, NTS>(bt, r / nstrips, x, r % nstrips, lane);
}


// LDS-staged wide tile (the shape a real kernel can use): one wave = 1024 px x K pairs.  Every line is fetched with ONE
// 16-byte load per lane (1 KB contiguous per wave), parked in LDS, and read back in the 4-px-per-lane layout so that
// every store instruction still writes 1 KB contiguous.  SHARE: waves of one workgroup handle vertically adjacent
// strips of the same column and pass the boundary chroma row through LDS instead of re-reading it.
template <int K, int WPB, int SHARE>
__device__ __forceinline__ void tile_lds(const Batch &bt, int z, int xb, int s, int lane, int wv, bool active) {
  __shared__ unsigned lds_y[WPB][2][256 + 4];
  __shared__ unsigned lds_c[WPB + 1][K + 1][256 + 4];
  const unsigned char *src = bt.src[z];
  unsigned char *dst = bt.dst[z];
  const unsigned char *cb = src + (size_t)W * H;
  const int p0 = s * K, p1 = p0 + K < PAIRS ? p0 + K : PAIRS;
  const int xw = xb * 1024, xl = xw + lane * 16;
  const bool in = active && xl < W;
  // chroma rows p0-1 .. p1-1 -> lds_c[wv][0..K]
  if (in) {
    if (!SHARE || wv == 0) {
      const int r = p0 > 0 ? p0 - 1 : 0;
      *(u32x4 *)&lds_c[wv][0][lane * 4] = *(const u32x4 *)(cb + (size_t)r * W + xl);
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int p = p0 + k, cr = p < H / 2 ? p : H / 2 - 1;
      const u32x4 v = *(const u32x4 *)(cb + (size_t)cr * W + xl);
      *(u32x4 *)&lds_c[wv][k + 1][lane * 4] = v;
      if (SHARE && k == K - 1) *(u32x4 *)&lds_c[wv + 1][0][lane * 4] = v;
    }
  }
  if (SHARE) __syncthreads();
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int p = p0 + k;
    if (p >= p1 || !active) break;
    const int l0 = 2 * p - 1, l1 = 2 * p;
    const int r0 = l0 >= 0 ? l0 : 0, r1 = l1 < H ? l1 : H - 1;
    if (in) {
      *(u32x4 *)&lds_y[wv][0][lane * 4] = __builtin_nontemporal_load((const u32x4 *)(src + (size_t)r0 * W + xl));
      *(u32x4 *)&lds_y[wv][1][lane * 4] = __builtin_nontemporal_load((const u32x4 *)(src + (size_t)r1 * W + xl));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int x0 = xw + g * 256 + lane * 4;
      if (x0 < W) {
        const unsigned y0 = lds_y[wv][0][g * 64 + lane], y1 = lds_y[wv][1][g * 64 + lane];
        const unsigned ca = lds_c[wv][k][g * 64 + lane] + lds_c[wv][k][g * 64 + lane + 1];
        const unsigned cc = lds_c[wv][k + 1][g * 64 + lane] + lds_c[wv][k + 1][g * 64 + lane + 1];
        if (l0 >= 0) nt_store(dst + (size_t)l0 * W * 4 + (size_t)x0 * 4, y0, y0 ^ ca, y0 + cc, y0 - cc);
        if (l1 < H) nt_store(dst + (size_t)l1 * W * 4 + (size_t)x0 * 4, y1, y1 ^ cc, y1 + ca, y1 - ca);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  }
}

// ORDER 0: 3D grid (x fastest, strip, frame), WPB waves = WPB vertically adjacent strips.
// ORDER 4: 1D grid, (column, band of BAND strips) units dealt round-robin to the XCDs: vertically adjacent strips of a
// column run back to back on ONE XCD (chroma row re-reads hit its L2) while all XCDs sweep the same region of memory.
template <int K, int WPB, int SHARE, int ORDER, int BAND>
__global__ __launch_bounds__(64 * WPB) void k_lds(Batch bt, int nxb, int nstrips) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int x, S;
  if (ORDER == 0) {
    x = blockIdx.x; S = blockIdx.z * nstrips + blockIdx.y * WPB + wv;
    const bool act = blockIdx.y * WPB + wv < nstrips;
    tile_lds<K, WPB, SHARE>(bt, blockIdx.z, x, blockIdx.y * WPB + wv, lane, wv, act);
  } else {
    const int l = blockIdx.x, xcd = l & 7, i = l >> 3;
    const int unit = (i / BAND) * 8 + xcd, sb = i % BAND;
    x = unit % nxb; S = ((unit / nxb) * BAND + sb) * WPB + wv;
    const bool act = S < nstrips * NB;
    tile_lds<K, WPB, SHARE>(bt, act ? S / nstrips : 0, x, S % nstrips, lane, wv, act);
  }
}

// shipped-shape tile (4 px per lane, K pairs) under the ORDER-4 mapping
template <int XG, int K, int LW, int BAND>
__global__ __launch_bounds__(64) void k_band(Batch bt, int nxb, int nstrips) {
  const int l = blockIdx.x, xcd = l & 7, i = l >> 3;
  const int unit = (i / BAND) * 8 + xcd, sb = i % BAND;
  const int x = unit % nxb, S = (unit / nxb) * BAND + sb;
  if (S >= nstrips * NB) return;
  tile<XG, K, LW, 1, 1>(bt, S / nstrips, x, S % nstrips, threadIdx.x);
}

// ideal mix: same bytes, perfectly linear 16-byte accesses (the ceiling for this read:write ratio)
template <int NTS>
__global__ __launch_bounds__(256) void k_ideal(Batch bt) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;     // one lane = 16 luma bytes + 8 chroma bytes in, 64 bytes out
  const int z = blockIdx.y;
  const size_t n = (size_t)W * H / 16;
  if (i >= n) return;
  const u32x4 a = *(const u32x4 *)(bt.src[z] + i * 16);
  const u2u c = *(const u2u *)(bt.src[z] + (size_t)W * H + i * 8);
  unsigned char *d = bt.dst[z] + ((size_t)blockIdx.x * 256 * 64) + (threadIdx.x >> 6) * 4096 + (threadIdx.x & 63) * 16;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    if (NTS) nt_store(d + q * 1024, a[q], a[q] ^ c.a, a[q] + c.b, a[q] - c.a); else pl_store(d + q * 1024, a[q], a[q] ^ c.a, a[q] + c.b, a[q] - c.a);
  }
}

int main(int argc, char **argv) {
  const size_t out_bytes = (size_t)W * H * 4, in_bytes = (size_t)W * H * 3 / 2;
  const int RIN = 32, ROUT = 16;
  unsigned char *dst, *src;
  CK(hipMalloc(&dst, out_bytes * ROUT)); CK(hipMalloc(&src, in_bytes * RIN + 4096));
  CK(hipMemset(src, 1, in_bytes * RIN + 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char *only = argc > 1 ? argv[1] : nullptr;
  auto timeit = [&](const char *name, auto launch) {
    if (only && !strstr(name, only)) return;
    int step = 0;
    auto mk = [&]() { Batch b; for (int i = 0; i < NB; i++) { b.src[i] = src + (size_t)((step * NB + i) % RIN) * in_bytes; b.dst[i] = dst + (size_t)((step * NB + i) % ROUT) * out_bytes; } step++; return b; };
    for (int i = 0; i < 8; i++) launch(mk());
    CK(hipDeviceSynchronize());
    double best = 1e30, sum = 0;
    const int reps = 5, iters = 40;
    for (int r = 0; r < reps; r++) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < iters; i++) launch(mk());
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double us = ms * 1e3 / iters; sum += us; if (us < best) best = us;
    }
    CK(hipGetLastError());
    double us = sum / reps;
    printf("%-44s %8.2f us/launch (best %7.2f)  %6.2f us/frame  %7.1f GB/s\n", name, us, best, us / NB, (double)(in_bytes + out_bytes) * NB / us / 1e3);
    fflush(stdout);
  };
#define SK(XG, K, LW, NTL, NTS, ORDER, WPB, label) \
  timeit(label, [&](Batch b) { \
    const int nxb = (W + XG * 256 - 1) / (XG * 256), nstrips = (PAIRS + K - 1) / K, total = nxb * nstrips * NB; \
    if (ORDER == 0) hipLaunchKernelGGL((k_skel<XG, K, LW, NTL, NTS, ORDER, WPB>), dim3(nxb, (nstrips + WPB - 1) / WPB, NB), dim3(64 * WPB), 0, 0, b, nxb, nstrips, total); \
    else if (ORDER == 3) hipLaunchKernelGGL((k_skel<XG, K, LW, NTL, NTS, ORDER, WPB>), dim3(256 * 8 / WPB * PERSIST_MULT), dim3(64 * WPB), 0, 0, b, nxb, nstrips, total); \
    else { int nb = (total + WPB - 1) / WPB; nb = (nb + 7) / 8 * 8; hipLaunchKernelGGL((k_skel<XG, K, LW, NTL, NTS, ORDER, WPB>), dim3(nb), dim3(64 * WPB), 0, 0, b, nxb, nstrips, total); } \
  })
#define PERSIST_MULT 1
  timeit("ideal linear mix nt", [&](Batch b) { hipLaunchKernelGGL(k_ideal<1>, dim3((W * H / 16 + 255) / 256, NB), dim3(256), 0, 0, b); });
  timeit("ideal linear mix plain", [&](Batch b) { hipLaunchKernelGGL(k_ideal<0>, dim3((W * H / 16 + 255) / 256, NB), dim3(256), 0, 0, b); });
  SK(1, 3, 0, 1, 1, 0, 1, "shipped: XG1 K3 3D wpb1");
  SK(1, 3, 0, 0, 1, 0, 1, "XG1 K3 3D wpb1 plain-luma-loads");
  SK(1, 3, 0, 1, 0, 0, 1, "XG1 K3 3D wpb1 plain-stores");
  SK(1, 3, 0, 1, 1, 0, 4, "XG1 K3 3D wpb4");
  SK(1, 2, 0, 1, 1, 0, 1, "XG1 K2 3D wpb1");
  SK(1, 4, 0, 1, 1, 0, 1, "XG1 K4 3D wpb1");
  SK(1, 6, 0, 1, 1, 0, 1, "XG1 K6 3D wpb1");
  SK(1, 3, 0, 1, 1, 2, 1, "XG1 K3 1D-linear wpb1");
  SK(1, 3, 0, 1, 1, 1, 1, "XG1 K3 1D-xcd wpb1");
  SK(1, 6, 0, 1, 1, 1, 1, "XG1 K6 1D-xcd wpb1");
  SK(1, 3, 0, 1, 1, 1, 4, "XG1 K3 1D-xcd wpb4");
  SK(1, 3, 0, 1, 1, 3, 1, "XG1 K3 persistent-xcd wpb1 (2048 waves)");
  SK(1, 3, 0, 1, 1, 3, 4, "XG1 K3 persistent-xcd wpb4 (2048 waves)");
  SK(2, 3, 0, 1, 1, 0, 1, "XG2 K3 3D wpb1");
  SK(2, 3, 0, 1, 1, 1, 1, "XG2 K3 1D-xcd wpb1");
  SK(4, 2, 0, 1, 1, 0, 1, "XG4 K2 3D wpb1");
  SK(4, 3, 0, 1, 1, 0, 1, "XG4 K3 3D wpb1");
  SK(4, 3, 0, 1, 1, 1, 1, "XG4 K3 1D-xcd wpb1");
  SK(4, 2, 1, 1, 1, 0, 1, "XG4 K2 wide16 3D wpb1");
  SK(4, 3, 1, 1, 1, 0, 1, "XG4 K3 wide16 3D wpb1");
  SK(4, 3, 1, 0, 1, 0, 1, "XG4 K3 wide16 3D wpb1 plain-luma");
  SK(4, 3, 1, 1, 1, 1, 1, "XG4 K3 wide16 1D-xcd wpb1");
  SK(4, 6, 1, 1, 1, 1, 1, "XG4 K6 wide16 1D-xcd wpb1");
  SK(4, 3, 1, 1, 1, 3, 1, "XG4 K3 wide16 persistent-xcd wpb1");
  SK(4, 1, 1, 1, 1, 0, 1, "XG4 K1 wide16 3D wpb1");
  SK(4, 1, 1, 1, 1, 1, 1, "XG4 K1 wide16 1D-xcd wpb1");

#define LD(K, WPB, SHARE, ORDER, BAND, label) \
  timeit(label, [&](Batch b) { \
    const int nxb = (W + 1023) / 1024, nstrips = (PAIRS + K - 1) / K; \
    if (ORDER == 0) hipLaunchKernelGGL((k_lds<K, WPB, SHARE, ORDER, BAND>), dim3(nxb, (nstrips + WPB - 1) / WPB, NB), dim3(64 * WPB), 0, 0, b, nxb, nstrips); \
    else { const int groups = (nstrips * NB + WPB - 1) / WPB, units = nxb * ((groups + BAND - 1) / BAND), up = (units + 7) / 8 * 8; \
      hipLaunchKernelGGL((k_lds<K, WPB, SHARE, ORDER, BAND>), dim3(up * BAND), dim3(64 * WPB), 0, 0, b, nxb, nstrips); } \
  })
#define BD(XG, K, LW, BAND, label) \
  timeit(label, [&](Batch b) { \
    const int nxb = (W + XG * 256 - 1) / (XG * 256), nstrips = (PAIRS + K - 1) / K; \
    const int units = nxb * ((nstrips * NB + BAND - 1) / BAND), up = (units + 7) / 8 * 8; \
    hipLaunchKernelGGL((k_band<XG, K, LW, BAND>), dim3(up * BAND), dim3(64), 0, 0, b, nxb, nstrips); \
  })
  BD(1, 3, 0, 8, "band: XG1 K3 band8");
  BD(1, 3, 0, 32, "band: XG1 K3 band32");
  BD(1, 2, 0, 8, "band: XG1 K2 band8");
  BD(1, 1, 0, 8, "band: XG1 K1 band8");
  BD(1, 1, 0, 32, "band: XG1 K1 band32");
  BD(4, 1, 1, 8, "band: XG4 K1 wide16 band8");
  BD(4, 1, 1, 32, "band: XG4 K1 wide16 band32");
  BD(4, 2, 1, 8, "band: XG4 K2 wide16 band8");
  LD(1, 1, 0, 0, 1, "lds: K1 wpb1 3D");
  LD(2, 1, 0, 0, 1, "lds: K2 wpb1 3D");
  LD(1, 1, 0, 4, 8, "lds: K1 wpb1 band8");
  LD(1, 1, 0, 4, 32, "lds: K1 wpb1 band32");
  LD(2, 1, 0, 4, 8, "lds: K2 wpb1 band8");
  LD(1, 4, 1, 0, 1, "lds: K1 wpb4 share 3D");
  LD(1, 8, 1, 0, 1, "lds: K1 wpb8 share 3D");
  LD(1, 4, 1, 4, 4, "lds: K1 wpb4 share band4");
  LD(1, 4, 0, 0, 1, "lds: K1 wpb4 noshare 3D");
  LD(2, 4, 1, 0, 1, "lds: K2 wpb4 share 3D");
  return 0;
}
