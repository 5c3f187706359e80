// scripts/membench2.hip - step-by-step reconstruction of the C2 kernel's memory pattern (no colour math)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store(void *p, unsigned a, unsigned b, unsigned c, unsigned d) { u32x4 v = {a, b, c, d}; __builtin_nontemporal_store(v, (u32x4 *)p); }
#define W 3840
#define H 2160
// A: single line per lane, 4 px: Y 4B, C 4B (row y>>1), store 16B
template <int BX, int BY>
__global__ __launch_bounds__(256) void kA(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst) {
  int x0 = (blockIdx.x * BX + threadIdx.x) * 4, y = blockIdx.y * BY + threadIdx.y;
  if (x0 >= W || y >= H) return;
  unsigned a = *(const unsigned *)(src + (size_t)y * W + x0);
  unsigned c = *(const unsigned *)(src + (size_t)W * H + (size_t)(y >> 1) * W + x0);
  nt_store(dst + (size_t)y * W * 4 + x0 * 4, a, a ^ c, a + c, a - c);
}
// B: pair per lane (rows 2p-1, 2p; chroma rows p-1, p), 4 px
template <int BX, int BY, int NBR>
__global__ __launch_bounds__(256) void kB(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst) {
  int x0 = (blockIdx.x * BX + threadIdx.x) * 4, p = blockIdx.y * BY + threadIdx.y;
  if (x0 >= W || p > H / 2) return;
  int l0 = 2 * p - 1, l1 = 2 * p;
  int ra = p > 0 ? p - 1 : 0, rb = l1 < H ? p : p - 1;
  const unsigned char *cb = src + (size_t)W * H;
  unsigned ca = *(const unsigned *)(cb + (size_t)ra * W + x0), cc = *(const unsigned *)(cb + (size_t)rb * W + x0);
  if (NBR) {
    int xn = x0 + 4 < W ? x0 + 4 : x0;
    ca += *(const unsigned short *)(cb + (size_t)ra * W + xn);
    cc += *(const unsigned short *)(cb + (size_t)rb * W + xn);
  }
  if (l0 >= 0) { unsigned a = *(const unsigned *)(src + (size_t)l0 * W + x0); nt_store(dst + (size_t)l0 * W * 4 + x0 * 4, a, a ^ ca, a + cc, a - ca); }
  if (l1 < H) { unsigned a = *(const unsigned *)(src + (size_t)l1 * W + x0); nt_store(dst + (size_t)l1 * W * 4 + x0 * 4, a, a ^ cc, a + ca, a - cc); }
}
// C: linear 1D mapping of the same work as A (lane -> 4 px chunk index), to separate 2D-grid effects
__global__ __launch_bounds__(256) void kC(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)W * H / 4) return;
  int y = (int)(i / (W / 4)), x0 = (int)(i % (W / 4)) * 4;
  unsigned a = *(const unsigned *)(src + (size_t)y * W + x0);
  unsigned c = *(const unsigned *)(src + (size_t)W * H + (size_t)(y >> 1) * W + x0);
  nt_store(dst + (size_t)y * W * 4 + x0 * 4, a, a ^ c, a + c, a - c);
}
// D: like A but 8 px per lane via two 16 B stores 1 KB apart (lane-contiguous stores), Y as 2 x 4 B
template <int BX, int BY>
__global__ __launch_bounds__(256) void kD(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst) {
  int xb = blockIdx.x * BX * 8 + threadIdx.x * 4, y = blockIdx.y * BY + threadIdx.y;
  if (y >= H) return;
#pragma unroll
  for (int g = 0; g < 2; g++) {
    int x0 = xb + g * BX * 4;
    if (x0 < W) {
      unsigned a = *(const unsigned *)(src + (size_t)y * W + x0);
      unsigned c = *(const unsigned *)(src + (size_t)W * H + (size_t)(y >> 1) * W + x0);
      nt_store(dst + (size_t)y * W * 4 + x0 * 4, a, a ^ c, a + c, a - c);
    }
  }
}
int main() {
  const size_t out_bytes = (size_t)W * H * 4, in_bytes = (size_t)W * H * 3 / 2;
  const int RING = 16;
  unsigned char *dst, *src;
  CK(hipMalloc(&dst, out_bytes * RING)); CK(hipMalloc(&src, in_bytes * 2 * RING));
  CK(hipMemset(src, 1, in_bytes * 2 * RING));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char *name, auto launch) {
    for (int i = 0; i < RING; i++) launch(i);
    CK(hipDeviceSynchronize());
    const int iters = 64;
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) launch(i % RING);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / iters;
    printf("%-34s %8.2f us  %8.1f GB/s\n", name, us, (double)(in_bytes + out_bytes) / us / 1e3);
  };
#define S(r) (src + (size_t)(r) * in_bytes * 2)
#define D(r) (dst + (size_t)(r) * out_bytes)
  timeit("A single-line 64x4", [&](int r) { hipLaunchKernelGGL((kA<64, 4>), dim3(W / 4 / 64, H / 4), dim3(64, 4), 0, 0, S(r), D(r)); });
  timeit("A single-line 192x1 (5 blk/row)", [&](int r) { hipLaunchKernelGGL((kA<192, 1>), dim3(W / 4 / 192, H), dim3(192, 1), 0, 0, S(r), D(r)); });
  timeit("A single-line 32x8", [&](int r) { hipLaunchKernelGGL((kA<32, 8>), dim3(W / 4 / 32, H / 8), dim3(32, 8), 0, 0, S(r), D(r)); });
  timeit("C linear 1D", [&](int r) { hipLaunchKernelGGL(kC, dim3((W * H / 4 + 255) / 256), dim3(256), 0, 0, S(r), D(r)); });
  timeit("B pair 64x4", [&](int r) { hipLaunchKernelGGL((kB<64, 4, 0>), dim3(W / 4 / 64, (H / 2 + 1 + 3) / 4), dim3(64, 4), 0, 0, S(r), D(r)); });
  timeit("B pair 64x4 +nbr loads", [&](int r) { hipLaunchKernelGGL((kB<64, 4, 1>), dim3(W / 4 / 64, (H / 2 + 1 + 3) / 4), dim3(64, 4), 0, 0, S(r), D(r)); });
  timeit("B pair 64x2 +nbr", [&](int r) { hipLaunchKernelGGL((kB<64, 2, 1>), dim3(W / 4 / 64, (H / 2 + 1 + 1) / 2), dim3(64, 2), 0, 0, S(r), D(r)); });
  timeit("D 2x4px lane-contig 64x4", [&](int r) { hipLaunchKernelGGL((kD<64, 4>), dim3((W / 8 + 63) / 64, H / 4), dim3(64, 4), 0, 0, S(r), D(r)); });
  return 0;
}
