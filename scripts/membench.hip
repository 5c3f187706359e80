// scripts/membench.hip - HBM micro-benchmarks used to calibrate the C2 kernel's roofline on the GPU box:
// write-only, read-only, copy and a 1:2.67 read:write mix, plain vs nontemporal stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store(uint4 *p, uint4 x) { u32x4 v = {x.x, x.y, x.z, x.w}; __builtin_nontemporal_store(v, (u32x4 *)p); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NT>
__global__ __launch_bounds__(256) void k_fill(uint4 *__restrict__ d, size_t n, unsigned v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    uint4 x = make_uint4(v, v + 1, v + 2, (unsigned)i);
    if (NT) nt_store(&d[i], x); else d[i] = x;
  }
}
__global__ __launch_bounds__(256) void k_read(const uint4 *__restrict__ s, size_t n, unsigned *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { uint4 x = s[i]; if ((x.x ^ x.y ^ x.z ^ x.w) == 0x12345677u) out[0] = 1; }
}
template <int NT>
__global__ __launch_bounds__(256) void k_copy(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { uint4 x = s[i]; if (NT) nt_store(&d[i], x); else d[i] = x; }
}
// mix: each lane reads 4 B + 2 B (like Y + chroma share) and writes 16 B: 6 B in, 16 B out per lane
template <int NT>
__global__ __launch_bounds__(256) void k_mix(const unsigned *__restrict__ y, const unsigned short *__restrict__ c, uint4 *__restrict__ d, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    unsigned a = y[i], b = c[i];
    uint4 x = make_uint4(a, a ^ b, a + b, a - b);
    if (NT) nt_store(&d[i], x); else d[i] = x;
  }
}
int main() {
  const size_t out_bytes = 3840ull * 2160 * 4, in_bytes = 3840ull * 2160 * 3 / 2;
  const int RING = 16;
  uint4 *dst; unsigned char *src; unsigned *flag;
  CK(hipMalloc(&dst, out_bytes * RING)); CK(hipMalloc(&src, out_bytes * RING)); CK(hipMalloc(&flag, 4));
  CK(hipMemset(src, 1, out_bytes * RING));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char *name, double bytes, auto launch) {
    for (int i = 0; i < RING; i++) launch(i);
    CK(hipDeviceSynchronize());
    const int iters = 64;
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) launch(i % RING);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / iters;
    printf("%-28s %8.2f us  %8.1f GB/s\n", name, us, bytes / us / 1e3);
  };
  size_t n_out = out_bytes / 16, n_in = in_bytes / 16;
  dim3 b(256);
  timeit("fill 33MB", out_bytes, [&](int r) { hipLaunchKernelGGL(k_fill<0>, dim3((n_out + 255) / 256), b, 0, 0, dst + r * n_out, n_out, r); });
  timeit("fill 33MB nt", out_bytes, [&](int r) { hipLaunchKernelGGL(k_fill<1>, dim3((n_out + 255) / 256), b, 0, 0, dst + r * n_out, n_out, r); });
  timeit("read 33MB", out_bytes, [&](int r) { hipLaunchKernelGGL(k_read, dim3((n_out + 255) / 256), b, 0, 0, (const uint4 *)src + r * n_out, n_out, flag); });
  timeit("read 12.4MB", in_bytes, [&](int r) { hipLaunchKernelGGL(k_read, dim3((n_in + 255) / 256), b, 0, 0, (const uint4 *)src + r * n_out, n_in, flag); });
  timeit("copy 33MB (r+w 66MB)", 2.0 * out_bytes, [&](int r) { hipLaunchKernelGGL(k_copy<0>, dim3((n_out + 255) / 256), b, 0, 0, (const uint4 *)src + r * n_out, dst + r * n_out, n_out); });
  timeit("copy 33MB nt", 2.0 * out_bytes, [&](int r) { hipLaunchKernelGGL(k_copy<1>, dim3((n_out + 255) / 256), b, 0, 0, (const uint4 *)src + r * n_out, dst + r * n_out, n_out); });
  timeit("mix 12.4MB in 33MB out", in_bytes + out_bytes, [&](int r) { hipLaunchKernelGGL(k_mix<0>, dim3((n_out + 255) / 256), b, 0, 0, (const unsigned *)(src + r * out_bytes), (const unsigned short *)(src + r * out_bytes + n_out * 4), dst + r * n_out, n_out); });
  timeit("mix nt", in_bytes + out_bytes, [&](int r) { hipLaunchKernelGGL(k_mix<1>, dim3((n_out + 255) / 256), b, 0, 0, (const unsigned *)(src + r * out_bytes), (const unsigned short *)(src + r * out_bytes + n_out * 4), dst + r * n_out, n_out); });
  // one big fill (16 frames in one launch) to see the launch-gap-free rate
  {
    size_t n = n_out * RING;
    CK(hipEventRecord(e0));
    for (int i = 0; i < 4; i++) hipLaunchKernelGGL(k_fill<0>, dim3((n + 255) / 256), b, 0, 0, dst, n, i);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %8.2f us  %8.1f GB/s\n", "fill 531MB x4", ms * 1e3 / 4, out_bytes * RING / (ms * 1e3 / 4) / 1e3);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 4; i++) hipLaunchKernelGGL(k_copy<0>, dim3((n + 255) / 256), b, 0, 0, (const uint4 *)src, dst, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %8.2f us  %8.1f GB/s\n", "copy 531MB x4 (r+w)", ms * 1e3 / 4, 2.0 * out_bytes * RING / (ms * 1e3 / 4) / 1e3);
  }
  return 0;
}
