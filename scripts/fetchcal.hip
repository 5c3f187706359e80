// scripts/fetchcal.hip - calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the load widths our kernels use.
// Each kernel streams a KNOWN number of bytes (a 512 MiB buffer, larger than the 256 MiB Infinity Cache, every byte once) with
// 4-, 8- or 16-byte loads per lane, or writes it with 4- / 16-byte (plain / nontemporal) stores.  Run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ...   and   rocprofv3 --pmc WRITE_SIZE --kernel-trace ...
// and divide the counter (KB) by the known bytes: the factor to apply to the same access width elsewhere
// (MI355X_MICROARCH.md gives x2 for 16-byte streaming reads and calls other widths uncalibrated).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_read4(const unsigned *__restrict__ s, size_t n, unsigned *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { unsigned x = s[i]; if (x == 0x12345677u) out[0] = 1; }
}
__global__ __launch_bounds__(256) void k_read8(const uint2 *__restrict__ s, size_t n, unsigned *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { uint2 x = s[i]; if ((x.x ^ x.y) == 0x12345677u) out[0] = 1; }
}
__global__ __launch_bounds__(256) void k_read16(const uint4 *__restrict__ s, size_t n, unsigned *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { uint4 x = s[i]; if ((x.x ^ x.y ^ x.z ^ x.w) == 0x12345677u) out[0] = 1; }
}
// the C2 strip kernel's shape: 4-byte luma words of two lines + one 8-byte chroma word per lane
__global__ __launch_bounds__(64) void k_read_c2like(const unsigned *__restrict__ y, const uint2 *__restrict__ c, size_t n, size_t stride_w, unsigned *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { unsigned a = y[i], b = y[i + stride_w]; uint2 u = c[i / 2]; if ((a ^ b ^ u.x ^ u.y) == 0x12345677u) out[0] = 1; }
}
__global__ __launch_bounds__(256) void k_write4(unsigned *__restrict__ d, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = (unsigned)i;
}
template <int NT>
__global__ __launch_bounds__(256) void k_write16(uint4 *__restrict__ d, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { u32x4 v = {(unsigned)i, 1u, 2u, 3u}; if (NT) __builtin_nontemporal_store(v, (u32x4 *)&d[i]); else *(u32x4 *)&d[i] = v; }
}

int main() {
  const size_t bytes = 512ull << 20;
  unsigned char *buf; unsigned *flag;
  CK(hipMalloc(&buf, bytes + (64 << 20))); CK(hipMalloc(&flag, 4));
  CK(hipMemset(buf, 1, bytes + (64 << 20)));
  CK(hipDeviceSynchronize());
  printf("known bytes per launch: %zu (every kernel below moves exactly this much, once)\n", bytes);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_read4, dim3((bytes / 4 + 255) / 256), dim3(256), 0, 0, (const unsigned *)buf, bytes / 4, flag);
    hipLaunchKernelGGL(k_read8, dim3((bytes / 8 + 255) / 256), dim3(256), 0, 0, (const uint2 *)buf, bytes / 8, flag);
    hipLaunchKernelGGL(k_read16, dim3((bytes / 16 + 255) / 256), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16, flag);
    // 2 x 4 B luma + 4 B (half of an 8-byte chroma word shared by two lanes' i/2) per lane ~ bytes/ (4+4+4) lanes; counted below
    { const size_t n = bytes / 12; hipLaunchKernelGGL(k_read_c2like, dim3((n + 63) / 64), dim3(64), 0, 0, (const unsigned *)buf, (const uint2 *)(buf + 2 * 4 * n), n, n, flag); }
    hipLaunchKernelGGL(k_write4, dim3((bytes / 4 + 255) / 256), dim3(256), 0, 0, (unsigned *)buf, bytes / 4);
    hipLaunchKernelGGL(k_write16<0>, dim3((bytes / 16 + 255) / 256), dim3(256), 0, 0, (uint4 *)buf, bytes / 16);
    hipLaunchKernelGGL(k_write16<1>, dim3((bytes / 16 + 255) / 256), dim3(256), 0, 0, (uint4 *)buf, bytes / 16);
    CK(hipDeviceSynchronize());
  }
  printf("done\n");
  return 0;
}
