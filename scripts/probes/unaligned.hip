// do unaligned global loads / LDS accesses return the bytes at their address on this device?  (development probe)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
__global__ void k_ld (const uint8_t *p, int off, uint32_t *o1, uint2 *o2, uint4 *o4)
{
  const int t = threadIdx.x + blockIdx.x * blockDim.x;
  uint32_t a; uint2 b; uint4 c;
  __builtin_memcpy (&a, p + off + 4 * t, 4);
  __builtin_memcpy (&b, p + off + 8 * t, 8);
  __builtin_memcpy (&c, p + off + 16 * t, 16);
  o1[t] = a; o2[t] = b; o4[t] = c;
}
__global__ void k_lds (const uint32_t *in, int off, uint32_t *out)
{
  __shared__ __attribute__ ((aligned (16))) uint8_t lds[4096];
  const int t = threadIdx.x;
  typedef volatile __attribute__ ((address_space (3))) uint16_t *l16;
  const uint32_t v = in[t];
  uint8_t *d = lds + 4 * t + off;
  *(l16) d = (uint16_t) v; *(l16) (d + 2) = (uint16_t) (v >> 16);
  __syncthreads ();
  out[t] = *(const volatile __attribute__ ((address_space (3))) uint32_t *) (lds + 4 * t);
}
int main ()
{
  const int N = 1 << 16;
  std::vector<uint8_t> h (N + 64);
  for (int i = 0; i < N + 64; i++) h[i] = (uint8_t) (i * 7 + (i >> 8) * 13);
  uint8_t *d; uint32_t *o1; uint2 *o2; uint4 *o4;
  hipMalloc (&d, N + 64); hipMalloc (&o1, 1024 * 4); hipMalloc (&o2, 1024 * 8); hipMalloc (&o4, 1024 * 16);
  hipMemcpy (d, h.data (), N + 64, hipMemcpyHostToDevice);
  for (int off = 0; off < 8; off++) {
    hipLaunchKernelGGL (k_ld, dim3 (4), dim3 (256), 0, 0, d, off, o1, o2, o4);
    std::vector<uint32_t> r1 (1024); std::vector<uint2> r2 (1024); std::vector<uint4> r4 (1024);
    hipMemcpy (r1.data (), o1, 1024 * 4, hipMemcpyDeviceToHost); hipMemcpy (r2.data (), o2, 1024 * 8, hipMemcpyDeviceToHost); hipMemcpy (r4.data (), o4, 1024 * 16, hipMemcpyDeviceToHost);
    int b1 = 0, b2 = 0, b4 = 0;
    for (int t = 0; t < 1024; t++) {
      b1 += memcmp (&r1[t], &h[off + 4 * t], 4) != 0; b2 += memcmp (&r2[t], &h[off + 8 * t], 8) != 0; b4 += memcmp (&r4[t], &h[off + 16 * t], 16) != 0;
    }
    printf ("global off %d: bad dword %d dwordx2 %d dwordx4 %d\n", off, b1, b2, b4);
  }
  uint32_t *in, *out; hipMalloc (&in, 256 * 4); hipMalloc (&out, 256 * 4);
  std::vector<uint32_t> hi (256); for (int i = 0; i < 256; i++) hi[i] = 0x01020304u * (i + 1);
  hipMemcpy (in, hi.data (), 1024, hipMemcpyHostToDevice);
  for (int off = 0; off <= 6; off += 2) {
    hipLaunchKernelGGL (k_lds, dim3 (1), dim3 (256), 0, 0, in, off, out);
    std::vector<uint32_t> ho (256); hipMemcpy (ho.data (), out, 1024, hipMemcpyDeviceToHost);
    std::vector<uint8_t> e (1024 + 16, 0); for (int t = 0; t < 256; t++) memcpy (&e[4 * t + off], &hi[t], 4);
    int bad = 0; for (int t = 2; t < 256; t++) bad += memcmp (&ho[t], &e[4 * t], 4) != 0;
    printf ("lds b16 stores at +%d: bad %d\n", off, bad);
  }
  return 0;
}
