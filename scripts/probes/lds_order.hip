// does an LDS read that follows an LDS write of the same wave (no s_waitcnt between) see the written data?  (development probe)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
__global__ void k (uint32_t *bad, int iters, int mode)
{
  __shared__ __attribute__ ((aligned (16))) uint32_t lds[4096];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t *base = lds + wave * 1024;
  int nbad = 0;
  for (int i = 0; i < 1024; i++) base[i] = 0;
  for (int it = 0; it < iters; it++) {
    const uint32_t v0 = (uint32_t) (it * 2654435761u) ^ (uint32_t) lane, v1 = v0 * 3u + 1u, v2 = v0 * 5u + 7u;
    const uint32_t a = (uint32_t) (uintptr_t) (__attribute__ ((address_space (3))) uint32_t *) (base + lane) ;
    uint32_t r0, r1, r2;
    if (mode == 0)
      asm volatile ("ds_write_b32 %3, %4\n\tds_write_b32 %3, %5 offset:256\n\tds_write_b32 %3, %6 offset:512\n\t"
                    "ds_read_b32 %0, %3\n\tds_read_b32 %1, %3 offset:256\n\tds_read_b32 %2, %3 offset:512\n\ts_waitcnt lgkmcnt(0)"
                    : "=&v" (r0), "=&v" (r1), "=&v" (r2) : "v" (a), "v" (v0), "v" (v1), "v" (v2) : "memory");
    else
      asm volatile ("ds_write2st64_b32 %3, %4, %5 offset0:1 offset1:2\n\tds_write_b32 %3, %6 offset:768\n\t"
                    "ds_read_b32 %0, %3 offset:256\n\tds_read_b32 %1, %3 offset:512\n\tds_read_b32 %2, %3 offset:768\n\ts_waitcnt lgkmcnt(0)"
                    : "=&v" (r0), "=&v" (r1), "=&v" (r2) : "v" (a), "v" (v0), "v" (v1), "v" (v2) : "memory");
    nbad += (r0 != v0) + (r1 != v1) + (r2 != v2);
  }
  atomicAdd (bad, (uint32_t) nbad);
}
int main ()
{
  uint32_t *d, h;
  if (hipMalloc (&d, 4) != hipSuccess) return 1;
  for (int mode = 0; mode < 2; mode++) {
    h = 0; (void) hipMemcpy (d, &h, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL (k, dim3 (1024), dim3 (256), 0, 0, d, 2000, mode);
    (void) hipMemcpy (&h, d, 4, hipMemcpyDeviceToHost);
    printf ("mode %d: stale reads %u\n", mode, h);
  }
  return 0;
}
