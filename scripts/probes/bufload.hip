// raw buffer loads: unaligned offsets, SGPR row offset, out-of-range reads (development probe)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
typedef int v4i __attribute__ ((ext_vector_type (4)));
__global__ void k (const uint8_t *p, int bytes, int off, int soff, uint32_t *o)
{
  const int t = threadIdx.x;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc ((void *) p, (short) 0, bytes, 0x00020000);
  o[t] = __builtin_amdgcn_raw_buffer_load_b32 (r, off + 4 * t, soff, 0);
}
int main ()
{
  const int N = 4096;
  std::vector<uint8_t> h (N);
  for (int i = 0; i < N; i++) h[i] = (uint8_t) (i * 7 + (i >> 8) * 13 + 1);
  uint8_t *d; uint32_t *o;
  (void) hipMalloc (&d, N); (void) hipMalloc (&o, 256 * 4);
  (void) hipMemcpy (d, h.data (), N, hipMemcpyHostToDevice);
  for (int off = 0; off < 4; off++)
    for (int soff : {0, 1000, 3002, 3500}) {
      hipLaunchKernelGGL (k, dim3 (1), dim3 (256), 0, 0, d, N, off, soff, o);
      std::vector<uint32_t> r (256);
      (void) hipMemcpy (r.data (), o, 1024, hipMemcpyDeviceToHost);
      int bad = 0, zero_oob = 0, oob = 0;
      for (int t = 0; t < 256; t++) {
        const int a = off + 4 * t + soff;
        uint32_t e = 0;
        if (a + 4 <= N) memcpy (&e, &h[a], 4); else oob++;
        if (a + 4 <= N) bad += r[t] != e; else zero_oob += r[t] == 0;
      }
      printf ("voffset %d + soffset %d: bad %d, out of range %d of which zero %d\n", off, soff, bad, oob, zero_oob);
    }
  return 0;
}
