// how fast are global / buffer loads whose lane addresses are off their natural alignment?  (development probe)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t __attribute__ ((aligned (1))) u32_any;
typedef uint32_t u32x2_any __attribute__ ((ext_vector_type (2), aligned (1)));
typedef uint32_t u32x4_any __attribute__ ((ext_vector_type (4), aligned (1)));
typedef uint32_t u32x3_any __attribute__ ((ext_vector_type (3), aligned (1)));
template <int W>
__global__ void k (const uint8_t *p, int off, int stride_lane, size_t per_block, uint32_t *out)
{
  const uint8_t *b = p + (size_t) blockIdx.x * per_block + off + (size_t) threadIdx.x * stride_lane;
  uint32_t acc = 0;
  for (size_t i = 0; i < per_block - 4096; i += (size_t) 256 * stride_lane) {
    if (W == 4) acc ^= *(const __attribute__ ((address_space (1))) u32_any *) (b + i);
    else if (W == 8) { const u32x2_any v = *(const __attribute__ ((address_space (1))) u32x2_any *) (b + i); acc ^= v.x ^ v.y; }
    else if (W == 12) { const u32x3_any v = *(const __attribute__ ((address_space (1))) u32x3_any *) (b + i); acc ^= v.x ^ v.y ^ v.z; }
    else { const u32x4_any v = *(const __attribute__ ((address_space (1))) u32x4_any *) (b + i); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  }
  if (acc == 0x12345678u) out[0] = acc;
}
int main ()
{
  const size_t bytes = (size_t) 1 << 30, blocks = 8192, per_block = bytes / blocks;
  uint8_t *d; uint32_t *o;
  if (hipMalloc (&d, bytes + 65536) != hipSuccess || hipMalloc (&o, 64) != hipSuccess) return 1;
  (void) hipMemset (d, 1, bytes + 65536);
  hipEvent_t e0, e1; (void) hipEventCreate (&e0); (void) hipEventCreate (&e1);
  for (int w : {16, 12})
    for (int sl : {16, 8})            /* bytes between neighbouring lanes: contiguous dwords, 8-byte steps, 2-byte steps (overlapping) */
      for (int off : {0, 1, 2, 4, 8}) {
        if (w == 12 && sl == 16) continue;
        for (int rep = 0; rep < 2; rep++) {
          (void) hipEventRecord (e0);
          if (w == 16) hipLaunchKernelGGL (k<16>, dim3 (blocks), dim3 (256), 0, 0, d, off, sl, per_block, o);
          else hipLaunchKernelGGL (k<12>, dim3 (blocks), dim3 (256), 0, 0, d, off, sl, per_block, o);
          (void) hipEventRecord (e1); (void) hipEventSynchronize (e1);
          float ms; (void) hipEventElapsedTime (&ms, e0, e1);
          const double lane_bytes = (double) (bytes - blocks * 4096) / sl * w;
          if (rep) printf ("width %d lane step %d offset %d: %.3f ms, %.0f G lane-bytes/s, %.1f G load instr/s\n", w, sl, off, ms, lane_bytes / ms / 1e6, lane_bytes / w / 64 / ms / 1e6);
        }
      }
  return 0;
}
