// video_bilinear_half.hip - k_bilinear420_half (video_bilinear_half.h): 4:2:0 -> bilinear halving -> matrix -> 4-byte RGB, eight consecutive
// outputs per lane straight from registers, a wave per 1024-pixel source column and strip of output rows.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "planner.h"
#include "tuning.h"
#include "video_kernels.h"
#include "video_device.h"
#include "video_fast.h"
#include "video_bilinear_half.h"

namespace gstamd {

// frames of one launch: workgroups [f * blocks_per_frame, (f + 1) * blocks_per_frame) are frame f (a multiple of 8 blocks, so a workgroup's
// XCD does not depend on the frame); wave w of workgroup b plays block (b / 8 * waves + w) * 8 + b % 8 of wide_block_map's XCD-aware order
template <int CH, int L>
__global__ __launch_bounds__ (256) void k_bilinear420_half (BilParams bp, BilBatch fb, int dstride, int tiles_x, int blocks_per_frame)
{
  const int frame = (int) blockIdx.x / blocks_per_frame, fblock = (int) blockIdx.x - frame * blocks_per_frame;
  /* pointers read out of an indexed kernel-argument array are generic to the compiler (FLAT instructions, which wait on both counters):
   * they are global memory, say so */
  typedef const __attribute__ ((address_space (1))) uint8_t *gptr_t;
  Planes pl;
  pl.p[0] = (const uint8_t *) (gptr_t) fb.p[0][frame], pl.p[1] = (const uint8_t *) (gptr_t) fb.p[1][frame];
  pl.p[2] = (const uint8_t *) (gptr_t) fb.p[2][frame], pl.p[3] = nullptr;
  pl.stride[0] = fb.stride[0], pl.stride[1] = fb.stride[1], pl.stride[2] = fb.stride[2], pl.stride[3] = 0;
  uint8_t *dst = (uint8_t *) (__attribute__ ((address_space (1))) uint8_t *) fb.dst[frame];
  const int wave = __builtin_amdgcn_readfirstlane ((int) threadIdx.x >> 6), waves = (int) blockDim.x >> 6;
  int tile, g;
  if (!wide_block_map (((fblock >> 3) * waves + wave) * 8 + (fblock & 7), tiles_x, bp.strips, &tile, &g))
    return;
  const int lane = (int) threadIdx.x & 63;
  const int y0 = (int) ((unsigned) g * (unsigned) bp.out_h / (unsigned) bp.strips);
  const int y1 = (int) ((unsigned) (g + 1) * (unsigned) bp.out_h / (unsigned) bp.strips);
  /* the strip's second vertical taps, one lane per row: a table read inside the row loop is a vector load followed by s_waitcnt vmcnt(0),
   * which also waits for the prefetch and the stores of the row before */
  const int yl = y0 + lane < bp.out_h ? y0 + lane : bp.out_h - 1;
  const int tab_p1 = (int) bp.vtaps[(size_t) yl * 2 + 1];
  /* a lane's eight pixels are 32 consecutive bytes: stored as they are, each 16-byte store instruction would write every other 16 bytes of a
   * 2 KB run - half-written 32-byte sectors that the memory system has to merge.  The wave trades the halves through 2 KB of LDS instead
   * (two 16-byte writes, two 16-byte reads per lane and row, no barrier: one wave) and stores two whole 1 KB runs. */
  extern __shared__ __attribute__ ((aligned (16))) uint8_t lds_all[];
  uint8_t *lds = lds_all + wave * 2048;
  const int x0 = tile * BILH_TILE_SRC + BILH_SRC_PER_LANE * lane;
  const int mode = bp.half;             /* 1: through LDS; 2: plain direct stores; 3: streaming direct stores (tuning builds) */
  const bool src_a = tile * BILH_TILE_SRC + BILH_SRC_PER_LANE * (lane >> 1) < bp.fp.width;
  const bool src_b = tile * BILH_TILE_SRC + BILH_SRC_PER_LANE * (32 + (lane >> 1)) < bp.fp.width;
  bilh_strip<CH, L> (bp, pl, dst, dstride, x0, y0, y1,
      [&] (int y) { return __builtin_amdgcn_readlane (tab_p1, y - y0); },
      [&] (uint8_t *d, bool active, int half, uint32_t a, uint32_t b, uint32_t e, uint32_t f) {
        if (mode == 2) {
          if (active)
            *(uint4 *) (d + 16 * half) = make_uint4 (a, b, e, f);
          return;
        }
        if (mode == 3) {
          if (active)
            store16_stream (d + 16 * half, a, b, e, f);
          return;
        }
        *(uint4 *) (lds + 32 * lane + 16 * half) = make_uint4 (a, b, e, f);
        if (half == 0)
          return;
        __builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier ();
        __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "wavefront");
        const uint4 va = *(const uint4 *) (lds + 16 * lane), vb = *(const uint4 *) (lds + 1024 + 16 * lane);
        /* d = the row's pointer at THIS lane's first output (at the ROW's first for lanes right of the picture): to the tile's start */
        uint8_t *row = active ? d - 32 * lane : d + 2 * BILH_TILE_SRC * tile;
        if (src_a)
          store16_stream (row + 16 * lane, va.x, va.y, va.z, va.w);
        if (src_b)
          store16_stream (row + 1024 + 16 * lane, vb.x, vb.y, vb.z, vb.w);
        __builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier ();
        __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "wavefront");
      });
}

static inline bool aligned (const void *p, size_t a) { return ((uintptr_t) p & (a - 1)) == 0; }

bool bilinear420_half_usable (const BilParams &bp, int n, const Planes *pl, uint8_t *const *dst, int dstride)
{
  if (!bp.half || (dstride % 16) != 0)
    return false;
  /* a wave needs a 1024-pixel source column to itself: one frame of less than ~4 M outputs is a few hundred waves with long serial walks, and the
   * rows kernel's narrower tiles win (4K -> 1080p, one frame: 12.6 us against 15.2; 8K -> 4K: 24.2 against 22.8; in lists this kernel, 14.3 against 18) */
  if (n == 1 && (long) bp.out_w * bp.out_h < 4000000 && !tuning_on ("GSTAMD_BIL_HALF_SMALL"))
    return false;
  for (int f = 0; f < n; f++) {
    const Planes &q = pl[f];
    if (!aligned (q.p[0], 16) || (q.stride[0] % 16) != 0 || !aligned (dst[f], 16) || q.stride[0] != pl[0].stride[0] || q.stride[1] != pl[0].stride[1] ||
        q.stride[2] != pl[0].stride[2])
      return false;
    if (bp.planar ? (!aligned (q.p[1], 8) || !aligned (q.p[2], 8) || (q.stride[1] % 8) != 0 || (q.stride[2] % 8) != 0) : (!aligned (q.p[1], 16) || (q.stride[1] % 16) != 0))
      return false;
  }
  return true;
}

// waves the device holds at once
static int bilh_wave_slots ()
{
  static int slots = 0;
  if (slots == 0) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice (&dev) != hipSuccess || hipDeviceGetAttribute (&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    const void *fn = (const void *) k_bilinear420_half<CHROMA_H_H2_CS, GSTAMD_LAYOUT (2, 1, 0)>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor (&per_cu, fn, 64, 2048) != hipSuccess || per_cu <= 0)
      per_cu = 16;
    (void) hipGetLastError ();
    slots = cus * per_cu;
  }
  return slots;
}

hipError_t launch_bilinear420_half (const BilParams &bp, int chroma_h, int n, const Planes *pl, uint8_t *const *dst, int dstride, hipStream_t stream)
{
  const int tiles = (bp.fp.width + BILH_TILE_SRC - 1) / BILH_TILE_SRC;
  BilParams rp = bp;
  int slots = bilh_wave_slots (), rows = n > 1 ? 8 : -1, wg = 1;
#ifdef GSTAMD_TUNING
  if (tuning_on ("GSTAMD_BIL_SLOTS"))
    slots = tuning_int ("GSTAMD_BIL_SLOTS", slots);
  if (tuning_on ("GSTAMD_BIL_HALF_ROWS"))
    rows = tuning_int ("GSTAMD_BIL_HALF_ROWS", rows);
  if (tuning_on ("GSTAMD_BIL_WG"))
    wg = tuning_int ("GSTAMD_BIL_WG", wg);
  if (tuning_on ("GSTAMD_BIL_HALF_STORE"))
    rp.half = tuning_int ("GSTAMD_BIL_HALF_STORE", 1);
#endif
  /* one frame: as many strips as make one resident round (with a margin: a round that tips over into a second one costs 20 %);
   * several frames: the rounds follow each other anyway, strips of eight rows */
  rp.strips = bilh_strips (bp.out_h, rows, tiles, slots - slots / 16);
#ifdef GSTAMD_TUNING
  if (tuning_on ("GSTAMD_BIL_VERBOSE"))
    fprintf (stderr, "k_bilinear420_half: %d frame(s), slots %d, tiles %d, strips %d\n", n, slots, tiles, rp.strips);
#endif
  const int blocks_per_frame = wide_grid_blocks (tiles, rp.strips) / wg;
  for (int base = 0; base < n; base += GSTAMD_BIL_MAX_BATCH) {
    const int nb = n - base < GSTAMD_BIL_MAX_BATCH ? n - base : GSTAMD_BIL_MAX_BATCH;
    BilBatch fb;
    memset (&fb, 0, sizeof (fb));
    for (int f = 0; f < nb; f++) {
      for (int k = 0; k < 3; k++)
        fb.p[k][f] = pl[base + f].p[k];
      fb.dst[f] = dst[base + f];
    }
    for (int k = 0; k < 3; k++)
      fb.stride[k] = pl[0].stride[k];
    dim3 grid ((unsigned) blocks_per_frame * (unsigned) nb), block (64 * wg);
#define W(pr, pg, pb) case GSTAMD_LAYOUT (pr, pg, pb): \
    if (chroma_h == CHROMA_H_H2_CS) \
      hipLaunchKernelGGL ((k_bilinear420_half<CHROMA_H_H2_CS, GSTAMD_LAYOUT (pr, pg, pb)>), grid, block, (size_t) wg * 2048, stream, rp, fb, dstride, tiles, blocks_per_frame); \
    else if (chroma_h == CHROMA_H_H2) \
      hipLaunchKernelGGL ((k_bilinear420_half<CHROMA_H_H2, GSTAMD_LAYOUT (pr, pg, pb)>), grid, block, (size_t) wg * 2048, stream, rp, fb, dstride, tiles, blocks_per_frame); \
    else \
      hipLaunchKernelGGL ((k_bilinear420_half<CHROMA_H_NONE, GSTAMD_LAYOUT (pr, pg, pb)>), grid, block, (size_t) wg * 2048, stream, rp, fb, dstride, tiles, blocks_per_frame); \
    break;
    switch (bp.fp.ayuv ? GSTAMD_LAYOUT_AYUV : GSTAMD_LAYOUT (bp.fp.pack_pos[1], bp.fp.pack_pos[2], bp.fp.pack_pos[3])) {
      W (0, 0, 4)      /* GSTAMD_LAYOUT_AYUV: no colour stage (FastParams::ayuv) */
      W (2, 1, 0)      /* BGRA, BGRx */
      W (0, 1, 2)      /* RGBA, RGBx */
      W (1, 2, 3)      /* ARGB, xRGB */
      W (3, 2, 1)      /* ABGR, xBGR */
      default:
        return hipErrorInvalidValue;
    }
#undef W
  }
  return hipGetLastError ();
}

}  // namespace gstamd
