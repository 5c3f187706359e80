// compositor_walk.h - scaled pads as a COLUMN WALK inside the blend kernel (BASELINE C4 variant A: sixteen 1080p pads, each halved by
// its GstVideoAggregatorConvertPad converter, gstvideoaggregator.c:479-513, and blended, compositor.c:1678-1697).
//
// What is computed is compositor_scaled.h's: the converter of such a pad is convert_scale_planes (video-converter.c:7757) =
// gst_video_scaler_2d (video-scaler.c:1451) with the vertical N-tap pass first (video_scale_v_ntap_u8 :987-1072: 16-bit wrapping sum of
// u8 x s16, (sum + 32) >> 6, clamped to u8), then the horizontal one on those bytes (video_scale_h_ntap_u8 :621-760), then the blend.
// k_aggregate_scaled evaluates that tile by tile with run-time tap loops (C4-A: 33 M VALU + 27 M SALU wave-instructions per frame,
// 103 us, 1.62 x the algorithmic bytes).  Here, for the case every mosaic / video wall is - EXACT HALVINGS with 8-tap filters (cubic,
// the converters' default) and scaled pads that do not overlap each other:
//
//   a WAVE owns a strip of <= 61 output columns of one pad (the 128 source columns under it: two per lane) and walks DOWN a run of
//   output rows.  The lane keeps the last eight source rows of its two columns in registers, unpacked into 16-bit lanes (bytes 0, 2 and
//   1, 3 of a pixel apart) the moment they arrive - every source byte is loaded once per walk and unpacked once, not once per tap.  Per
//   output row: two new source rows (one 8-byte buffer load each, requested a row ahead), the vertical pass as v_pk_mad_u16 with the
//   row's taps in SCALAR registers (op_sel picks the half: no splat), the 128 intermediate pixels through 1 KB of the wave's own LDS
//   (in-order LDS queue: no barrier anywhere), the horizontal pass from four 16-byte LDS reads with the lane's own taps, then the blend
//   on the packed halves the pass leaves behind (compositor_device.h px2 form) and one 4-byte store.
//   (Tried and dropped: a half-wave per row parity - ONE 16-byte load instruction per output row, partial sums exchanged with
//   v_permlane32_swap_b32.  Half the vector-memory instructions, 12 more VALU per row and 84 registers instead of 76: 56.5 us against 51.)
//
// Windows are REMAPPED on the host (capi_video.cpp walk_tables): output i of a pass reads ring positions 2 i + base .. + 7 with the
// pass's own taps moved to those positions (edge outputs, whose windows the resampler folds into the picture, have zeros where the
// ring lies outside it); a pass whose non-zero taps do not fit that ring is not eligible and the pad set takes k_aggregate_scaled.
// Unscaled pads may lie anywhere (below or above the scaled ones): the walk blends them in z order where they cross its strip, and a
// filler grid writes the canvas pixels no scaled pad covers.
#pragma once
#include "compositor_scaled.h"
#include "video_scale_col.h"

namespace gstamd {

#define GSTAMD_WALK_MAX_PADS 16
#define GSTAMD_WALK_TILE 61             /* outputs per wave: lane l reads intermediate columns 2 l .. 2 l + 7 of 128 */

struct WalkPad {
  const uint8_t *src;           // the frame as it arrived
  int sstride, src_w, src_h;
  int xpos, ypos, ow, oh;       // the scaled pad on the canvas
  int s_alpha, mode;
  const uint32_t *vt, *ht;      // [oh][4] / [ow][4]: eight s16 taps per output at ring positions 0 .. 7 (two per word, even position low)
  int vbase, hbase;             // ring position 0 of output i is source row / column 2 i + base
  int tiles, tile_w, chunks, chunk_rows;
  int first_block;
  int n_below;                  // plain pads [0, n_below) lie under this pad, the others above it
};

struct WalkParams {
  int ashift, overlay, bg_kind, checker_yuv;
  uint32_t bg_word;
  int n_walk, n_plain;
  int n_blocks;                 // walk blocks; the filler's come after them
  int xcd_span;                 // > 0: hardware block b is walk block (b & 7) * xcd_span + (b >> 3) for b < 8 * xcd_span
  int fill_tiles_x, fill_tiles_y;       // filler grid (64 x 16 pixel tiles), 0 x 0: the scaled pads cover the canvas
  WalkPad walk[GSTAMD_WALK_MAX_PADS];
  PadDev plain[GSTAMD_WALK_MAX_PADS];   // unscaled pads in z order
};

#ifdef __HIPCC__
// a * (low / high half of t) + c in both 16-bit lanes, wrapping: v_pk_mad_u16 with the tap selected by op_sel
__device__ __forceinline__ uint32_t walk_mad_lo (uint32_t a, uint32_t t, uint32_t c)
{
  uint32_t r;
  asm ("v_pk_mad_u16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v" (r) : "v" (a), "v" (t), "v" (c));
  return r;
}
__device__ __forceinline__ uint32_t walk_mad_hi (uint32_t a, uint32_t t, uint32_t c)
{
  uint32_t r;
  asm ("v_pk_mad_u16 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v" (r) : "v" (a), "v" (t), "v" (c));
  return r;
}
__device__ __forceinline__ uint32_t walk_mad_lo_s (uint32_t a, uint32_t t_uniform, uint32_t c)
{
  uint32_t r;
  asm ("v_pk_mad_u16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v" (r) : "v" (a), "s" (t_uniform), "v" (c));
  return r;
}
__device__ __forceinline__ uint32_t walk_mad_hi_s (uint32_t a, uint32_t t_uniform, uint32_t c)
{
  uint32_t r;
  asm ("v_pk_mad_u16 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v" (r) : "v" (a), "s" (t_uniform), "v" (c));
  return r;
}

// per 16-bit lane: clamp ((int16) acc >> 6, 0, 255) - the accumulators start at 32, so this is pk_lq_finish
__device__ __forceinline__ uint32_t walk_finish (uint32_t acc)
{
  typedef short s2 __attribute__ ((ext_vector_type (2)));
  s2 v = __builtin_bit_cast (s2, acc) >> (short) 6;
  v = __builtin_elementwise_min (__builtin_elementwise_max (v, (s2) (short) 0), (s2) (short) 255);
  return __builtin_bit_cast (uint32_t, v);
}

__device__ __forceinline__ uint32_t pk_add16 (uint32_t a, uint32_t b)
{
  typedef unsigned short us2 __attribute__ ((ext_vector_type (2)));
  return __builtin_bit_cast (uint32_t, (us2) (__builtin_bit_cast (us2, a) + __builtin_bit_cast (us2, b)));
}

// four wave-uniform table words (s_load_dwordx4)
__device__ __forceinline__ void walk_entry4 (const uint32_t *table, int idx, uint32_t *e)
{
  typedef const __attribute__ ((address_space (4))) uint32_t *cptr_t;
  cptr_t t = (cptr_t) (uintptr_t) table + (size_t) __builtin_amdgcn_readfirstlane (idx) * 4;
#pragma unroll
  for (int k = 0; k < 4; k++)
    e[k] = t[k];
}

// 8 bytes of a source row: the lane's two columns.  `col_bytes` may be negative in a pad's first strip (ring positions left of the picture:
// zero taps): buffer offsets do not wrap, a load that STARTS out of range returns zeros for all of its dwords - also for the one inside
// the row.  Such lanes load from the row's first byte and move the word up (walk_shift_up).
__device__ __forceinline__ void walk_request (colplane_t src, int row, int src_h, int sstride, int col_bytes, uint32_t *w)
{
  typedef uint32_t u32x2 __attribute__ ((ext_vector_type (2)));
  const int y = row < 0 ? 0 : (row > src_h - 1 ? src_h - 1 : row);              /* rows outside the picture meet zero taps only */
  const u32x2 d = __builtin_amdgcn_raw_buffer_load_b64 (src, (int) ((uint32_t) y * (uint32_t) sstride + (uint32_t) (col_bytes < 0 ? 0 : col_bytes)), 0, 0);
  w[0] = d.x, w[1] = d.y;
}

__device__ __forceinline__ void walk_shift_up (uint32_t *w, int shift)          /* shift: words the load was moved down by (0, 1, >= 2) */
{
  const uint32_t a = w[0];
  w[1] = shift == 1 ? a : (shift == 0 ? w[1] : 0u);
  w[0] = shift == 0 ? a : 0u;
}

// the walk of one wave.  ASH: alpha byte of the format family (0 / 24); FAST: every blend is the opaque-destination one (no
// transparent background, no SOURCE operator); PLAIN: there are unscaled pads
template <int ASH, int FAST, int PLAIN>
__device__ __forceinline__ void walk_wave (const WalkParams &p, const WalkPad &wp, int tile, int chunk, uint8_t *dst, int dstride, int dw, int dh, uint32_t *lds)
{
  const int lane = (int) threadIdx.x;
  const int x0 = tile * wp.tile_w;
  const int n_out = wp.ow - x0 < wp.tile_w ? wp.ow - x0 : wp.tile_w;
  int r0 = chunk * wp.chunk_rows, r1 = r0 + wp.chunk_rows < wp.oh ? r0 + wp.chunk_rows : wp.oh;
  if (r0 < -wp.ypos)
    r0 = -wp.ypos;                      /* rows above / below the canvas are nobody's */
  if (r1 > dh - wp.ypos)
    r1 = dh - wp.ypos;
  if (r0 >= r1 || wp.xpos + x0 >= dw || wp.xpos + x0 + n_out <= 0)
    return;
  // loop invariants the compiler would otherwise fetch from the kernel arguments again in every row (a scalar load and its wait per use)
  int k_vbase = wp.vbase, k_sstride = wp.sstride, k_src_h = wp.src_h, k_ypos = wp.ypos, k_bg = p.bg_kind, k_yuv = p.checker_yuv;
  uint32_t k_bgw = p.bg_word;
  const uint32_t *k_vt = wp.vt;
  asm volatile ("" : "+s" (k_vbase), "+s" (k_sstride), "+s" (k_src_h), "+s" (k_ypos), "+s" (k_bg), "+s" (k_yuv), "+s" (k_bgw), "+s" (k_vt));
  const colplane_t src = col_plane (wp.src, 0, (uint32_t) wp.sstride * (uint32_t) wp.src_h);
  const int col_bytes = 4 * (2 * x0 + wp.hbase + 2 * lane);
  const bool fix_left = __builtin_amdgcn_readfirstlane (2 * x0 + wp.hbase) < 0;         /* wave-uniform: a pad's first strip */
  const int shift = col_bytes < 0 ? -col_bytes >> 2 : 0;
  // the lane's horizontal taps (lanes past the strip's end repeat its last output)
  const int xl = lane < n_out ? lane : n_out - 1;
  typedef uint32_t u32x4g __attribute__ ((ext_vector_type (4)));
  const u32x4g ht = *(const u32x4g *) (wp.ht + 4 * (size_t) (x0 + xl));
  const int cx = wp.xpos + x0 + lane;                                   /* canvas column of the lane's output */
  const bool store = lane < n_out && cx >= 0 && cx < dw;
  // plain pads that cross the strip (wave-uniform mask)
  uint32_t crossing = 0;
  for (int i = 0; PLAIN && i < p.n_plain; i++) {
    const PadDev &q = p.plain[i];
    if (q.xpos < wp.xpos + x0 + n_out && q.xpos + q.width > wp.xpos + x0 && q.ypos < wp.ypos + r1 && q.ypos + q.height > wp.ypos + r0)
      crossing |= 1u << i;
  }
  const uint32_t alpha8081 = (uint32_t) wp.s_alpha * 0x8081u;

  uint32_t E[8][2], O[8][2];                    // [ring slot][column]
  uint32_t raw[2][2];                           // the two new rows of the next output row
  auto unpack = [&](int slot, uint32_t *w) {
    if (fix_left)
      walk_shift_up (w, shift);
#pragma unroll
    for (int c = 0; c < 2; c++) {
      E[slot][c] = cbperm (0u, w[c], 0x0c020c00u);
      O[slot][c] = cbperm (0u, w[c], 0x0c030c01u);
    }
  };
  // prime: ring positions 0 .. 5 of row r0, then the requests of its two new rows
  const int vb0 = 2 * r0 + k_vbase;
#pragma unroll
  for (int k = 0; k < 6; k += 2) {
    walk_request (src, vb0 + k, k_src_h, k_sstride, col_bytes, raw[0]);
    walk_request (src, vb0 + k + 1, k_src_h, k_sstride, col_bytes, raw[1]);
    unpack (k, raw[0]);
    unpack (k + 1, raw[1]);
  }
  walk_request (src, vb0 + 6, k_src_h, k_sstride, col_bytes, raw[0]);
  walk_request (src, vb0 + 7, k_src_h, k_sstride, col_bytes, raw[1]);
  uint32_t vt[4];
  walk_entry4 (k_vt, r0, vt);
  // the canvas: the lane's pixel of row r0, one row further per output row; the checker's column phase
  /* the lane's pixel leaves through a buffer store that is ISSUED by every lane (lanes without a pixel aim past the canvas and are dropped by the range
     check): a store under `if (store)` is a branch when no lane stores, and after such a join the compiler no longer knows how many vector-memory
     operations are in flight - it waits for vmcnt(0) at the next row's unpack, i.e. for this row's store to reach memory */
  const colplane_t canvas = col_plane (dst, 0, (uint32_t) dstride * (uint32_t) dh);
  const uint32_t out_lane = store ? 4u * (uint32_t) cx : 0xfffffff0u;
  uint32_t out_row = (uint32_t) (k_ypos + r0) * (uint32_t) dstride;
  uint8_t *outp = dst + (size_t) (k_ypos + r0) * dstride + 4 * (ptrdiff_t) cx;
  const uint32_t chk_x = ((uint32_t) cx >> 3) & 1u;
  const int wl = lane <= 60 ? lane : 60;
  typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
  typedef __attribute__ ((address_space (3))) u32x4 *l4_t;
  const l4_t lds_mine = (l4_t) (lds + 4 * lane), lds_win = (l4_t) (lds + 4 * wl);

  /* the loop is entered with the vector-memory queue in the shape its back edge leaves it in (two prefetches, then a store): a store no lane takes part
     in, so that the wait counts of the first of every four rows are the exact ones too */
  if (FAST && !PLAIN)
    __builtin_amdgcn_raw_buffer_store_b32 (0u, canvas, (int) 0xfffffff0u, 0, 2);
  for (int rb = r0; rb < r1; rb += 4) {
#pragma unroll
    for (int ph = 0; ph < 4; ph++) {
      const int r = rb + ph;
      if (r >= r1)
        break;
      // the two new rows into the places of the two oldest; the next row's requests go out before anything is computed
      unpack ((2 * ph + 6) & 7, raw[0]);
      unpack ((2 * ph + 7) & 7, raw[1]);
      const uint32_t t0 = vt[0], t1 = vt[1], t2 = vt[2], t3 = vt[3];
      if (r + 1 < r1) {
        walk_request (src, 2 * (r + 1) + k_vbase + 6, k_src_h, k_sstride, col_bytes, raw[0]);
        walk_request (src, 2 * (r + 1) + k_vbase + 7, k_src_h, k_sstride, col_bytes, raw[1]);
        walk_entry4 (k_vt, r + 1, vt);
      }
      // vertical pass: ring position k is slot (2 ph + k) & 7; the row's taps sit in scalar registers
      uint32_t ve[2], vo[2];
#pragma unroll
      for (int c = 0; c < 2; c++) {
        uint32_t ae = 0x00200020u, ao = 0x00200020u;
#define GSTAMD_WALK_V(K, T, HALF) \
        ae = walk_mad_##HALF##_s (E[(2 * ph + K) & 7][c], T, ae), ao = walk_mad_##HALF##_s (O[(2 * ph + K) & 7][c], T, ao);
        GSTAMD_WALK_V (0, t0, lo) GSTAMD_WALK_V (1, t0, hi) GSTAMD_WALK_V (2, t1, lo) GSTAMD_WALK_V (3, t1, hi)
        GSTAMD_WALK_V (4, t2, lo) GSTAMD_WALK_V (5, t2, hi) GSTAMD_WALK_V (6, t3, lo) GSTAMD_WALK_V (7, t3, hi)
#undef GSTAMD_WALK_V
        ve[c] = walk_finish (ae), vo[c] = walk_finish (ao);
      }
      // the intermediate row through LDS: pixel c = (e, o) at words 2 c, 2 c + 1; the lane's window starts at its own first column
      const u32x4 mine = {ve[0], vo[0], ve[1], vo[1]};
      *lds_mine = mine;
      __builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier ();
      __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "wavefront");
      const u32x4 q0 = lds_win[0], q1 = lds_win[1], q2 = lds_win[2], q3 = lds_win[3];
      __builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier ();
      __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "wavefront");
      uint32_t he = 0x00200020u, ho = 0x00200020u;
      he = walk_mad_lo (q0.x, ht.x, he), ho = walk_mad_lo (q0.y, ht.x, ho);
      he = walk_mad_hi (q0.z, ht.x, he), ho = walk_mad_hi (q0.w, ht.x, ho);
      he = walk_mad_lo (q1.x, ht.y, he), ho = walk_mad_lo (q1.y, ht.y, ho);
      he = walk_mad_hi (q1.z, ht.y, he), ho = walk_mad_hi (q1.w, ht.y, ho);
      he = walk_mad_lo (q2.x, ht.z, he), ho = walk_mad_lo (q2.y, ht.z, ho);
      he = walk_mad_hi (q2.z, ht.z, he), ho = walk_mad_hi (q2.w, ht.z, ho);
      he = walk_mad_lo (q3.x, ht.w, he), ho = walk_mad_lo (q3.y, ht.w, ho);
      he = walk_mad_hi (q3.z, ht.w, he), ho = walk_mad_hi (q3.w, ht.w, ho);
      const uint32_t se = walk_finish (he), so = walk_finish (ho);      /* the scaled pixel: bytes 0, 2 / 1, 3 in 16-bit lanes */
      // blend
      const int cy = k_ypos + r;
      uint32_t *out = (uint32_t *) outp;
      outp += dstride;
      uint32_t d;
      if (k_bg == 0) {                  /* checker_px: 160 where exactly one of (x & 8), (y & 8) is set, else 80 */
        const uint32_t val = (chk_x ^ (((uint32_t) cy >> 3) & 1u)) ? 160u : 80u;
        d = !k_yuv ? (ASH == 0 ? (0xffu | (val * 0x01010100u)) : ((val * 0x00010101u) | 0xff000000u))
            : (ASH == 0 ? (0x808000ffu | (val << 8)) : (0xff008080u | (val << 16)));
      } else {
        /* a constant word (black, white, transparent).  No load of the canvas in this loop, ever: on gfx9 one vmcnt counts everything, so a load on ANY
           path to the blend makes the compiler wait for vmcnt(0) there - for the next row's two prefetches too, and every row pays the memory latency
           the prefetch was meant to hide (round 6: C4-A 47.4 -> see DESIGN 12.7) */
        d = k_bgw;
      }
      if (FAST && (!PLAIN || !crossing)) {
        Px2 acc = px2_unpack (d);
        const uint32_t a = mul24 (ASH == 0 ? (se & 0xffu) : (so >> 16), alpha8081) >> 23;
        const uint32_t as = a | (a << 16), ias = 0x00ff00ffu - as, one = pk_one ();
        const uint32_t te = cpk_mad16 (se, as, cpk_mad16 (acc.e, ias, one));
        const uint32_t to = cpk_mad16 (so, as, cpk_mad16 (acc.o, ias, one));
        acc.e = pk16_shr8 (te + pk16_shr8 (te));
        acc.o = pk16_shr8 (to + pk16_shr8 (to));
        d = px2_pack (acc) | (0xffu << ASH);
      } else {
        const uint32_t s = se | (so << 8);
        uint32_t m = crossing;
        for (int i = 0; i < p.n_plain; i++, m >>= 1) {
          if (i == wp.n_below)
            d = apply_pad (d, s, wp.s_alpha, wp.mode, ASH, p.overlay);
          if (m & 1u) {
            const PadDev &q = p.plain[i];
            const int sx = cx - q.xpos, sy = cy - q.ypos;
            if (sx >= 0 && sy >= 0 && sx < q.width && sy < q.height)
              d = apply_pad (d, load_px1 (q.data + (size_t) sy * q.stride + 4 * (size_t) sx), q.s_alpha, q.mode, ASH, p.overlay);
          }
        }
        if (wp.n_below >= p.n_plain)
          d = apply_pad (d, s, wp.s_alpha, wp.mode, ASH, p.overlay);
      }
      if (FAST && !PLAIN)
        __builtin_amdgcn_raw_buffer_store_b32 (d, canvas, (int) out_lane, (int) __builtin_amdgcn_readfirstlane (out_row), 2 /* nt */);
      else if (store)
        __builtin_nontemporal_store (d, out);
      out_row += (uint32_t) dstride;
    }
  }
}

// canvas pixels no scaled pad covers: background + the plain pads (64 x 16 tiles, one lane per column, sixteen rows)
template <int ASH>
__device__ __forceinline__ void walk_fill (const WalkParams &p, int tile, uint8_t *dst, int dstride, int dw, int dh)
{
  const int tx = tile % p.fill_tiles_x, ty = tile / p.fill_tiles_x;
  const int x = tx * 64 + (int) threadIdx.x;
  const int y0 = ty * 16, y1 = y0 + 16 < dh ? y0 + 16 : dh;
  // the tile lies inside one scaled pad: nothing to do (wave-uniform)
  for (int i = 0; i < p.n_walk; i++) {
    const WalkPad &w = p.walk[i];
    if (tx * 64 >= w.xpos && tx * 64 + 64 <= w.xpos + w.ow && y0 >= w.ypos && y1 <= w.ypos + w.oh)
      return;
  }
  if (x >= dw)
    return;
  for (int y = y0; y < y1; y++) {
    bool covered = false;
    for (int i = 0; i < p.n_walk; i++) {
      const WalkPad &w = p.walk[i];
      covered = covered || (x >= w.xpos && x < w.xpos + w.ow && y >= w.ypos && y < w.ypos + w.oh);
    }
    if (covered)
      continue;
    uint32_t *out = (uint32_t *) (dst + (size_t) y * dstride) + x;
    uint32_t d = p.bg_kind == 0 ? checker_px (x, y, ASH, p.checker_yuv) : (p.bg_kind == 1 ? p.bg_word : *out);
    for (int i = 0; i < p.n_plain; i++) {
      const PadDev &q = p.plain[i];
      const int sx = x - q.xpos, sy = y - q.ypos;
      if (sx >= 0 && sy >= 0 && sx < q.width && sy < q.height)
        d = apply_pad (d, load_px1 (q.data + (size_t) sy * q.stride + 4 * (size_t) sx), q.s_alpha, q.mode, ASH, p.overlay);
    }
    *out = d;
  }
}
#endif

}  // namespace gstamd
