// compositor_device.h - per-pixel device code of the compositor blend kernels (bodies only; see
// video_device.h for the host-emulation arrangement).
//
// Reference semantics reproduced bit-exactly (paths under
// /root/reference/subprojects/gst-plugins-base/gst/compositor/):
//   BLEND_A32 clipping                      blend.c:41-99
//   _blend_loop_* / _overlay_loop_*         blend.c:101-158
//   compositor_orc_blend_argb/bgra          compositororc.orc:158-195, 225-265  (C: compositororc-dist.c:1900-2160)
//   compositor_orc_source_argb/bgra         compositororc.orc:196-224, 266-294
//   compositor_orc_overlay_* (+_addition)   compositororc.orc:295-559 (divluw: compositororc-dist.c:3345)
//   fill_checker_*_c / fill_color_*         blend.c:177-240
//   _draw_background + blend_pads           compositor.c:1619-1697
//
// "argb" family = alpha in memory byte 0 (ARGB, ABGR, AYUV); "bgra" family = alpha in byte 3
// (BGRA, RGBA): blend.h:56-61.  A pixel is the little-endian uint32 of its 4 memory bytes.
#pragma once
#include <stdint.h>
#include <stddef.h>

#include "../../include/gstamd_video.h"

#ifdef __HIPCC__
#define GSTAMD_CD __device__ __forceinline__
#else
#define GSTAMD_CD inline
#endif

namespace gstamd {

#define GSTAMD_MAX_FUSED_PADS 32

struct PadDev {
  const uint8_t *data;
  int width, height, stride;
  int xpos, ypos;
  int s_alpha;      // CLAMP ((gint) (src_alpha * 255), 0, 255)
  int mode;         // GstCompositorBlendMode
};

struct AggregateParams {
  int ashift;       // 0 ("argb" family) or 24 ("bgra" family): bit position of alpha in the LE word
  int overlay;      // 1: transparent background -> overlay functions
  int bg_kind;      // 0 checker, 1 solid colour word, 2 keep destination (continuation chunk)
  int checker_yuv;  // checker for AYUV: Y = tab, U = V = 128
  uint32_t bg_word;
  int n_pads;
  int fast;         // every pad takes the opaque blend (px2_blend): !overlay and no SOURCE operator
  PadDev pads[GSTAMD_MAX_FUSED_PADS];
};

GSTAMD_CD uint32_t div255w (uint32_t x) { return ((x & 0xffffu) * 0x8081u) >> 23; }

// compositor_orc_blend_*: opaque destination
GSTAMD_CD uint32_t px_blend (uint32_t s, uint32_t d, uint32_t alpha, int ashift)
{
  const uint32_t a = div255w (((s >> ashift) & 0xff) * alpha);
  const uint32_t ia = (0xffu - a) & 0xffffu;
  uint32_t r = 0;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const uint32_t sw = (((s >> (8 * c)) & 0xff) * a) & 0xffffu;
    const uint32_t dw = (((d >> (8 * c)) & 0xff) * ia) & 0xffffu;
    r |= (div255w ((dw + sw) & 0xffffu) & 0xff) << (8 * c);
  }
  return r | (0xffu << ashift);
}

// compositor_orc_source_*: copy colour, scale alpha
GSTAMD_CD uint32_t px_source (uint32_t s, uint32_t alpha, int ashift)
{
  const uint32_t a = div255w (((s >> ashift) & 0xff) * alpha) & 0xff;
  return (s & ~(0xffu << ashift)) | (a << ashift);
}

// compositor_orc_overlay_* and _addition: destination may be transparent
GSTAMD_CD uint32_t px_overlay (uint32_t s, uint32_t d, uint32_t alpha, int ashift, bool addition)
{
  const uint32_t alpha_s = div255w (((s >> ashift) & 0xff) * alpha);
  const uint32_t alpha_s_inv = (0xffu - alpha_s) & 0xffffu;
  const uint32_t da = (d >> ashift) & 0xff;
  uint32_t alpha_d = div255w ((da * alpha_s_inv) & 0xffffu);
  const uint32_t norm = (alpha_d + alpha_s) & 0xffffu;          // final alpha (over) / alpha factor (add)
  uint32_t r = 0;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const uint32_t sw = (((s >> (8 * c)) & 0xff) * alpha_s) & 0xffffu;
    const uint32_t dw = (((d >> (8 * c)) & 0xff) * alpha_d) & 0xffffu;
    const uint32_t sum = (dw + sw) & 0xffffu;
    uint32_t q;
    if ((norm & 0xff) == 0)
      q = 255;
    else {
      q = sum / (norm & 0xff);                                  // divluw
      q = q > 255 ? 255 : q;
    }
    r |= (q & 0xff) << (8 * c);
  }
  const uint32_t out_a = addition ? ((da + alpha_s) & 0xff) : (norm & 0xff);
  return (r & ~(0xffu << ashift)) | (out_a << ashift);
}

// one pad applied to one destination pixel value (BlendFunction semantics, any background)
GSTAMD_CD uint32_t apply_pad (uint32_t d, uint32_t s, int s_alpha, int mode, int ashift, int overlay)
{
  if (mode == GSTAMD_COMPOSITOR_BLEND_MODE_SOURCE)
    return s_alpha == 255 ? s : px_source (s, (uint32_t) s_alpha, ashift);
  if (!overlay)
    return px_blend (s, d, (uint32_t) s_alpha, ashift);
  return px_overlay (s, d, (uint32_t) s_alpha, ashift, mode == GSTAMD_COMPOSITOR_BLEND_MODE_ADD);
}

GSTAMD_CD uint32_t checker_px (int x, int y, int ashift, int yuv)
{
  const uint32_t val = (((y & 0x8) >> 3) + ((x & 0x8) >> 3)) == 1 ? 160u : 80u;   // tab = {80,160,80,160}
  if (!yuv)
    return (ashift == 0) ? (0xffu | (val << 8) | (val << 16) | (val << 24)) : (val | (val << 8) | (val << 16) | (0xffu << 24));
  return (ashift == 0) ? (0xffu | (val << 8) | (128u << 16) | (128u << 24)) : (128u | (128u << 8) | (val << 16) | (0xffu << 24));
}

// fused aggregate: value of destination pixel (x, y) after background + all pads of this chunk
GSTAMD_CD uint32_t aggregate_px (const AggregateParams &p, uint32_t dest_in, int x, int y)
{
  uint32_t d = p.bg_kind == 0 ? checker_px (x, y, p.ashift, p.checker_yuv) : (p.bg_kind == 1 ? p.bg_word : dest_in);
  for (int i = 0; i < p.n_pads; i++) {
    const PadDev &pad = p.pads[i];
    const int sx = x - pad.xpos, sy = y - pad.ypos;
    if (sx >= 0 && sy >= 0 && sx < pad.width && sy < pad.height) {
      const uint32_t s = *(const uint32_t *) (pad.data + (size_t) sy * pad.stride + 4 * (size_t) sx);
      d = apply_pad (d, s, pad.s_alpha, pad.mode, p.ashift, p.overlay);
    }
  }
  return d;
}

// ------------------------------------------------------------------------------------------------
// 4 pixels per lane.  Opaque-destination blend (compositor_orc_blend_*) on two packed halves:
// e = bytes 0,2 and o = bytes 1,3 of the pixel, each in its own 16-bit lane of a 32-bit register.
// All reference intermediates are < 2^16 here (s*a + d*(255-a) <= 255*255), so the "& 0xffff" wraps of
// the ORC program never trigger and one 24-bit multiply serves two channels; div255w (x) =
// (x * 0x8081) >> 23 equals (x + 1 + ((x + 1) >> 8)) >> 8 for every x <= 65025 (checked exhaustively
// in tests/test_compositor.py), which needs no multiply and works per 16-bit lane.
struct Px2 {
  uint32_t e, o;
};

GSTAMD_CD uint32_t pk16_shr8 (uint32_t x)
{
#ifdef __HIPCC__
  typedef unsigned short us2 __attribute__ ((ext_vector_type (2)));
  us2 v = __builtin_bit_cast (us2, x);
  v = v >> (unsigned short) 8;
  return __builtin_bit_cast (uint32_t, v);
#else
  return (x >> 8) & 0x00ff00ffu;
#endif
}

GSTAMD_CD Px2 px2_unpack (uint32_t v)
{
  Px2 r;
  r.e = v & 0x00ff00ffu;
  r.o = (v >> 8) & 0x00ff00ffu;
  return r;
}

GSTAMD_CD uint32_t px2_pack (const Px2 &v) { return v.e | (v.o << 8); }

GSTAMD_CD uint32_t pk16_div255 (uint32_t x)
{
  const uint32_t t = x + 0x00010001u;
  return pk16_shr8 (t + pk16_shr8 (t));
}

// 24-bit x 24-bit multiply (v_mul_u32_u24); every operand below is < 2^24
GSTAMD_CD uint32_t mul24 (uint32_t a, uint32_t b)
{
#ifdef __HIPCC__
  uint32_t r;
  asm ("v_mul_u32_u24 %0, %1, %2" : "=v" (r) : "v" (a), "v" (b));
  return r;
#else
  return a * b;
#endif
}

// same with a wave-uniform second factor (pad alpha, constants)
GSTAMD_CD uint32_t mul24_uniform (uint32_t a, uint32_t b)
{
#ifdef __HIPCC__
  uint32_t r;
  asm ("v_mul_u32_u24 %0, %1, %2" : "=v" (r) : "v" (a), "s" (b));
  return r;
#else
  return a * b;
#endif
}

// a * b + c per 16-bit lane (v_pk_mad_u16); nothing here exceeds 65535, so the lanes never wrap
GSTAMD_CD uint32_t cpk_mad16 (uint32_t a, uint32_t b, uint32_t c)
{
#ifdef __HIPCC__
  typedef unsigned short us2 __attribute__ ((ext_vector_type (2)));
  return __builtin_bit_cast (uint32_t, (us2) (__builtin_bit_cast (us2, a) * __builtin_bit_cast (us2, b) + __builtin_bit_cast (us2, c)));
#else
  const uint32_t lo = ((a & 0xffffu) * (b & 0xffffu) + (c & 0xffffu)) & 0xffffu;
  const uint32_t hi = ((a >> 16) * (b >> 16) + (c >> 16)) & 0xffffu;
  return lo | (hi << 16);
#endif
}

// byte `ashift / 8` of s times a wave-uniform 24-bit factor: the byte select rides in the multiply (SDWA)
template <int ASH>
GSTAMD_CD uint32_t mul24_alpha_byte (uint32_t s, uint32_t k)
{
#ifdef __HIPCC__
  uint32_t r;
  if (ASH == 0)
    asm ("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v" (r) : "v" (s), "v" (k));
  else
    asm ("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v" (r) : "v" (s), "v" (k));
  return r;
#else
  return ((s >> ASH) & 0xffu) * k;
#endif
}

GSTAMD_CD uint32_t cbperm (uint32_t hi, uint32_t lo, uint32_t sel)
{
#ifdef __HIPCC__
  return __builtin_amdgcn_perm (hi, lo, sel);
#else
  uint32_t r = 0;
  for (int i = 0; i < 4; i++) {
    const uint32_t q = (sel >> (8 * i)) & 0xff;
    const uint32_t b = q < 4 ? (lo >> (8 * q)) & 0xff : (q < 8 ? (hi >> (8 * (q - 4))) & 0xff : (q == 0x0c ? 0u : 0xffu));
    r |= b << (8 * i);
  }
  return r;
#endif
}

// alpha8081 = pad alpha * 0x8081 (< 2^24): a = div255w (sA * alpha) = (sA * alpha * 0x8081) >> 23 in one multiply.  Both channels
// of a half then take   t = s * a + d * (255 - a) + 1   as two packed multiply-adds (the + 1 of the division rides in the first),
// and div255 finishes as (t + (t >> 8)) >> 8.  The destination's alpha lane is NOT forced to 0xff here: it never feeds another
// lane, every blend recomputes it from bytes <= 255, and aggregate_span4 sets it once after the last pad (16 VALU per
// pixel-blend instead of 24).
// 0x00010001 in a register the compiler cannot see through: as a literal it splits `d * ias + 1` into v_pk_mul_lo_u16 and a
// v_pk_add_u16 with an inline 1 (VOP3P takes no 32-bit literal) - 18 instructions per pixel-blend instead of 16
GSTAMD_CD uint32_t pk_one ()
{
#ifdef __HIPCC__
  uint32_t r;
  asm ("v_mov_b32 %0, 0x10001" : "=v" (r));
  return r;
#else
  return 0x00010001u;
#endif
}

template <int ASH>
GSTAMD_CD void px2_blend_lazy (Px2 &d, uint32_t s, uint32_t alpha8081)
{
  const uint32_t a = mul24_alpha_byte<ASH> (s, alpha8081) >> 23;
  const uint32_t as = a | (a << 16), ias = 0x00ff00ffu - as;
  const uint32_t se = cbperm (0, s, 0x0c020c00u), so = cbperm (0, s, 0x0c030c01u);
  const uint32_t one = pk_one ();
  const uint32_t te = cpk_mad16 (se, as, cpk_mad16 (d.e, ias, one));
  const uint32_t to = cpk_mad16 (so, as, cpk_mad16 (d.o, ias, one));
  d.e = pk16_shr8 (te + pk16_shr8 (te));
  d.o = pk16_shr8 (to + pk16_shr8 (to));
}

GSTAMD_CD void px2_blend (Px2 &d, uint32_t s, uint32_t alpha8081, int ashift)
{
  if (ashift == 0) {
    px2_blend_lazy<0> (d, s, alpha8081);
    d.e |= 0xffu;
  } else {
    px2_blend_lazy<24> (d, s, alpha8081);
    d.o |= 0x00ff0000u;
  }
}

struct __attribute__ ((aligned (4))) Px4Words {
  uint32_t v[4];
};

// 16 bytes at a 4-byte aligned address of a pad image (device: global address space, so the pointer that came
// through LDS does not turn the access into a flat load)
GSTAMD_CD Px4Words load_px4 (const uint8_t *p)
{
  Px4Words r;
#ifdef __HIPCC__
  const __attribute__ ((address_space (1))) uint32_t *g = (const __attribute__ ((address_space (1))) uint32_t *) (uintptr_t) p;
#else
  const uint32_t *g = (const uint32_t *) p;
#endif
#if defined(__HIPCC__) && defined(GSTAMD_AGG_NT_LOADS)
  typedef unsigned int u32x4_a4l __attribute__ ((ext_vector_type (4), aligned (4)));
  const u32x4_a4l v = __builtin_nontemporal_load ((const __attribute__ ((address_space (1))) u32x4_a4l *) g);
  r.v[0] = v.x, r.v[1] = v.y, r.v[2] = v.z, r.v[3] = v.w;
#else
  r.v[0] = g[0];
  r.v[1] = g[1];
  r.v[2] = g[2];
  r.v[3] = g[3];
#endif
  return r;
}

GSTAMD_CD uint32_t load_px1 (const uint8_t *p)
{
#ifdef __HIPCC__
  return *(const __attribute__ ((address_space (1))) uint32_t *) (uintptr_t) p;
#else
  return *(const uint32_t *) p;
#endif
}

// A pad that intersects one block's strip (destination row y, columns [bx0, bx1)): everything the lanes need,
// resolved once per block (k_aggregate builds the list with one lane per pad and a ballot).
struct PadHit {
  const uint8_t *row;   // first byte of the pad's source row under destination row y
  int xpos, width;
  int s_alpha, mode;
};

// branch-free on purpose: on the device one lane per pad runs this and the whole descriptor should arrive in
// one round trip instead of a load-compare-load chain
GSTAMD_CD bool pad_hit_test (const AggregateParams &p, int k, int bx0, int bx1, int y, PadHit *h)
{
  const PadDev pad = p.pads[k];
  const int sy = y - pad.ypos;
  h->row = pad.data + (ptrdiff_t) sy * pad.stride;
  h->xpos = pad.xpos;
  h->width = pad.width;
  h->s_alpha = pad.s_alpha;
  h->mode = pad.mode;
  return (sy >= 0) & (sy < pad.height) & (pad.xpos < bx1) & (pad.xpos + pad.width > bx0);
}

// a value that is the same in every lane of the wave (hit-list entries): on the device move it to a scalar register
// so that the arithmetic on it runs on the scalar unit and does not take VALU issue slots
GSTAMD_CD int uniform_i32 (int v)
{
#ifdef __HIPCC__
  return __builtin_amdgcn_readfirstlane (v);
#else
  return v;
#endif
}

GSTAMD_CD const uint8_t *uniform_ptr (const uint8_t *p)
{
#ifdef __HIPCC__
  const uint64_t v = (uint64_t) (uintptr_t) p;
  const uint32_t lo = (uint32_t) __builtin_amdgcn_readfirstlane ((int) (uint32_t) v);
  const uint32_t hi = (uint32_t) __builtin_amdgcn_readfirstlane ((int) (uint32_t) (v >> 32));
  return (const uint8_t *) (uintptr_t) (((uint64_t) hi << 32) | lo);
#else
  return p;
#endif
}

GSTAMD_CD int span4_clamp (int sx, int w)
{
  return sx < 0 ? 0 : (sx > w - 4 ? w - 4 : sx);
}

#define AGG_DEPTH 4

GSTAMD_CD void issue_order_fence ()
{
#ifdef __HIPCC__
  __builtin_amdgcn_sched_barrier (0);
#endif
}

// the (clamped) 16 bytes of hit min (k, n_hits - 1) under destination pixels x .. x+3
template <int ABL>
GSTAMD_CD Px4Words span4_fetch (const PadHit *hits, int k, int n_hits, int x)
{
  const int kc = k < n_hits ? k : n_hits - 1;
  if (ABL == 2 || ABL == 4) {
    Px4Words r;
    r.v[0] = r.v[1] = r.v[2] = r.v[3] = (uint32_t) (x + kc) * 0x01010101u;
    return r;
  }
  const uint8_t *row = uniform_ptr (hits[kc].row);
  const int xpos = uniform_i32 (hits[kc].xpos), w = uniform_i32 (hits[kc].width);
  return load_px4 (row + 4 * (size_t) (unsigned) span4_clamp (x - xpos, w));
}

// pixels x .. x+3 of destination row y from the block's hit list.  p.fast = every pad of the chunk takes the
// opaque blend (no transparent background, no SOURCE operator), decided on the host.
template <int ABL, int ASH>
GSTAMD_CD void aggregate_span4 (const AggregateParams &p, const PadHit *hits, int n_hits, uint32_t *d, int x, int y)
{
  const int ashift = ASH;       // compile-time alpha position on this path (0 or 24)
#pragma unroll
  for (int i = 0; i < 4; i++)
    d[i] = p.bg_kind == 0 ? checker_px (x + i, y, p.ashift, p.checker_yuv) : (p.bg_kind == 1 ? p.bg_word : d[i]);
  if (ABL == 3)
    n_hits = 0;
  if (!p.fast || n_hits == 0) {
    for (int k = 0; k < n_hits; k++) {
      const PadHit h = hits[k];
      const int sx = x - h.xpos;
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (sx + i >= 0 && sx + i < h.width)
          d[i] = apply_pad (d[i], load_px1 (h.row + 4 * (size_t) (sx + i)), h.s_alpha, h.mode, p.ashift, p.overlay);
    }
    return;
  }
  Px2 acc[4];
#pragma unroll
  for (int i = 0; i < 4; i++)
    acc[i] = px2_unpack (d[i]);
  // One 16-byte load per hit, always issued (clamped into the pad row; the host guarantees width >= 4 on this path)
  // and requested AGG_DEPTH hits ahead of the blend chain that consumes it: a wave keeps AGG_DEPTH KB in flight,
  // which is what it takes to cover HBM latency with 8 waves per SIMD (1-deep left the VALU ~40 % idle).
  Px4Words buf[AGG_DEPTH];
#pragma unroll
  for (int j = 0; j < AGG_DEPTH; j++) {
    buf[j] = span4_fetch<ABL> (hits, j, n_hits, x);
    issue_order_fence ();       // oldest request first: the loop's s_waitcnt vmcnt(AGG_DEPTH) relies on it
  }
  for (int k0 = 0; k0 < n_hits; k0 += AGG_DEPTH) {
#pragma unroll
    for (int j = 0; j < AGG_DEPTH; j++) {
      const int k = k0 + j;
      const Px4Words cur = buf[j];
      buf[j] = span4_fetch<ABL> (hits, k + AGG_DEPTH, n_hits, x);
      if (k < n_hits) {
        const int sx = x - uniform_i32 (hits[k].xpos), w = uniform_i32 (hits[k].width);
        const uint32_t alpha8081 = (uint32_t) uniform_i32 (hits[k].s_alpha) * 0x8081u;
        if ((unsigned) sx <= (unsigned) (w - 4)) {
#pragma unroll
          for (int i = 0; i < 4; i++)
            if (ABL == 1 || ABL == 4)
              acc[i].e ^= cur.v[i];
            else
              px2_blend_lazy<ASH> (acc[i], cur.v[i], alpha8081);
        } else if ((unsigned) (sx + 3) < (unsigned) (w + 3)) {           // lane straddles a pad edge
          const uint8_t *row = uniform_ptr (hits[k].row);
#pragma unroll
          for (int i = 0; i < 4; i++)
            if (sx + i >= 0 && sx + i < w)
              px2_blend_lazy<ASH> (acc[i], load_px1 (row + 4 * (size_t) (sx + i)), alpha8081);
        }
      }
    }
  }
  /* the opaque blend leaves alpha 0xff (compositor_orc_blend_*: the destination's alpha byte is overwritten); the background's
   * alpha is 0xff on this path as well (checker / opaque colour / an earlier chunk's output), so forcing it once is the same */
#pragma unroll
  for (int i = 0; i < 4; i++)
    d[i] = px2_pack (acc[i]) | (0xffu << ashift);
}

// ------------------------------------------------------------------------------------------------
// Several rows per wave (k_aggregate_rows).  A wave owns a strip of 256 destination columns over `rows` consecutive rows and
// walks ONE flat list of (row, pad) entries: the pad descriptors are fetched once per wave instead of once per row, and the
// load pipeline (AGG_DEPTH requests in flight per lane) runs across row boundaries instead of draining at the end of every row -
// with one row per wave the wave's life was three dependent HBM round trips (descriptors, the first pad words, the last ones)
// for 16 bytes of output per lane.  An entry flagged AGG_ROW_END finishes its row: the lanes store it and start the next one
// from the background.  Rows no pad touches carry one AGG_ROW_SKIP entry.
#define AGG_LIST_MAX 128
#define AGG_ROW_END 0x100
#define AGG_ROW_SKIP 0x200

struct RowHit {
  const uint8_t *row;   // first byte of the pad's source row under this destination row
  int xpos, width;
  int ctl;              // pad alpha (bits 0-7) | AGG_ROW_END | AGG_ROW_SKIP
};

GSTAMD_CD int agg_rows_per_pass (int n_xhits, int rows)
{
  const int per = AGG_LIST_MAX / (n_xhits > 0 ? n_xhits : 1);
  return per < rows ? per : rows;
}

GSTAMD_CD bool pad_xhit (const PadDev &pad, int wx0, int wx1) { return (pad.xpos < wx1) & (pad.xpos + pad.width > wx0); }

GSTAMD_CD void agg_store4 (uint8_t *p, const uint32_t d[4], int nv)
{
#ifdef __HIPCC__
  typedef unsigned int u32x4_a4s __attribute__ ((ext_vector_type (4), aligned (4)));
  if (nv == 4) {
    const u32x4_a4s v = {d[0], d[1], d[2], d[3]};
    __builtin_nontemporal_store (v, (u32x4_a4s *) p);
    return;
  }
#endif
  for (int i = 0; i < nv; i++)
    ((uint32_t *) p)[i] = d[i];
}

GSTAMD_CD void agg_background4 (const AggregateParams &p, Px2 acc[4], int x, int y)
{
#pragma unroll
  for (int i = 0; i < 4; i++)
    acc[i] = px2_unpack (p.bg_kind == 0 ? checker_px (x + i, y, p.ashift, p.checker_yuv) : p.bg_word);
}

GSTAMD_CD Px4Words rows_fetch (const RowHit *list, int k, int n, int x)
{
  const int kc = k < n ? k : n - 1;
  const uint8_t *row = uniform_ptr (list[kc].row);
  const int xpos = uniform_i32 (list[kc].xpos), w = uniform_i32 (list[kc].width);
  return load_px4 (row + 4 * (size_t) (unsigned) span4_clamp (x - xpos, w));
}

// pixels x .. x+3 (the first nv of them inside the rectangle) of the rows starting at y, from the wave's entry list.  Opaque
// blends only (p.fast), background checker or colour.
template <int ASH, int DEPTH>
GSTAMD_CD void aggregate_rows4 (const AggregateParams &p, const RowHit *list, int n, uint8_t *dst, int dstride, int x, int y, int nv)
{
  Px2 acc[4];
  Px4Words buf[DEPTH];
  agg_background4 (p, acc, x, y);
#pragma unroll
  for (int j = 0; j < DEPTH; j++) {
    buf[j] = rows_fetch (list, j, n, x);
    issue_order_fence ();
  }
  for (int k0 = 0; k0 < n; k0 += DEPTH) {
#pragma unroll
    for (int j = 0; j < DEPTH; j++) {
      const int k = k0 + j;
      const Px4Words cur = buf[j];
      buf[j] = rows_fetch (list, k + DEPTH, n, x);
      if (k < n) {
        const int ctl = uniform_i32 (list[k].ctl);
        if (!(ctl & AGG_ROW_SKIP)) {
          const int sx = x - uniform_i32 (list[k].xpos), w = uniform_i32 (list[k].width);
          const uint32_t alpha8081 = (uint32_t) (ctl & 0xff) * 0x8081u;
          if ((unsigned) sx <= (unsigned) (w - 4)) {
#pragma unroll
            for (int i = 0; i < 4; i++)
              px2_blend_lazy<ASH> (acc[i], cur.v[i], alpha8081);
          } else if ((unsigned) (sx + 3) < (unsigned) (w + 3)) {         // lane straddles a pad edge
            const uint8_t *row = uniform_ptr (list[k].row);
#pragma unroll
            for (int i = 0; i < 4; i++)
              if (sx + i >= 0 && sx + i < w)
                px2_blend_lazy<ASH> (acc[i], load_px1 (row + 4 * (size_t) (sx + i)), alpha8081);
          }
        }
        if (ctl & AGG_ROW_END) {
          uint32_t d[4];
#pragma unroll
          for (int i = 0; i < 4; i++)
            d[i] = px2_pack (acc[i]) | (0xffu << ASH);
          agg_store4 (dst + (ptrdiff_t) y * dstride + 4 * (ptrdiff_t) x, d, nv);
          y++;
          agg_background4 (p, acc, x, y);
        }
      }
    }
  }
}

// the entry list of rows y .. y + ny - 1 for the strip [wx0, wx1): host form (the kernel builds the same list with one lane per
// pad and a ballot per row)
GSTAMD_CD int agg_build_list_host (const AggregateParams &p, int wx0, int wx1, int y, int ny, RowHit *list)
{
  int n = 0;
  for (int r = 0; r < ny; r++) {
    const int first = n;
    for (int k = 0; k < p.n_pads; k++) {
      const PadDev &pad = p.pads[k];
      const int sy = y + r - pad.ypos;
      if (pad_xhit (pad, wx0, wx1) && sy >= 0 && sy < pad.height) {
        list[n].row = pad.data + (ptrdiff_t) sy * pad.stride;
        list[n].xpos = pad.xpos;
        list[n].width = pad.width;
        list[n].ctl = pad.s_alpha;
        n++;
      }
    }
    if (n == first) {
      list[n].row = p.pads[0].data;
      list[n].xpos = 0;
      list[n].width = 4;
      list[n].ctl = AGG_ROW_SKIP;
      n++;
    }
    list[n - 1].ctl |= AGG_ROW_END;
  }
  return n;
}

// ------------------------------------------------------------------------------------------------
// k_aggregate_strip: a strip of 256 columns x a few rows per wave with everything that is the same for the whole wave - which pad
// comes next, its row pointer, position, width and alpha - kept on the scalar side.  k_aggregate spent 559 vector instructions per
// wave-row on C4 (18.1 M per frame, 81 % of the VALU issue slots: the kernel was bound by them, not by HBM), of which the blend
// arithmetic is 4 x 16 per pad; the rest was the per-row prologue (descriptor fetch, hit test, list in LDS) and, per pad, reading
// the list entry back from LDS and broadcasting it (v_readfirstlane is a vector instruction).  Here the pads that touch the strip's
// columns are ONE bit mask (a ballot, once per wave), the pads of a row are that mask and a ballot of the row test (3 vector
// instructions per row), and walking the set bits, loading the pad descriptor (s_load from the kernel arguments) and computing
// the row pointer are scalar instructions, which issue beside the vector ones.
#define AGGS_SKIP 1     // nothing to blend (a row no pad touches)
#define AGGS_END 2      // last entry of its row: store the row, start the next from the background
#define AGGS_NONE 4     // past the last row of the strip

struct AggsEntry {
  const uint8_t *row;   // the pad's source row under this destination row
  int xpos, width;
  uint32_t alpha8081;
  int ctl;
};

// what a lane knows about "its" pad (lane k < n_pads holds pad k): enough for the row test
struct AggsLanePad {
  int xhit;             // the pad touches the strip's columns
  int ypos, height;
};

struct AggsCursor {
  uint32_t xmask;       // pads that touch the strip's columns
  uint32_t left;        // pads of row y not yet handed out
  int y, y_end;
  int fresh;            // row y has not handed out anything yet
};

// pads (of xmask) that cover destination row y.  Device: one lane per pad and a ballot; host: the loop it stands for.
GSTAMD_CD uint32_t aggs_rowmask (const AggregateParams &p, const AggsLanePad &lp, uint32_t xmask, int y)
{
#ifdef __HIPCC__
  (void) p;
  (void) xmask;
  return (uint32_t) __ballot (lp.xhit && (unsigned) (y - lp.ypos) < (unsigned) lp.height);
#else
  (void) lp;
  uint32_t m = 0;
  for (int k = 0; k < p.n_pads; k++)
    if (((xmask >> k) & 1u) && (unsigned) (y - p.pads[k].ypos) < (unsigned) p.pads[k].height)
      m |= 1u << k;
  return m;
#endif
}

GSTAMD_CD int aggs_first_bit (uint32_t m)
{
#ifdef __HIPCC__
  return __builtin_ctz (m);
#else
  int k = 0;
  while (!((m >> k) & 1u))
    k++;
  return k;
#endif
}

// next (row, pad) of the strip: pad index (or -1), flags and row of the entry
GSTAMD_CD void aggs_advance (const AggregateParams &p, const AggsLanePad &lp, AggsCursor &c, int *k, int *ctl, int *y)
{
  if (c.y >= c.y_end) {
    *k = -1, *ctl = AGGS_NONE, *y = c.y_end - 1;
    return;
  }
  *y = c.y;
  if (c.left == 0) {            /* only a fresh row gets here: no pad touches it */
    *k = -1, *ctl = AGGS_SKIP | AGGS_END;
  } else {
    *k = aggs_first_bit (c.left);
    c.left &= c.left - 1;
    *ctl = c.left == 0 ? AGGS_END : 0;
  }
  if (*ctl & AGGS_END) {
    c.y++;
    c.left = c.y < c.y_end ? aggs_rowmask (p, lp, c.xmask, c.y) : 0;
  }
}

// the entry of pad k on row y (k < 0: a dummy that reads pad 0's first row, which exists on this path)
template <int NPX>
GSTAMD_CD AggsEntry aggs_entry (const PadDev &pad, int k, int ctl, int y)
{
  AggsEntry e;
  e.ctl = ctl;
  if (k < 0) {
    e.row = pad.data;
    e.xpos = 0;
    e.width = NPX;              /* the fetch reads NPX pixels from the row's start: every pad is at least that wide on this path */
    e.alpha8081 = 0;
  } else {
    e.row = pad.data + (ptrdiff_t) (y - pad.ypos) * pad.stride;
    e.xpos = pad.xpos;
    e.width = pad.width;
    e.alpha8081 = (uint32_t) pad.s_alpha * 0x8081u;
  }
  return e;
}

// NPX / 4 16-byte pieces of the entry's row under destination pixels x .. x + NPX - 1 (clamped into the row: width >= NPX here)
template <int NPX>
GSTAMD_CD void aggs_fetch (const AggsEntry &e, int x, Px4Words *out)
{
  int sx = x - e.xpos;
  const int hi = e.width - NPX;
  sx = sx < 0 ? 0 : (sx > hi ? hi : sx);
#pragma unroll
  for (int q = 0; q < NPX / 4; q++)
    out[q] = load_px4 (e.row + 4u * (uint32_t) sx + 16u * q);
}

// pixels x .. x + NPX - 1 (the first nv inside the rectangle) of rows y0 .. y1-1.  Opaque blends only (p.fast), checker or colour
// background.  NPX = 4 or 8 pixels per lane (a wave covers 256 or 512 columns).
template <int ASH, int DEPTH, int NPX>
GSTAMD_CD void aggregate_strip (const AggregateParams &p, const AggsLanePad &lp, uint32_t xmask, uint8_t *dst, int dstride, int x, int y0, int y1, int nv)
{
  AggsCursor c;
  c.xmask = xmask;
  c.y = y0;
  c.y_end = y1;
  c.left = aggs_rowmask (p, lp, xmask, y0);
  // one entry ahead of the ring: its pad descriptor is on its way from the kernel arguments while the entry before it is set up
  int pk, pctl, py;
  aggs_advance (p, lp, c, &pk, &pctl, &py);
  PadDev pend = p.pads[pk < 0 ? 0 : pk];
  AggsEntry ring[DEPTH];
  Px4Words buf[DEPTH][NPX / 4];
#pragma unroll
  for (int j = 0; j < DEPTH; j++) {
    ring[j] = aggs_entry<NPX> (pend, pk, pctl, py);
    aggs_advance (p, lp, c, &pk, &pctl, &py);
    pend = p.pads[pk < 0 ? 0 : pk];
    aggs_fetch<NPX> (ring[j], x, buf[j]);
    issue_order_fence ();
  }
  Px2 acc[NPX];
  int y = y0;
#pragma unroll
  for (int q = 0; q < NPX / 4; q++)
    agg_background4 (p, acc + 4 * q, x + 4 * q, y);
  for (;;) {
#pragma unroll
    for (int j = 0; j < DEPTH; j++) {
      const AggsEntry e = ring[j];
      Px4Words cur[NPX / 4];
#pragma unroll
      for (int q = 0; q < NPX / 4; q++)
        cur[q] = buf[j][q];
      ring[j] = aggs_entry<NPX> (pend, pk, pctl, py);
      aggs_advance (p, lp, c, &pk, &pctl, &py);
      pend = p.pads[pk < 0 ? 0 : pk];
      aggs_fetch<NPX> (ring[j], x, buf[j]);
      if (e.ctl & AGGS_NONE)
        return;
      if (!(e.ctl & AGGS_SKIP)) {
        const int sx = x - e.xpos, w = e.width;
        if ((unsigned) sx <= (unsigned) (w - NPX)) {
#pragma unroll
          for (int i = 0; i < NPX; i++)
            px2_blend_lazy<ASH> (acc[i], cur[i >> 2].v[i & 3], e.alpha8081);
        } else if ((unsigned) (sx + NPX - 1) < (unsigned) (w + NPX - 1)) {       // lane straddles a pad edge
#pragma unroll
          for (int i = 0; i < NPX; i++)
            if (sx + i >= 0 && sx + i < w)
              px2_blend_lazy<ASH> (acc[i], load_px1 (e.row + 4 * (ptrdiff_t) (sx + i)), e.alpha8081);
        }
      }
      if (e.ctl & AGGS_END) {
#pragma unroll
        for (int q = 0; q < NPX / 4; q++) {
          uint32_t d[4];
#pragma unroll
          for (int i = 0; i < 4; i++)
            d[i] = px2_pack (acc[4 * q + i]) | (0xffu << ASH);
          const int nq = nv - 4 * q;
          agg_store4 (dst + (ptrdiff_t) y * dstride + 4 * (ptrdiff_t) (x + 4 * q), d, nq < 0 ? 0 : (nq > 4 ? 4 : nq));
        }
        y++;
#pragma unroll
        for (int q = 0; q < NPX / 4; q++)
          agg_background4 (p, acc + 4 * q, x + 4 * q, y);
      }
    }
  }
}

template <int ASH, int DEPTH>
GSTAMD_CD void aggregate_strip4 (const AggregateParams &p, const AggsLanePad &lp, uint32_t xmask, uint8_t *dst, int dstride, int x, int y0, int y1, int nv)
{
  aggregate_strip<ASH, DEPTH, 4> (p, lp, xmask, dst, dstride, x, y0, y1, nv);
}

// ------------------------------------------------------------------------------------------------
// k_aggregate_direct (round 3): the opaque-blend path with NO hit list in LDS and every request of a wave in flight before its
// first blend.  Lane k tests pad k (one vector pass over all pads, as k_aggregate does) and keeps the pad's row pointer, position,
// width and alpha factor; the ballot of the tests is a scalar bit mask.  The wave then walks the set bits on the scalar unit
// (s_ff1 + clear), fetches the hit's values with v_readlane (lane index in an SGPR) and requests its 16 bytes into register set j -
// j is the position among the HITS, so AGG_DIRECT_SLOTS sets serve any number of pads (more hits than sets: another round) - and
// only then waits and blends, again walking the mask.  Against k_aggregate: no LDS round trip and no wave barrier between the pad
// test and the first request, no 64-bit vector address arithmetic (row pointer in SGPRs + 32-bit lane offset), no clamped duplicate
// requests past the last hit, no ring moves; up to 12 KB requested per wave ahead of the arithmetic instead of 4.
// scripts/c4_probe.hip measured this request structure (its "skeleton", depth 0) at 30.1-30.4 us per C4 frame against 34.1 us with
// one request in flight per lane; k_aggregate with its arithmetic removed took 33.3 us, and its SQ counters showed 18.1 M vector
// instructions per frame of which 8.3 M are blend arithmetic.
#define AGG_DIRECT_SLOTS 12

// The requests are inline assembly on the device: written as plain loads the compiler sinks every conditional load to its use (one
// request in flight: the 34 us of the probe's depth-1 form) or, with the loads kept in place, waits for the previous one before
// each request.  So: `agg_request` issues global_load_dwordx4 with the uniform row pointer in SGPRs and a 32-bit lane offset - the
// compiler does not know a load is outstanding - and `agg_arrived` is the one s_waitcnt vmcnt(0) ahead of the blends, tied to all
// register sets so that nothing that reads them can be scheduled above it.  The sets stay 128-bit tuples until after the wait (a
// tuple split would be a register copy of data that has not arrived).  In-order return makes the compiler's own counted waits for
// its own (younger) loads conservative, never too short; a compiler-visible memory operation that is only POSSIBLY outstanding when
// the requests start makes it put counted waits between them, hence no such operation in k_aggregate_direct ahead of this code.
#ifdef __HIPCC__
typedef unsigned int AggVec __attribute__ ((ext_vector_type (4)));
template <int NT>
GSTAMD_CD void agg_request (AggVec &b, const uint8_t *row, uint32_t byte_off)
{
  /* s_nop 4: the row pointer usually arrives in its SGPRs by v_readlane immediately before; a VMEM instruction reading an SGPR that a
   * VALU instruction wrote needs 5 wait states on gfx9 hardware, and the compiler's hazard recognizer does not look into inline
   * assembly (without it the load used the stale high half: memory access fault on the first MI355X run) */
  if (NT)       /* streaming: every source pixel is read once */
    asm volatile ("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 nt" : "=v" (b) : "v" (byte_off), "s" (row));
  else
    asm volatile ("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v" (b) : "v" (byte_off), "s" (row));
}
GSTAMD_CD void agg_arrived (AggVec *b)
{
  asm volatile ("s_waitcnt vmcnt(0)" : "+v" (b[0]), "+v" (b[1]), "+v" (b[2]), "+v" (b[3]), "+v" (b[4]), "+v" (b[5]), "+v" (b[6]), "+v" (b[7]),
      "+v" (b[8]), "+v" (b[9]), "+v" (b[10]), "+v" (b[11]));
}
#define AGGV(b, i) ((b)[i])
#else
struct AggVec {
  uint32_t v[4];
};
template <int NT>
GSTAMD_CD void agg_request (AggVec &b, const uint8_t *row, uint32_t byte_off)
{
  const Px4Words w = load_px4 (row + (size_t) byte_off);
  for (int i = 0; i < 4; i++)
    b.v[i] = w.v[i];
}
GSTAMD_CD void agg_arrived (AggVec *) {}
#define AGGV(b, i) ((b).v[i])
#endif

// what the wave knows about the pads under its strip and row: bit k of `mask` = pad k is hit; on the device lane k holds pad k's
// values and at () is four v_readlane, on the host they are computed from the pad table
struct DirectPads {
  unsigned long long mask;
#ifdef __HIPCC__
  uint32_t row_lo, row_hi, alpha8081;
  int xpos, width;
  int stride;           // aggregate_direct4_rows only: the pad's pitch, for the rows after the first
#else
  const AggregateParams *p;
  int y;
#endif
};

struct DirectHit {
  const uint8_t *row;
  int xpos, width;
  uint32_t alpha8081;
};

GSTAMD_CD DirectHit direct_hit (const DirectPads &dp, int k)
{
  DirectHit h;
#ifdef __HIPCC__
  const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane ((int) dp.row_lo, k), hi = (uint32_t) __builtin_amdgcn_readlane ((int) dp.row_hi, k);
  h.row = (const uint8_t *) (uintptr_t) (((uint64_t) hi << 32) | lo);
  h.xpos = __builtin_amdgcn_readlane (dp.xpos, k);
  h.width = __builtin_amdgcn_readlane (dp.width, k);
  h.alpha8081 = (uint32_t) __builtin_amdgcn_readlane ((int) dp.alpha8081, k);
#else
  const PadDev &pad = dp.p->pads[k];
  h.row = pad.data + (ptrdiff_t) (dp.y - pad.ypos) * pad.stride;
  h.xpos = pad.xpos;
  h.width = pad.width;
  h.alpha8081 = (uint32_t) pad.s_alpha * 0x8081u;
#endif
  return h;
}

// pad k's row under canvas row (first row of the wave) + r
GSTAMD_CD DirectHit direct_hit_row (const DirectPads &dp, int k, int r)
{
  DirectHit h = direct_hit (dp, k);
#ifdef __HIPCC__
  h.row += (ptrdiff_t) r * __builtin_amdgcn_readlane (dp.stride, k);
#else
  h.row += (ptrdiff_t) r * dp.p->pads[k].stride;
#endif
  return h;
}

GSTAMD_CD int mask_first (unsigned long long m) { return __builtin_ctzll (m); }

GSTAMD_CD bool pad_hits_strip (const PadDev &pad, int wx0, int wx1, int y)
{
  const int sy = y - pad.ypos;
  return (sy >= 0) & (sy < pad.height) & (pad.xpos < wx1) & (pad.xpos + pad.width > wx0);
}

// ---- opaque culling (gstamd_compositor_aggregate_opaque) --------------------------------------------------------------------------
// The reference blends every pad over what is under it (blend_pads compositor.c:1678-1697) even where the result is the pad's own pixel:
// OVER at pad alpha 1.0 on a pixel of alpha 255 is (s * 255 + d * 0) / 255 = s with the destination alpha forced to 0xff (BLEND_A32 blend.c:96-132,
// compositor_orc_blend_argb / _bgra).  Where such pixels cover the whole 256-pixel strip of a row, nothing under that pad is read.
struct OpacityMaps {
  const unsigned long long *map[GSTAMD_MAX_FUSED_PADS];   // pad k: one word per pad row, bit b = the pixels [64 b, 64 b + 64) of the row all have alpha 255
                                                          // (gstamd_compositor_pad_opacity_map); NULL: not known
  uint32_t all;                                           // bit k: every pixel of pad k has alpha 255 (a pad converted from a format without alpha)
};

// bit b of a pad row's word (host form; k_opacity_map computes it with a ballot)
GSTAMD_CD unsigned long long opacity_row_bits (const uint8_t *row, int w, int ashift)
{
  unsigned long long bits = 0;
  for (int b = 0; 64 * b < w && b < 64; b++) {
    bool ok = true;
    for (int x = 64 * b; x < 64 * b + 64 && x < w; x++)
      ok &= ((load_px1 (row + 4 * (size_t) x) >> ashift) & 0xffu) == 0xffu;
    if (ok)
      bits |= 1ull << b;
  }
  return bits;
}

// does this pad hide everything under the strip [wx0, wx1) of canvas row y?  (callers: the pad hits the strip and the launch is the fast form -
// no SOURCE operator, where OVER and ADD are the same arithmetic)
GSTAMD_CD bool pad_covers_strip (const PadDev &pad, const unsigned long long *map, bool all, int wx0, int wx1, int y)
{
  if (pad.s_alpha != 255 || pad.xpos > wx0 || pad.xpos + pad.width < wx1)
    return false;
  const int sy = y - pad.ypos;
  if (sy < 0 || sy >= pad.height)
    return false;
  if (all)
    return true;
  if (!map)
    return false;
  const int b0 = (wx0 - pad.xpos) >> 6, b1 = (wx1 - 1 - pad.xpos) >> 6;
  if (b1 > 63)
    return false;
  const unsigned long long need = (b1 - b0 == 63 ? ~0ull : ((1ull << (b1 - b0 + 1)) - 1ull)) << b0;
  return (map[sy] & need) == need;
}

// the hit mask without the pads under the topmost covering one
GSTAMD_CD unsigned long long cull_mask (unsigned long long hits, unsigned long long covers)
{
  return covers ? hits & ~((1ull << (63 - __builtin_clzll (covers))) - 1ull) : hits;
}

#ifndef __HIPCC__
GSTAMD_CD unsigned long long direct_cover_mask_host (const AggregateParams &p, const OpacityMaps &om, int wx0, int wx1, int y)
{
  unsigned long long c = 0;
  for (int k = 0; k < p.n_pads; k++)
    if (pad_hits_strip (p.pads[k], wx0, wx1, y) && pad_covers_strip (p.pads[k], om.map[k], (om.all >> k) & 1u, wx0, wx1, y))
      c |= 1ull << k;
  return c;
}

GSTAMD_CD DirectPads direct_pads_host (const AggregateParams &p, int wx0, int wx1, int y)
{
  DirectPads dp;
  dp.mask = 0;
  dp.p = &p;
  dp.y = y;
  for (int k = 0; k < p.n_pads; k++)
    if (pad_hits_strip (p.pads[k], wx0, wx1, y))
      dp.mask |= 1ull << k;
  return dp;
}
#endif

// the four background pixels of a lane (checker: tab[((y & 8) >> 3) + ((x & 8) >> 3)] = 160 where exactly one of the bits is set)
GSTAMD_CD void background4 (const AggregateParams &p, uint32_t *d, int x, int y)
{
  if (p.bg_kind == 1) {
    d[0] = d[1] = d[2] = d[3] = p.bg_word;
  } else if (p.bg_kind == 0) {
    if ((x & 3) == 0) {          /* the four pixels share one 8-pixel square */
      d[0] = d[1] = d[2] = d[3] = checker_px (x, y, p.ashift, p.checker_yuv);
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++)
        d[i] = checker_px (x + i, y, p.ashift, p.checker_yuv);
    }
  }
}

template <int ASH, int NT, int KEEP>
GSTAMD_CD void aggregate_direct4 (const AggregateParams &p, const DirectPads &dp, uint32_t *d, int x, int y)
{
  background4 (p, d, x, y);
  /* (unpacked before the requests: on a continuation chunk d is the canvas, a load the compiler waits for - ahead of the requests
   * it is a wait for that load alone, after them it would drain them) */
  Px2 acc[4];
#pragma unroll
  for (int i = 0; i < 4; i++)
    acc[i] = px2_unpack (d[i]);
  unsigned long long m = dp.mask;
  uint32_t touched = 0;
  while (m) {
    AggVec buf[AGG_DIRECT_SLOTS];
    // requests: one 16-byte load per hit (clamped into the pad row like span4_fetch), all of them ahead of the first use
    unsigned long long mr = m;
#pragma unroll
    for (int j = 0; j < AGG_DIRECT_SLOTS; j++)
      if (mr) {
        const DirectHit h = direct_hit (dp, mask_first (mr));
        mr &= mr - 1;
        agg_request<NT> (buf[j], h.row, 4u * (uint32_t) span4_clamp (x - h.xpos, h.width));
      }
    agg_arrived (buf);
#pragma unroll
    for (int j = 0; j < AGG_DIRECT_SLOTS; j++)
      if (m) {
        const DirectHit h = direct_hit (dp, mask_first (m));
        m &= m - 1;
        const int sx = x - h.xpos, w = h.width;
        if ((unsigned) sx <= (unsigned) (w - 4)) {
          if (KEEP)
            touched = 0xf;
#pragma unroll
          for (int i = 0; i < 4; i++)
            px2_blend_lazy<ASH> (acc[i], AGGV (buf[j], i), h.alpha8081);
        } else if ((unsigned) (sx + 3) < (unsigned) (w + 3)) {           // lane straddles a pad edge
#pragma unroll
          for (int i = 0; i < 4; i++)
            if (sx + i >= 0 && sx + i < w) {
              if (KEEP)
                touched |= 1u << i;
              px2_blend_lazy<ASH> (acc[i], load_px1 (h.row + 4 * (size_t) (sx + i)), h.alpha8081);
            }
        }
      }
  }
  /* alpha forced once after the last pad, as in aggregate_span4: the opaque blend leaves 0xff, and so do the backgrounds of this path.
   * A continuation chunk (KEEP = bg_kind 2) starts from the canvas, whose alpha an earlier chunk's SOURCE pads may have lowered: there only the pixels a
   * pad of this chunk touched are forced */
#pragma unroll
  for (int i = 0; i < 4; i++)
    if (!KEEP || ((touched >> i) & 1))
      d[i] = px2_pack (acc[i]) | (0xffu << ASH);
}

// R canvas rows per wave (k_aggregate_direct_cull): the culled walk is short - one or two pads a strip where opaque pads pile up - and a wave's life is
// the three dependent trips to memory (descriptors, opacity words, pad pixels), so a wave takes R rows through them together: the request slots are
// split R ways (a row with more hits than its share takes another round), masks[r] = hit mask of row y0 + r (0: a row past the rectangle).
template <int ASH, int NT, int KEEP, int R>
GSTAMD_CD void aggregate_direct4_rows (const AggregateParams &p, const DirectPads &dp, const unsigned long long *masks, uint32_t (*d)[4], int x, int y0)
{
  constexpr int S = AGG_DIRECT_SLOTS / R;
  Px2 acc[R][4];
  unsigned long long m[R];
  uint32_t touched[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    background4 (p, d[r], x, y0 + r);
#pragma unroll
    for (int i = 0; i < 4; i++)
      acc[r][i] = px2_unpack (d[r][i]);
    m[r] = masks[r];
    touched[r] = 0;
  }
  for (;;) {
    unsigned long long any = 0;
#pragma unroll
    for (int r = 0; r < R; r++)
      any |= m[r];
    if (!any)
      break;
    AggVec buf[AGG_DIRECT_SLOTS];
#pragma unroll
    for (int r = 0; r < R; r++) {
      unsigned long long mr = m[r];
#pragma unroll
      for (int j = 0; j < S; j++)
        if (mr) {
          const DirectHit h = direct_hit_row (dp, mask_first (mr), r);
          mr &= mr - 1;
          agg_request<NT> (buf[r * S + j], h.row, 4u * (uint32_t) span4_clamp (x - h.xpos, h.width));
        }
    }
    agg_arrived (buf);
#pragma unroll
    for (int r = 0; r < R; r++) {
#pragma unroll
      for (int j = 0; j < S; j++)
        if (m[r]) {
          const DirectHit h = direct_hit_row (dp, mask_first (m[r]), r);
          m[r] &= m[r] - 1;
          const int sx = x - h.xpos, w = h.width;
          if ((unsigned) sx <= (unsigned) (w - 4)) {
            if (KEEP)
              touched[r] = 0xf;
#pragma unroll
            for (int i = 0; i < 4; i++)
              px2_blend_lazy<ASH> (acc[r][i], AGGV (buf[r * S + j], i), h.alpha8081);
          } else if ((unsigned) (sx + 3) < (unsigned) (w + 3)) {           // lane straddles a pad edge
#pragma unroll
            for (int i = 0; i < 4; i++)
              if (sx + i >= 0 && sx + i < w) {
                if (KEEP)
                  touched[r] |= 1u << i;
                px2_blend_lazy<ASH> (acc[r][i], load_px1 (h.row + 4 * (size_t) (sx + i)), h.alpha8081);
              }
          }
        }
    }
  }
#pragma unroll
  for (int r = 0; r < R; r++)
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (!KEEP || ((touched[r] >> i) & 1))
        d[r][i] = px2_pack (acc[r][i]) | (0xffu << ASH);
}

}  // namespace gstamd
