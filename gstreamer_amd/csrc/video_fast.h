// video_fast.h - the speed-of-light variants of the unscaled NV12/NV21 -> 4-byte RGB conversion
// (BASELINE config 2).  Same integers as convert_body in video_device.h at every step, restructured so the
// kernels are bounded by HBM and not by VALU issue (a wave64 VALU instruction occupies its SIMD for 4 cycles:
// at 6 TB/s every instruction per pixel costs about 0.2 us per 4K frame):
//
//   * a lane converts 4-pixel spans of a chroma LINE PAIR (2p-1, 2p) - the pairing the reference's
//     do_upsample_lines produces (video-converter.c:2991-3021) - so each chroma row is loaded and
//     horizontally filtered once for the two luma lines that use it;
//   * chroma lives in packed 16-bit lanes {U, V} of one VGPR; the vertical 3:1 blends of both lines are four
//     v_pk_mad_u16 with the result left in the high byte of each lane (no shift, the "- 128" folded into the constant);
//   * mulhsw (splatbw (x - 128), p) of video_orc_convert_AYUV_ARGB is the high word of the exact product
//     splat * p: v_mul_i32_i24 on a sign-extended WORD operand (SDWA), the adds take WORD_1 of the products and
//     write the halves of two registers, v_sat_pk_u8_i16 saturates two channels at once and writes the pixel's
//     bytes in place: 13 VALU instructions per pixel for matrix + saturate + pack;
//   * edge rules (first / last chroma sample, first / last line) come out of the same formulas on clamped
//     indices: (a + a + 1) >> 1 == (3a + a + 2) >> 2 == a.
//
// The 16-bit wrap of ORC's addw cannot trigger for the matrices these kernels accept; the planner
// proves that (fast_matrix_ok) before selecting them, otherwise convert_body runs.
#pragma once
#include "video_device.h"

namespace gstamd {

struct FastParams {
  int width, height;
  int p8[5];            // p1..p5 << 8
  uint32_t pack_sel;    // v_perm selector: dest byte pos[A] <- 0xff, pos[R] <- lo.0, pos[G] <- lo.1, pos[B] <- hi.0
  int u_first;          // 1: NV12 (U,V), 0: NV21 (V,U)
  int pc[5];            // raw p1..p5 (word form, fast_emit4_l)
  int crow_lo, crow_hi; // chroma rows that exist, relative to the plane pointer (0 .. rows - 1; a source crop widens it)
  int pack_pos[4];      // destination byte of A, R, G, B
  int px_bytes;         // 4; 3: RGB / BGR destination - the pixel is formed as RGBx / BGRx and its three colour bytes are stored
  const uint8_t *lut;   // NULL, or a 256-byte table in device memory every colour byte of the finished pixel goes through (GammaPlan::lut_direct:
  int lut_keep;         // decode table . encode table); lut_keep: the destination byte that is alpha / filler and stays
  MatrixParams m8;      // ayuv == 2: the 8-bit convert stage behind the scaler (video_converter_matrix8 / _table between two YUV colorimetries: apply_matrix)
  int ayuv;             // 2: as 1, then m8.  1: no colour stage - the pixel is A (0xff) Y U V as it comes out of the scaler (the pack image of a planar destination, an AYUV
                        // destination): the layout argument GSTAMD_LAYOUT_AYUV of the bilinear 4:2:0 kernels
  int store_policy;     // how the finished pixels leave the CU (store16_policy): 0 streaming (nt: the line stays in the XCD's L2 until evicted or flushed
                        // at the end of the kernel), 1 write-through (sc0 sc1: nothing of the frame is dirty in L2 when the kernel ends)
};

// the table's copy in LDS while a kernel of the ABL == 2 ("table after the pack") variants runs: 64 words, filled by the kernel's one wave
#ifdef __HIPCC__
extern __shared__ uint32_t fast_lut_lds[];
#else
static uint32_t fast_lut_lds[64];
#endif
#define GSTAMD_FAST_LUT 2

GSTAMD_HD uint32_t fast_lut3_px (uint32_t px, int keep)
{
  const uint8_t *t = (const uint8_t *) fast_lut_lds;
  uint32_t r = 0;
#pragma unroll
  for (int b = 0; b < 4; b++) {
    const uint32_t v = (px >> (8 * b)) & 0xffu;
    r |= (b == keep ? v : (uint32_t) t[v]) << (8 * b);
  }
  return r;
}

inline void fast_params_finish (FastParams &fp, const int p[5], const int pack_pos[4], int u_first)
{
  fp.crow_lo = 0;
  fp.crow_hi = ((fp.height + 1) >> 1) - 1;
  for (int i = 0; i < 5; i++) {
    fp.p8[i] = p[i] * 256;
    fp.pc[i] = p[i];
  }
  fp.pack_sel = (0x0du << (8 * pack_pos[0])) | (0x00u << (8 * pack_pos[1])) | (0x01u << (8 * pack_pos[2])) | (0x04u << (8 * pack_pos[3]));
  fp.u_first = u_first;
  for (int i = 0; i < 4; i++)
    fp.pack_pos[i] = pack_pos[i];
  fp.px_bytes = 4;
  fp.ayuv = 0;
  fp.m8.kind = MATRIX_NONE;
  fp.store_policy = 0;
  fp.lut = nullptr;
  fp.lut_keep = 0;
}

// RGB / BGR destination: the kernel forms RGBx / BGRx pixels (alpha in byte 3) and stores their three colour bytes
inline void fast_params_rgb24 (FastParams &fp, const int p[5], const int pos[4], int u_first)
{
  const int pos4[4] = {3, pos[1], pos[2], pos[3]};
  const int lo = fp.crow_lo, hi = fp.crow_hi;
  fast_params_finish (fp, p, pos4, u_first);
  fp.crow_lo = lo;
  fp.crow_hi = hi;
  fp.px_bytes = 3;
}

#ifdef __HIPCC__
GSTAMD_HD uint32_t bperm (uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm (hi, lo, sel); }
#else
// v_perm_b32: selector byte 0-3 -> byte of lo, 4-7 -> byte of hi, 0x0c -> 0x00, >= 0x0d -> 0xff
inline uint32_t bperm (uint32_t hi, uint32_t lo, uint32_t sel)
{
  uint32_t r = 0;
  for (int i = 0; i < 4; i++) {
    const uint32_t s = (sel >> (8 * i)) & 0xff;
    uint32_t b;
    if (s < 4)
      b = (lo >> (8 * s)) & 0xff;
    else if (s < 8)
      b = (hi >> (8 * (s - 4))) & 0xff;
    else if (s == 0x0c)
      b = 0;
    else
      b = 0xff;
    r |= b << (8 * i);
  }
  return r;
}
#endif

// (a * b) >> 32 of the sign-extended low 24 bits of both operands: one v_mul_hi_i32_i24.  Written as
// asm on the device because the C form makes hipcc sign-extend the operands with extra v_bfe_i32.
GSTAMD_HD int mulhi24 (int a, int b)
{
#ifdef __HIPCC__
  int r;
  asm ("v_mul_hi_i32_i24 %0, %1, %2" : "=v" (r) : "v" (a), "s" (b));   /* b: wave-uniform coefficient */
  return r;
#else
  const int a24 = (int) ((uint32_t) a << 8) >> 8, b24 = (int) ((uint32_t) b << 8) >> 8;
  return (int) (((long long) a24 * (long long) b24) >> 32);
#endif
}

// per-16-bit-lane logical shift right (v_pk_lshrrev_b16)
template <int N>
GSTAMD_HD uint32_t pk_shr (uint32_t x)
{
#ifdef __HIPCC__
  typedef unsigned short us2 __attribute__ ((ext_vector_type (2)));
  us2 v = __builtin_bit_cast (us2, x);
  v = v >> (unsigned short) N;
  return __builtin_bit_cast (uint32_t, v);
#else
  return (x >> N) & (0x0000ffffu >> N) * 0x00010001u;
#endif
}

// write-once output: nontemporal (streaming) 16-byte store.  On MI355X a plain store costs the
// 12.4 MB-in / 33 MB-out byte mix 10.8 us per 4K frame, the nontemporal one 7.5 us (scripts/membench.hip).
GSTAMD_HD void store16_stream (uint8_t *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
#ifdef __HIPCC__
  typedef unsigned int u32x4 __attribute__ ((ext_vector_type (4)));
  u32x4 v = {a, b, c, d};
  __builtin_nontemporal_store (v, (u32x4 *) p);
#else
  *(uint4 *) p = gstamd_make_uint4 (a, b, c, d);
#endif
}

// The same with the store policy of the launch (FastParams::store_policy, wave-uniform).  A kernel's end writes back what its stores left dirty in the
// eight L2s - "+ B / 6 TB/s when the predecessor leaves B bytes dirty" (MI355X_MICROARCH.md, boundary row) - and for ONE 4K frame per launch that is
// most of the 33 MB the frame wrote: a launch per frame (what a live pipeline runs) pays it every frame, a 32-frame list once.  Write-through stores
// (sc0 sc1) leave nothing behind.
GSTAMD_HD void store16_policy (uint8_t *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d, int policy)
{
#ifdef __HIPCC__
  typedef unsigned int u32x4 __attribute__ ((ext_vector_type (4)));
  u32x4 v = {a, b, c, d};
  if (policy == 1)
    asm volatile ("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v" (p), "v" (v) : "memory");
  else if (policy == 2)
    asm volatile ("global_store_dwordx4 %0, %1, off sc1" : : "v" (p), "v" (v) : "memory");
  else if (policy == 3)
    asm volatile ("global_store_dwordx4 %0, %1, off sc0 sc1 nt" : : "v" (p), "v" (v) : "memory");
  else if (policy == 4)
    *(u32x4 *) p = v;
  else
    __builtin_nontemporal_store (v, (u32x4 *) p);
#else
  (void) policy;
  *(uint4 *) p = gstamd_make_uint4 (a, b, c, d);
#endif
}

// four 3-byte pixels (colour bytes 0..2 of a..d) as one streaming 12-byte store (pack_RGB / pack_BGR, video-format.c:1540, 1577)
GSTAMD_HD void store12_stream (uint8_t *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
  const uint32_t w0 = bperm (b, a, 0x04020100u), w1 = bperm (c, b, 0x05040201u), w2 = bperm (d, c, 0x06050402u);
#ifdef __HIPCC__
  typedef unsigned int u32x3 __attribute__ ((ext_vector_type (3)));
  u32x3 v = {w0, w1, w2};
  __builtin_nontemporal_store (v, (u32x3 *) p);
#else
  uint32_t *q = (uint32_t *) p;
  q[0] = w0;
  q[1] = w1;
  q[2] = w2;
#endif
}

GSTAMD_HD int med3_0_255 (int v)
{
#ifdef __HIPCC__
  int r;
  asm ("v_med3_i32 %0, %1, 0, %2" : "=v" (r) : "v" (v), "s" (255));
  return r;
#else
  return v < 0 ? 0 : (v > 255 ? 255 : v);
#endif
}

// one output pixel.  yx = 4 luma bytes ^ 0x80808080, cx = packed chroma {lane0, lane1} ^ 0x00800080: the
// "- 128" of the reference (subb 128, bytewise wrap) is that XOR, done once per word instead of per byte.
GSTAMD_HD uint32_t fast_pixel (const FastParams &fp, uint32_t yx, uint32_t ysel, uint32_t cx, uint32_t usel, uint32_t vsel)
{
  const int sy = (int) bperm (yx, yx, ysel);      // ((Y ^ 0x80) * 0x10100): splatbw << 8, a signed 24-bit value
  const int su = (int) bperm (cx, cx, usel);
  const int sv = (int) bperm (cx, cx, vsel);
  const int wy = mulhi24 (sy, fp.p8[0]);
  const int r = med3_0_255 (wy + mulhi24 (sv, fp.p8[1]) + 128);
  const int b = med3_0_255 (wy + mulhi24 (su, fp.p8[2]) + 128);
  const int g = med3_0_255 (wy + mulhi24 (su, fp.p8[3]) + mulhi24 (sv, fp.p8[4]) + 128);
  // bytes: lo = {r, g, 0, 0}, hi = {b, ...}; pack_sel routes r/g/b and the constant 0xff alpha
  return bperm ((uint32_t) b, ((uint32_t) g << 8) | (uint32_t) r, fp.pack_sel);
}

// ---- word form of the same pixel: sub-dword operand selects (SDWA) instead of byte shuffles ---------------
// mulhsw (t, p) is the high word of the exact 32-bit product t * p, so the products stay whole
// (v_mul_i32_i24 on a sign-extended WORD of the splat register) and the adds pick their WORD_1 directly; sums
// land in the halves of two registers, v_sat_pk_u8_i16 does the signed-saturate + 128 of two channels at once.
// 13 VALU ops per pixel instead of 20; same integers at every step (the planner's no-wrap proof covers the sums).
template <int W>
GSTAMD_HD int mul_word (uint32_t v, int coef)            // sext16 (word W of v) * coef
{
#ifdef __HIPCC__
  int r;
  if (W == 0)
    asm ("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v" (r) : "v" (v), "s" (coef));
  else
    asm ("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v" (r) : "v" (v), "s" (coef));
  return r;
#else
  return (int) (int16_t) (v >> (16 * W)) * coef;
#endif
}

GSTAMD_HD int add_hiword (int a, int prod)                // a + (prod >> 16)
{
#ifdef __HIPCC__
  int r;
  asm ("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v" (r) : "v" (a), "v" (prod));
  return r;
#else
  return a + (prod >> 16);
#endif
}

template <int W>
GSTAMD_HD void add_hiword_into (uint32_t &dst, int a, int prod)   // word W of dst = low16 (a + (prod >> 16)), other word kept
{
#ifdef __HIPCC__
  if (W == 0)
    asm ("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:WORD_1" : "+v" (dst) : "v" (a), "v" (prod));
  else
    asm ("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:WORD_1" : "+v" (dst) : "v" (a), "v" (prod));
#else
  const uint32_t r = (uint32_t) (a + (prod >> 16)) & 0xffffu;
  dst = W == 0 ? (dst & 0xffff0000u) | r : (dst & 0x0000ffffu) | (r << 16);
#endif
}

GSTAMD_HD uint32_t sat_pk_u8 (uint32_t v)                  // {sat_u8 (i16 lane 0), sat_u8 (i16 lane 1)} in bytes 0, 1
{
#ifdef __HIPCC__
  uint32_t r;
  asm ("v_sat_pk_u8_i16 %0, %1" : "=v" (r) : "v" (v));
  return r;
#else
  const int a = (int16_t) (v & 0xffff), b = (int16_t) (v >> 16);
  return (uint32_t) (a < 0 ? 0 : a > 255 ? 255 : a) | ((uint32_t) (b < 0 ? 0 : b > 255 ? 255 : b) << 8);
#endif
}

GSTAMD_HD int add_hiwords (int pa, int pb)                 // (pa >> 16) + (pb >> 16)
{
#ifdef __HIPCC__
  int r;
  asm ("v_add_u32_sdwa %0, sext(%1), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v" (r) : "v" (pa), "v" (pb));
  return r;
#else
  return (pa >> 16) + (pb >> 16);
#endif
}

template <int W>
GSTAMD_HD void add_hiwords_into (uint32_t &dst, int pa, int pb)   // word W of dst = low16 ((pa >> 16) + (pb >> 16)), other word kept
{
#ifdef __HIPCC__
  if (W == 0)
    asm ("v_add_u32_sdwa %0, sext(%1), sext(%2) dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v" (dst) : "v" (pa), "v" (pb));
  else
    asm ("v_add_u32_sdwa %0, sext(%1), sext(%2) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v" (dst) : "v" (pa), "v" (pb));
#else
  const uint32_t r = (uint32_t) ((pa >> 16) + (pb >> 16)) & 0xffffu;
  dst = W == 0 ? (dst & 0xffff0000u) | r : (dst & 0x0000ffffu) | (r << 16);
#endif
}

// ---- layout-specialised pixels: no final byte shuffle, no chroma XOR -------------------------------------------
// PR / PG / PB = destination byte of R, G, B (alpha takes the fourth).  Bytes 0, 1 are the saturated i16 lanes of
// q[0], bytes 2, 3 those of q[1]; every add writes its sum straight into the lane of its channel, the second
// v_sat_pk writes the upper half of the pixel (SDWA dst_sel).  q[][] lives across pixels: the alpha lane is set once.
GSTAMD_HD void sat_pk_u8_hi (uint32_t &dst, uint32_t v)     // bytes 2, 3 of dst = sat_u8 of v's i16 lanes; bytes 0, 1 kept
{
#ifdef __HIPCC__
  asm ("v_sat_pk_u8_i16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v" (dst) : "v" (v));
#else
  dst = (dst & 0xffffu) | (sat_pk_u8 (v) << 16);
#endif
}

// both u16 lanes: a * k + c (mod 2^16); k_splat carries the multiplier in both halves
GSTAMD_HD uint32_t pk_mad (uint32_t a, uint32_t k_splat, uint32_t c)
{
#ifdef __HIPCC__
  uint32_t r;           // written out: the compiler would expand a power-of-two multiplier into shift + add
  asm ("v_pk_mad_u16 %0, %1, %2, %3" : "=v" (r) : "v" (a), "s" (k_splat), "v" (c));
  return r;
#else
  const uint32_t lo = ((a & 0xffffu) * (k_splat & 0xffffu) + (c & 0xffffu)) & 0xffffu;
  const uint32_t hi = ((a >> 16) * (k_splat >> 16) + (c >> 16)) & 0xffffu;
  return lo | (hi << 16);
#endif
}

// vertical 2x chroma blend of two filtered rows a, b (8-bit values in u16 lanes), both roles at once.  The sums are
// kept scaled by 64 so the blended value sits in the HIGH byte of each lane - no shift - and the constant carries the
// rounding (2 * 64) plus 0x8000, which is the "- 128" (XOR 0x80) the colour matrix wants on that byte:
//   x0.hi = ((3a + b + 2) >> 2) ^ 0x80,  x1.hi = ((a + 3b + 2) >> 2) ^ 0x80;   64 * (4 * 255 + 2) < 2^16.
GSTAMD_HD void blend_rows_hi (uint32_t a, uint32_t b, uint32_t &x0, uint32_t &x1)
{
  x0 = pk_mad (a, 0x00c000c0u, pk_mad (b, 0x00400040u, 0x80808080u));
  x1 = pk_mad (b, 0x00c000c0u, pk_mad (a, 0x00400040u, 0x80808080u));
}

#define GSTAMD_LAYOUT(pr, pg, pb) ((pr) | ((pg) << 2) | ((pb) << 4))
#define GSTAMD_LAYOUT_AYUV GSTAMD_LAYOUT (0, 0, 4)      /* 64: not a byte order of R, G, B - the pixel leaves as A Y U V (FastParams::ayuv) */
// the A Y U V word from operands that are XOR 0x80: byte 0 of ys = Y ^ 0x80, bytes 0 and 2 of c = U ^ 0x80, V ^ 0x80
#define GSTAMD_AYUV_X80(ys, c) ((bperm ((c), (ys), 0x0604000cu) ^ 0x80808000u) | 0xffu)
// ... through the convert stage of a plan that has one (wave-uniform)
#define GSTAMD_AYUV_OUT(fp, px) ((fp).ayuv == 2 ? apply_matrix ((fp).m8, (px)) : (px))

template <int L>
GSTAMD_HD void layout_init (uint32_t (&q)[4][2])
{
  constexpr int PA = (L & GSTAMD_LAYOUT_AYUV) ? 0 : 6 - (L & 3) - ((L >> 2) & 3) - ((L >> 4) & 3);          /* (A Y U V: the accumulators are not used) */
#pragma unroll
  for (int j = 0; j < 4; j++) {
    q[j][0] = q[j][1] = 0;
    q[j][PA >> 1] = 255u << (16 * (PA & 1));
#ifdef __HIPCC__
    // opaque to the optimiser: otherwise it re-materialises these constants in front of every tied ("+v") use
    asm volatile ("" : "+v" (q[j][0]), "+v" (q[j][1]));
#endif
  }
}

// four pixels of one line.  yw: luma bytes y0..y3; xh[j]: blended chroma of pixel j in the high bytes of the u16 lanes
// (already XOR 0x80).  Step-major order keeps dependent SDWA instructions apart (gfx950 needs a wait state between them).
template <int L, int ABL>
GSTAMD_HD void fast_emit4_l (const FastParams &fp, uint8_t *__restrict__ d, bool store, uint32_t yw, const uint32_t *xh, uint32_t (&q)[4][2])
{
  constexpr int PR = L & 3, PG = (L >> 2) & 3, PB = (L >> 4) & 3;
  uint32_t o[4];
  if (ABL == 1) {
#pragma unroll
    for (int j = 0; j < 4; j++)
      o[j] = yw ^ xh[j];
  } else {
    const uint32_t yx = yw ^ 0x80808080u;
    const uint32_t ys01 = bperm (yx, yx, 0x01010000u), ys23 = bperm (yx, yx, 0x03030202u);
    const uint32_t csel = fp.u_first ? 0x03030101u : 0x01010303u;
    uint32_t cs[4];
    int wy[4], prv[4], pbu[4], pgu[4], pgv[4], g0[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
      cs[j] = bperm (xh[j], xh[j], csel);
    wy[0] = mul_word<0> (ys01, fp.pc[0]) + 0x00800000;
    wy[1] = mul_word<1> (ys01, fp.pc[0]) + 0x00800000;
    wy[2] = mul_word<0> (ys23, fp.pc[0]) + 0x00800000;
    wy[3] = mul_word<1> (ys23, fp.pc[0]) + 0x00800000;
#pragma unroll
    for (int j = 0; j < 4; j++)
      pgu[j] = mul_word<0> (cs[j], fp.pc[3]);
#pragma unroll
    for (int j = 0; j < 4; j++)
      prv[j] = mul_word<1> (cs[j], fp.pc[1]);
#pragma unroll
    for (int j = 0; j < 4; j++)
      g0[j] = add_hiwords (wy[j], pgu[j]);
#pragma unroll
    for (int j = 0; j < 4; j++)
      pgv[j] = mul_word<1> (cs[j], fp.pc[4]);
#pragma unroll
    for (int j = 0; j < 4; j++)
      add_hiwords_into<PR & 1> (q[j][PR >> 1], wy[j], prv[j]);
#pragma unroll
    for (int j = 0; j < 4; j++)
      pbu[j] = mul_word<0> (cs[j], fp.pc[2]);
#pragma unroll
    for (int j = 0; j < 4; j++)
      add_hiword_into<PG & 1> (q[j][PG >> 1], g0[j], pgv[j]);
#pragma unroll
    for (int j = 0; j < 4; j++)
      add_hiwords_into<PB & 1> (q[j][PB >> 1], wy[j], pbu[j]);
#pragma unroll
    for (int j = 0; j < 4; j++)
      o[j] = sat_pk_u8 (q[j][0]);
#pragma unroll
    for (int j = 0; j < 4; j++)
      sat_pk_u8_hi (o[j], q[j][1]);
  }
  if (ABL == GSTAMD_FAST_LUT) {
#pragma unroll
    for (int j = 0; j < 4; j++)
      o[j] = fast_lut3_px (o[j], fp.lut_keep);
  }
  if (store) {
    if (fp.px_bytes == 3)                   // wave-uniform
      store12_stream (d, o[0], o[1], o[2], o[3]);
    else
      store16_policy (d, o[0], o[1], o[2], o[3], fp.store_policy);
  }
}

// Horizontal chroma filter of 4 pixels without edge cases.  raw = the 2 samples {c0 c1 c0 c1} under the span, nxt = the
// dword right of it (its first sample is used), prv_hi = the dword left of it (its second sample is used).  Callers
// pass the row's last sample again right of the row end and its first sample left of the start, and
// (a + a + 1) >> 1 == (3a + a + 2) >> 2 == a, so the formulas of video_chroma_up_h2_cs_u8 / up_h2_u8
// (video-chroma.c:687, 277) reproduce their own edge rules.  out[i] = packed {c0, c1} in u16 lanes.
template <int CH>
GSTAMD_HD void chroma_filter4 (uint32_t raw, uint32_t nxt, uint32_t prv_hi, uint32_t *out)
{
  const uint32_t s1 = bperm (raw, raw, 0x0c010c00u), s2 = bperm (raw, raw, 0x0c030c02u);
  if (CH == CHROMA_H_NONE) {
    out[0] = out[1] = s1;
    out[2] = out[3] = s2;
    return;
  }
  const uint32_t s3 = bperm (nxt, nxt, 0x0c010c00u);
  if (CH == CHROMA_H_H2_CS) {
    out[0] = s1;
    out[1] = pk_shr<1> (s1 + s2 + 0x00010001u);
    out[2] = s2;
    out[3] = pk_shr<1> (s2 + s3 + 0x00010001u);
  } else {
    const uint32_t s0 = bperm (prv_hi, prv_hi, 0x0c030c02u);
    out[0] = pk_shr<2> (s0 + 3u * s1 + 0x00020002u);
    out[1] = pk_shr<2> (3u * s1 + s2 + 0x00020002u);
    out[2] = pk_shr<2> (s1 + 3u * s2 + 0x00020002u);
    out[3] = pk_shr<2> (3u * s2 + s3 + 0x00020002u);
  }
}

// ---- strip variant: one lane walks K consecutive line pairs of its 4-pixel column ---------------------------
// Loads of pair p+1 are issued before pair p is computed (software prefetch: with ~16 waves per SIMD in the
// whole grid the hardware alone cannot overlap the load, VALU and store phases), and the horizontally
// filtered chroma row of pair p is reused as the upper row of pair p+1 (each chroma row is loaded and
// filtered exactly once per strip).
struct ChromaRaw4 {
  uint32_t raw, nxt, prv_hi;
};

struct __attribute__ ((aligned (4))) u32x2u { uint32_t a, b; };

GSTAMD_HD uint32_t load_stream32 (const uint8_t *__restrict__ p)
{
#ifdef __HIPCC__
  return __builtin_nontemporal_load ((const uint32_t *) p);       // luma is read exactly once: streaming load
#else
  return *(const uint32_t *) p;
#endif
}

// the 4 chroma bytes under pixels x0 .. x0+3 of a row (cw samples) plus the neighbours the filter needs
template <int CH>
GSTAMD_HD void chroma_load4 (const uint8_t *__restrict__ row, int cw, int x0, ChromaRaw4 &r)
{
  const int k0 = x0 >> 1;
  const uint8_t *p = row + 2 * (size_t) k0;
  r.nxt = r.prv_hi = 0;
  // the sample right of the span rides in the same (dword-aligned) load unless the span ends the row
  if (CH != CHROMA_H_NONE && k0 + 2 < cw) {
    const u32x2u m = *(const u32x2u *) p;
    r.raw = m.a;
    r.nxt = m.b;
  } else {
    r.raw = *(const uint32_t *) p;
    if (CH != CHROMA_H_NONE)
      r.nxt = *(const uint16_t *) (row + 2 * (size_t) (cw - 1));
  }
  if (CH == CHROMA_H_H2)
    r.prv_hi = (uint32_t) * (const uint16_t *) (row + 2 * (size_t) (k0 > 0 ? k0 - 1 : 0)) << 16;
}

// pairs [p_begin, p_end) of the column x0 .. x0+3; pair p = lines (2p-1, 2p), chroma rows (p-1, p), all clamped into
// the frame: a frame's first / last pair has one line only, its two chroma rows are then the same row, the blend is
// the identity and the missing line is computed but not stored.
template <int CH, int L, int ABL>
GSTAMD_HD void fast_strip (const FastParams &fp, const Planes &pl, uint8_t *__restrict__ dst, int dstride, int x0, int p_begin,
    int p_end)
{
  const int w = fp.width, h = fp.height, cw = (w + 1) >> 1;
  const uint8_t *yb = pl.p[0] + x0, *cbase = pl.p[1];
  const int ys = pl.stride[0], cs = pl.stride[1];
  uint8_t *db = dst + (size_t) fp.px_bytes * (size_t) x0;
  uint32_t q[4][2];
  layout_init<L> (q);

  uint32_t cprev[4], ccur[4], c0[4], c1[4];
  ChromaRaw4 craw;
  uint32_t y0, y1;
  {                                       // upper chroma row of the first pair
    ChromaRaw4 r0;
    chroma_load4<CH> (cbase + (ptrdiff_t) (p_begin - 1 > fp.crow_lo ? p_begin - 1 : fp.crow_lo) * cs, cw, x0, r0);
    chroma_filter4<CH> (r0.raw, r0.nxt, r0.prv_hi, cprev);
  }
  {                                       // prefetch pair p_begin
    const int p = p_begin, l0 = 2 * p - 1, l1 = 2 * p;
    chroma_load4<CH> (cbase + (ptrdiff_t) (p < fp.crow_hi ? p : fp.crow_hi) * cs, cw, x0, craw);
    y0 = load_stream32 (yb + (size_t) (l0 >= 0 ? l0 : 0) * ys);
    y1 = load_stream32 (yb + (size_t) (l1 < h ? l1 : h - 1) * ys);
  }
  for (int p = p_begin; p < p_end; p++) {
    const int l0 = 2 * p - 1, l1 = 2 * p;
    const bool have0 = l0 >= 0, have1 = l1 < h;
    const uint32_t cy0 = y0, cy1 = y1;
    const ChromaRaw4 cr_now = craw;
    if (p + 1 < p_end) {                  // issue the next pair's loads before this pair's math
      const int pn = p + 1, n0 = 2 * pn - 1, n1 = 2 * pn;
      chroma_load4<CH> (cbase + (ptrdiff_t) (pn < fp.crow_hi ? pn : fp.crow_hi) * cs, cw, x0, craw);
      y0 = load_stream32 (yb + (size_t) n0 * ys);
      y1 = load_stream32 (yb + (size_t) (n1 < h ? n1 : h - 1) * ys);
    }
    chroma_filter4<CH> (cr_now.raw, cr_now.nxt, cr_now.prv_hi, ccur);
#pragma unroll
    for (int j = 0; j < 4; j++)
      blend_rows_hi (cprev[j], ccur[j], c0[j], c1[j]);
    fast_emit4_l<L, ABL> (fp, db + (size_t) (have0 ? l0 : l1) * dstride, have0, cy0, c0, q);
    fast_emit4_l<L, ABL> (fp, db + (size_t) (have1 ? l1 : l0) * dstride, have1, cy1, c1, q);
#pragma unroll
    for (int i = 0; i < 4; i++)
      cprev[i] = ccur[i];
  }
}

// ---- wide variant: one wave = a 1024-pixel run of ONE line pair, staged through LDS ---------------------
// Every source row segment is fetched with ONE 16-byte load per lane (1 KB contiguous per wave instruction), parked in
// LDS and read back in the 4-pixels-per-lane layout of fast_emit4_l, so each store instruction still writes 1 KB
// contiguous and four of them cover a 4 KB run of the destination row.  Measured on MI355X with the memory skeleton
// (scripts/membench3.hip): 7.9 us per 4K frame against 8.5 us for the strip shape and 7.6 us for a perfectly linear
// kernel moving the same bytes.  Arithmetic and results are those of fast_strip.
#define GSTAMD_WIDE_PX 1024
struct WideLds {
  uint32_t y[2][256];
  uint32_t c[2][4 + 256 + 4];     // [3] = dword left of the run, [4 ..] = the run, then the dword right of it
};

#ifdef __HIPCC__
typedef unsigned int gstamd_u32x4 __attribute__ ((ext_vector_type (4)));
#endif

// 16 bytes of a row at byte offset xl (row + xl 16-byte aligned when `vec`), limited to the first `nd` dwords
template <bool STREAM>
GSTAMD_HD void wide_load16 (const uint8_t *__restrict__ p, int nd, bool vec, uint32_t *out)
{
  if (nd >= 4 && vec) {
#ifdef __HIPCC__
    const gstamd_u32x4 v = STREAM ? __builtin_nontemporal_load ((const gstamd_u32x4 *) p) : *(const gstamd_u32x4 *) p;
    out[0] = v.x;
    out[1] = v.y;
    out[2] = v.z;
    out[3] = v.w;
#else
    const uint4 v = *(const uint4 *) p;
    out[0] = v.x;
    out[1] = v.y;
    out[2] = v.z;
    out[3] = v.w;
#endif
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++)
      out[j] = j < nd ? *(const uint32_t *) (p + 4 * j) : 0u;
  }
}

GSTAMD_HD void wide_put16 (uint32_t *l, const uint32_t *v)
{
  *(uint4 *) l = gstamd_make_uint4 (v[0], v[1], v[2], v[3]);
}

// One lane's share of the rows a pair needs, held in registers between the fetch (global loads, issued one pair ahead)
// and the commit (LDS writes, after the previous pair has been emitted).
struct WideRegs {
  uint32_t y0[4], y1[4], c[4];
  uint32_t cn, cp;                 // chroma dwords right / left of the run (only some lanes, see wide_fetch_chroma)
};

// chroma row `crow` of the run starting at xw: this lane's 16 bytes plus, on the lanes that own them, the pad dwords
template <int CH>
GSTAMD_HD void wide_fetch_chroma (const FastParams &fp, const Planes &pl, int xw, int crow, int lane, bool vec, WideRegs &r)
{
  const int w = fp.width, xl = xw + 16 * lane;
  if (xl >= w)
    return;
  const int nd = (w - xl) >> 2;            // dwords of this lane that lie inside the row (w % 4 == 0)
  const uint8_t *row = pl.p[1] + (ptrdiff_t) crow * pl.stride[1];
  wide_load16<false> (row + xl, nd, vec, r.c);
  if (CH != CHROMA_H_NONE && (lane == 63 || nd <= 4)) {
    // the sample right of this lane's last one: the next dword of the row, or the row's last sample again
    const int xn = xl + 4 * (nd < 4 ? nd : 4);
    r.cn = xn < w ? *(const uint32_t *) (row + xn) : (uint32_t) * (const uint16_t *) (row + w - 2);
  }
  if (CH == CHROMA_H_H2 && lane == 0)         // the sample left of the run (high half of the dword), clamped at the row start
    r.cp = xw > 0 ? *(const uint32_t *) (row + xw - 4) : 0u;
}

template <int CH>
GSTAMD_HD void wide_commit_chroma (const FastParams &fp, int xw, int lane, const WideRegs &r, uint32_t *crow_lds)
{
  const int w = fp.width, xl = xw + 16 * lane;
  if (xl >= w)
    return;
  const int nd = (w - xl) >> 2;
  wide_put16 (&crow_lds[4 + 4 * lane], r.c);
  if (CH != CHROMA_H_NONE && (lane == 63 || nd <= 4))
    crow_lds[4 + 4 * lane + (nd < 4 ? nd : 4)] = r.cn;
  if (CH == CHROMA_H_H2 && lane == 0)
    crow_lds[3] = xw > 0 ? r.cp : r.c[0] << 16;
}

// rows of pair p (lines 2p-1, 2p clamped into the frame; chroma row p clamped): the pair's upper chroma row p-1 is
// already in LDS (previous pair of the strip, or the strip prologue)
template <int CH>
GSTAMD_HD void wide_fetch (const FastParams &fp, const Planes &pl, int xw, int p, int lane, bool vec, WideRegs &r)
{
  const int w = fp.width, h = fp.height;
  const int xl = xw + 16 * lane;
  if (xl >= w)
    return;
  const int nd = (w - xl) >> 2;
  const int l0 = 2 * p - 1, l1 = 2 * p;
  const int r0 = l0 >= 0 ? l0 : 0, r1 = l1 < h ? l1 : h - 1;
  wide_load16<true> (pl.p[0] + (size_t) r0 * pl.stride[0] + xl, nd, vec, r.y0);
  wide_load16<true> (pl.p[0] + (size_t) r1 * pl.stride[0] + xl, nd, vec, r.y1);
  wide_fetch_chroma<CH> (fp, pl, xw, p < fp.crow_hi ? p : fp.crow_hi, lane, vec, r);
}

template <int CH>
GSTAMD_HD void wide_commit (const FastParams &fp, int xw, int lane, const WideRegs &r, WideLds *lds, int cslot)
{
  if (xw + 16 * lane >= fp.width)
    return;
  wide_put16 (&lds->y[0][4 * lane], r.y0);
  wide_put16 (&lds->y[1][4 * lane], r.y1);
  wide_commit_chroma<CH> (fp, xw, lane, r, lds->c[cslot]);
}

// phase 2, lane `lane`: four 4-pixel groups (x = xw + 256 g + 4 lane) of both lines, from LDS.  The first and the last
// pair of a frame have one line only; the same chroma row is then staged twice (clamped row index), and (3a + a + 2) >> 2 == a makes
// the blend the identity, so no case distinction is needed in the arithmetic.
template <int CH, int L, int ABL>
GSTAMD_HD void wide_emit (const FastParams &fp, uint8_t *__restrict__ dst, int dstride, int xw, int p, int lane, const WideLds *lds, int aslot)
{
  const int w = fp.width, h = fp.height;
  const int l0 = 2 * p - 1, l1 = 2 * p;
  const bool have0 = l0 >= 0, have1 = l1 < h;
  uint8_t *d0 = dst + (size_t) (have0 ? l0 : l1) * dstride + 4 * (size_t) (xw + 4 * lane);
  uint8_t *d1 = dst + (size_t) (have1 ? l1 : l0) * dstride + 4 * (size_t) (xw + 4 * lane);
  uint32_t q[4][2];
  layout_init<L> (q);
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const int x0 = xw + 256 * g + 4 * lane, idx = 64 * g + lane;
    uint32_t ca[4], cb[4], c0[4], c1[4];
    // lanes right of the row end compute on whatever LDS holds (in bounds) and skip the store
    const uint32_t *ra = lds->c[aslot], *rb = lds->c[aslot ^ 1];
    chroma_filter4<CH> (ra[4 + idx], ra[5 + idx], ra[3 + idx], ca);
    chroma_filter4<CH> (rb[4 + idx], rb[5 + idx], rb[3 + idx], cb);
#pragma unroll
    for (int j = 0; j < 4; j++)
      blend_rows_hi (ca[j], cb[j], c0[j], c1[j]);
    // a frame's first / last pair has one line only: the other one is computed (on in-bounds data) and not stored
    fast_emit4_l<L, ABL> (fp, d0 + 1024 * g, have0 && x0 < w, lds->y[0][idx], c0, q);
    fast_emit4_l<L, ABL> (fp, d1 + 1024 * g, have1 && x0 < w, lds->y[1][idx], c1, q);
  }
}

// Block order of the wide kernel (1-D grid).  One wave = one strip: a 1024-px column x K consecutive line pairs, walked
// pair by pair with the next pair's rows in flight.  Hardware places block b on XCD b % 8, each XCD with its own L2.
// Units of (column, band of GSTAMD_WIDE_BAND consecutive strips) are dealt round-robin to the XCDs and every XCD walks
// its unit strip by strip: the chroma row two neighbouring strips share is re-read from that XCD's L2, not from HBM,
// while all eight XCDs still sweep the same region of the frame together.  Strips of the batch's frames are numbered
// consecutively (S = frame * strips_per_frame + strip).  Returns false for padding blocks.
#define GSTAMD_WIDE_BAND 32
GSTAMD_HD bool wide_block_map (int block, int nxb, int total_strips, int *x, int *S)
{
  const int xcd = block & 7, i = block >> 3;
  const int unit = (i / GSTAMD_WIDE_BAND) * 8 + xcd, sb = i % GSTAMD_WIDE_BAND;
  *x = unit % nxb;
  *S = (unit / nxb) * GSTAMD_WIDE_BAND + sb;
  return *S < total_strips;
}

inline int wide_grid_blocks (int nxb, int total_strips)
{
  const int units = nxb * ((total_strips + GSTAMD_WIDE_BAND - 1) / GSTAMD_WIDE_BAND);
  return (units + 7) / 8 * 8 * GSTAMD_WIDE_BAND;
}

}  // namespace gstamd
