// video_swizzle34.h - byte permutations between 3- and 4-byte pixels: RGB / BGR / v308 / IYU2 <-> the 4-byte RGB and YUV orders, and RGB <-> BGR,
// when the chain has neither a matrix nor an alpha operation (unpack_RGB / pack_RGB & co, video-format.c:1521-1595, are byte moves; a
// 3-byte source unpacks with alpha 0xff).  Four pixels per lane: 12 or 16 source bytes as aligned words, every destination word one or two
// v_perm_b32 of neighbouring source words (the word pairs are fixed by the two pixel sizes, the selectors come from the formats).
#pragma once
#include "video_device.h"

namespace gstamd {

struct Swz34Params {
  uint32_t sel_a[4], sel_b[4];  // per destination word: selector of its first / second v_perm (0x0c0c0c0c: no second one)
  uint8_t map[4];               // destination byte j of a pixel <- source byte map[j] of the pixel, 0xff: the constant 0xff
  const uint8_t *src;
  uint8_t *dst;
  int sstride, dstride, width;
};

GSTAMD_HD uint32_t swz_perm (uint32_t hi, uint32_t lo, uint32_t sel)
{
#ifdef __HIPCC__
  return __builtin_amdgcn_perm (hi, lo, sel);
#else
  uint32_t r = 0;
  for (int j = 0; j < 4; j++) {
    const uint32_t s = (sel >> (8 * j)) & 0xffu;
    const uint32_t b = s < 4 ? (lo >> (8 * s)) & 0xffu : (s < 8 ? (hi >> (8 * (s - 4))) & 0xffu : (s >= 0x0d ? 0xffu : 0u));
    r |= b << (8 * j);
  }
  return r;
#endif
}

// first source word of the pair (w[a], w[a + 1]) destination word m takes its bytes from; second pair (-1: none)
template <int SB, int DB> GSTAMD_VP constexpr int swz_pair_a (int m) { return SB == 3 && DB == 4 ? (m < 2 ? 0 : m == 2 ? 1 : 2) : (SB == 4 ? m : (m == 0 ? 0 : m == 1 ? 0 : 1)); }
template <int SB, int DB> GSTAMD_VP constexpr int swz_pair_b (int m) { return SB == 3 && DB == 3 && m == 1 ? 2 : -1; }

// selectors for a format pair (host)
template <int SB, int DB>
inline void swz34_selectors (const uint8_t map[4], Swz34Params *p)
{
  for (int m = 0; m < DB; m++) {        /* 4 pixels x DB bytes = DB words */
    uint32_t sa = 0, sb = 0;
    const int pa = swz_pair_a<SB, DB> (m), pb = swz_pair_b<SB, DB> (m);
    for (int j = 0; j < 4; j++) {
      const int k = 4 * m + j, px = k / DB, db = k % DB;
      uint32_t a = 0x0c, b = 0x0c;
      if (map[db] == 0xff) {
        a = 0x0d;
      } else {
        const int s = px * SB + map[db], ws = s / 4, wb = s % 4;
        if (ws == pa)
          a = (uint32_t) wb;
        else if (ws == pa + 1)
          a = 4u + (uint32_t) wb;
        else if (pb >= 0 && ws == pb)
          b = (uint32_t) wb;
        else if (pb >= 0 && ws == pb + 1)
          b = 4u + (uint32_t) wb;
      }
      sa |= a << (8 * j);
      sb |= b << (8 * j);
    }
    p->sel_a[m] = sa;
    p->sel_b[m] = sb;
  }
  for (int j = 0; j < 4; j++)
    p->map[j] = map[j];
}

template <int SB, int DB>
GSTAMD_HD void swizzle34_body (const Swz34Params &p, int lane, int y)
{
  const int x0 = 4 * lane;
  if (x0 >= p.width)
    return;
  const uint8_t *s = p.src + (size_t) y * p.sstride + (size_t) x0 * SB;
  uint8_t *d = p.dst + (size_t) y * p.dstride + (size_t) x0 * DB;
  if (x0 + 4 <= p.width) {
    uint32_t w[5];
#pragma unroll
    for (int i = 0; i < SB; i++)
      w[i] = ((const uint32_t *) s)[i];
    w[SB] = 0;
    if (SB == 3)
      w[4] = 0;
#pragma unroll
    for (int m = 0; m < DB; m++) {
      const int pa = swz_pair_a<SB, DB> (m), pb = swz_pair_b<SB, DB> (m);
      uint32_t o = swz_perm (w[pa + 1], w[pa], p.sel_a[m]);
      if (pb >= 0)
        o |= swz_perm (w[pb + 1], w[pb], p.sel_b[m]);
      ((uint32_t *) d)[m] = o;
    }
    return;
  }
  for (int i = 0; x0 + i < p.width; i++)
    for (int j = 0; j < DB; j++)
      d[i * DB + j] = p.map[j] == 0xff ? 0xff : s[i * SB + p.map[j]];
}

}  // namespace gstamd
