// video_scale420_fused.h - BASELINE C3 in ONE kernel: horizontal N-tap pass from a regular 4:2:0 frame (the line-pair walk of
// video_hscale420.h) and the vertical N-tap pass, with the horizontally filtered lines kept in an LDS ring instead of the AYUV
// image in HBM that the two-pass form writes and reads back (video-converter.c chain_scale :1685-1717 runs the two scalers as
// separate line stages; the integers of every stage are the same here: clamped u8 AYUV after the horizontal pass,
// video_scale_h_ntap_u8 video-scaler.c:621-760, then video_scale_v_ntap_u8 :987-1072 on those bytes).
//
// Work split.  A workgroup of NWAVES waves owns a column tile of <= 256 output pixels and a chunk of output rows.  Source lines are
// handled in GROUPS of four (lines 4g-1 .. 4g+2 = the line pairs 2g and 2g+1 of the chroma upsampler): a wave filters the four
// lines of a group horizontally exactly like k_hscale420_reg and, instead of storing AYUV pixels, packs the four lines' results
// of one output and one channel into ONE word (byte b = line 4g-1+b, XOR 0x80).  The ring slot of a group is [channel][slot i]
// [lane] words, output x = t0 + lane + 64 i.  The vertical pass is then a byte dot product down the ring: a window of n taps
// starting at line `off` covers ceil ((n + s) / 4) groups from g = (off + 1) >> 2 with the taps shifted by s = (off + 1) & 3
// places in zero-padded int8 words (zero taps add nothing to the 16-bit wrapping sum) - the same construction as the horizontal
// pass's tap words.  Sum of taps = 64 in every phase (checked on the host), so sum (px * tap) = sum ((px - 128) * tap) + 128 * 64.
//
// A round: every wave produces the groups of its residue class that the round's rows need (loads of the NEXT group are in
// flight meanwhile), barrier, every wave filters one output row of the tile vertically (taps wave-uniform), post stage, store,
// barrier.  HBM traffic: the source once per chunk (+ (n_taps - step) lines of overlap between chunks) and the output once.
#pragma once
#include "video_hscale420.h"

#define GSTAMD_FUSED_GROUP_WORDS (12 * 64)

namespace gstamd {

struct Fused420Params {
  H420RegParams h;              // source planes, horizontal tables, tile_w, out_w (dst / dstride / lines_per_wave unused)
  int n_taps_h;
  const int32_t *vgroup;        // [out_h] first line group of the row's window
  const uint32_t *vtapw;        // [out_h][ngv] int8 x 4 tap words aligned to the groups
  int ngv;
  int out_h;
  int rows_per_chunk;           // output rows per workgroup (blockIdx.y)
  int first_rows;               // rows of a chunk's first round (later rounds: one row per wave): chosen so that the first round, which
                                // also fills the ring's run-in, has one group per wave like the others (no wave waits for a second)
  int ring;                     // ring slots (groups) in LDS
  int sched;                    // 1: two barriers per round, a staged line PAIR per wave.  2: one barrier per round - the ring also holds the
                                // NEXT round's groups, so between two barriers a wave filters its row of round k and produces its groups of
                                // round k + 1, odd waves in the other order (the vector ALU work of one half overlaps the LDS / memory waits
                                // of the other); the stage area of a wave is ONE line (the LDS that pays for the longer ring)
  int n_groups;                 // line groups of the picture: (height / 2 + 2) / 2
#ifdef GSTAMD_TUNING
  unsigned long long *trace;    // profiling builds: [workgroup][wave][32] s_memtime stamps of the kernel's stages, or NULL
#endif
};

// per-lane registers of the horizontal part
template <int NW>
struct Fused420Lane {
  Dot4Taps<NW> ft;
  uint32_t P[8], Q[8];
  H420Pair pr;                  // the pair in flight
  H420Raw pre;                  // chroma row 2g-1 of the next group
  uint32_t gw[12];              // [channel][slot]: four lines of one output and channel
  int x0;
};

// one line of the pair: outputs t0 + lane + 64 i from the byte planes, results into byte B of the group words
template <int NW, int B>
GSTAMD_HD void fused_filter_line (const uint32_t *line, const Dot4Taps<NW> &ft, uint32_t *gw)
{
  const int pw = GSTAMD_H420_PLANE_BYTES / 4;
#pragma unroll
  for (int i0 = 0; i0 < 4; i0 += 2) {
    uint32_t wy[2][NW], wu[2][NW], wv[2][NW];
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const uint32_t *b = line + ft.w0[i0 + j];
#pragma unroll
      for (int k = 0; k < NW; k++) {
#if defined(GSTAMD_FUSED_ABL) && (GSTAMD_FUSED_ABL == 2 || GSTAMD_FUSED_ABL == 3)
        /* profiling builds only (-DGSTAMD_FUSED_ABL=n, results WRONG): 2 = no LDS reads in the horizontal filter, 3 = no arithmetic either */
        wy[j][k] = (uint32_t) ft.w0[i0 + j] + k, wu[j][k] = wy[j][k] * 3u, wv[j][k] = wy[j][k] * 5u;
#else
        wy[j][k] = h420r_lds (b + k);
        wu[j][k] = h420r_lds (b + pw + k);
        wv[j][k] = h420r_lds (b + 2 * pw + k);
#endif
      }
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int i = i0 + j;
      int ay = 128 * 64 + 32, au = 128 * 64 + 32, av = 128 * 64 + 32;
#pragma unroll
      for (int k = 0; k < NW; k++) {
        const uint32_t t = ft.t[i][k];
#if defined(GSTAMD_FUSED_ABL) && GSTAMD_FUSED_ABL == 3
        ay ^= (int) wy[j][k], au ^= (int) wu[j][k], av ^= (int) (wv[j][k] + t);
#else
        ay = dot4_i8 (wy[j][k], t, ay);
        au = dot4_i8 (wu[j][k], t, au);
        av = dot4_i8 (wv[j][k], t, av);
#endif
      }
      if (B == 0) {
        gw[i] = h420r_finish (ay);
        gw[4 + i] = h420r_finish (au);
        gw[8 + i] = h420r_finish (av);
      } else {
        gw[i] |= h420r_finish (ay) << (8 * B);
        gw[4 + i] |= h420r_finish (au) << (8 * B);
        gw[8 + i] |= h420r_finish (av) << (8 * B);
      }
    }
  }
}

// loads a group needs before its first pair: chroma row 2g-1 (clamped) and the pair 2g
template <int NW, int SEMI>
GSTAMD_HD void fused_request_group (const H420RegParams &p, int g, Fused420Lane<NW> &s)
{
  h420r_load_raw<SEMI> (p, h420r_crow (p, 2 * g - 1), s.x0 >> 1, s.pre);
  h420r_request<SEMI> (p, 2 * g, s.x0, s.pr);           /* g <= n_groups - 1, so pair 2g exists */
}

// the phases of one group; between two consecutive phases the wave synchronises its LDS traffic (wave_lds_sync)
//   A: filter row 2g-1 into P, stage pair 2g (P older, Q fresh), request pair 2g+1
//   B: filter the two staged lines -> bytes 0, 1
//   C: stage pair 2g+1 (Q older, P fresh), request the next group of this wave (g_next, clamped by the caller)
//   D: filter -> bytes 2, 3, then the group words go to the ring slot
template <int NW, int CH, int SEMI>
GSTAMD_HD void fused_phase_a (const H420RegParams &p, Fused420Lane<NW> &s, uint32_t *stage, int g, int lane)
{
  h420_filter_raw2<CH> (SEMI != 0, p.u_first != 0, s.pre, s.P);
  h420r_stage_pair<CH, SEMI> (p, s.pr, s.P, s.Q, stage, 4 * lane);
  const int last = p.height / 2;                        /* a picture of `height` lines has the pairs 0 .. height / 2 */
  h420r_request<SEMI> (p, 2 * g + 1 < last ? 2 * g + 1 : last, s.x0, s.pr);     /* past the picture: harmless loads, zero taps */
}

template <int NW>
GSTAMD_HD void fused_phase_b (Fused420Lane<NW> &s, const uint32_t *stage)
{
  fused_filter_line<NW, 0> (stage, s.ft, s.gw);
  fused_filter_line<NW, 1> (stage + GSTAMD_H420_LINE_WORDS, s.ft, s.gw);
}

template <int NW, int CH, int SEMI>
GSTAMD_HD void fused_phase_c (const H420RegParams &p, Fused420Lane<NW> &s, uint32_t *stage, int g_next, int lane)
{
  h420r_stage_pair<CH, SEMI> (p, s.pr, s.Q, s.P, stage, 4 * lane);
  fused_request_group<NW, SEMI> (p, g_next, s);
}

template <int NW>
GSTAMD_HD void fused_phase_d (Fused420Lane<NW> &s, const uint32_t *stage, uint32_t *slot, int lane)
{
  fused_filter_line<NW, 2> (stage, s.ft, s.gw);
  fused_filter_line<NW, 3> (stage + GSTAMD_H420_LINE_WORDS, s.ft, s.gw);
#pragma unroll
  for (int k = 0; k < 12; k++)
    slot[64 * k + lane] = s.gw[k] ^ 0x80808080u;
}

// ---- schedule 2: the same arithmetic line by line (one staged line per wave) ---------------------------------------------------------
// line B of group g (B = 0 .. 3 = lines 4g-1 .. 4g+2): stage it.  B 0 / 1 are pair 2g (chroma: P older, Q fresh), 2 / 3 pair 2g+1 (Q older,
// P fresh).  After the pair's second line its registers are free: the next pair / the next group of this wave is requested there.
template <int NW, int CH, int SEMI, int B>
GSTAMD_HD void fused2_stage (const H420RegParams &p, Fused420Lane<NW> &s, uint32_t *stage, int g, int g_next, int lane)
{
  const int pw = GSTAMD_H420_PLANE_BYTES / 4, w0 = 4 * lane;
  if (B == 0) {
    h420_filter_raw2<CH> (SEMI != 0, p.u_first != 0, s.pre, s.P);
    h420_filter_raw2<CH> (SEMI != 0, p.u_first != 0, s.pr.raw, s.Q);
    h420r_stage_luma (s.pr.la, stage + w0);
    h420_blend_store (s.P, s.Q, stage + pw + w0, stage + 2 * pw + w0);
  } else if (B == 1) {
    h420r_stage_luma (s.pr.lb, stage + w0);
    h420_blend_store (s.Q, s.P, stage + pw + w0, stage + 2 * pw + w0);
    const int last = p.height / 2;
    h420r_request<SEMI> (p, 2 * g + 1 < last ? 2 * g + 1 : last, s.x0, s.pr);
  } else if (B == 2) {
    h420_filter_raw2<CH> (SEMI != 0, p.u_first != 0, s.pr.raw, s.P);
    h420r_stage_luma (s.pr.la, stage + w0);
    h420_blend_store (s.Q, s.P, stage + pw + w0, stage + 2 * pw + w0);
  } else {
    h420r_stage_luma (s.pr.lb, stage + w0);
    h420_blend_store (s.P, s.Q, stage + pw + w0, stage + 2 * pw + w0);
    fused_request_group<NW, SEMI> (p, g_next, s);
  }
}

// filter the staged line into byte B of the group words; after the fourth line the words go to the ring slot
template <int NW, int B>
GSTAMD_HD void fused2_filter (Fused420Lane<NW> &s, const uint32_t *stage, uint32_t *slot, int lane)
{
  fused_filter_line<NW, B> (stage, s.ft, s.gw);
  if (B == 3) {
#pragma unroll
    for (int k = 0; k < 12; k++)
      slot[64 * k + lane] = s.gw[k] ^ 0x80808080u;
  }
}

// one window word of the vertical pass: the lane's 12 (channel, slot) words of ring slot `slot` against tap word t
GSTAMD_HD void fused_vstep (const uint32_t *ring, int slot, int lane, uint32_t t, int *acc)
{
  const uint32_t *w = ring + (size_t) slot * GSTAMD_FUSED_GROUP_WORDS + lane;
  uint32_t v[12];
#pragma unroll
  for (int k = 0; k < 12; k++)
    v[k] = h420r_lds (w + 64 * k);
#pragma unroll
  for (int k = 0; k < 12; k++)
    acc[k] = dot4_i8 (v[k], t, acc[k]);
}

// vertical pass of output row j for the lane's four outputs, post stage, store: the row's first ring slot and its NGV window words given
template <int NGV>
GSTAMD_HD void fused_vrow_words (const Fused420Params &p, const uint32_t *ring, const Dst &dst, const PostFast &pf, int j, int t0, int t1, int lane, int slot,
    const uint32_t *tw)
{
  int acc[12];
#pragma unroll
  for (int k = 0; k < 12; k++)
    acc[k] = 128 * 64 + 32;
#pragma unroll
  for (int r = 0; r < NGV; r++) {
    fused_vstep (ring, slot, lane, tw[r], acc);
    slot = slot + 1 == p.ring ? 0 : slot + 1;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int x = t0 + lane + 64 * i;
    const uint32_t px = 0xffu | (h420r_finish (acc[i]) << 8) | (h420r_finish (acc[4 + i]) << 16) | (h420r_finish (acc[8 + i]) << 24);
    if (x < t1)
      store_px (dst, x, j, post_px (dst, pf, px));
  }
}

// the same with the tables read here.  NGV > 0: window words known at compile time.
template <int NGV>
GSTAMD_HD void fused_vrow (const Fused420Params &p, const uint32_t *ring, const Dst &dst, const PostFast &pf, int j, int t0, int t1, int lane)
{
  const uint32_t *tw = p.vtapw + (size_t) j * p.ngv;
  if (NGV > 0) {
    uint32_t w[NGV > 0 ? NGV : 1];
#pragma unroll
    for (int r = 0; r < NGV; r++)
      w[r] = tw[r];
    fused_vrow_words<NGV> (p, ring, dst, pf, j, t0, t1, lane, p.vgroup[j] % p.ring, w);
    return;
  }
  int acc[12];
#pragma unroll
  for (int k = 0; k < 12; k++)
    acc[k] = 128 * 64 + 32;
  int slot = p.vgroup[j] % p.ring;
  for (int r = 0; r < p.ngv; r++) {
    fused_vstep (ring, slot, lane, tw[r], acc);
    slot = slot + 1 == p.ring ? 0 : slot + 1;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int x = t0 + lane + 64 * i;
    const uint32_t px = 0xffu | (h420r_finish (acc[i]) << 8) | (h420r_finish (acc[4 + i]) << 16) | (h420r_finish (acc[8 + i]) << 24);
    if (x < t1)
      store_px (dst, x, j, post_px (dst, pf, px));
  }
}

// groups a round has to have in the ring: [*gl, *gh] for the rows [jr, jl]
GSTAMD_HD void fused_round_groups (const Fused420Params &p, int jr, int jl, int *gl, int *gh)
{
  *gl = p.vgroup[jr];
  const int h = p.vgroup[jl] + p.ngv - 1;
  *gh = h < p.n_groups - 1 ? h : p.n_groups - 1;
}

}  // namespace gstamd
