// video_planes.h - convert_scale_planes of the reference (video-converter.c:7757, setup_scale :7958-8200) for planar and
// semi-planar formats: every destination plane is produced from one source plane on its own - copied, halved / doubled
// by the video_orc_planar_chroma_* helpers (video-orc.orc:1271-1334) or sent through gst_video_scaler_2d as 1-byte
// (GRAY8), 2-byte (the UV plane of the NV12 family) or 3-byte (RGB / BGR) pixels; a packed 4:2:2 line is scaled
// vertically as plain bytes.  The scaler arithmetic is the 4 x u8 one of video_device.h
// applied to the bytes the plane has (hscale_px / vscale_px on words whose upper bytes are zero).
#pragma once
#include <algorithm>
#include "video_device.h"
#include "video_scale_fast.h"

namespace gstamd {

struct SrcPlane {
  const uint8_t *p;
  int stride;
  int n;                // bytes per pixel: 1, 2 or 3 (RGB / BGR)
  int pairs;            // n == 2 and every pixel sits on an even address: one 16-bit load per pixel
  GSTAMD_HD uint32_t at (int x, int y) const
  {
    const uint8_t *q = p + (size_t) y * stride + (size_t) x * n;
    if (pairs)
      return *(const uint16_t *) q;
    uint32_t v = q[0];
    if (n >= 2)
      v |= (uint32_t) q[1] << 8;
    if (n >= 3)
      v |= (uint32_t) q[2] << 16;
    return v;
  }
};

struct DstPlane {
  uint8_t *p;
  int stride;
  int n;
  GSTAMD_HD void put (int x, int y, uint32_t px) const
  {
    uint8_t *q = p + (size_t) y * stride + (size_t) x * n;
    q[0] = (uint8_t) px;
    if (n >= 2)
      q[1] = (uint8_t) (px >> 8);
    if (n >= 3)
      q[2] = (uint8_t) (px >> 16);
  }
};

GSTAMD_HD uint32_t avgub (uint32_t a, uint32_t b) { return (a + b + 1) >> 1; }

// the pass-free plane kinds (n == 1): one lane = one output byte
GSTAMD_HD void plane_simple_body (int kind, const SrcPlane &s, const DstPlane &d, int ow, int oh, int x, int y)
{
  if (x >= ow || y >= oh)
    return;
  uint32_t v;
  switch (kind) {
    case PLANE_H_HALVE:      /* video_orc_planar_chroma_444_422 */
      v = avgub (s.at (2 * x, y), s.at (2 * x + 1, y));
      break;
    case PLANE_H_DOUBLE:     /* video_orc_planar_chroma_422_444 */
      v = s.at (x >> 1, y);
      break;
    case PLANE_V_HALVE:      /* video_orc_planar_chroma_422_420 */
      v = avgub (s.at (x, 2 * y), s.at (x, 2 * y + 1));
      break;
    case PLANE_V_DOUBLE:     /* video_orc_planar_chroma_420_422 */
      v = s.at (x, y >> 1);
      break;
    case PLANE_HV_HALVE:     /* video_orc_planar_chroma_444_420: vertical averages first, then the pair */
      v = avgub (avgub (s.at (2 * x, 2 * y), s.at (2 * x, 2 * y + 1)), avgub (s.at (2 * x + 1, 2 * y), s.at (2 * x + 1, 2 * y + 1)));
      break;
    case PLANE_HV_DOUBLE:    /* video_orc_planar_chroma_420_444 */
      v = s.at (x >> 1, y >> 1);
      break;
    case PLANE_FILL:         /* convert_plane_fill: the chroma plane of a destination whose source has none */
      v = 0x80;
      break;
    default:                 /* PLANE_COPY */
      v = s.at (x, y);
      break;
  }
  d.put (x, y, v);
}

GSTAMD_HD void plane_hscale_body (const SrcPlane &s, const ScaleDev &sd, const DstPlane &d, int ow, int rows, int x, int y)
{
  if (x >= ow || y >= rows)
    return;
  const RowOfSrc<SrcPlane> row = {s, y};
  d.put (x, y, hscale_px (row, sd, x));
}

GSTAMD_HD void plane_vscale_body (const SrcPlane &s, const ScaleDev &sd, const DstPlane &d, int width, int oh, int x, int y)
{
  if (x >= width || y >= oh)
    return;
  d.put (x, y, vscale_px (s, sd, x, y));
}

// ---- one launch per frame: every plane, both passes ---------------------------------------------------------------------------
// The passes above are one lane per output byte with the first pass's plane going through HBM, a frame is 2 .. 6 launches, and every lane
// walks a chain of dependent loads (offset table -> tap table -> pixels) - the kernels are latency-bound at full occupancy.
// k_plane_frame covers the frame with 64 x 16 output tiles of all its planes (PlaneJobs: where each plane's tiles start) and gives every
// tile ONE round of global loads: the source rectangle the tile's outputs depend on and the table rows of its output rows / columns go
// to LDS with plain coalesced copies (phase 0); the first pass runs from LDS into LDS (phase 1: the source rows the vertical pass
// needs, horizontally scaled - or the columns the horizontal pass needs, vertically scaled; the order and the per-pass rounding are the
// plan's, hscale_px / vscale_px the functions), the second from LDS to the plane (phase 2).  Plans whose tiles do not fit
// PLN_LDS_BYTES, and the merged packed-4:2:2 scalers, keep the passes above.
#define PLN_TW 64
#define PLN_TH 16
#define PLN_THREADS 256
#define PLN_LDS_BYTES 61440
#define PLN_MAX_JOBS 3
#define PLN_PHASES 3
#define PLN_STAGE_ILP 6

struct PlaneJob {
  int kind;             // PlaneKind
  SrcPlane s;
  DstPlane d;
  int iw, ih, ow, oh;
  int n_pass, h_first;
  ScaleDev pass[2];
  int tile0, tiles_x;   // first tile of the plane in the launch's grid, tiles per tile row
  int wide;             // destination rows and the tile's 4-pixel groups sit on whole words: 4 outputs per store
  int wide_src;         // source rows on 8 bytes: the pass-free kinds read 4 / 8 source bytes per lane (plane_simple4)
  int quad;             // 0: not for k_plane_quad; 1 + QUAD_4 / QUAD_8 / QUAD_16 / QUAD_S8: how (plane_job_quad)
  int dstep;            // plane_quad_dstep of the plan (k_plane_quad's eight-byte form only)
};

struct PlaneJobs {
  PlaneJob job[PLN_MAX_JOBS];
  int n;
};

// source indices [lo, hi) a pass reads for its outputs [o0, o1)
GSTAMD_HD void plane_pass_span (const ScaleDev &sd, bool horizontal, int o0, int o1, int *lo, int *hi)
{
  if (sd.kind == SCALE_2TAP && horizontal) {
    *lo = (o0 * sd.inc) >> 16;
    *hi = (((o1 - 1) * sd.inc) >> 16) + 2;
    return;
  }
  const int n = sd.kind == SCALE_NEAREST ? 1 : (sd.kind == SCALE_2TAP ? 2 : sd.n_taps);
  *lo = (int) sd.offset[o0];
  *hi = (int) sd.offset[o1 - 1] + n;
}

// the tile's source rectangle in LDS, bytes as they are in the plane
struct LdsBytes {
  const uint8_t *b;
  int x0, y0, pitch, n;
  GSTAMD_HD uint32_t at (int x, int y) const
  {
    const uint8_t *q = b + (y - y0) * pitch + (x - x0) * n;
    uint32_t v = q[0];
    if (n >= 2)
      v |= (uint32_t) q[1] << 8;
    if (n >= 3)
      v |= (uint32_t) q[2] << 16;
    return v;
  }
};

// the first pass's result in LDS, one word per pixel
struct LdsPlane {
  const uint32_t *w;
  int x0, y0, pitch;
  GSTAMD_HD uint32_t at (int x, int y) const { return w[(y - y0) * pitch + (x - x0)]; }
};

// what a tile keeps in LDS and where
struct PlaneTile {
  int x0, y0, x1, y1;           // outputs
  int sx0, sx1, sy0, sy1;       // source rectangle
  int pitch;                    // bytes per staged source row (a multiple of 4)
  int skew;                     // bytes of a staged row in front of pixel sx0 (the rows are staged from a 4-byte boundary when the plane allows)
  int raw_off, mid_off, tab_off[2];     // byte offsets: source bytes, first-pass words, the passes' table rows (offset words, then taps)
  int mid_pitch;                // words per row of the first pass's result
};

// Where pass k's table rows sit: [n offsets as words][n x n_taps taps as int16], for the pass's outputs [o0, o1)
GSTAMD_HD int plane_tab_bytes (const ScaleDev &sd, int n_out)
{
  const int taps = sd.kind == SCALE_NTAP ? sd.n_taps : (sd.kind == SCALE_2TAP ? 2 : 0);
  return (4 * n_out + 2 * n_out * taps + 3) & ~3;
}

GSTAMD_HD void plane_tile_geometry (const PlaneJob &J, int tile, PlaneTile &T)
{
  const int tx = tile % J.tiles_x, ty = tile / J.tiles_x;
  T.x0 = tx * PLN_TW, T.y0 = ty * PLN_TH;
  T.x1 = T.x0 + PLN_TW < J.ow ? T.x0 + PLN_TW : J.ow;
  T.y1 = T.y0 + PLN_TH < J.oh ? T.y0 + PLN_TH : J.oh;
  const ScaleDev &ph = J.pass[J.h_first ? 0 : 1], &pv = J.pass[J.h_first ? 1 : 0];
  plane_pass_span (ph, true, T.x0, T.x1, &T.sx0, &T.sx1);
  plane_pass_span (pv, false, T.y0, T.y1, &T.sy0, &T.sy1);
  T.sx1 = T.sx1 < J.iw ? T.sx1 : J.iw;                  /* a tap of weight 0 may point past the plane: staged as 0 */
  T.sy1 = T.sy1 < J.ih ? T.sy1 : J.ih;
  T.skew = J.wide_src ? (T.sx0 * J.s.n) & 3 : 0;
  T.pitch = (T.skew + (T.sx1 - T.sx0) * J.s.n + 7) & ~3;          /* + room for the one pixel past the row the 2-tap form reads */
  T.raw_off = 0;
  const int raw_rows = T.sy1 - T.sy0 + 1;               /* + one zero row for the same reason */
  T.mid_off = T.raw_off + raw_rows * T.pitch;
  T.mid_pitch = J.h_first ? PLN_TW : T.sx1 - T.sx0 + 1;
  const int mid_rows = J.h_first ? T.sy1 - T.sy0 + 1 : PLN_TH;
  T.tab_off[0] = T.mid_off + 4 * mid_rows * T.mid_pitch;
  const int n0 = J.h_first ? T.x1 - T.x0 : T.y1 - T.y0;
  T.tab_off[1] = T.tab_off[0] + plane_tab_bytes (J.pass[0], n0);
}

// pass k of the tile with its tables in LDS: the ScaleDev's pointers rebased so that the pass's own output indices keep working
GSTAMD_HD ScaleDev plane_lds_pass (const PlaneJob &J, const PlaneTile &T, const uint8_t *lds, int k)
{
  ScaleDev sd = J.pass[k];
  const bool horizontal = (k == 0) == (J.h_first != 0);
  const int o0 = horizontal ? T.x0 : T.y0, n_out = horizontal ? T.x1 - T.x0 : T.y1 - T.y0;
  const int taps = sd.kind == SCALE_NTAP ? sd.n_taps : (sd.kind == SCALE_2TAP ? 2 : 0);
  const uint32_t *off = (const uint32_t *) (lds + T.tab_off[k]);
  sd.offset = off - o0;
  sd.taps = (const int16_t *) (off + n_out) - (ptrdiff_t) o0 * taps;
  return sd;
}

GSTAMD_HD uint32_t avgub_w (uint32_t a, uint32_t b) { return (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu); }        /* avgub on four bytes */
GSTAMD_HD uint32_t even_bytes (uint32_t lo, uint32_t hi) { return (lo & 0xffu) | ((lo >> 8) & 0xff00u) | ((hi & 0xffu) << 16) | ((hi << 8) & 0xff000000u); }
GSTAMD_HD uint32_t odd_bytes (uint32_t lo, uint32_t hi) { return ((lo >> 8) & 0xffu) | ((lo >> 16) & 0xff00u) | ((hi << 8) & 0xff0000u) | (hi & 0xff000000u); }

// the pass-free plane kinds on four outputs of a 1-byte plane at once (x a multiple of 4, rows of both planes on 8 / 4 bytes): the same
// averages (video_orc_planar_chroma_*: avgub) on packed bytes
GSTAMD_HD bool plane_simple4 (const PlaneJob &J, int x, int y)
{
  if (J.s.n != 1)
    return false;
  const uint8_t *sp = J.s.p;
  const size_t st = (size_t) J.s.stride;
  uint32_t v;
  switch (J.kind) {
    case PLANE_FILL:
      v = 0x80808080u;
      break;
    case PLANE_COPY:
      v = *(const uint32_t *) (sp + (size_t) y * st + x);
      break;
    case PLANE_V_HALVE:
      v = avgub_w (*(const uint32_t *) (sp + (size_t) (2 * y) * st + x), *(const uint32_t *) (sp + (size_t) (2 * y + 1) * st + x));
      break;
    case PLANE_H_HALVE: {
      const uint2 a = *(const uint2 *) (sp + (size_t) y * st + 2 * x);
      v = avgub_w (even_bytes (a.x, a.y), odd_bytes (a.x, a.y));
      break;
    }
    case PLANE_HV_HALVE: {
      const uint2 a = *(const uint2 *) (sp + (size_t) (2 * y) * st + 2 * x), b = *(const uint2 *) (sp + (size_t) (2 * y + 1) * st + 2 * x);
      const uint32_t lo = avgub_w (a.x, b.x), hi = avgub_w (a.y, b.y);
      v = avgub_w (even_bytes (lo, hi), odd_bytes (lo, hi));
      break;
    }
    case PLANE_V_DOUBLE:
      v = *(const uint32_t *) (sp + (size_t) (y >> 1) * st + x);
      break;
    case PLANE_H_DOUBLE:
    case PLANE_HV_DOUBLE: {
      const uint32_t h = *(const uint16_t *) (sp + (size_t) (J.kind == PLANE_HV_DOUBLE ? y >> 1 : y) * st + (x >> 1));
      v = (h & 0xffu) * 0x0101u | ((h >> 8) * 0x0101u) << 16;
      break;
    }
    default:
      return false;
  }
  *(uint32_t *) (J.d.p + (size_t) y * J.d.stride + x) = v;
  return true;
}

// four outputs of a row (or what is left of the row) to the plane
GSTAMD_HD void plane_put4 (const PlaneJob &J, int x, int y, int x1, const uint32_t *px)
{
  if (J.wide && x + 4 <= x1) {
    uint8_t *q = J.d.p + (size_t) y * J.d.stride + (size_t) x * J.d.n;
    if (J.d.n == 1) {
      *(uint32_t *) q = (px[0] & 0xffu) | ((px[1] & 0xffu) << 8) | ((px[2] & 0xffu) << 16) | (px[3] << 24);
      return;
    }
    if (J.d.n == 2) {
      uint2 o;
      o.x = (px[0] & 0xffffu) | (px[1] << 16), o.y = (px[2] & 0xffffu) | (px[3] << 16);
      *(uint2 *) q = o;
      return;
    }
  }
  for (int i = 0; i < 4 && x + i < x1; i++)
    J.d.put (x + i, y, px[i]);
}

GSTAMD_VP bool plane_small_kind (int k) { return k == SCALE_NEAREST || k == SCALE_2TAP; }

// scale2x2_px in the packed 16-bit lane arithmetic of the 4-byte scalers (h2tap_eo / v2tap_pk, video_scale_fast.h: the same integers as
// hscale_px / v2tap_px - scale2x2_tile_lane is held to them by the packed-format tests); a plane's pixel is 1 .. 3 bytes in a zero-extended word
GSTAMD_HD uint32_t plane_scale2x2_pk (const SrcPlane &s, const ScaleDev &sh, const ScaleDev &sv, int h_first, int x, int y)
{
  const bool v2 = sv.kind == SCALE_2TAP, h2 = sh.kind == SCALE_2TAP;
  const int ya = (int) sv.offset[y];
  const uint32_t p1s = v2 ? ((uint32_t) (uint16_t) sv.taps[(size_t) y * 2 + 1]) * 0x00010001u : 0u;
  uint32_t e, o;
  if (h2) {
    const int tmp = x * sh.inc;
    const int idx = tmp >> 16;
    const uint32_t fr = (uint32_t) (tmp >> 8) & 0xffu;
    if (!v2) {
      h2tap_eo (s.at (idx, ya), s.at (idx + 1, ya), fr, e, o);
    } else if (h_first) {
      uint32_t e2, o2;
      h2tap_eo (s.at (idx, ya), s.at (idx + 1, ya), fr, e, o);
      h2tap_eo (s.at (idx, ya + 1), s.at (idx + 1, ya + 1), fr, e2, o2);
      e = v2tap_pk (e, e2, p1s);
      o = v2tap_pk (o, o2, p1s);
    } else {
      const uint32_t a1 = s.at (idx, ya), a2 = s.at (idx, ya + 1), b1 = s.at (idx + 1, ya), b2 = s.at (idx + 1, ya + 1);
      const uint32_t ae = v2tap_pk (a1 & 0x00ff00ffu, a2 & 0x00ff00ffu, p1s), ao = v2tap_pk (pk_shr<8> (a1), pk_shr<8> (a2), p1s);
      const uint32_t be = v2tap_pk (b1 & 0x00ff00ffu, b2 & 0x00ff00ffu, p1s), bo = v2tap_pk (pk_shr<8> (b1), pk_shr<8> (b2), p1s);
      const uint32_t nf = 256u - fr;
      e = pk_shr<8> (umul24 (ae, nf) + umul24 (be, fr));
      o = pk_shr<8> (umul24 (ao, nf) + umul24 (bo, fr));
    }
  } else {
    const uint32_t a1 = s.at ((int) sh.offset[x], ya);
    e = a1 & 0x00ff00ffu;
    o = pk_shr<8> (a1);
    if (v2) {
      const uint32_t a2 = s.at ((int) sh.offset[x], ya + 1);
      e = v2tap_pk (e, a2 & 0x00ff00ffu, p1s);
      o = v2tap_pk (o, pk_shr<8> (a2), p1s);
    }
  }
  return e | (o << 8);
}

// planes that need nothing staged: pass-free kinds, one pass, or nearest / 2-tap in both directions.  One call per lane: outputs
// x .. x + 3 of a row of the tile.
// a pass that reads at most two source pixels per output: nearest, the 2-tap functions, or an N-tap filter of two taps (what `linear` with max-taps 2
// gives the planes the reference's 2-tap functions do not serve - the UV plane of NV12)
GSTAMD_VP bool plane_short_pass (const PlaneJob &J, int q) { return plane_small_kind (J.pass[q].kind) || (J.pass[q].kind == SCALE_NTAP && J.pass[q].n_taps <= 2); }

GSTAMD_VP bool plane_job_is_direct (const PlaneJob &J)
{
  return J.kind != PLANE_SCALE || J.n_pass < 2 || (plane_short_pass (J, 0) && plane_short_pass (J, 1));
}

// the two passes composed per pixel on the plane itself (first pass's result = the clamped byte the reference keeps in its temporary line)
struct PlaneHRows {
  SrcPlane s;
  const ScaleDev *sh;
  GSTAMD_HD uint32_t at (int x, int y) const
  {
    const RowOfSrc<SrcPlane> row = {s, y};
    return hscale_px (row, *sh, x);
  }
};

struct PlaneVRow {
  SrcPlane s;
  const ScaleDev *sv;
  int y;
  GSTAMD_HD uint32_t at (int x) const { return vscale_px (s, *sv, x, y); }
};

GSTAMD_HD void plane_direct_body (const PlaneJob &J, int tile, int tid)
{
  const int tx = tile % J.tiles_x, ty = tile / J.tiles_x;
  const int x0 = tx * PLN_TW, y0 = ty * PLN_TH;
  const int x1 = x0 + PLN_TW < J.ow ? x0 + PLN_TW : J.ow, y1 = y0 + PLN_TH < J.oh ? y0 + PLN_TH : J.oh;
  const int x = x0 + 4 * (tid % (PLN_TW / 4)), y = y0 + tid / (PLN_TW / 4);
  if (y >= y1 || x >= x1)
    return;
  if (J.kind != PLANE_SCALE) {
    if (J.wide && J.wide_src && x + 4 <= x1 && plane_simple4 (J, x, y))
      return;
    for (int i = 0; i < 4 && x + i < x1; i++)
      plane_simple_body (J.kind, J.s, J.d, J.ow, J.oh, x + i, y);
    return;
  }
  uint32_t px[4] = {0, 0, 0, 0};
  if (J.n_pass == 2) {
    /* nearest / 2-tap in both directions (the elements' default method): an output is a function of at most four source pixels - straight
       from memory, no staging, no barrier (scale2x2_px: the two passes with the first one's rounding in between, in the plan's order) */
    const ScaleDev &sh = J.pass[J.h_first ? 0 : 1], &sv = J.pass[J.h_first ? 1 : 0];
    if (plane_small_kind (sh.kind) && plane_small_kind (sv.kind)) {
      for (int i = 0; i < 4 && x + i < x1; i++)
        px[i] = plane_scale2x2_pk (J.s, sh, sv, J.h_first, x + i, y);
    } else if (J.h_first) {          /* two-tap N-tap passes: V (H (plane)) or H (V (plane)) per pixel, at most four source pixels each */
      const PlaneHRows rows = {J.s, &sh};
      for (int i = 0; i < 4 && x + i < x1; i++)
        px[i] = vscale_px (rows, sv, x + i, y);
    } else {
      const PlaneVRow row = {J.s, &sv, y};
      for (int i = 0; i < 4 && x + i < x1; i++)
        px[i] = hscale_px (row, sh, x + i);
    }
  } else {
    const RowOfSrc<SrcPlane> row = {J.s, y};
    for (int i = 0; i < 4 && x + i < x1; i++)
      px[i] = J.h_first ? hscale_px (row, J.pass[0], x + i) : vscale_px (J.s, J.pass[0], x + i, y);
  }
  plane_put4 (J, x, y, x1, px);
}

// ---- k_plane_quad: planes of two short passes, 4 or 8 output bytes per lane from 8- / 16-byte windows; pass-free planes 16 per lane ---------
// plane_direct_body asks memory for every source byte on its own (4 outputs x 4 source pixels = 16 byte loads per lane, plus the table
// rows): at 4K -> 1080p the kernel issues ~20 load instructions per wave and the chip's vector-memory issue rate (one wave instruction
// per ~37 clocks and CU, profiles/r04) bounds it at 6.5 us per NV12 frame against 1.9 us of traffic.  When both passes read at most two
// source pixels per output and B consecutive output BYTES of a row (B = 4 or 8: pixels of a one-byte plane, half as many of a two-byte
// one) depend on at most 2 B source bytes per row - any ratio up to 2:1 shrinking, every enlargement - a lane loads that window of its
// two source rows with one (unaligned) load each, picks an output's two taps with v_perm_b32 into the halves of a register, and a pass
// is a v_dot2 on it: H = (a * w0 + b * w1 + rnd) >> shift with (256 - f, f, 0, 8) for the 2-tap function (ldreslinl), (256, 0, 0, 8)
// for nearest, (t0, t1, 32, 6) and a clamp for a two-tap N-tap filter; V likewise (the 2-tap V keeps its own rounding: v2tap_px).  The
// passes run in the plan's order with the first one's result as the byte the reference keeps in its temporary line.
// A wave works on `rows` consecutive output rows of its 64 lanes' columns: what depends on the column only - source indices, weight
// pairs, the v_perm selectors (for the 16-byte window two per output: a | b << 16 = perm (w1, w0, selA) | perm (w3, w2, selB), selector
// 0x0c yields zero for the half a tap is not in) - is set up once; the vertical pass's table row is wave-uniform (scalar loads).  The
// last lane of a row whose group would cross the row's end moves back to end with it (it recomputes bytes of its neighbour: same
// values).  The pass-free kinds (the 2:1 luma next to a scaled UV plane) go 16 output bytes per lane.  A frame list rebases the planes
// where a pointer is formed (ds, dd).
struct QuadGrid {
  int n[PLN_MAX_JOBS];           // workgroups of each job (jobs in PlaneJobs order), 0: none
  int bx[PLN_MAX_JOBS];          // workgroups per row of workgroups (64 lanes x 4 waves each)
  int rows[PLN_MAX_JOBS];        // rows a wave walks
  int nt;                        // bit 0: nontemporal loads, bit 1: nontemporal stores
};

// workgroup b of the launch -> job and workgroup within the job: plane after plane.  (Alternating the planes' workgroups along the grid in
// proportion to their counts - a plane that waits on dependent loads next to one that streams - measured the same or worse: 30.5 against
// 29.6 us per list of eight 4K NV12 frames, profiles/r04/f8scale_variants.log.)
GSTAMD_HD void quad_grid_find (const QuadGrid &g, int b, int *job, int *local)
{
  for (int k = 0; k < PLN_MAX_JOBS - 1; k++) {
    if (b < g.n[k]) {
      *job = k, *local = b;
      return;
    }
    b -= g.n[k];
  }
  *job = PLN_MAX_JOBS - 1, *local = b;
}

enum { QUAD_4 = 0, QUAD_8 = 1, QUAD_16 = 2, QUAD_S8 = 3 };           // PlaneJob::quad - 1: two-pass planes 4 / 8 output bytes per lane and row, pass-free ones 16 / 8
GSTAMD_VP int quad_mode_bytes (int mode) { return mode == QUAD_S8 ? 8 : 4 << mode; }

// host: may the plane go this way with B = `bytes` output bytes per lane?  (table contents: spans of the groups, tap magnitudes)
inline bool plane_quad_ok (const PlanePlan &pp, int bytes = 4)
{
  if (pp.kind != PLANE_SCALE || pp.passes.size () != 2 || (pp.n_elems != 1 && pp.n_elems != 2 && !(pp.n_elems == 4 && bytes == 8)) || pp.iw * pp.n_elems < 2 * bytes ||
      pp.ow * pp.n_elems < bytes || pp.ih < 1)
    return false;
  const ScalePass *ph = pp.passes[0].horizontal ? &pp.passes[0] : &pp.passes[1], *pv = pp.passes[0].horizontal ? &pp.passes[1] : &pp.passes[0];
  if (!ph->horizontal || pv->horizontal)
    return false;
  for (const ScalePass *q : {ph, pv}) {
    if (q->merged != 0)
      return false;
    if (q->kind == SCALE_NTAP) {
      if (q->n_taps != 2)
        return false;
      for (size_t i = 0; i + 1 < q->taps.size (); i += 2)
        if (abs ((int) q->taps[i]) + abs ((int) q->taps[i + 1]) > 128)          /* no 16-bit wrap in a * t0 + b * t1 + 32 */
          return false;
    } else if (q->kind != SCALE_NEAREST && q->kind != SCALE_2TAP) {
      return false;
    }
  }
  /* the source pixels of every group of `per` consecutive outputs - a group starts at ANY pixel (the row's last one is moved back): first
     tap of the first .. second tap of the last, within 2 * bytes */
  const int per = bytes / pp.n_elems;
  for (int x = 0; x + per <= pp.ow; x++) {
    int lo, hi;
    if (ph->kind == SCALE_2TAP) {
      lo = (x * ph->inc) >> 16, hi = (((x + per - 1) * ph->inc) >> 16) + 1;
    } else {
      lo = (int) ph->offset[x], hi = (int) ph->offset[x + per - 1] + 1;
      for (int k = 1; k < per; k++)
        if ((int) ph->offset[x + k] < lo)
          return false;
    }
    if ((hi - lo + 1) * pp.n_elems > 2 * bytes)
      return false;
  }
  return true;
}

GSTAMD_HD int quad_dot2 (uint32_t pair, uint32_t w, int acc)     // lo(pair) * lo(w) + hi(pair) * hi(w) + acc, 16-bit signed halves
{
#ifdef __HIPCC__
  typedef short s2 __attribute__ ((ext_vector_type (2)));
  /* clamp = true keeps the three-operand v_dot2_i32_i16 (its accumulator may be a scalar register); without it the compiler picks v_dot2c, whose
   * accumulator is the destination, and loads it with a v_mov in front of every dot product.  The sums here are far from 2^31: it never acts. */
  return __builtin_amdgcn_sdot2 (__builtin_bit_cast (s2, pair), __builtin_bit_cast (s2, w), acc, true);
#else
  return (int) (int16_t) (pair & 0xffffu) * (int) (int16_t) (w & 0xffffu) + (int) (int16_t) (pair >> 16) * (int) (int16_t) (w >> 16) + acc;
#endif
}

// (the opaque register barrier keeps a preceding shift and this clamp apart: fused into v_ashr_pk_u8_i32 the second lane of a pair came back
// wrong on gfx950 - video_device.h lq_round)
GSTAMD_HD int quad_clamp255 (int v)
{
#ifdef __HIPCC__
  asm volatile ("" : "+v" (v));
#endif
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// (s1 * t0 + s2 * t1 + 32) >> 6, clamped, per 16-bit half (taps splat over both halves)
GSTAMD_HD uint32_t quad_vntap_pk (uint32_t s1, uint32_t s2, uint32_t t0s, uint32_t t1s)
{
#ifdef __HIPCC__
  typedef short s2v __attribute__ ((ext_vector_type (2)));
  s2v m = __builtin_bit_cast (s2v, pk_mad16 (s1, t0s, pk_mad16 (s2, t1s, 0x00200020u)));
  m = m >> (short) 6;
  m = __builtin_elementwise_min (__builtin_elementwise_max (m, (s2v) (short) 0), (s2v) (short) 255);
  return __builtin_bit_cast (uint32_t, m);
#else
  uint32_t r = 0;
  for (int h = 0; h < 2; h++) {
    const int a = (int) ((s1 >> (16 * h)) & 0xffffu), b = (int) ((s2 >> (16 * h)) & 0xffffu);
    const int acc = (int) (int16_t) (uint16_t) (a * (int) (int16_t) (t0s & 0xffffu) + b * (int) (int16_t) (t1s & 0xffffu) + 32);
    r |= (uint32_t) quad_clamp255 (acc >> 6) << (16 * h);
  }
  return r;
#endif
}

// W dwords from byte `off` of a plane row / to byte `off` of one, any alignment
template <int W>
struct QuadWords { uint32_t w[W]; };

template <int W>
GSTAMD_HD QuadWords<W> quad_load (const uint8_t *row, int off, int nt = 0)
{
  QuadWords<W> r;
#ifdef __HIPCC__
  typedef unsigned int uv __attribute__ ((ext_vector_type (W), aligned (1)));
  const uv v = nt ? __builtin_nontemporal_load ((const uv *) (row + off)) : *(const uv *) (row + off);
#pragma unroll
  for (int i = 0; i < W; i++)
    r.w[i] = v[i];
#else
  memcpy (r.w, row + off, 4 * W);
#endif
  return r;
}

template <int W>
GSTAMD_HD void quad_store (uint8_t *row, int off, const uint32_t *w, int nt = 0)
{
#ifdef __HIPCC__
  typedef unsigned int uv __attribute__ ((ext_vector_type (W), aligned (1)));
  uv v;
#pragma unroll
  for (int i = 0; i < W; i++)
    v[i] = w[i];
  if (nt)
    __builtin_nontemporal_store (v, (uv *) (row + off));
  else
    *(uv *) (row + off) = v;
#else
  memcpy (row + off, w, 4 * W);
#endif
}

// a source plane as a raw buffer (base, bytes): a window load takes a wave-uniform row offset plus the lane's byte offset, any alignment,
// and what lies past the plane's end reads as zero (per dword) - the row's last lane needs no special case
#ifdef __HIPCC__
typedef __amdgpu_buffer_rsrc_t quadplane_t;
#else
struct quadplane_t { const uint8_t *p; uint32_t bytes; };
#endif
GSTAMD_HD quadplane_t quad_plane (const uint8_t *base, uint32_t bytes)
{
#ifdef __HIPCC__
  return __builtin_amdgcn_make_buffer_rsrc ((void *) base, (short) 0, (int) bytes, 0x00020000);
#else
  quadplane_t pl = {base, bytes};
  return pl;
#endif
}
template <int W>
GSTAMD_HD QuadWords<W> quad_window (quadplane_t pl, uint32_t row_off, uint32_t off)
{
  QuadWords<W> r;
#ifdef __HIPCC__
  if constexpr (W == 4) {
    typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
    const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128 (pl, (int) off, (int) row_off, 0);
    r.w[0] = d.x, r.w[1] = d.y, r.w[2] = d.z, r.w[3] = d.w;
  } else {
    typedef uint32_t u32x2 __attribute__ ((ext_vector_type (2)));
    const u32x2 d = __builtin_amdgcn_raw_buffer_load_b64 (pl, (int) off, (int) row_off, 0);
    r.w[0] = d.x, r.w[1] = d.y;
  }
#else
  const unsigned long long at = (unsigned long long) row_off + off;
  for (int i = 0; i < W; i++) {
    r.w[i] = 0;
    if (at + 4 * i + 4 <= pl.bytes)
      memcpy (&r.w[i], pl.p + at + 4 * i, 4);
  }
#endif
  return r;
}

// selector for the window's bytes base .. base + 7: tap a (window byte da) into byte 0, tap b (db) into byte 2, zero elsewhere.  dd = da | db << 16;
// per half: t = d - base (wrapping), u = min (t, 8), selector byte = u + 4 * (u >> 3): 0 .. 7 stay, everything else becomes 0x0c - packed
// 16-bit instructions, no compares (v_cndmask behind a v_cmp runs at a fifth of the rate of plain VALU here, scripts/valubench.hip)
GSTAMD_HD uint32_t pk_min16 (uint32_t a, uint32_t b)
{
#ifdef __HIPCC__
  typedef unsigned short us2 __attribute__ ((ext_vector_type (2)));
  return __builtin_bit_cast (uint32_t, __builtin_elementwise_min (__builtin_bit_cast (us2, a), __builtin_bit_cast (us2, b)));
#else
  const uint32_t lo = (a & 0xffffu) < (b & 0xffffu) ? (a & 0xffffu) : (b & 0xffffu), hi = (a >> 16) < (b >> 16) ? (a >> 16) : (b >> 16);
  return lo | (hi << 16);
#endif
}

GSTAMD_HD uint32_t quad_sel (uint32_t dd, int base)
{
  const uint32_t u = pk_min16 (pk_sub16 (dd, (uint32_t) base * 0x00010001u), 0x00080008u);
  return pk_mad16 (pk_shr<3> (u), 0x00040004u, u) | 0x0c000c00u;
}

#ifndef QUAD_CH
#define QUAD_CH 2               // rows whose loads leave together
#endif
// N = bytes per pixel (1, 2), B = output bytes per lane and row (4, 8), DS = 0, or the plane's uniform distance in pixels between the first
// taps of neighbouring outputs (PlaneJob::dstep: 2 = a 2:1 reduction - the window bytes of an output are then known at compile time, no
// selectors to set up and one v_perm per tap pair).  lane: output bytes B * lane .. of rows y0 .. y0 + rows - 1
template <int N, int B, int DS, int MODE>
GSTAMD_HD void plane_quad_body (const PlaneJob &J, int lane, int y0, int rows, long long ds, long long dd, int nt)
{
  const int PER = B / N, HALVES = B / 4;          /* pixels per lane; 8-byte halves of the window */
  int x = lane * PER;
  if (x >= J.ow)
    return;
  x = x + PER > J.ow ? J.ow - PER : x;          /* the row's last group ends with the row */
  /* MODE >= 0: the passes' order and kinds as compile-time constants (quad_mode_of) - with run-time kinds every output byte of every row
     went through four wave-uniform branches, ~50 taken branches per wave and row: more clocks than the arithmetic */
  const bool hfirst = MODE < 0 ? J.h_first != 0 : (MODE & 1) != 0;
  const ScaleDev &sh = J.pass[hfirst ? 0 : 1], &sv = J.pass[hfirst ? 1 : 0];
  const int vkind = MODE < 0 ? sv.kind : ((MODE >> 1) & 3) == 0 ? (int) SCALE_NEAREST : ((MODE >> 1) & 3) == 1 ? (int) SCALE_2TAP : (int) SCALE_NTAP;
  int idx[PER];
  uint32_t w[PER];
  const bool h_ntap = MODE < 0 ? sh.kind == SCALE_NTAP : (MODE & 8) != 0;
  if (!h_ntap && sh.kind == SCALE_2TAP) {
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const int tmp = (x + k) * sh.inc;
      idx[k] = tmp >> 16;
      w[k] = ((uint32_t) (tmp >> 8) & 0xffu) * 0xffffu + 256u;           /* (256 - f) | f << 16 */
    }
  } else {
    /* the lane's table rows with one load each (PER consecutive words; the compiler leaves indexed loads one word at a time) */
    if (DS == 0) {
      const QuadWords<PER> to = quad_load<PER> ((const uint8_t *) sh.offset, 4 * x);
#pragma unroll
      for (int k = 0; k < PER; k++)
        idx[k] = (int) to.w[k];
    } else {
      idx[0] = (int) sh.offset[x];
    }
    QuadWords<PER> tw;
    if (h_ntap)
      tw = quad_load<PER> ((const uint8_t *) sh.taps, 4 * x);           /* an output's two taps as they lie in the table */
#pragma unroll
    for (int k = 0; k < PER; k++)
      w[k] = h_ntap ? tw.w[k] : 256u;                                   /* nearest: a * 256 >> 8 */
  }
  const int h_rnd = h_ntap ? 32 : 0, h_shift = h_ntap ? 6 : 8;
  /* the window: 2 B bytes from the first output's first tap; pulled back where it would leave the row (a word that straddles the plane's
     end reads as zero as a whole: the row's own bytes must not depend on what follows it) - a 2:1 plane's windows end with the row */
  uint32_t off = (uint32_t) idx[0] * N;
  int back = 0;
  if (DS == 0 && (int) off + 2 * B > J.iw * N)
    back = (int) off + 2 * B - J.iw * N, off = (uint32_t) (J.iw * N - 2 * B);
  uint32_t sel[HALVES][B];
  if (DS == 0) {
#pragma unroll
    for (int c = 0; c < B; c++) {         /* output byte c: component c % N of pixel c / N */
      const uint32_t d = (uint32_t) ((idx[c / N] - idx[0]) * N + (c % N) + back) * 0x00010001u + ((uint32_t) N << 16);
#pragma unroll
      for (int h = 0; h < HALVES; h++)
        sel[h][c] = quad_sel (d, 8 * h);
    }
  }
  const quadplane_t src = quad_plane (J.s.p + ds, (uint32_t) ((size_t) J.s.stride * (size_t) J.ih));
  uint8_t *dp = J.d.p + dd;
  for (int r0 = 0; r0 < rows && y0 + r0 < J.oh; r0 += QUAD_CH) {
    /* every load of the chunk's rows leaves before the first result is needed (the stores of a row and the loads of the next may
       alias as far as the compiler knows: it would not move them itself) */
    QuadWords<2 * HALVES> ra[QUAD_CH], rb[QUAD_CH];
    int vt0[QUAD_CH], vt1[QUAD_CH];
#pragma unroll
    for (int j = 0; j < QUAD_CH; j++) {
      const int y = y0 + r0 + j;
      if (r0 + j < rows && y < J.oh) {
        const int ya = (int) sv.offset[y], yb = vkind == SCALE_NEAREST ? ya : (ya + 1 < J.ih ? ya + 1 : J.ih - 1);
        vt0[j] = vkind == SCALE_NTAP ? (int) sv.taps[(size_t) y * 2] : 0, vt1[j] = vkind == SCALE_NEAREST ? 0 : (int) sv.taps[(size_t) y * 2 + 1];
#ifdef GSTAMD_TUNING
        if (nt & 4) {                   /* profiling builds (results WRONG): no window loads */
#pragma unroll
          for (int i = 0; i < 2 * HALVES; i++)
            ra[j].w[i] = rb[j].w[i] = (uint32_t) (lane + i);
          continue;
        }
#endif
        ra[j] = quad_window<2 * HALVES> (src, (uint32_t) ya * (uint32_t) J.s.stride, off);
        rb[j] = vkind == SCALE_NEAREST ? ra[j] : quad_window<2 * HALVES> (src, (uint32_t) yb * (uint32_t) J.s.stride, off);
      }
    }
#pragma unroll
    for (int j = 0; j < QUAD_CH; j++) {
      const int y = y0 + r0 + j;
      if (r0 + j < rows && y < J.oh) {
        uint32_t o[B];
#pragma unroll
        for (int c = 0; c < B; c++) {
          const int k = c / N;
          uint32_t pa, pb;                /* tap a | tap b << 16, rows ya and yb */
          if (DS == 0) {
            pa = bperm (ra[j].w[1], ra[j].w[0], sel[0][c]), pb = bperm (rb[j].w[1], rb[j].w[0], sel[0][c]);
            if (HALVES == 2) {
              pa |= bperm (ra[j].w[2 * HALVES - 1], ra[j].w[2 * HALVES - 2], sel[HALVES - 1][c]);
              pb |= bperm (rb[j].w[2 * HALVES - 1], rb[j].w[2 * HALVES - 2], sel[HALVES - 1][c]);
            }
          } else {
            /* taps at window bytes d and d + N, d = k DS N + c % N: both in the word pair (d / 4, d / 4 + 1) - or in one word */
            const int d = k * DS * N + (c % N), wi = d / 4 < 2 * HALVES - 1 ? d / 4 : 2 * HALVES - 2, dl = d - 4 * wi;
            const uint32_t cs = 0x0c000c00u | (uint32_t) dl | ((uint32_t) (dl + N) << 16);
            pa = bperm (ra[j].w[wi + 1], ra[j].w[wi], cs), pb = bperm (rb[j].w[wi + 1], rb[j].w[wi], cs);
          }
          int q;
          if (hfirst) {
            int ha = (quad_dot2 (pa, w[k], h_rnd)) >> h_shift, hb = (quad_dot2 (pb, w[k], h_rnd)) >> h_shift;
            if (h_ntap)
              ha = quad_clamp255 (ha), hb = quad_clamp255 (hb);
            if (vkind == SCALE_NEAREST)
              q = ha;
            else if (vkind == SCALE_2TAP)
              q = (((((hb - ha) * vt1[j] + 128) >> 8) & 0xff) + ha) & 0xff;                        /* v2tap_px */
            else
              q = quad_clamp255 ((int) (int16_t) (uint16_t) (ha * vt0[j] + hb * vt1[j] + 32) >> 6);
          } else {
            uint32_t v;
            if (vkind == SCALE_NEAREST)
              v = pa;
            else if (vkind == SCALE_2TAP)
              v = v2tap_pk (pa, pb, (uint32_t) (uint16_t) vt1[j] * 0x00010001u);
            else
              v = quad_vntap_pk (pa, pb, (uint32_t) (uint16_t) vt0[j] * 0x00010001u, (uint32_t) (uint16_t) vt1[j] * 0x00010001u);
            q = quad_dot2 (v, w[k], h_rnd) >> h_shift;
            if (h_ntap)
              q = quad_clamp255 (q);
          }
          o[c] = (uint32_t) q;
        }
        uint32_t st[HALVES];
#pragma unroll
        for (int h = 0; h < HALVES; h++)
          st[h] = (o[4 * h] | (o[4 * h + 1] << 8)) | ((o[4 * h + 2] | (o[4 * h + 3] << 8)) << 16);
#ifdef GSTAMD_TUNING
        if ((nt & 8) && st[0] != 0x12345678u)          /* profiling builds (results WRONG): no stores */
          continue;
#endif
        quad_store<HALVES> (dp + (size_t) y * J.d.stride, x * N, st, nt & 2);
      }
    }
  }
}

// host: PlaneJob::dstep of a plane that goes through plane_quad_body - 2 when every output's first horizontal tap lies two pixels after its
// left neighbour's (a 2:1 reduction), else 0
inline int plane_quad_dstep (const PlanePlan &pp)
{
  if (pp.kind != PLANE_SCALE || pp.passes.size () != 2)
    return 0;
  const ScalePass *ph = pp.passes[0].horizontal ? &pp.passes[0] : &pp.passes[1];
  for (int x = 0; x + 1 < pp.ow; x++) {
    const int a = ph->kind == SCALE_2TAP ? (x * ph->inc) >> 16 : (int) ph->offset[x], b = ph->kind == SCALE_2TAP ? ((x + 1) * ph->inc) >> 16 : (int) ph->offset[x + 1];
    if (b - a != 2)
      return 0;
  }
  const int last = ph->kind == SCALE_2TAP ? ((pp.ow - 1) * ph->inc) >> 16 : (int) ph->offset[pp.ow - 1];
  return last + 2 <= pp.iw ? 2 : 0;           /* the last output's window ends inside the row */
}

// the pass-free kinds of a one-byte plane on sixteen outputs: copy, and the 2:1 averages in one or both directions (video_orc_planar_chroma_*:
// avgub).  hs / vs: the kind halves horizontally / vertically.  The source bytes of a row first (s16_load: up to four 16-byte loads), then
// the averages and the store (s16_put).  Plain locals and uniform branches, no switch writing into a struct (that ends in scratch).
GSTAMD_VP bool plane_simple16_kind (int kind) { return kind == PLANE_COPY || kind == PLANE_V_HALVE || kind == PLANE_H_HALVE || kind == PLANE_HV_HALVE; }

struct S16Row { QuadWords<4> a0, a1, b0, b1; };

GSTAMD_HD void s16_load (const uint8_t *sp, size_t st, int hs, int vs, int x, int y, S16Row &q, int nt)
{
  const uint8_t *r0 = sp + (size_t) (y << vs) * st;
  q.a0 = quad_load<4> (r0, x << hs, nt);
  q.a1 = hs ? quad_load<4> (r0, 2 * x + 16, nt) : q.a0;
  q.b0 = vs ? quad_load<4> (r0 + st, x << hs, nt) : q.a0;
  q.b1 = hs && vs ? quad_load<4> (r0 + st, 2 * x + 16, nt) : q.a0;
}

GSTAMD_HD uint32_t avg_pairs (uint32_t lo, uint32_t hi) { return avgub_w (bperm (hi, lo, 0x06040200u), bperm (hi, lo, 0x07050301u)); }     /* avgub of the byte pairs of 8 bytes */

GSTAMD_HD void s16_put (uint8_t *dp, size_t dst, int hs, int vs, int x, int y, const S16Row &q, int nt)
{
  uint32_t m0[4], m1[4], v[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    m0[i] = vs ? avgub_w (q.a0.w[i], q.b0.w[i]) : q.a0.w[i];
    m1[i] = vs ? avgub_w (q.a1.w[i], q.b1.w[i]) : q.a1.w[i];
  }
  v[0] = hs ? avg_pairs (m0[0], m0[1]) : m0[0];
  v[1] = hs ? avg_pairs (m0[2], m0[3]) : m0[1];
  v[2] = hs ? avg_pairs (m1[0], m1[1]) : m0[2];
  v[3] = hs ? avg_pairs (m1[2], m1[3]) : m0[3];
  quad_store<4> (dp + (size_t) y * dst, x, v, nt);
}

GSTAMD_HD void plane_simple16_body (const PlaneJob &J, int lane, int y0, int rows, long long ds, long long dd, int nt)
{
  int x = 16 * lane;
  if (x >= J.ow)
    return;
  x = x + 16 > J.ow ? J.ow - 16 : x;
  const uint8_t *sp = J.s.p + ds;
  uint8_t *dp = J.d.p + dd;
  const int hs = J.kind == PLANE_H_HALVE || J.kind == PLANE_HV_HALVE ? 1 : 0, vs = J.kind == PLANE_V_HALVE || J.kind == PLANE_HV_HALVE ? 1 : 0;
  const size_t st = (size_t) J.s.stride, dst = (size_t) J.d.stride;
  for (int r0 = 0; r0 < rows && y0 + r0 < J.oh; r0 += 2) {          /* two rows' loads, then their averages and stores */
    S16Row qa, qb;
    const bool two = r0 + 1 < rows && y0 + r0 + 1 < J.oh;
    s16_load (sp, st, hs, vs, x, y0 + r0, qa, nt & 1);
    if (two)
      s16_load (sp, st, hs, vs, x, y0 + r0 + 1, qb, nt & 1);
    s16_put (dp, dst, hs, vs, x, y0 + r0, qa, nt & 2);
    if (two)
      s16_put (dp, dst, hs, vs, x, y0 + r0 + 1, qb, nt & 2);
  }
}

// the same on eight outputs per lane: a horizontally halving kind then reads 16 bytes per lane and row, lane after lane without gaps
GSTAMD_HD void plane_simple8_body (const PlaneJob &J, int lane, int y0, int rows, long long ds, long long dd, int nt)
{
  int x = 8 * lane;
  if (x >= J.ow)
    return;
  x = x + 8 > J.ow ? J.ow - 8 : x;
  const uint8_t *sp = J.s.p + ds;
  uint8_t *dp = J.d.p + dd;
  const int hs = J.kind == PLANE_H_HALVE || J.kind == PLANE_HV_HALVE ? 1 : 0, vs = J.kind == PLANE_V_HALVE || J.kind == PLANE_HV_HALVE ? 1 : 0;
  const size_t st = (size_t) J.s.stride;
  for (int r = 0; r < rows && y0 + r < J.oh; r++) {
    const int y = y0 + r;
    const uint8_t *r0 = sp + (size_t) (y << vs) * st;
    uint32_t a[4], b[4], v[2];
    if (hs) {
      const QuadWords<4> qa = quad_load<4> (r0, 2 * x, nt & 1), qb = vs ? quad_load<4> (r0 + st, 2 * x, nt & 1) : qa;
#pragma unroll
      for (int i = 0; i < 4; i++)
        a[i] = qa.w[i], b[i] = qb.w[i];
    } else {
      const QuadWords<2> qa = quad_load<2> (r0, x, nt & 1), qb = vs ? quad_load<2> (r0 + st, x, nt & 1) : qa;
      a[0] = qa.w[0], a[1] = qa.w[1], b[0] = qb.w[0], b[1] = qb.w[1];
      a[2] = a[3] = b[2] = b[3] = 0;
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
      a[i] = vs ? avgub_w (a[i], b[i]) : a[i];
    v[0] = hs ? avg_pairs (a[0], a[1]) : a[0];
    v[1] = hs ? avg_pairs (a[2], a[3]) : a[1];
    quad_store<2> (dp + (size_t) y * J.d.stride, x, v, nt & 2);
  }
}

// host: PlaneJob::quad of a job whose pointers are set (quad_ok / oct_ok: plane_quad_ok (pp, 4) / (pp, 8) of its plan); max_mode caps the form
inline int plane_job_quad (const PlaneJob &J, bool quad_ok, bool oct_ok, int max_mode = 2)
{
  if (J.kind != PLANE_SCALE)
    return J.s.n == 1 && plane_simple16_kind (J.kind) && J.ow >= 16 && max_mode >= 2 ? 1 + (max_mode >= 3 ? QUAD_S8 : QUAD_16) : 0;
  if (oct_ok && (max_mode >= 1 || J.s.n == 4))
    return 1 + QUAD_8;
  return quad_ok && J.s.n != 4 ? 1 + QUAD_4 : 0;
}

// host: the reference's plane scaler on a 4-byte packed format (convert_scale_planes: the same 8-bit format on both sides, no colour step - the
// bytes go through raw) with two short passes is one plane of four-byte pixels to k_plane_quad: *pp = that plane's plan
inline bool plane_raw4_plan (const VideoPlan &p, PlanePlan *pp)
{
  if (p.ref_fastpath != "convert_scale_planes" || p.plane_mode || p.gamma.on || p.front.kind != UNPACK_PACKED4 || p.front.hi_depth != 0 || p.passes.size () != 2 ||
      p.dither.on || p.out_planar || p.matrix.kind != MATRIX_NONE || p.post.matrix.kind != MATRIX_NONE || p.post.alpha_kind != ALPHA_NONE)
    return false;
  pp->kind = PLANE_SCALE;
  pp->src_plane = pp->dst_plane = 0;
  pp->n_elems = 4;
  pp->iw = p.front.width, pp->ih = p.front.height, pp->ow = p.out_info.width, pp->oh = p.out_info.height;
  pp->passes = p.passes;
  /* enlargements stay with k_bilinear4_rows (four outputs x four rows per lane share their source pixels; here a lane makes two pixels from two
     16-byte windows: 1080p -> 4K 40 us against 27) */
  if ((long long) pp->ow * pp->oh > (long long) pp->iw * pp->ih)
    return false;
  return plane_quad_ok (*pp, 8);
}

// host: a 4-byte 8-bit packed source shrunk by two short passes into a planar / semi-planar 8-bit destination (BGRA 4K -> NV12 1080p: screen
// capture into an encoder).  The chain is unpack (a byte permutation) -> scale -> matrix -> chroma down -> pack; the scaler works per byte, so
// it runs on the RAW pixels (k_plane_quad, four-byte pixels) into the plan's AYUV-sized image and the UNSCALED kernels - k_encode420 or
// k_convert_pack, with the source format at the destination's size - finish from there: two launches, 12 + 4 us instead of the one-lane-per-pixel
// scaler's 20 + the packer's 9.  *pp = the scaler's plane; *enc420 = the block encoder applies (planner.cpp's fast_enc420 test without the
// "no scaler" clause).
inline bool plane_raw4_pack_plan (const VideoPlan &p, PlanePlan *pp, bool *enc420)
{
  if (!p.out_planar || p.plane_mode || p.gamma.on || p.deep16 || p.deep_out || p.front.kind != UNPACK_PACKED4 || p.front.hi_depth != 0 || p.passes.size () != 2 ||
      p.matrix_before_scale || p.pack.virtual_line || p.dither.on || p.post.pack_pos[0] != 0 || p.post.pack_pos[1] != 1 || p.post.pack_pos[2] != 2 || p.post.pack_pos[3] != 3)
    return false;
  if (p.pack.dither.on && p.pack.dither.method != GSTAMD_DITHER_NONE && p.pack.dither.method != GSTAMD_DITHER_BAYER)
    return false;
  pp->kind = PLANE_SCALE;
  pp->src_plane = pp->dst_plane = 0;
  pp->n_elems = 4;
  pp->iw = p.front.width, pp->ih = p.front.height, pp->ow = p.out_info.width, pp->oh = p.out_info.height;
  pp->passes = p.passes;
  if ((long long) pp->ow * pp->oh > (long long) pp->iw * pp->ih || !plane_quad_ok (*pp, 8))
    return false;
  bool fits = kind_has_planes (p.fout->kind) && p.fout->w_sub == 1 && p.fout->h_sub == 1 && !p.fin->yuv && p.matrix.kind == MATRIX_TABLE && (pp->ow % 4) == 0 &&
      !p.pack.dither.on;
  for (int k = 0; k < 3; k++)
    for (int j = 0; j < 3; j++)
      fits = fits && abs (p.matrix.im[k][j]) <= 255;
  *enc420 = fits;
  return true;
}

GSTAMD_HD bool sh_is_2tap (const PlaneJob &J) { return J.pass[J.h_first ? 0 : 1].kind == SCALE_2TAP; }

// the compile-time form of a job's passes for plane_quad_body: bit 0 horizontal first, bits 1-2 the vertical kind (0 nearest, 1 the 2-tap
// function, 2 a two-tap N-tap filter), bit 3 the horizontal pass is an N-tap filter.  The forms the elements' methods produce are
// instantiated (bilinear: both orders with the 2-tap functions or with N-tap pairs); anything else runs the generic body (-1).
GSTAMD_HD int quad_mode_of (const PlaneJob &J)
{
  const ScaleDev &sh = J.pass[J.h_first ? 0 : 1], &sv = J.pass[J.h_first ? 1 : 0];
  const int vk = sv.kind == SCALE_NEAREST ? 0 : sv.kind == SCALE_2TAP ? 1 : 2;
  return (J.h_first ? 1 : 0) | (vk << 1) | (sh.kind == SCALE_NTAP ? 8 : 0);
}

template <int N, int B, int DS>
GSTAMD_HD void plane_quad_modes (const PlaneJob &J, int lane, int y0, int rows, long long ds, long long dd, int nt)
{
  const int mode = quad_mode_of (J);
  if (B == 8 && mode == (0 | (2 << 1) | 8))                 /* V then H, N-tap pairs */
    plane_quad_body<N, B, DS, (0 | (2 << 1) | 8)> (J, lane, y0, rows, ds, dd, nt);
  else if (B == 8 && mode == (1 | (2 << 1) | 8))            /* H then V, N-tap pairs */
    plane_quad_body<N, B, DS, (1 | (2 << 1) | 8)> (J, lane, y0, rows, ds, dd, nt);
  else if (B == 8 && mode == (0 | (1 << 1) | 8))            /* V 2-tap function then H N-tap pair: the UV plane of NV12 under `bilinear` */
    plane_quad_body<N, B, DS, (0 | (1 << 1) | 8)> (J, lane, y0, rows, ds, dd, nt);
  else if (B == 8 && mode == (1 | (1 << 1) | 8))            /* H N-tap pair then V 2-tap function */
    plane_quad_body<N, B, DS, (1 | (1 << 1) | 8)> (J, lane, y0, rows, ds, dd, nt);
  else if (B == 8 && mode == (0 | (1 << 1) | 0) && sh_is_2tap (J))          /* V then H, the 2-tap functions */
    plane_quad_body<N, B, DS, (0 | (1 << 1) | 0)> (J, lane, y0, rows, ds, dd, nt);
  else if (B == 8 && mode == (1 | (1 << 1) | 0) && sh_is_2tap (J))          /* H then V, the 2-tap functions */
    plane_quad_body<N, B, DS, (1 | (1 << 1) | 0)> (J, lane, y0, rows, ds, dd, nt);
  else
    plane_quad_body<N, B, DS, -1> (J, lane, y0, rows, ds, dd, nt);
}

// what a lane of k_plane_quad does for a job
GSTAMD_HD void plane_rows_body (const PlaneJob &J, int lane, int y0, int rows, long long ds = 0, long long dd = 0, int nt = 0)
{
  const int mode = J.quad - 1;
  if (mode == QUAD_16)
    plane_simple16_body (J, lane, y0, rows, ds, dd, nt);
  else if (mode == QUAD_S8)
    plane_simple8_body (J, lane, y0, rows, ds, dd, nt);
  else if (mode == QUAD_8 && J.s.n == 4 && J.dstep == 2)
    plane_quad_modes<4, 8, 2> (J, lane, y0, rows, ds, dd, nt);
  else if (mode == QUAD_8 && J.s.n == 4)
    plane_quad_modes<4, 8, 0> (J, lane, y0, rows, ds, dd, nt);
  else if (mode == QUAD_8 && J.s.n == 1 && J.dstep == 2)
    plane_quad_modes<1, 8, 2> (J, lane, y0, rows, ds, dd, nt);
  else if (mode == QUAD_8 && J.s.n == 1)
    plane_quad_modes<1, 8, 0> (J, lane, y0, rows, ds, dd, nt);
  else if (mode == QUAD_8 && J.dstep == 2)
    plane_quad_modes<2, 8, 2> (J, lane, y0, rows, ds, dd, nt);
  else if (mode == QUAD_8)
    plane_quad_modes<2, 8, 0> (J, lane, y0, rows, ds, dd, nt);
  else if (J.s.n == 1)
    plane_quad_body<1, 4, 0, -1> (J, lane, y0, rows, ds, dd, nt);
  else
    plane_quad_body<2, 4, 0, -1> (J, lane, y0, rows, ds, dd, nt);
}

// does the plane go through plane_direct_body (host and device)
// two N-tap passes.  phase 0: source rectangle + table rows -> LDS; 1: first pass, LDS -> LDS; 2: second pass -> plane.  A barrier separates
// the phases.
GSTAMD_HD void plane_tile_body (const PlaneJob &J, uint8_t *lds, int tile, int tid, int phase)
{
  PlaneTile T;
  plane_tile_geometry (J, tile, T);
  if (phase == 0) {
    /* the source rectangle, row by row, as aligned words (T.skew: the bytes in front of the rectangle's first pixel that come along); the
       loads of a lane all leave before the first of them is stored (each is a full memory round trip otherwise); the row past the
       rectangle is zero, what lies past a row's end or the plane's is never weighted */
    const int rows = T.sy1 - T.sy0, wpr = T.pitch / 4, total = (rows + 1) * wpr;
    const size_t plane_bytes = (size_t) J.s.stride * (size_t) (J.ih - 1) + (size_t) J.iw * J.s.n;          /* up to the picture's last byte */
    for (int i0 = tid; i0 < total; i0 += PLN_STAGE_ILP * PLN_THREADS) {
      uint32_t v[PLN_STAGE_ILP];
#pragma unroll
      for (int k = 0; k < PLN_STAGE_ILP; k++) {
        const int i = i0 + k * PLN_THREADS;
        const int rr = i / wpr, w4 = 4 * (i % wpr);
        v[k] = 0;
        if (i < total && rr < rows) {
          const size_t at = (size_t) (T.sy0 + rr) * J.s.stride + (size_t) T.sx0 * J.s.n - T.skew + w4;
          if (J.wide_src && at + 4 <= plane_bytes)
            v[k] = *(const uint32_t *) (J.s.p + at);
          else
            for (int b = 0; b < 4 && at + b < plane_bytes; b++)
              v[k] |= (uint32_t) J.s.p[at + b] << (8 * b);
        }
      }
#pragma unroll
      for (int k = 0; k < PLN_STAGE_ILP; k++) {
        const int i = i0 + k * PLN_THREADS;
        if (i < total)
          *(uint32_t *) (lds + T.raw_off + 4 * i) = v[k];
      }
    }
    for (int k = 0; k < 2; k++) {
      const ScaleDev &sd = J.pass[k];
      const bool horizontal = (k == 0) == (J.h_first != 0);
      const int o0 = horizontal ? T.x0 : T.y0, n_out = horizontal ? T.x1 - T.x0 : T.y1 - T.y0;
      const int taps = sd.kind == SCALE_NTAP ? sd.n_taps : (sd.kind == SCALE_2TAP ? 2 : 0);
      uint32_t *off = (uint32_t *) (lds + T.tab_off[k]);
      int16_t *tp = (int16_t *) (off + n_out);
      if (!(sd.kind == SCALE_2TAP && horizontal))       /* the horizontal 2-tap form has no tables (ldreslinl) */
        for (int i = tid; i < n_out; i += PLN_THREADS)
          off[i] = sd.offset[o0 + i];
      if (taps && sd.taps)
        for (int i = tid; i < n_out * taps; i += PLN_THREADS)
          tp[i] = sd.taps[(size_t) o0 * taps + i];
    }
    return;
  }
  const LdsBytes raw = {lds + T.raw_off + T.skew, T.sx0, T.sy0, T.pitch, J.s.n};
  uint32_t *mid = (uint32_t *) (lds + T.mid_off);
  if (phase == 1) {
    const ScaleDev p0 = plane_lds_pass (J, T, lds, 0);
    if (J.h_first) {
      const int rows = T.sy1 - T.sy0;
      for (int i = tid; i < rows * PLN_TW; i += PLN_THREADS) {
        const int rr = i / PLN_TW, c = i % PLN_TW;
        uint32_t v = 0;
        if (T.x0 + c < T.x1) {
          const RowOfSrc<LdsBytes> row = {raw, T.sy0 + rr};
          v = hscale_px (row, p0, T.x0 + c);
        }
        mid[i] = v;
      }
      for (int i = tid; i < PLN_TW; i += PLN_THREADS)          /* the row a zero-weight tap may point at */
        mid[rows * PLN_TW + i] = 0;
    } else {
      const int cols = T.sx1 - T.sx0;
      for (int i = tid; i < PLN_TH * (cols + 1); i += PLN_THREADS) {
        const int rr = i / (cols + 1), c = i % (cols + 1);
        mid[i] = (c < cols && T.y0 + rr < T.y1) ? vscale_px (raw, p0, T.sx0 + c, T.y0 + rr) : 0u;
      }
    }
    return;
  }
  const int r = tid / (PLN_TW / 4), x = T.x0 + 4 * (tid % (PLN_TW / 4)), y = T.y0 + r;
  if (y >= T.y1 || x >= T.x1)
    return;
  const ScaleDev p1 = plane_lds_pass (J, T, lds, 1);
  uint32_t px[4] = {0, 0, 0, 0};
  if (J.h_first) {
    const LdsPlane lp = {mid, T.x0, T.sy0, PLN_TW};
    for (int i = 0; i < 4 && x + i < T.x1; i++)
      px[i] = vscale_px (lp, p1, x + i, y);
  } else {
    const RowOfLds row = {mid + r * T.mid_pitch, T.sx0};
    for (int i = 0; i < 4 && x + i < T.x1; i++)
      px[i] = hscale_px (row, p1, x + i);
  }
  plane_put4 (J, x, y, T.x1, px);
}

// LDS bytes the plane's largest tile needs (host, from the pass tables); 0 for planes without a second pass
inline size_t plane_job_lds_bytes (const PlanePlan &pp)
{
  if (pp.kind != PLANE_SCALE || pp.passes.size () != 2)
    return 0;
  if ((pp.passes[0].kind == SCALE_NEAREST || pp.passes[0].kind == SCALE_2TAP) && (pp.passes[1].kind == SCALE_NEAREST || pp.passes[1].kind == SCALE_2TAP))
    return 0;                   /* the direct form: nothing staged */
  const bool h_first = pp.passes[0].horizontal;
  const ScalePass &ph = pp.passes[h_first ? 0 : 1], &pv = pp.passes[h_first ? 1 : 0];
  const auto span = [](const ScalePass &p, bool horizontal, int o0, int o1) {
    if (p.kind == SCALE_2TAP && horizontal)
      return ((((o1 - 1) * p.inc) >> 16) + 2) - ((o0 * p.inc) >> 16);
    const int n = p.kind == SCALE_NEAREST ? 1 : (p.kind == SCALE_2TAP ? 2 : p.n_taps);
    return (int) p.offset[o1 - 1] + n - (int) p.offset[o0];
  };
  const auto tabs = [](const ScalePass &p, int n_out) {
    const int taps = p.kind == SCALE_NTAP ? p.n_taps : (p.kind == SCALE_2TAP ? 2 : 0);
    return (size_t) ((4 * n_out + 2 * n_out * taps + 3) & ~3);
  };
  int cols = 0, rows = 0;
  for (int o0 = 0; o0 < pp.ow; o0 += PLN_TW)
    cols = std::max (cols, span (ph, true, o0, std::min (o0 + PLN_TW, pp.ow)));
  for (int o0 = 0; o0 < pp.oh; o0 += PLN_TH)
    rows = std::max (rows, span (pv, false, o0, std::min (o0 + PLN_TH, pp.oh)));
  const size_t pitch = (size_t) ((3 + cols * pp.n_elems + 7) & ~3);
  const size_t raw = (size_t) (rows + 1) * pitch;
  const size_t mid = 4 * (h_first ? (size_t) (rows + 1) * PLN_TW : (size_t) PLN_TH * (cols + 1));
  return raw + mid + tabs (pp.passes[0], h_first ? PLN_TW : PLN_TH) + tabs (pp.passes[1], h_first ? PLN_TH : PLN_TW);
}

}  // namespace gstamd
