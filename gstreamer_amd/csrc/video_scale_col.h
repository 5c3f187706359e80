// video_scale_col.h - the column-walk scaler: both N-tap passes (horizontal first) of a 2x horizontally subsampled planar / semi-planar
// 4:2:0 source in ONE kernel (BASELINE C3: 8K I420 -> 1080p RGBA Lanczos; every Lanczos / cubic / sinc down-scale of a decoder frame).
// The integers of every stage are the reference's: unpack + chroma upsample (video-chroma.c:277-327, 687-699; do_upsample_lines
// video-converter.c:2991 filters horizontally first, then blends the two chroma rows 3:1), video_scale_h_ntap_u8 (video-scaler.c:621-760:
// 16-bit wrapping sum, (sum + 32) >> 6, clamped u8 AYUV) on every source line, video_scale_v_ntap_u8 (:987-1072) on those bytes, then the
// convert stage (chain order: video-converter.c:1685-1714).
//
// Work split.  A WAVE owns a column tile (<= 64 opl outputs wide, the source span under it is 256 opl pixels = 4 opl per lane) and a run of
// output rows, and walks down the source in GROUPS of four lines (lines 4g-1 .. 4g+2 = the line pairs 2g, 2g+1 of the chroma upsampler):
//   stage   every lane brings its 4 opl pixels of the four lines into three BYTE planes (Y, U, V, XOR 0x80) in the wave's LDS slice: the two
//           new chroma rows are h-filtered in byte lanes (v_perm selectors made per lane at the start carry the edge rules), the third row
//           is carried from the previous group, the four 3:1 blends share their inner average;
//   filter  every lane owns opl output columns: their windows of the four lines against the lane's int8 tap words (v_dot4_i32_i8), the four
//           lines' results of one output and channel packed into ONE word (byte = line) - the group's ring word;
//   rows    the vertical pass is a byte dot product down the lane's own ring words (no other lane ever reads them), post stage, store.
// The loads of the next group are in flight (second register set) while a group is staged and filtered.  Nothing in the loop waits for
// another wave: the waves of a workgroup sit on ONE column tile and split the workgroup's rows top to bottom; the few groups the last rows
// of a wave share with the first rows of the wave below are not filtered twice - the lower wave makes them first anyway and leaves a copy
// in LDS (a flag per wave; by the time the upper wave has walked down its own rows the copy is long there).  One barrier at the start
// (the flags' initial state), none afterwards.
#pragma once
#include "video_hscale420.h"

#define GSTAMD_COL_MAX_WAVES 8
#define GSTAMD_COL_MAX_FRAMES 16
#define GSTAMD_COL_FLAG_BYTES 64

namespace gstamd {

// frame list (GstBufferList analogue): workgroup z converts frame z
struct ColFrames {
  const uint8_t *y[GSTAMD_COL_MAX_FRAMES];
  const uint8_t *c0[GSTAMD_COL_MAX_FRAMES];     // planar: U plane; semi-planar: the interleaved plane
  const uint8_t *c1[GSTAMD_COL_MAX_FRAMES];     // planar: V plane
  uint8_t *dst[GSTAMD_COL_MAX_FRAMES];
};

// a source plane as the kernel reads it: a raw buffer (base, bytes) - loads take a 32-bit lane offset plus a scalar row offset, and reads
// past the end return zero (lanes right of the picture and lines below it only ever meet zero taps)
#ifdef __HIPCC__
typedef __amdgpu_buffer_rsrc_t colplane_t;
#else
struct colplane_t { const uint8_t *p; uint32_t bytes; };
#endif
struct ColSrc {
  colplane_t y, c0, c1;         // c0 / c1: based at chroma row ColParams::crow_lo (rows above a crop origin have negative numbers)
  colplane_t out;               // the destination frame
};

GSTAMD_HD colplane_t col_plane (const uint8_t *base, long long first_byte, uint32_t bytes)
{
#ifdef __HIPCC__
  return __builtin_amdgcn_make_buffer_rsrc ((void *) (base + first_byte), (short) 0, (int) bytes, 0x00020000);
#else
  colplane_t pl = {base + first_byte, bytes};
  return pl;
#endif
}

GSTAMD_HD void col_store32 (colplane_t pl, uint32_t lane_off, uint32_t row_off, const uint32_t *v, int n)
{
#ifdef __HIPCC__
  typedef uint32_t u32x2 __attribute__ ((ext_vector_type (2)));
  if (n == 2) {
    const u32x2 d = {v[0], v[1]};
    __builtin_amdgcn_raw_buffer_store_b64 (d, pl, (int) lane_off, (int) row_off, 0);
  } else {
    __builtin_amdgcn_raw_buffer_store_b32 (v[0], pl, (int) lane_off, (int) row_off, 0);
  }
#else
  if ((unsigned long long) lane_off + row_off + 4 * n <= pl.bytes)
    __builtin_memcpy (const_cast<uint8_t *> (pl.p) + lane_off + row_off, v, 4 * (size_t) n);
#endif
}

// dwords at byte lane_off + row_off (+ 4) of the plane; row_off is wave-uniform (the instruction's scalar offset)
GSTAMD_HD uint32_t col_load32 (colplane_t pl, uint32_t lane_off, uint32_t row_off)
{
#ifdef __HIPCC__
  return (uint32_t) __builtin_amdgcn_raw_buffer_load_b32 (pl, (int) lane_off, (int) row_off, 0);
#else
  uint32_t v = 0;
  if ((unsigned long long) lane_off + row_off + 4 <= pl.bytes)
    __builtin_memcpy (&v, pl.p + lane_off + row_off, 4);
  return v;
#endif
}

struct ColParams {
  int ystride, cstride;
  int width, height;            // source picture (pixels / lines), width % 4 == 0
  int u_first;                  // semi-planar: U is the first byte of a pair (NV12)
  int crow_lo, crow_hi;         // chroma rows the upsampler may touch (frame rows around a crop)
  const int32_t *tiles;         // ColTables::tiles
  const uint32_t *hout, *vrow;  // ColTables::hout / vrow
  int out_w, out_h;
  int n_tiles, n_chunks;        // grid: n_tiles x n_chunks workgroups per frame
  int rows_per_wg, rows_per_wave;
  int rows_last, nwaves;        // a workgroup's last wave owns rows_last <= rows_per_wave rows (it also makes the groups past its last row that the
                                // other waves get from the wave below them): rows_per_wg = rows_per_wave * (nwaves - 1) + rows_last
  int dstride;
  int pubn;                     // ring slots of a wave's hand-over area
#ifdef GSTAMD_COL_TRACE
  unsigned long long *trace;    // profiling builds: [workgroup][wave][8] cycles per phase of the walk
#endif
};

// the instantiated forms (outputs per lane, tap words per output, window groups per row, window form, 8-byte aligned shared window); each exists
// for the co-sited and the non-co-sited horizontal chroma filter and for planar and semi-planar sources.  A plan runs on the smallest form
// that holds it: tap words and groups beyond its own are zero.
#define GSTAMD_COL_FORMS(V) \
  V (1, 3, 3, -1, 0) V (2, 3, 3, -1, 0) V (2, 3, 3, 1, 1) V (2, 3, 3, 0, 0) \
  V (1, 4, 4, -1, 0) V (2, 4, 4, -1, 0) V (2, 4, 4, 1, 1) V (2, 4, 4, 0, 0)

struct ColForm { int opl, nw, ngv, wstep, a8; };
// the form that serves (opl, nw, ngv, wstep, a8), false: none
inline bool col_form_for (int opl, int nw, int ngv, int wstep, int a8, ColForm *out)
{
  static const ColForm forms[] = {
#define V(o, n, g, w, a) {o, n, g, w, a},
    GSTAMD_COL_FORMS (V)
#undef V
  };
  const ColForm *best = nullptr;
  for (const ColForm &f : forms)
    if (f.opl == opl && f.wstep == wstep && f.a8 <= a8 && f.nw >= nw && f.ngv >= ngv &&
        (!best || f.nw + f.ngv < best->nw + best->ngv || (f.nw + f.ngv == best->nw + best->ngv && f.a8 > best->a8)))
      best = &f;
  if (best && out)
    *out = *best;
  return best != nullptr;
}

// tables + form for a plan's two passes.  opl_pref: 0 = the default order (two outputs per lane first), else that many only
inline bool col_choose (const ScalePass &h, const ScalePass &v, int width, int height, int opl_pref, bool share, ColTables *t, ColForm *form)
{
  for (int opl = 2; opl >= 1; opl--) {
    if (opl_pref && opl != opl_pref)
      continue;
    for (int sh = share ? 1 : 0; sh >= 0; sh--)
      if (make_col_tables (h, v, width, height, opl, sh != 0, t) && col_form_for (opl, t->nw, t->ngv, t->wstep, t->a8, form)) {
        col_align_rows (t, form->ngv);
        return true;
      }
  }
  return false;
}

// rows of a workgroup's last wave: it makes the groups its rows share with the rows BELOW the workgroup itself (the other waves get
// theirs from the wave below), so it owns that many rows fewer - the waves of a workgroup then finish together
inline int col_rows_last (const ColTables &t, int rows_per_wave, int out_h)
{
  const int extra = (int) (((long long) t.pubn * out_h + t.n_groups / 2) / (t.n_groups > 0 ? t.n_groups : 1));
  return rows_per_wave - extra > 1 ? rows_per_wave - extra : (rows_per_wave > 1 ? 1 : rows_per_wave);
}

template <int OPL>
struct ColGeom {
  static constexpr int PXL = 4 * OPL;                   // source pixels per lane and line
  static constexpr int SPAN = 64 * PXL;
  static constexpr int PP = SPAN + 16;                  // bytes of a staged plane
  static constexpr int LINEB = 3 * PP;
  static constexpr int STAGEB = 4 * LINEB + 32;         // + room for the zero-tap words a window may read past the last plane
  static constexpr int SLOTW = 3 * OPL * 64;            // words of a ring slot: [channel][lane][output of the lane]
};

// LDS of a wave: the staged group, then the hand-over slots (the ring of line groups itself lives in registers)
GSTAMD_H420_HOSTDEV size_t col_wave_bytes (int opl, int ngv, int pubn)
{
  const int slotw = 3 * opl * 64, stage = 4 * 3 * (256 * opl + 16) + 32;
  (void) ngv;
  return (size_t) ((stage + pubn * slotw * 4 + 15) & ~15);
}

// per-lane constants of a wave
template <int OPL, int NW>
struct ColLane {
  uint32_t tw[OPL][NW];
  int hinit[OPL];
  int wb[OPL];                  // byte offset of the output's first window word inside a staged plane (shared windows: wb[0] for both)
  int lb[4][OPL];               // wb + line * LINEB, kept apart (see col_setup)
  int st;                       // byte offset of the lane's staged pixels inside a plane
  int xl;                       // first source pixel the lane loads
  int xo;                       // first output column of the lane
  int ca[OPL];                  // first chroma sample of the lane's chroma load(s)
  uint32_t sx[OPL], sy[OPL];    // v_perm selectors of the horizontal chroma filter (U; planar: both planes)
  uint32_t sxv[OPL], syv[OPL];  // semi-planar: the V bytes
  uint32_t hu[OPL], hv[OPL];    // h-filtered chroma row 2g-1 (carried from the previous group)
};

// the lane's ring: the words of the last NGV line groups of its output column(s), oldest first - registers, moved down one place per group
template <int OPL, int NGV>
struct ColRingRegs {
  uint32_t w[NGV][3][OPL];
};

// one group's loads
template <int OPL>
struct ColRaw {
  uint32_t y[4][OPL];
  uint32_t c[2][2 * OPL];       // chroma rows 2g, 2g+1.  planar: [0, OPL) U, [OPL, 2 OPL) V; semi-planar: OPL x 8 interleaved bytes
};

GSTAMD_HD int col_clamp (int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// byte offset of row `row` of a plane: wave-uniform, stays on the scalar unit
GSTAMD_HD uint32_t col_row_off (int row, int stride)
{
#ifdef __HIPCC__
  return (uint32_t) __builtin_amdgcn_readfirstlane (row * stride);
#else
  return (uint32_t) (row * stride);
#endif
}

// tile entry / row entry (wave-uniform index): read through the constant address space, so that the compiler issues scalar loads
// (s_load_dwordx8) and places their waits itself - a row's entry is requested a whole group ahead of its use
GSTAMD_HD void col_entry8 (const uint32_t *table, int idx, uint32_t *e)
{
#ifdef __HIPCC__
  typedef const __attribute__ ((address_space (4))) uint32_t *cptr_t;
  cptr_t t = (cptr_t) (uintptr_t) table + (size_t) __builtin_amdgcn_readfirstlane (idx) * 8;
#pragma unroll
  for (int k = 0; k < 8; k++)
    e[k] = t[k];
#else
  for (int k = 0; k < 8; k++)
    e[k] = table[(size_t) idx * 8 + k];
#endif
}

// byte dot product with accumulator, ALWAYS this one opcode (never the compiler's own choice between v_dot4_i32_i8 and v_dot4c_i32_i8): a dot
// product's result may be read back-to-back only as the accumulator of the SAME opcode (every other reader needs three wait states,
// which the compiler cannot add around instructions it does not see - col_fin4 / col_fin_px start with them).  b_is_uniform: the taps
// sit in a scalar register.
GSTAMD_HD int col_dot4 (uint32_t a, uint32_t b, int c)
{
#ifdef __HIPCC__
  int r;
  asm ("v_dot4_i32_i8 %0, %1, %2, %3" : "=v" (r) : "v" (a), "v" (b), "v" (c));
  return r;
#else
  return dot4_i8 (a, b, c);
#endif
}
GSTAMD_HD int col_dot4s (uint32_t a, uint32_t b_uniform, int c)
{
#ifdef __HIPCC__
  int r;
  asm ("v_dot4_i32_i8 %0, %1, %2, %3" : "=v" (r) : "v" (a), "s" (b_uniform), "v" (c));
  return r;
#else
  return dot4_i8 (a, b_uniform, c);
#endif
}

// selector byte of chroma sample s for a lane whose load starts at sample a: planar byte s - a, semi-planar byte 2 (s - a) + comp
GSTAMD_HD uint32_t col_sel_byte (int s, int a, int cw, int semi, int comp)
{
  const int pos = col_clamp (s, 0, cw - 1) - a;
  return (uint32_t) (semi ? 2 * pos + comp : pos);
}

template <int OPL, int NW, int CH, int SEMI>
GSTAMD_HD void col_setup (const ColParams &p, const int32_t *tile, int lane, ColLane<OPL, NW> &L)
{
  typedef ColGeom<OPL> G;
  const int o0 = tile[0], n = tile[1], s0 = tile[2], p0 = tile[3];
  const int cw = p.width >> 1;
  int xl = p0 + G::PXL * lane;
  if (xl + G::PXL > p.width)
    xl = p.width - G::PXL;              // lanes past the picture: harmless loads, their staged bytes meet zero taps only
  L.xl = xl;
  L.st = (p0 - s0) + G::PXL * lane;
  const int uo = SEMI ? (p.u_first ? 0 : 1) : 0;
#pragma unroll
  for (int j = 0; j < OPL; j++) {
    const int k = (xl + 4 * j) >> 1;
    const int a = col_clamp (CH == CHROMA_H_H2 ? k - 1 : k, 0, cw - 4);
    L.ca[j] = a;
    // pixels 4j .. 4j+3 of the lane: X = {c[k], c[k], c[k+1], c[k+1]}; co-sited Y = {c[k], c[k+1], c[k+1], c[k+2]} (even pixel: the sample, odd:
    // (c[j] + c[j+1] + 1) >> 1), else Y = {c[k-1], c[k+1], c[k], c[k+2]} ((3 X + Y + 2) >> 2); samples clamp at the picture's edges
    const uint32_t k0 = col_sel_byte (k, a, cw, SEMI, uo), k1 = col_sel_byte (k + 1, a, cw, SEMI, uo), k2 = col_sel_byte (k + 2, a, cw, SEMI, uo),
        km = col_sel_byte (k - 1, a, cw, SEMI, uo);
    L.sx[j] = k0 | (k0 << 8) | (k1 << 16) | (k1 << 24);
    if (CH == CHROMA_H_H2)
      L.sy[j] = km | (k1 << 8) | (k0 << 16) | (k2 << 24);
    else if (CH == CHROMA_H_H2_CS)
      L.sy[j] = k0 | (k1 << 8) | (k1 << 16) | (k2 << 24);
    else
      L.sy[j] = L.sx[j];
    L.sxv[j] = L.sx[j] ^ 0x01010101u;   // the other byte of every pair
    L.syv[j] = L.sy[j] ^ 0x01010101u;
    L.hu[j] = L.hv[j] = 0;
  }
  // the lane's outputs: o0 + OPL * lane + i; lanes past the tile's end repeat its last output(s) (same bytes to the same address)
  const int units = n / OPL;
  const int u = lane < units ? lane : units - 1;
  L.xo = o0 + OPL * u;
#pragma unroll
  for (int i = 0; i < OPL; i++) {
    const uint32_t *e = p.hout + (size_t) (o0 + OPL * u + i) * 8;
    L.wb[i] = (int) e[0];
    L.hinit[i] = (int) e[1];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      // one address register per staged line: all three planes of a line are within reach of its LDS instructions' immediate offsets
      // (8 bits of dwords for the paired reads).  Opaque, or the compiler folds the lines back onto one base and adds a constant per read.
      L.lb[k][i] = L.wb[i] + k * ColGeom<OPL>::LINEB;
#ifdef __HIPCC__
      asm volatile ("" : "+v" (L.lb[k][i]));
#endif
    }
#pragma unroll
    for (int k = 0; k < NW; k++)
      L.tw[i][k] = e[2 + k];
  }
}

template <int OPL, int SEMI>
GSTAMD_HD void col_load_crow (const ColParams &p, const ColSrc &s, const int *ca, int row, uint32_t *c)
{
  const uint32_t ro = col_row_off (col_clamp (row, p.crow_lo, p.crow_hi) - p.crow_lo, p.cstride);
  if (SEMI) {
#pragma unroll
    for (int j = 0; j < OPL; j++) {
      c[2 * j] = col_load32 (s.c0, (uint32_t) (2 * ca[j]), ro);
      c[2 * j + 1] = col_load32 (s.c0, (uint32_t) (2 * ca[j] + 4), ro);
    }
  } else {
#pragma unroll
    for (int j = 0; j < OPL; j++) {
      c[j] = col_load32 (s.c0, (uint32_t) ca[j], ro);
      c[OPL + j] = col_load32 (s.c1, (uint32_t) ca[j], ro);
    }
  }
}

// loads of group g: lines 4g-1 .. 4g+2 (clamped into the picture: lines outside it meet zero taps only), chroma rows 2g, 2g+1
template <int OPL, int NW, int SEMI>
GSTAMD_HD void col_request (const ColParams &p, const ColSrc &s, const ColLane<OPL, NW> &L, int g, ColRaw<OPL> &r)
{
#if defined(GSTAMD_COL_ABL) && GSTAMD_COL_ABL == 1      /* profiling builds only (results WRONG): no source loads after a wave's first */
  if (g != -12345)
    return;
#endif
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t ro = col_row_off (col_clamp (4 * g - 1 + k, 0, p.height - 1), p.ystride);
#pragma unroll
    for (int j = 0; j < OPL; j++)
      r.y[k][j] = col_load32 (s.y, (uint32_t) (L.xl + 4 * j), ro);
  }
  col_load_crow<OPL, SEMI> (p, s, L.ca, 2 * g, r.c[0]);
  col_load_crow<OPL, SEMI> (p, s, L.ca, 2 * g + 1, r.c[1]);
}

// horizontal chroma filter of one raw row: 4 OPL pixels of U and of V in byte lanes
template <int OPL, int NW, int CH, int SEMI>
GSTAMD_HD void col_hup (const ColLane<OPL, NW> &L, const uint32_t *c, uint32_t *u, uint32_t *v)
{
#pragma unroll
  for (int j = 0; j < OPL; j++) {
    uint32_t xu, yu, xv, yv;
    if (SEMI) {
      xu = bperm (c[2 * j + 1], c[2 * j], L.sx[j]), xv = bperm (c[2 * j + 1], c[2 * j], L.sxv[j]);
      if (CH != CHROMA_H_NONE)
        yu = bperm (c[2 * j + 1], c[2 * j], L.sy[j]), yv = bperm (c[2 * j + 1], c[2 * j], L.syv[j]);
      else
        yu = xu, yv = xv;
    } else {
      xu = bperm (c[j], c[j], L.sx[j]), xv = bperm (c[OPL + j], c[OPL + j], L.sx[j]);
      if (CH != CHROMA_H_NONE)
        yu = bperm (c[j], c[j], L.sy[j]), yv = bperm (c[OPL + j], c[OPL + j], L.sy[j]);
      else
        yu = xu, yv = xv;
    }
    if (CH == CHROMA_H_H2_CS)
      u[j] = lerp_u8 (xu, yu, 0x01010101u), v[j] = lerp_u8 (xv, yv, 0x01010101u);
    else if (CH == CHROMA_H_H2)
      u[j] = blend31_u8 (xu, yu), v[j] = blend31_u8 (xv, yv);
    else
      u[j] = xu, v[j] = xv;
  }
}

// the lane's 4 OPL bytes of one staged plane line.  Tiles whose LDS placement is not a multiple of the lane's width (the first tile:
// s0 < 0) store 16-bit halves - an 8-byte store off its alignment is replayed for 64 cycles.
template <int OPL, int ALIGNED>
GSTAMD_HD void col_lds_store (uint8_t *d, const uint32_t *w)
{
  if constexpr (ALIGNED) {
    if constexpr (OPL == 2)
      *(uint2 *) d = gstamd_make_uint2 (w[0], w[1]);
    else
      *(uint32_t *) d = w[0];
  } else {
#pragma unroll
    for (int j = 0; j < OPL; j++) {
#ifdef __HIPCC__
      typedef volatile __attribute__ ((address_space (3))) uint16_t *lds16_t;       /* volatile: not to be merged back into one misaligned store */
      *(lds16_t) (d + 4 * j) = (uint16_t) w[j];
      *(lds16_t) (d + 4 * j + 2) = (uint16_t) (w[j] >> 16);
#else
      *(uint16_t *) (d + 4 * j) = (uint16_t) w[j];
      *(uint16_t *) (d + 4 * j + 2) = (uint16_t) (w[j] >> 16);
#endif
    }
  }
}

// stage group g from its loads: Y, U, V byte planes of the four lines.  Chroma rows: A = 2g-1 (carried, h-filtered), B = 2g, C = 2g+1;
// line 4g-1 = (3 A + B + 2) >> 2, 4g = (A + 3 B + 2) >> 2, 4g+1 = (3 B + C + 2) >> 2, 4g+2 = (B + 3 C + 2) >> 2 per byte, each as
// lerp (heavy, (heavy + light) >> 1) - the inner average is the same for the two lines of a pair.
template <int OPL, int NW, int CH, int SEMI, int ALIGNED>
GSTAMD_HD void col_stage (ColLane<OPL, NW> &L, const ColRaw<OPL> &r, uint8_t *stage)
{
  typedef ColGeom<OPL> G;
  uint32_t bu[OPL], bv[OPL], cu[OPL], cv[OPL];
  col_hup<OPL, NW, CH, SEMI> (L, r.c[0], bu, bv);
  col_hup<OPL, NW, CH, SEMI> (L, r.c[1], cu, cv);
  uint32_t lu[4][OPL], lv[4][OPL], ly[4][OPL];
#pragma unroll
  for (int j = 0; j < OPL; j++) {
    const uint32_t mu = lerp_u8 (L.hu[j], bu[j], 0u), mv = lerp_u8 (L.hv[j], bv[j], 0u);
    const uint32_t nu = lerp_u8 (bu[j], cu[j], 0u), nv = lerp_u8 (bv[j], cv[j], 0u);
    lu[0][j] = lerp_u8 (L.hu[j], mu, 0x01010101u) ^ 0x80808080u, lv[0][j] = lerp_u8 (L.hv[j], mv, 0x01010101u) ^ 0x80808080u;
    lu[1][j] = lerp_u8 (bu[j], mu, 0x01010101u) ^ 0x80808080u, lv[1][j] = lerp_u8 (bv[j], mv, 0x01010101u) ^ 0x80808080u;
    lu[2][j] = lerp_u8 (bu[j], nu, 0x01010101u) ^ 0x80808080u, lv[2][j] = lerp_u8 (bv[j], nv, 0x01010101u) ^ 0x80808080u;
    lu[3][j] = lerp_u8 (cu[j], nu, 0x01010101u) ^ 0x80808080u, lv[3][j] = lerp_u8 (cv[j], nv, 0x01010101u) ^ 0x80808080u;
    L.hu[j] = cu[j], L.hv[j] = cv[j];
#pragma unroll
    for (int k = 0; k < 4; k++)
      ly[k][j] = r.y[k][j] ^ 0x80808080u;
  }
  uint8_t *d = stage + L.st;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    col_lds_store<OPL, ALIGNED> (d + k * G::LINEB, ly[k]);
    col_lds_store<OPL, ALIGNED> (d + k * G::LINEB + G::PP, lu[k]);
    col_lds_store<OPL, ALIGNED> (d + k * G::LINEB + 2 * G::PP, lv[k]);
  }
}

// chroma row 2g-1 of a wave's first group: loaded and h-filtered on the spot
template <int OPL, int NW, int CH, int SEMI>
GSTAMD_HD void col_first_row (const ColParams &p, const ColSrc &s, ColLane<OPL, NW> &L, int g)
{
  uint32_t c[2 * OPL];
  col_load_crow<OPL, SEMI> (p, s, L.ca, 2 * g - 1, c);
  col_hup<OPL, NW, CH, SEMI> (L, c, L.hu, L.hv);
}

// Two gfx950 hazards the compiler does not see inside inline asm (both found on the device: bytes off by a few units in rows whose last
// tap word is not zero): (1) a VALU instruction that writes PART of a register (SDWA dst_sel other than DWORD) needs one wait state before
// a VALU instruction reads that register; (2) the result of a dot-product instruction (v_dot4c_i32_i8) may be read by a VALU instruction
// of another kind only three wait states later - an earlier read sees the accumulator WITHOUT the last product.  The sequences below are
// single asm blocks that start with s_nop 2 (their inputs come straight from dot products), keep an independent instruction between every
// pair of the first kind, and end in s_nop 0 for whatever reads the result next.
//
// four 16-bit wrapping sums -> ((int16) sum >> 6) clamped to a byte each, byte k = sum k, XOR 0x80: a ring word
GSTAMD_HD uint32_t col_fin4 (int a0, int a1, int a2, int a3)
{
#ifdef __HIPCC__
  uint32_t t01, t23, w;
  const int six = 6;
  asm ("s_nop 2\n\t"
       "v_ashrrev_i16_sdwa %1, %3, %4 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
       "v_ashrrev_i16_sdwa %2, %3, %6 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
       "v_ashrrev_i16_sdwa %1, %3, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:WORD_0\n\t"
       "v_ashrrev_i16_sdwa %2, %3, %7 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:WORD_0\n\t"
       "v_sat_pk_u8_i16 %0, %1\n\t"
       "v_sat_pk_u8_i16_sdwa %0, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
       "s_nop 0"
       : "=&v" (w), "=&v" (t01), "=&v" (t23) : "s" (six), "v" (a0), "v" (a1), "v" (a2), "v" (a3));
  return w ^ 0x80808080u;
#else
  const int a[4] = {a0, a1, a2, a3};
  uint32_t w = 0;
  for (int k = 0; k < 4; k++) {
    const int v = ((int) (int16_t) (uint16_t) a[k]) >> 6;
    w |= (uint32_t) (v < 0 ? 0 : (v > 255 ? 255 : v)) << (8 * k);
  }
  return w ^ 0x80808080u;
#endif
}

// the vertical pass's three sums of one output -> the AYUV pixel (alpha 0xff: the taps of every phase sum to >= 64)
GSTAMD_HD uint32_t col_fin_px (int ay, int au, int av)
{
#ifdef __HIPCC__
  uint32_t t01 = 0xffu, t23, w;
  const int six = 6;
  asm ("s_nop 2\n\t"
       "v_ashrrev_i16_sdwa %2, %3, %5 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
       "v_ashrrev_i16_sdwa %1, %3, %4 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:WORD_0\n\t"
       "v_ashrrev_i16_sdwa %2, %3, %6 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:WORD_0\n\t"
       "v_sat_pk_u8_i16 %0, %1\n\t"
       "v_sat_pk_u8_i16_sdwa %0, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
       "s_nop 0"
       : "=&v" (w), "+v" (t01), "=&v" (t23) : "s" (six), "v" (ay), "v" (au), "v" (av));
  return w;
#else
  return 0xffu | (h420r_finish (ay) << 8) | (h420r_finish (au) << 16) | (h420r_finish (av) << 24);
#endif
}

// N words of LDS from p; A8: p is 8-byte aligned (ds_read_b64 at 256 B / clk; off that alignment such a read is replayed for 64 cycles)
template <int N, int A8>
GSTAMD_HD void col_lds_words (const uint8_t *p, uint32_t *o)
{
#ifdef __HIPCC__
  typedef const __attribute__ ((address_space (3))) uint32_t *lptr_t;
  typedef uint32_t u32x2 __attribute__ ((ext_vector_type (2)));
  typedef const __attribute__ ((address_space (3))) u32x2 *lptr2_t;
  if (A8) {
#pragma unroll
    for (int k = 0; k + 1 < N; k += 2) {
      const u32x2 v = *(lptr2_t) (p + 4 * k);
      o[k] = v.x, o[k + 1] = v.y;
    }
    if (N & 1)
      o[N - 1] = *(lptr_t) (p + 4 * (N - 1));
  } else {
#pragma unroll
    for (int k = 0; k < N; k++)
      o[k] = *(lptr_t) (p + 4 * k);
  }
#else
  for (int k = 0; k < N; k++)
    o[k] = *(const uint32_t *) (p + 4 * k);
#endif
}

// horizontal pass of the staged group: the lane's outputs on the four lines, packed into the group's ring words gw[channel][output]
template <int OPL, int NW, int WSTEP, int A8>
GSTAMD_HD void col_hfilter (const ColLane<OPL, NW> &L, const uint8_t *stage, uint32_t gw[3][OPL])
{
  typedef ColGeom<OPL> G;
  constexpr int SH = WSTEP >= 0 ? WSTEP : 0;
  constexpr int NWIN = WSTEP >= 0 ? 1 : OPL;            /* windows a lane reads per line and plane */
  constexpr int NWORDS = NW + SH;
  int acc[4][3][OPL];
  uint32_t w[2][3][NWIN][NWORDS];
  // the windows of line k + 1 are requested before the dot products of line k start
  auto fetch = [&](int k, uint32_t (*d)[NWIN][NWORDS]) {
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int i = 0; i < NWIN; i++) {
#if defined(GSTAMD_COL_ABL) && GSTAMD_COL_ABL == 2      /* no window reads */
        for (int q = 0; q < NWORDS; q++)
          d[c][i][q] = (uint32_t) L.lb[k][i] * 3u + q + c;
#else
        col_lds_words<NWORDS, A8> (stage + L.lb[k][i] + c * G::PP, d[c][i]);
#endif
      }
  };
  fetch (0, w[0]);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (k < 3)
      fetch (k + 1, w[(k + 1) & 1]);
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int i = 0; i < OPL; i++) {
        // shared windows: words [0, NW) for the even output, [WSTEP, WSTEP + NW) for the odd one
        const uint32_t *ww = w[k & 1][c][WSTEP >= 0 ? 0 : i] + (WSTEP >= 0 && i ? SH : 0);
        int a = L.hinit[i];
#pragma unroll
        for (int q = 0; q < NW; q++) {
#if defined(GSTAMD_COL_ABL) && GSTAMD_COL_ABL == 5      /* no horizontal dot products */
          a ^= (int) ww[q];
#else
          a = col_dot4 (ww[q], L.tw[i][q], a);
#endif
        }
        acc[k][c][i] = a;
      }
  }
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int i = 0; i < OPL; i++)
      gw[c][i] = col_fin4 (acc[0][c][i], acc[1][c][i], acc[2][c][i], acc[3][c][i]);
}

// a ring slot (or a slot of the hand-over area): [channel][lane][output of the lane]
template <int OPL>
GSTAMD_HD void col_slot_store (uint32_t *slot, int lane, const uint32_t gw[3][OPL])
{
#pragma unroll
  for (int c = 0; c < 3; c++) {
    uint32_t *d = slot + (c * 64 + lane) * OPL;
    if constexpr (OPL == 2)
      *(uint2 *) d = gstamd_make_uint2 (gw[c][0], gw[c][1]);
    else
      *d = gw[c][0];
  }
}

template <int OPL>
GSTAMD_HD void col_slot_load (const uint32_t *slot, int lane, uint32_t gw[3][OPL])
{
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const uint32_t *d = slot + (c * 64 + lane) * OPL;
    if constexpr (OPL == 2) {
      const uint2 v = *(const uint2 *) d;
      gw[c][0] = v.x, gw[c][1] = v.y;
    } else {
#ifdef __HIPCC__
      gw[c][0] = *(const __attribute__ ((address_space (3))) uint32_t *) d;
#else
      gw[c][0] = *d;
#endif
    }
  }
}

// the newest group's words into the ring: the older ones move down one place (registers: an array indexed at run time would live in
// scratch memory - 5 x the kernel's time, as tried)
template <int OPL, int NGV>
GSTAMD_HD void col_ring_put (ColRingRegs<OPL, NGV> &rg, const uint32_t gw[3][OPL])
{
#pragma unroll
  for (int k = 0; k + 1 < NGV; k++)
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int i = 0; i < OPL; i++)
        rg.w[k][c][i] = rg.w[k + 1][c][i];
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int i = 0; i < OPL; i++)
      rg.w[NGV - 1][c][i] = gw[c][i];
}

// one output row (its window ends with the newest group in the ring): vertical pass down the lane's ring words - e[3 + k] is the tap
// word of the k-th oldest of the last NGV groups (ColTables::ngv_aligned) -, post stage, store.  e = the row's entry of ColParams::vrow
template <int OPL, int NGV, int POST>
GSTAMD_HD void col_vrow (const ColRingRegs<OPL, NGV> &rg, const uint32_t *e, const ColParams &p, const ColSrc &s, const Dst &dst, const PostFast &pf, int x, int j)
{
  int acc[3][OPL];
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int i = 0; i < OPL; i++)
      acc[c][i] = (int) e[2];
#pragma unroll
  for (int k = 0; k < NGV; k++)
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int i = 0; i < OPL; i++)
        acc[c][i] = col_dot4s (rg.w[k][c][i], e[3 + k], acc[c][i]);
  uint32_t px[OPL];
#pragma unroll
  for (int i = 0; i < OPL; i++) {
    const uint32_t q = col_fin_px (acc[0][i], acc[1][i], acc[2][i]);
    if constexpr (POST) {               /* VideoPlan::fast_post: the no-wrap AYUV -> ARGB matrix, alpha stays 0xff */
      const uint32_t z = q ^ 0x80808080u;
      px[i] = fast_pixel (pf.fp, z, 0x0c01010cu, z, 0x0c02020cu, 0x0c03030cu);
    } else {
      px[i] = dst.final ? pack_px (dst.pack_pos, apply_color (dst.post, q)) : q;
    }
  }
  col_store32 (s.out, (uint32_t) (4 * x), col_row_off (j, p.dstride), px, OPL);
}

// what a wave does: its rows [r0, r1), the groups it makes [ga, ge), the groups it takes from the wave below (ge .. gl), the groups it
// leaves for the wave above (ga .. gp; none: gp < ga)
struct ColWavePlan {
  int r0, r1;
  int ga, ge, gl, gp;
};

GSTAMD_HD bool col_wave_plan (const ColParams &p, int chunk, int wave, ColWavePlan *w)
{
  const int ra = chunk * p.rows_per_wg, rb = ra + p.rows_per_wg < p.out_h ? ra + p.rows_per_wg : p.out_h;
  w->r0 = ra + wave * p.rows_per_wave;
  if (w->r0 >= rb)
    return false;
  const int mine = wave == p.nwaves - 1 ? p.rows_last : p.rows_per_wave;
  w->r1 = w->r0 + mine < rb ? w->r0 + mine : rb;
  uint32_t e[8];
  col_entry8 (p.vrow, w->r0, e);
  w->ga = (int) e[0];
  col_entry8 (p.vrow, w->r1 - 1, e);
  w->gl = (int) e[1];
  if (w->r1 == rb) {
    w->ge = w->gl + 1;                  // the workgroup's last wave makes every group its rows need
  } else {
    col_entry8 (p.vrow, w->r1, e);
    w->ge = (int) e[0];
    if (w->ge > w->gl + 1)
      w->ge = w->gl + 1;
  }
  if (w->r0 == ra) {
    w->gp = w->ga - 1;
  } else {
    col_entry8 (p.vrow, w->r0 - 1, e);
    w->gp = (int) e[1];
  }
  return true;
}

// ---- one wave's walk ---------------------------------------------------------------------------------------------------------------------
// X supplies the lanes: on the device `each (f)` calls f once with the thread's own registers, in the host emulator (tests/emu) it
// loops over 64 lane states - ONE control flow for both.  sync (): the wave's LDS traffic is ordered (in-order LDS queue; a compiler
// fence on the device).  publish / wait_flag: the hand-over flag of a wave (workgroup-scope release / acquire).
template <int OPL, int NW, int NGV, int CH, int SEMI, int WSTEP, int A8, int POST, class X>
GSTAMD_HD void col_wave (X &x, const ColParams &p, const ColSrc &s, const int32_t *tile, const ColWavePlan &wp, uint8_t *wave_lds, uint8_t *below_lds,
    uint32_t *flags, int wave, const Dst &dst, const PostFast &pf)
{
  typedef ColGeom<OPL> G;
  typedef ColLane<OPL, NW> Lane;
  typedef ColRaw<OPL> Raw;
  typedef ColRingRegs<OPL, NGV> Ring;
  uint8_t *stage = wave_lds;
  uint32_t *pub = (uint32_t *) (wave_lds + G::STAGEB);
  const uint32_t *pub_below = (const uint32_t *) (below_lds + G::STAGEB);
  const bool aligned = ((tile[3] - tile[2]) % G::PXL) == 0;
  x.each ([&](int lane, Lane &L, Raw &ra, Ring &) {
    col_setup<OPL, NW, CH, SEMI> (p, tile, lane, L);
    col_request<OPL, NW, SEMI> (p, s, L, wp.ga, ra);
    col_first_row<OPL, NW, CH, SEMI> (p, s, L, wp.ga);
  });
  int r = wp.r0;
  uint32_t e[8];
  col_entry8 (p.vrow, r, e);
#ifdef GSTAMD_COL_TRACE
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = x.now ();
#define COL_STAMP(k) do { const unsigned long long t_ = x.now (); tacc[k] += t_ - tprev; tprev = t_; } while (0)
#else
#define COL_STAMP(k) do { } while (0)
#endif
  // the rows whose window ends with group g
  auto rows = [&](int g) {
    while (r < wp.r1 && (int) e[1] == g) {
#if !(defined(GSTAMD_COL_ABL) && GSTAMD_COL_ABL == 3)   /* 3: no rows */
      x.each ([&](int, Lane &L, Raw &, Ring &rg) { col_vrow<OPL, NGV, POST> (rg, e, p, s, dst, pf, L.xo, r); });
#endif
      r++;
      if (r < wp.r1)
        col_entry8 (p.vrow, r, e);
    }
  };
  int g = wp.ga;
  // group g: staged from its loads, whose registers then take the next group's requests - those have the horizontal pass, the rows
  // and the other waves' turns to arrive (one register set: a second one bought nothing but a wave of occupancy less)
  COL_STAMP (0);                        /* prologue: tables, first loads */
  while (g < wp.ge) {
    const int gn = g + 1 < wp.ge ? g + 1 : wp.ge - 1;
#ifdef GSTAMD_COL_TRACE
    x.wait_loads ();
    COL_STAMP (1);                      /* waiting for the group's source loads */
#endif
#if defined(GSTAMD_COL_ABL) && GSTAMD_COL_ABL == 4      /* no staging */
    if (g == -12345)
#else
    if (aligned)
#endif
      x.each ([&](int, Lane &L, Raw &ra, Ring &) {
        col_stage<OPL, NW, CH, SEMI, 1> (L, ra, stage);
        col_request<OPL, NW, SEMI> (p, s, L, gn, ra);
      });
#if defined(GSTAMD_COL_ABL) && GSTAMD_COL_ABL == 4
    else if (g == -12346)
#else
    else
#endif
      x.each ([&](int, Lane &L, Raw &ra, Ring &) {
        col_stage<OPL, NW, CH, SEMI, 0> (L, ra, stage);
        col_request<OPL, NW, SEMI> (p, s, L, gn, ra);
      });
    x.sync ();
    COL_STAMP (2);                      /* staging */
    x.each ([&](int lane, Lane &L, Raw &, Ring &rg) {
      uint32_t gw[3][OPL];
      col_hfilter<OPL, NW, WSTEP, A8> (L, stage, gw);
      col_ring_put<OPL, NGV> (rg, gw);
      if (g <= wp.gp)
        col_slot_store<OPL> (pub + (size_t) (g - wp.ga) * G::SLOTW, lane, gw);
    });
    x.sync ();
    if (g == wp.gp)
      x.publish (flags, wave);
    COL_STAMP (3);                      /* horizontal pass */
    rows (g);
    COL_STAMP (4);                      /* rows */
    g++;
  }
  if (wp.gp >= wp.ga)
    x.publish (flags, wave);            // (already done inside the loop whenever rows_per_wave >= ColTables::min_rows_per_wave: never leave the wave above waiting)
  if (wp.gl >= wp.ge) {
    x.wait_flag (flags, wave + 1);
    for (; g <= wp.gl; g++) {
      x.each ([&](int lane, Lane &, Raw &, Ring &rg) {
        uint32_t gw[3][OPL];
        col_slot_load<OPL> (pub_below + (size_t) (g - wp.ge) * G::SLOTW, lane, gw);
        col_ring_put<OPL, NGV> (rg, gw);
      });
      rows (g);
    }
  }
  COL_STAMP (5);                        /* hand-over */
#ifdef GSTAMD_COL_TRACE
  x.trace_out (p.trace, tacc, wp.ge - wp.ga);
#endif
#undef COL_STAMP
}

}  // namespace gstamd
