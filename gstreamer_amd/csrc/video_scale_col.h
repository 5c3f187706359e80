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

// 16 / 12 bytes at byte `off` of the plane (any alignment; out of range: zeros, no fault)
GSTAMD_HD void col_load128 (colplane_t pl, uint32_t off, uint32_t *v)
{
#ifdef __HIPCC__
  typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
  const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128 (pl, (int) off, 0, 0);
  v[0] = d.x, v[1] = d.y, v[2] = d.z, v[3] = d.w;
#else
  v[0] = v[1] = v[2] = v[3] = 0;
  if ((unsigned long long) off + 16 <= pl.bytes)
    __builtin_memcpy (v, pl.p + off, 16);
  else if (off < pl.bytes)
    __builtin_memcpy (v, pl.p + off, (pl.bytes - off) & ~3u);       /* range checking is per dword */
#endif
}

GSTAMD_HD void col_load96 (colplane_t pl, uint32_t off, uint32_t *v)
{
#ifdef __HIPCC__
  typedef uint32_t u32x3 __attribute__ ((ext_vector_type (3)));
  const u32x3 d = __builtin_amdgcn_raw_buffer_load_b96 (pl, (int) off, 0, 0);
  v[0] = d.x, v[1] = d.y, v[2] = d.z;
#else
  v[0] = v[1] = v[2] = 0;
  if ((unsigned long long) off + 12 <= pl.bytes)
    __builtin_memcpy (v, pl.p + off, 12);
  else if (off < pl.bytes)
    __builtin_memcpy (v, pl.p + off, (pl.bytes - off) & ~3u);
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

// the instantiated forms (outputs per lane, tap words per output, window groups per row, window form, 8-byte aligned shared window - 2: and
// every lane's window starts with the lane's own staged words, the REGISTER-WINDOW form, see col_hfilter_regs); each exists
// for the co-sited and the non-co-sited horizontal chroma filter and for planar and semi-planar sources.  A plan runs on the smallest form
// that holds it: tap words and groups beyond its own are zero.
#define GSTAMD_COL_FORMS(V) \
  V (1, 3, 3, -1, 0) V (2, 3, 3, -1, 0) V (2, 3, 3, 1, 1) V (2, 3, 3, 0, 0) V (2, 3, 3, 1, 2) \
  V (1, 4, 4, -1, 0) V (2, 4, 4, -1, 0) V (2, 4, 4, 1, 1) V (2, 4, 4, 0, 0) V (2, 4, 4, 1, 2)

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
inline bool col_choose (const ScalePass &h, const ScalePass &v, int width, int height, int opl_pref, bool share, ColTables *t, ColForm *form, bool regwin = true)
{
  for (int opl = 2; opl >= 1; opl--) {
    if (opl_pref && opl != opl_pref)
      continue;
    for (int sh = share ? 1 : 0; sh >= 0; sh--)
      if (make_col_tables (h, v, width, height, opl, sh != 0, t) && (regwin || t->a8 < 2 || (t->a8 = 1)) &&
          col_form_for (opl, t->nw, t->ngv, t->wstep, t->a8, form)) {
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

template <int OPL, int RW = 0>
struct ColGeom {
  static constexpr int PXL = 4 * OPL;                   // source pixels per lane and line (the lane's share of a staged line)
  static constexpr int SPAN = 64 * PXL;
  static constexpr int PP = SPAN + 16;                  // bytes of a staged plane
  static constexpr int LINEB = RW ? PP : 3 * PP;        // register windows: only the luma goes through LDS (loader lanes -> owner lanes)
  static constexpr int STAGEB = 4 * LINEB + 32;         // + room for the zero-tap words a window may read past the last plane
  static constexpr int SLOTW = 3 * OPL * 64;            // words of a hand-over slot: [channel][lane][output of the lane]
  // loads.  Vector memory instructions cost the same whatever they carry per lane (~14 G wave-instructions / s on the chip: one per ~37
  // cycles and CU), so a group's 1 KB x OPL of luma is OPL instructions of 16 bytes a lane (16 OPL lanes per line, 4 / OPL lines per
  // instruction) and its raw chroma rows go through a small LDS area: lanes 8 bytes apart read 12 bytes each (the 4 extra bytes carry the
  // neighbour samples of the row's last lane).  Planar: one instruction per plane, lanes [0, 16 OPL) row 2g, [16 OPL, 32 OPL) row 2g + 1;
  // semi-planar: 32 OPL lanes per row, OPL instructions.
  static constexpr int YI = OPL;
  static constexpr int YLPL = 16 * OPL;                 // lanes per line
  static constexpr int YLINES = 4 / OPL;                // lines per instruction
};
template <int OPL, int SEMI>
struct ColChroma {
  static constexpr int CB = SEMI ? 2 : 1;               // bytes per chroma sample position
  static constexpr int CI = SEMI ? OPL : 2;             // load instructions per group
  static constexpr int CLPR = 16 * OPL * CB;            // lanes per raw row
  static constexpr int CROWB = 128 * OPL * CB + 16;     // bytes of a raw row in LDS
  static constexpr int CROWS = SEMI ? 2 : 4;            // semi-planar: rows 2g, 2g+1; planar: U 2g, U 2g+1, V 2g, V 2g+1
};
#define GSTAMD_COL_CRAW_BYTES(opl) (4 * (128 * (opl) + 16) + 32)

// LDS of a wave: the staged group, the raw chroma rows, then the hand-over slots (the ring of line groups itself lives in registers)
GSTAMD_H420_HOSTDEV size_t col_wave_bytes (int opl, int ngv, int pubn, int rw = 0)
{
  const int slotw = 3 * opl * 64, stage = 4 * (rw ? 1 : 3) * (256 * opl + 16) + 32;
  (void) ngv;
  return (size_t) ((stage + GSTAMD_COL_CRAW_BYTES (opl) + pubn * slotw * 4 + 15) & ~15);
}

// per-lane constants of a wave
template <int OPL, int NW>
struct ColLane {
  uint32_t tw[OPL][NW];
  int hinit[OPL];
  int wb[OPL];                  // byte offset of the output's first window word inside a staged plane (shared windows: wb[0] for both)
  int lb[4][OPL];               // wb + line * LINEB, kept apart (see col_setup)
  int xo;                       // first output column of the lane
  // loads (the lane as one of 16 OPL per luma line / as one of the lanes of a raw chroma row)
  int yvo, ykl;                 // luma: byte offset inside the row, line of the group (first instruction; + YLINES per instruction)
  int yst;                      // luma: LDS byte of the lane's 16 pixels (first instruction; + YLINES * LINEB per instruction)
  int cvo[2], crow[2], cst[2];  // chroma instruction t: byte offset inside the row (idle lanes: out of range), row of the pair, LDS byte in the raw area
  // staging (the lane as the owner of 4 OPL pixels of every line)
  int st;                       // byte offset of the lane's staged pixels inside a plane
  int cr[OPL];                  // byte offset inside a raw chroma row of the 8 bytes the lane's chroma of pixels 4j .. 4j+3 comes from
  uint32_t sx[OPL], sy[OPL];    // v_perm selectors of the horizontal chroma filter (U; planar: both planes)
  uint32_t sxv[OPL], syv[OPL];  // semi-planar: the V bytes
  uint32_t hu[OPL], hv[OPL];    // h-filtered chroma row 2g-1 (carried from the previous group)
  uint32_t win[4][3][OPL];      // register windows (col_hfilter_regs): the lane's own words of the four lines' Y / U / V bytes
};

// the lane's ring: the words of the last NGV line groups of its output column(s), oldest first - registers, moved down one place per group
template <int OPL, int NGV>
struct ColRingRegs {
  uint32_t w[NGV][3][OPL];
};

// one group's loads
template <int OPL>
struct ColRaw {
  uint32_t y[OPL][4];
  uint32_t c[2][3];
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

// N words of LDS from p; A8: p is 8-byte aligned (ds_read_b64 at 256 B / clk; off that alignment such a read is replayed for 64 cycles);
// A8 < 0: p may be only 2-byte aligned
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
  } else if (A8 < 0) {
    typedef uint32_t __attribute__ ((aligned (2))) u32_a2;
#pragma unroll
    for (int k = 0; k < N; k++)
      o[k] = *(const __attribute__ ((address_space (3))) u32_a2 *) (p + 4 * k);
  } else {
#pragma unroll
    for (int k = 0; k < N; k++)
      o[k] = *(lptr_t) (p + 4 * k);
  }
#else
  for (int k = 0; k < N; k++)
    __builtin_memcpy (&o[k], p + 4 * k, 4);
#endif
}

// selector byte of chroma sample s among the 8 bytes that start at sample sb of a raw row: planar byte s - sb, semi-planar 2 (s - sb) + comp
GSTAMD_HD uint32_t col_sel_byte (int s, int sb, int cw, int semi, int comp)
{
  const int pos = col_clamp (col_clamp (s, 0, cw - 1) - sb, 0, semi ? 3 : 7);
  return (uint32_t) (semi ? 2 * pos + comp : pos);
}

template <int OPL, int NW, int CH, int SEMI, int RW = 0>
GSTAMD_HD void col_setup (const ColParams &p, const int32_t *tile, int lane, ColLane<OPL, NW> &L)
{
  typedef ColGeom<OPL, RW> G;
  typedef ColChroma<OPL, SEMI> C;
  const int o0 = tile[0], n = tile[1], s0 = tile[2], p0 = tile[3];
  const int cw = p.width >> 1, delta = p0 - s0;
  // the lane as the owner of 4 OPL pixels of a line: from the tile's first loaded pixel on - or (register windows) from the pixel its own
  // window starts with: window space begins s0 - the tile's first window, which may lie left of the picture - and unit 0's window at byte wsh
  const int wsh = RW ? (int) p.hout[(size_t) o0 * 8] : 0;
  const int ob = RW ? s0 + wsh : p0;
  // ---- the lane as a loader
  L.ykl = lane / G::YLPL;
  L.yvo = p0 + 16 * (lane % G::YLPL);
  L.yst = L.ykl * G::LINEB + delta + 16 * (lane % G::YLPL);
  int rawbase = p0 / 2 - 1 > 0 ? p0 / 2 - 1 : 0;                /* first sample of a raw row: one left of the span's first (the non-co-sited filter's neighbour) */
  if (p0 + G::SPAN > p.width)                                   /* a tile at the right edge: whole dwords up to the row's end (width % 8 == 0), see ColTables */
    rawbase -= (rawbase - cw) & (4 / C::CB - 1);
#pragma unroll
  for (int t = 0; t < 2; t++) {
    const int u = SEMI ? t * 64 + lane : lane;                  /* planar: instruction t is plane t */
    const int row = u / C::CLPR, j = u % C::CLPR;
    const bool live = t < C::CI && row < 2;
    L.crow[t] = live ? row : 0;
    L.cvo[t] = live ? rawbase * C::CB + 8 * j : 0x40000000;     /* idle lanes: out of range, no memory request */
    L.cst[t] = live ? ((SEMI ? row : 2 * t + row) * C::CROWB + 8 * j) : C::CROWS * C::CROWB;
  }
  // ---- the lane as the owner of 4 OPL pixels per line
  L.st = (ob - s0) + G::PXL * lane;
  const int uo = SEMI ? (p.u_first ? 0 : 1) : 0;
#pragma unroll
  for (int j = 0; j < OPL; j++) {
    const int k = (ob + G::PXL * lane + 4 * j) >> 1;
    // planar: the aligned dword pair around sample k - 1; semi-planar: the pairs k - 1 .. k + 2 themselves (8 bytes: no slack to align in -
    // the first tile, whose raw rows start at sample 0 instead of -1, reads them off a 2-byte boundary)
    int cr = SEMI ? (k - 1 - rawbase) * 2 : ((k - 1 - rawbase) & ~3);
    if (cr < 0)
      cr = 0;
    L.cr[j] = cr;
    const int sb = rawbase + cr / C::CB;
    // pixels 4j .. 4j+3 of the lane: X = {c[k], c[k], c[k+1], c[k+1]}; co-sited Y = {c[k], c[k+1], c[k+1], c[k+2]} (even pixel: the sample, odd:
    // (c[j] + c[j+1] + 1) >> 1), else Y = {c[k-1], c[k+1], c[k], c[k+2]} ((3 X + Y + 2) >> 2); samples clamp at the picture's edges
    const uint32_t k0 = col_sel_byte (k, sb, cw, SEMI, uo), k1 = col_sel_byte (k + 1, sb, cw, SEMI, uo), k2 = col_sel_byte (k + 2, sb, cw, SEMI, uo),
        km = col_sel_byte (k - 1, sb, cw, SEMI, uo);
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
  if (RW && lane >= units)
    L.xo = 0x10000000;          /* register windows: such a lane would repeat the last output from OTHER words - its stores fall out of range (dropped) */
#pragma unroll
  for (int i = 0; i < OPL; i++) {
    const uint32_t *e = p.hout + (size_t) (o0 + OPL * u + i) * 8;
    L.wb[i] = (int) e[0];
    L.hinit[i] = (int) e[1];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      // one address register per staged line: all three planes of a line are within reach of its LDS instructions' immediate offsets
      // (8 bits of dwords for the paired reads).  Opaque, or the compiler folds the lines back onto one base and adds a constant per read.
      L.lb[k][i] = L.wb[i] + k * G::LINEB;
#ifdef __HIPCC__
      asm volatile ("" : "+v" (L.lb[k][i]));
#endif
    }
#pragma unroll
    for (int k = 0; k < NW; k++)
      L.tw[i][k] = e[2 + k];
  }
}

// chroma rows crow0, crow0 + 1 (a group's: 2g, 2g + 1), clamped into the rows the upsampler may touch
template <int OPL, int NW, int SEMI>
GSTAMD_HD void col_request_chroma (const ColParams &p, const ColSrc &s, const ColLane<OPL, NW> &L, int crow0, ColRaw<OPL> &r)
{
  typedef ColChroma<OPL, SEMI> C;
#pragma unroll
  for (int t = 0; t < C::CI; t++) {
    const int row = col_clamp (crow0 + L.crow[t], p.crow_lo, p.crow_hi) - p.crow_lo;
    col_load96 (SEMI || t == 0 ? s.c0 : s.c1, (uint32_t) (row * p.cstride + L.cvo[t]), r.c[t]);
  }
}

// loads of group g: lines 4g-1 .. 4g+2 (clamped into the picture: lines outside it meet zero taps only), chroma rows 2g, 2g+1
template <int OPL, int NW, int SEMI>
GSTAMD_HD void col_request (const ColParams &p, const ColSrc &s, const ColLane<OPL, NW> &L, int g, ColRaw<OPL> &r)
{
  typedef ColGeom<OPL> G;          /* (the members used here do not depend on the window form) */
#if defined(GSTAMD_COL_ABL) && GSTAMD_COL_ABL == 1      /* profiling builds only (results WRONG): no source loads after a wave's first */
  if (g != -12345)
    return;
#endif
#pragma unroll
  for (int t = 0; t < G::YI; t++) {
    const int y = col_clamp (4 * g - 1 + L.ykl + t * G::YLINES, 0, p.height - 1);
    col_load128 (s.y, (uint32_t) (y * p.ystride + L.yvo), r.y[t]);
  }
  col_request_chroma<OPL, NW, SEMI> (p, s, L, 2 * g, r);
}

// n dwords at LDS byte d; tiles whose LDS placement is off the natural alignment (the first tile: s0 < 0) store 16-bit halves - a wide
// store off its alignment is replayed for 64 cycles
template <int N, int ALIGNED>
GSTAMD_HD void col_lds_store (uint8_t *d, const uint32_t *w)
{
  if constexpr (ALIGNED) {
    if constexpr (N == 4)
      *(uint4 *) d = gstamd_make_uint4 (w[0], w[1], w[2], w[3]);
    else if constexpr (N == 2)
      *(uint2 *) d = gstamd_make_uint2 (w[0], w[1]);
    else
      *(uint32_t *) d = w[0];
  } else {
#pragma unroll
    for (int j = 0; j < N; j++) {
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

// first half of the staging, the lane as a loader: the luma bytes (XOR 0x80) straight into their plane, the raw chroma rows into the raw area
template <int OPL, int NW, int SEMI, int ALIGNED, int RW = 0>
GSTAMD_HD void col_stage_loads (const ColLane<OPL, NW> &L, const ColRaw<OPL> &r, uint8_t *stage, uint8_t *craw, bool with_luma)
{
  typedef ColGeom<OPL, RW> G;
  typedef ColChroma<OPL, SEMI> C;
  if (with_luma) {
#pragma unroll
    for (int t = 0; t < G::YI; t++) {
      const uint32_t w[4] = {r.y[t][0] ^ 0x80808080u, r.y[t][1] ^ 0x80808080u, r.y[t][2] ^ 0x80808080u, r.y[t][3] ^ 0x80808080u};
      col_lds_store<4, ALIGNED> (stage + L.yst + t * G::YLINES * G::LINEB, w);
    }
  }
#pragma unroll
  for (int t = 0; t < C::CI; t++) {
    // 12 bytes, the last four the same as the next lane's first four
    col_lds_store<2, 1> (craw + L.cst[t], r.c[t]);
    col_lds_store<1, 1> (craw + L.cst[t] + 8, r.c[t] + 2);
  }
}

// horizontal chroma filter of one raw row (in the raw area): 4 OPL pixels of U and of V in byte lanes
template <int OPL, int NW, int CH, int SEMI, int ALIGNED>
GSTAMD_HD void col_hup (const ColLane<OPL, NW> &L, const uint8_t *craw, int row, uint32_t *u, uint32_t *v)
{
  typedef ColChroma<OPL, SEMI> C;
#pragma unroll
  for (int j = 0; j < OPL; j++) {
    uint32_t cu[2], cv[2];
    col_lds_words<2, (SEMI && !ALIGNED) ? -1 : 0> (craw + row * C::CROWB + L.cr[j], cu);
    if (SEMI)
      cv[0] = cu[0], cv[1] = cu[1];
    else
      col_lds_words<2, 0> (craw + (2 + row) * C::CROWB + L.cr[j], cv);
    const uint32_t xu = bperm (cu[1], cu[0], L.sx[j]), xv = bperm (cv[1], cv[0], SEMI ? L.sxv[j] : L.sx[j]);
    uint32_t yu = xu, yv = xv;
    if (CH != CHROMA_H_NONE)
      yu = bperm (cu[1], cu[0], L.sy[j]), yv = bperm (cv[1], cv[0], SEMI ? L.syv[j] : L.sy[j]);
    if (CH == CHROMA_H_H2_CS)
      u[j] = lerp_u8 (xu, yu, 0x01010101u), v[j] = lerp_u8 (xv, yv, 0x01010101u);
    else if (CH == CHROMA_H_H2)
      u[j] = blend31_u8 (xu, yu), v[j] = blend31_u8 (xv, yv);
    else
      u[j] = xu, v[j] = xv;
  }
}

// second half of the staging, the lane as the owner of 4 OPL pixels of every line: U and V byte planes of the four lines.  Chroma rows:
// A = 2g-1 (carried, h-filtered), B = 2g, C = 2g+1; line 4g-1 = (3 A + B + 2) >> 2, 4g = (A + 3 B + 2) >> 2, 4g+1 = (3 B + C + 2) >> 2,
// 4g+2 = (B + 3 C + 2) >> 2 per byte, each as lerp (heavy, (heavy + light) >> 1) - the inner average is the same for the two lines of a pair.
template <int OPL, int NW, int CH, int SEMI, int ALIGNED, int RW = 0>
GSTAMD_HD void col_stage_chroma (ColLane<OPL, NW> &L, const uint8_t *craw, uint8_t *stage)
{
  typedef ColGeom<OPL> G;
  uint32_t bu[OPL], bv[OPL], cu[OPL], cv[OPL];
  col_hup<OPL, NW, CH, SEMI, ALIGNED> (L, craw, 0, bu, bv);
  col_hup<OPL, NW, CH, SEMI, ALIGNED> (L, craw, 1, cu, cv);
  uint32_t lu[4][OPL], lv[4][OPL];
#pragma unroll
  for (int j = 0; j < OPL; j++) {
    const uint32_t mu = lerp_u8 (L.hu[j], bu[j], 0u), mv = lerp_u8 (L.hv[j], bv[j], 0u);
    const uint32_t nu = lerp_u8 (bu[j], cu[j], 0u), nv = lerp_u8 (bv[j], cv[j], 0u);
    lu[0][j] = lerp_u8 (L.hu[j], mu, 0x01010101u) ^ 0x80808080u, lv[0][j] = lerp_u8 (L.hv[j], mv, 0x01010101u) ^ 0x80808080u;
    lu[1][j] = lerp_u8 (bu[j], mu, 0x01010101u) ^ 0x80808080u, lv[1][j] = lerp_u8 (bv[j], mv, 0x01010101u) ^ 0x80808080u;
    lu[2][j] = lerp_u8 (bu[j], nu, 0x01010101u) ^ 0x80808080u, lv[2][j] = lerp_u8 (bv[j], nv, 0x01010101u) ^ 0x80808080u;
    lu[3][j] = lerp_u8 (cu[j], nu, 0x01010101u) ^ 0x80808080u, lv[3][j] = lerp_u8 (cv[j], nv, 0x01010101u) ^ 0x80808080u;
    L.hu[j] = cu[j], L.hv[j] = cv[j];
  }
  if constexpr (RW) {           /* the words stay with the lane: they ARE its window's first words (col_hfilter_regs) */
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int j = 0; j < OPL; j++)
        L.win[k][1][j] = lu[k][j], L.win[k][2][j] = lv[k][j];
    return;
  }
  uint8_t *d = stage + L.st;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    col_lds_store<OPL, ALIGNED> (d + k * G::LINEB + G::PP, lu[k]);
    col_lds_store<OPL, ALIGNED> (d + k * G::LINEB + 2 * G::PP, lv[k]);
  }
}

// chroma row 2g-1 of a wave's first group, h-filtered: what the staging of group g - 1 would have left in hu / hv.  After col_stage_loads
// of the rows (2g-2, 2g-1) and a sync.
template <int OPL, int NW, int CH, int SEMI, int ALIGNED>
GSTAMD_HD void col_first_row (ColLane<OPL, NW> &L, const uint8_t *craw)
{
  col_hup<OPL, NW, CH, SEMI, ALIGNED> (L, craw, 1, L.hu, L.hv);
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

// The same pass with the windows in REGISTERS (two outputs per lane sharing a window one word apart, the tile's staged bytes at their
// natural place: window word q of lane l is own word q for q < 2, else word (q - 2) & 1 of lane l + 1 + (q - 2) / 2).  The lane's own two
// words of a line and plane it already holds (U / V: col_stage_chroma left them in L.win; Y: one 8-byte LDS read of what the loader lanes
// staged); the neighbours' come over the lanes with DPP wave shifts (v_mov_b32_dpp wave_shl:1 - lane l reads lane l + 1, the last lane
// zero: its window words there only meet zero taps) instead of the five-word LDS read per line and plane of col_hfilter: 424 -> 152 bytes
// of LDS traffic per lane and group, 36 more VALU.  nb (v, k, c, j, dist): the word [k][c][j] of lane + dist; on the device that is the
// shift of v (the caller hands over the lane's own word for dist 1 and the dist-1 result for dist 2), in the emulator a look at that lane.
template <int OPL, int NW, class NB>
GSTAMD_HD void col_hfilter_regs (const ColLane<OPL, NW> &L, NB &nb, uint32_t gw[3][OPL])
{
  static_assert (OPL == 2 && NW <= 4, "register windows: two outputs per lane, windows of up to five words");
  constexpr int NWORDS = NW + 1;
  int acc[4][3][OPL];
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      uint32_t w[5];
      w[0] = L.win[k][c][0], w[1] = L.win[k][c][1];
      w[2] = nb (w[0], k, c, 0, 1);
      w[3] = NWORDS > 3 ? nb (w[1], k, c, 1, 1) : 0u;
      w[4] = NWORDS > 4 ? nb (w[2], k, c, 0, 2) : 0u;
#pragma unroll
      for (int i = 0; i < OPL; i++) {
        int a = L.hinit[i];
#pragma unroll
        for (int q = 0; q < NW; q++)
          a = col_dot4 (w[q + i], L.tw[i][q], a);
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
  if (w->r1 == rb || p.pubn == 0) {
    w->ge = w->gl + 1;                  // the workgroup's last wave makes every group its rows need (pubn == 0: every wave does - no hand-over area)
  } else {
    col_entry8 (p.vrow, w->r1, e);
    w->ge = (int) e[0];
    if (w->ge > w->gl + 1)
      w->ge = w->gl + 1;
  }
  if (w->r0 == ra || p.pubn == 0) {
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
  constexpr int RW = OPL == 2 && WSTEP == 1 && A8 == 2;
  typedef ColGeom<OPL, RW> G;
  typedef ColLane<OPL, NW> Lane;
  typedef ColRaw<OPL> Raw;
  typedef ColRingRegs<OPL, NGV> Ring;
  uint8_t *stage = wave_lds, *craw = wave_lds + G::STAGEB;
  uint32_t *pub = (uint32_t *) (wave_lds + G::STAGEB + GSTAMD_COL_CRAW_BYTES (OPL));
  const uint32_t *pub_below = (const uint32_t *) (below_lds + G::STAGEB + GSTAMD_COL_CRAW_BYTES (OPL));
  const bool aligned = ((tile[3] - tile[2]) % 16) == 0 && tile[3] > 0 && tile[3] + G::SPAN <= p.width;
  // chroma row 2 ga - 1 the way the staging of group ga - 1 would have left it, then the first group's loads
  x.each ([&](int lane, Lane &L, Raw &ra, Ring &) {
    col_setup<OPL, NW, CH, SEMI, RW> (p, tile, lane, L);
    col_request_chroma<OPL, NW, SEMI> (p, s, L, 2 * wp.ga - 2, ra);
    if (aligned)
      col_stage_loads<OPL, NW, SEMI, 1, RW> (L, ra, stage, craw, false);
    else
      col_stage_loads<OPL, NW, SEMI, 0, RW> (L, ra, stage, craw, false);
    col_request<OPL, NW, SEMI> (p, s, L, wp.ga, ra);
  });
  x.sync ();
  x.each ([&](int, Lane &L, Raw &, Ring &) {
    if (aligned)
      col_first_row<OPL, NW, CH, SEMI, 1> (L, craw);
    else
      col_first_row<OPL, NW, CH, SEMI, 0> (L, craw);
  });
  x.sync ();
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
    // the loads' registers go to LDS (luma: final; chroma: raw rows) and take the next group's requests - those have the horizontal pass, the
    // rows and the other waves' turns to arrive
    if (aligned)
      x.each ([&](int, Lane &L, Raw &ra, Ring &) {
        col_stage_loads<OPL, NW, SEMI, 1, RW> (L, ra, stage, craw, true);
        col_request<OPL, NW, SEMI> (p, s, L, gn, ra);
      });
    else
      x.each ([&](int, Lane &L, Raw &ra, Ring &) {
        col_stage_loads<OPL, NW, SEMI, 0, RW> (L, ra, stage, craw, true);
        col_request<OPL, NW, SEMI> (p, s, L, gn, ra);
      });
    x.sync ();
    if (aligned)
      x.each ([&](int, Lane &L, Raw &, Ring &) { col_stage_chroma<OPL, NW, CH, SEMI, 1, RW> (L, craw, stage); });
    else
      x.each ([&](int, Lane &L, Raw &, Ring &) { col_stage_chroma<OPL, NW, CH, SEMI, 0, RW> (L, craw, stage); });
    x.sync ();
    COL_STAMP (2);                      /* staging */
    if constexpr (RW) {
      // the lane's own luma words of the four lines (what the loader lanes staged), then the windows over the lanes
      x.each ([&](int, Lane &L, Raw &, Ring &) {
#pragma unroll
        for (int k = 0; k < 4; k++)
          col_lds_words<2, 1> (stage + L.st + k * G::LINEB, L.win[k][0]);
      });
      x.each_nb ([&](int lane, Lane &L, Raw &, Ring &rg, auto &nb) {
        uint32_t gw[3][OPL];
        col_hfilter_regs<OPL, NW> (L, nb, gw);
        col_ring_put<OPL, NGV> (rg, gw);
        if (g <= wp.gp)
          col_slot_store<OPL> (pub + (size_t) (g - wp.ga) * G::SLOTW, lane, gw);
      });
    } else {
      x.each ([&](int lane, Lane &L, Raw &, Ring &rg) {
        uint32_t gw[3][OPL];
        col_hfilter<OPL, NW, WSTEP, A8> (L, stage, gw);
        col_ring_put<OPL, NGV> (rg, gw);
        if (g <= wp.gp)
          col_slot_store<OPL> (pub + (size_t) (g - wp.ga) * G::SLOTW, lane, gw);
      });
    }
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
