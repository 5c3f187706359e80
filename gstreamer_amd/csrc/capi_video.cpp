// capi_video.cpp - C ABI (include/gstamd_video.h) over the planner and the HIP kernels.
// There is deliberately NO CPU implementation behind these entry points: without a HIP device the
// frame calls fail with GSTAMD_ERR_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <array>
#include <vector>

#include "../../include/gstamd_video.h"
#include "planner.h"
#include "tuning.h"
#include "video_kernels.h"
#include "video_fast.h"
#include "video_encode_fast.h"
#include "video_scale420_fused.h"
#include "video_scale_col.h"
#include "video_dither_ed.h"
#include "video_deep_pack.h"

using namespace gstamd;

static thread_local std::string g_last_error;

static int set_error (int code, const std::string &msg)
{
  g_last_error = msg;
  return code;
}

static int hip_fail (hipError_t e, const char *what)
{
  return set_error (GSTAMD_ERR_HIP, std::string (what) + ": " + hipGetErrorString (e));
}

struct GstAmdVideoConverter {
  VideoPlan plan;
  std::mutex lock;
  bool tables_ready = false;
  int device = -1;
  int *vpair_dev = nullptr;
  struct PassDev {
    uint32_t *offset = nullptr;
    int16_t *taps = nullptr;
    uint32_t *tapw = nullptr;
  } pass_dev[2];
  uint8_t *tmp = nullptr;       // intermediate image between two scaler passes
  size_t tmp_size = 0;
  int tmp_w = 0, tmp_h = 0;
  struct PlaneDev { uint32_t *offset = nullptr; int16_t *taps = nullptr; };
  std::vector<std::vector<PlaneDev>> plane_dev;      // plane mode: tables of every pass of every plane
  uint8_t *plane_tmp = nullptr;           // plane mode: intermediate plane of a two-pass scale
  size_t plane_tmp_bytes = 0;
  size_t plane_lds_bytes = 0;
  bool plane_frame_ok = false;            // the frame's planes go through k_plane_frame (one launch)
  std::vector<int> plane_dstep;
  bool raw4_quad = false;                 // plane_raw4_plan: the 4-byte plane scaler goes through k_plane_quad
  int raw4_dstep = 0;
  bool raw4_pack = false, raw4_pack_enc420 = false;          // plane_raw4_pack_plan
  int raw4_pack_dstep = 0;
  std::vector<bool> plane_quad, plane_oct;        // per plane of the plan: plane_quad_ok / plane_oct_ok (k_plane_quad takes it where the destination rows allow)
  void *ed_carry = nullptr;                // error-diffusion dither on rectangles taller than one band: the band's last line of errors (video_dither_ed.h)
  uint8_t *pre_img = nullptr;             // enlarging from a planar / packed 4:2:2 source: the source frame after front + colour stage, at its own size
  size_t pre_img_size = 0;
  uint8_t *pk_img = nullptr;              // planar destinations: the chain's AYUV image before chroma downsample + pack
  int pk_img_frames = 1;                  // ... as many of them, one behind the other, as a frame list may use at once (video_frame_list_scratch)
  TileGeom geom[2] = {{0, 0}, {0, 0}};   // wave-tile geometry of the horizontal passes
  bool reg420 = false;                    // first pass horizontal from a 4:2:0 source whose chroma pairing is the closed form of h420r_rows
  int reg_lo = 0, reg_hi = 0;
  // k_scale_col (video_scale_col.h): both N-tap passes of a regular 4:2:0 source in one kernel, a wave per column tile
  bool col_ok = false;
  ColTables col;
  ColForm col_form;
  int32_t *col_tiles_dev = nullptr;
  uint32_t *col_hout_dev = nullptr, *col_vrow_dev = nullptr;
  // compositor_walk.h: a pad converter's two 8-tap passes remapped onto rings that advance two source lines / columns per output
  int walk_state = 0;                   // 0: not looked at, 1: tables on the device, -1: the plan does not fit the walk
  uint32_t *walk_vt_dev = nullptr, *walk_ht_dev = nullptr;
  int walk_vbase = 0, walk_hbase = 0;
  int col_waves = 0, col_per_cu = 0, col_cus = 0;
  int col_per_cu_solo = 0;              // the same without the hand-over area (every wave makes all the groups its rows need)
  // k_scale420_fused (video_scale420_fused.h): both N-tap passes of a regular 4:2:0 source in one kernel
  bool fused_ok = false;
  Fused420Tables fused;
  int32_t *vgroup_dev = nullptr;
  uint32_t *vtapw_dev = nullptr;
  int fused_waves = 0, fused_rpc = 0, fused_ring = 0, fused_first = 0, fused_sched = 1;
  uint8_t *deep_a = nullptr, *deep_b = nullptr;        /* scratch images of a scaled 10-bit conversion (convert_deep_scaled) */
  size_t deep_a_size = 0, deep_b_size = 0;
  /* gamma-mode = remap (GammaPlan): the two sub-conversions, the tables and the 8-bit images either side of the 16-bit part */
  GstAmdVideoConverter *sub_in = nullptr, *sub_out = nullptr;
  GstAmdVideoConverter *field[2] = {nullptr, nullptr};        // an interleaved frame's two field conversions (VideoPlan::interlaced): they do all the work
  const uint8_t *post_lut = nullptr;    /* this converter is the direct conversion of a GammaPlan::lut_direct plan: the composed table (device) the parent wants on
                                         * the colour bytes; a kernel that applies it itself says so in post_lut_done, otherwise the parent runs k_lut3 */
  int post_lut_keep = 0;
  bool post_lut_done = false;
  int list_launches = 0;                /* launches of the last _frames call that each served a whole (chunk of the) list; 0: frame by frame */
  bool hook_on = false;                 /* this converter is the direct conversion of a fused gamma plan: k_convert_gamma with `hook` */
  const DeepPackParams *deep_hook = nullptr;         /* this converter is the sub-conversion (narrowed image -> planes) of a shrinking 10-bit plan whose pixels
                                                      * come straight from the 10-bit source: k_deep_scale_pack (video_deep_pack.h) instead of reading an image */
  GammaDev hook;
  uint16_t *gamma_dec_dev = nullptr;
  uint16_t *gamma_dec16_dev = nullptr, *gamma_enc16_dev = nullptr;          /* the 65536-entry tables of a remap with 16-bit ends */
  uint8_t *gamma_enc_dev = nullptr;
  uint8_t *gamma_comp_dev = nullptr;      /* GammaPlan::comp */
  uint8_t *gamma_mid_a = nullptr, *gamma_mid_b = nullptr;
  /* The scratch images above (tmp, plane_tmp, pk_img, ed_carry, deep_a / deep_b, gamma_mid_a / gamma_mid_b) belong to the frames of ONE
   * stream: a second stream's frames would run over them while the first one's kernels are still reading.  Every stream a frame is sent
   * on gets a set of its own (bind_scratch): the members hold the set of `bound_stream`, the others wait in `parked`. */
  struct ScratchSet {
    uint8_t *tmp, *plane_tmp, *pk_img, *deep_a, *deep_b, *gamma_mid_a, *gamma_mid_b, *pre_img;
    void *ed_carry;
    size_t deep_a_size, deep_b_size, pre_img_size;
    int pk_img_frames;
  };
  bool bound = false;
  void *bound_stream = nullptr;
  std::vector<std::pair<void *, ScratchSet>> parked;
};

namespace gstamd {
hipError_t launch_deep_scale_pack (const PackPlanarParams &pk, const DeepPackParams &dp, uint8_t *const planes[3], const int strides[3], hipStream_t stream);
bool deep_scale_pack_usable (const PackPlanarParams &pk, const DeepPackParams &dp, uint8_t *const planes[3], const int strides[3]);
hipError_t launch_deep_scale4 (const DeepPackParams &dp, const Deep16Params &dd, const PostParams &post, uint8_t *dst, int dstride, hipStream_t stream);
bool deep_scale4_usable (const DeepPackParams &dp, const uint8_t *dst, int dstride);
hipError_t launch_deep_scale_pack16 (const PackPlanarParams &pk, int hi_depth, const DitherParams &dt, const DeepPackParams &dp, uint8_t *const planes[3],
    const int strides[3], hipStream_t stream);
bool deep_scale_pack16_usable (const PackPlanarParams &pk, const DeepPackParams &dp, uint8_t *const planes[3], const int strides[3]);
size_t fused420_lds_bytes (int ring, int nwaves, int sched);
int fused420_blocks_per_cu (int nwaves, size_t lds, int sched);
hipError_t launch_scale420_fused (const Fused420Params &p, int chroma_h, int nw, int nwaves, uint8_t *dst, int dstride, const ColorParams &post,
    const int pack_pos[4], const PostFast &pf, hipStream_t stream);
size_t col_lds_bytes (const ColForm &f, int pubn, int nwaves);
int col_blocks_per_cu (const ColForm &f, int chroma_h, int semi, int pubn, int nwaves);
hipError_t launch_scale_col (const ColParams &p, const ColForm &f, int chroma_h, int semi, int nwaves, const ColFrames &fr, int n_frames, int dstride,
    const ColorParams &post, const int pack_pos[4], const PostFast &pf, hipStream_t stream);
}

// geometry of the fused scaler: waves per workgroup, output rows per workgroup (every workgroup of the launch resident at once where
// possible), ring slots.  GSTAMD_FUSED_WAVES / GSTAMD_FUSED_ROWS select other (equally correct) shapes for tuning sessions and tests.
static bool fused_pick_geometry (GstAmdVideoConverter *c, int tiles)
{
  const int out_h = c->plan.out_info.height;
  const int ew = tuning_int ("GSTAMD_FUSED_WAVES", 0), er = tuning_int ("GSTAMD_FUSED_ROWS", 0);
  const int first = ew > 0 ? ew : 16;      /* MI355X, C3: 16 waves 30.8 us, 8: 34.5, 4: 32.7, two-pass 33.7 (profiles/r02_c3_variants.log) */
  int n_cu = 256, dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice (&dev) == hipSuccess && hipGetDeviceProperties (&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    n_cu = prop.multiProcessorCount;
  /* schedule 2 (one barrier per round, the longer ring; Fused420Params::sched) where its LDS fits, schedule 1 otherwise;
     GSTAMD_FUSED_SCHED pins one for tuning sessions and tests */
  const int es = tuning_int ("GSTAMD_FUSED_SCHED", 0);
  for (int sched = es == 1 ? 1 : 2; sched >= (es == 2 ? 2 : 1); sched--)
    for (int nwaves = first; nwaves >= 2; nwaves /= 2) {
      if (nwaves > 16)
        continue;
      const int first = tuning_on ("GSTAMD_FUSED_FIRST") ? std::max (1, tuning_int ("GSTAMD_FUSED_FIRST", 1)) : fused420_first_rows (c->fused, nwaves);
      auto ring_of = [&](int rows) { return sched == 2 ? fused420_ring_groups2 (c->fused, rows, nwaves, first) : fused420_ring_groups (c->fused, rows, nwaves, first); };
      int ring = ring_of (out_h);
      size_t lds = fused420_lds_bytes (ring, nwaves, sched);
      if (lds > 160 * 1024)
        continue;
      const int per_cu = fused420_blocks_per_cu (nwaves, lds, sched);
      if (per_cu <= 0)
        continue;
      const int chunks = std::max (1, per_cu * n_cu / std::max (1, tiles));
      int rpc = er > 0 ? er : (out_h + chunks - 1) / chunks;
      rpc = std::max (rpc, nwaves);
      ring = ring_of (rpc);
      if (fused420_lds_bytes (ring, nwaves, sched) > lds)
        continue;
      c->fused_first = std::min (first, nwaves);
      c->fused_waves = nwaves;
      c->fused_rpc = rpc;
      c->fused_ring = ring;
      c->fused_sched = sched;
      if (tuning_on ("GSTAMD_FUSED_DEBUG"))
        fprintf (stderr, "fused geometry: sched %d waves %d rows/chunk %d first %d ring %d lds %zu per_cu %d tiles %d\n", sched, nwaves, rpc, c->fused_first, ring,
            fused420_lds_bytes (ring, nwaves, sched), per_cu, tiles);
      return true;
    }
  return false;
}

// waves per workgroup of the column-walk scaler: the most for which a CU still holds its full share of waves (occupancy query), so that
// the workgroups' row runs are long and few groups are filtered twice at their seams.  GSTAMD_COL_WAVES pins it.
static bool col_pick_waves (GstAmdVideoConverter *c)
{
  const VideoPlan &p = c->plan;
  const int semi = p.front.kind == UNPACK_SEMI ? 1 : 0;
  int dev = 0;
  hipDeviceProp_t prop;
  c->col_cus = 256;
  if (hipGetDevice (&dev) == hipSuccess && hipGetDeviceProperties (&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    c->col_cus = prop.multiProcessorCount;
  /* MI355X, C3 (profiles/r04/col_sweep.log): four-wave workgroups of the two-outputs-per-lane form 16.4 us per 8K frame in lists of 8, 17.7 in
     lists of 4 (eight waves: 18.8 / 20.1, two: 18.8 / 20.8); single frames 25-27 us whatever the form */
  const int pin = tuning_int ("GSTAMD_COL_WAVES", 0);
  int best_waves = 0, best_per_cu = 0;
  /* four waves where a CU takes such a workgroup at all (MI355X, C3 in lists of 8, profiles/r04/col_waves.log: one wave 173 us per launch, two 149,
     three 143, four 137, five 154, six 155), fewer only where it does not */
  for (int nwaves = pin > 0 ? GSTAMD_COL_MAX_WAVES : 4; nwaves >= 1 && !best_waves; nwaves--) {
    if (pin > 0 && nwaves != pin)
      continue;
    const int per_cu = col_blocks_per_cu (c->col_form, p.front.chroma_h, semi, c->col.pubn, nwaves);
    if (per_cu > 0)
      best_waves = nwaves, best_per_cu = per_cu;
  }
  /* (the register-window form fits five four-wave workgroups a CU by the occupancy query's count - a launch sized for five ran 15.7 us per frame
     against 14.3-14.5 for every other geometry, profiles/r05: sized for four) */
  if (best_waves * best_per_cu > 16)
    best_per_cu = std::max (1, 16 / best_waves);
  c->col_waves = best_waves;
  c->col_per_cu = best_per_cu;
  c->col_per_cu_solo = best_waves > 0 ? col_blocks_per_cu (c->col_form, p.front.chroma_h, semi, 0, best_waves) : 0;
  return best_waves > 0;
}

// rows per wave / workgroup for a launch of n_frames frames: every workgroup of the launch resident at once where the picture allows it
static void col_geometry (const GstAmdVideoConverter *c, int n_frames, ColParams *q)
{
  const int out_h = c->plan.out_info.height, tiles = (int) c->col.tiles.size () / 4;
  const int pin = tuning_int ("GSTAMD_COL_CHUNKS", 0);
  /* solo: no hand-over between the waves of a workgroup - each makes every group its rows need (the groups at a seam twice), and the LDS the
     hand-over slots took buys resident waves.  Worth it when a wave's run of rows is long against the groups a seam shares. */
  const int solo_pin = tuning_int ("GSTAMD_COL_SOLO", -1);
  bool solo = false;
  if (c->col_per_cu_solo > c->col_per_cu && solo_pin != 0) {
    const int cap_solo = std::max (1, c->col_per_cu_solo * c->col_cus);
    const int chunks_solo = std::max (1, cap_solo / std::max (1, tiles * n_frames));
    const int rpw_solo = std::max (1, (out_h + chunks_solo * c->col_waves - 1) / (chunks_solo * c->col_waves));
    /* groups a run of rpw rows needs against the ones a seam adds */
    const double groups_per_row = (double) c->col.n_groups / std::max (1, out_h);
    solo = solo_pin > 0;                /* measured (C3, lists of 16): 16.5 us against 15.5 with the hand-over - the kernel is bound by its LDS and VALU
                                           work, not by resident waves; kept as a knob */
    (void) groups_per_row;
    (void) rpw_solo;
  }
  const int capacity = std::max (1, (solo ? c->col_per_cu_solo : c->col_per_cu) * c->col_cus);
  int chunks = pin > 0 ? pin : std::max (1, capacity / std::max (1, tiles * n_frames));
  /* a workgroup's rows: (waves - 1) runs of rpw rows and the last wave's shorter run (col_rows_last) */
  int rpw = std::max (1, (out_h + chunks * c->col_waves - 1) / (chunks * c->col_waves));
  for (;; rpw++) {
    if (!solo)
      rpw = std::max (rpw, c->col.min_rows_per_wave);
    const int per_wg = rpw * (c->col_waves - 1) + (solo ? rpw : col_rows_last (c->col, rpw, out_h));
    if ((long long) per_wg * chunks >= out_h)
      break;
  }
  q->rows_per_wave = rpw;
  q->nwaves = c->col_waves;
  q->rows_last = solo ? rpw : col_rows_last (c->col, rpw, out_h);
  q->rows_per_wg = rpw * (c->col_waves - 1) + q->rows_last;
  q->n_chunks = (out_h + q->rows_per_wg - 1) / q->rows_per_wg;
  q->n_tiles = tiles;
  q->pubn = solo ? 0 : c->col.pubn;
}

static bool fast_pair_usable (const VideoPlan &p, const Planes &pl, const uint8_t *dst, int dstride, int dalign = 16)
{
  return p.passes.empty () && p.fast_pair && ((uintptr_t) dst % dalign) == 0 && (dstride % dalign) == 0 &&
      ((uintptr_t) pl.p[0] % 4) == 0 && (pl.stride[0] % 4) == 0 && ((uintptr_t) pl.p[1] % 4) == 0 && (pl.stride[1] % 4) == 0;
}

static FastParams make_fast_params (const VideoPlan &p, bool rgb24 = false)
{
  FastParams fp;
  fp.width = p.front.width;
  fp.height = p.front.height;
  fast_params_finish (fp, p.matrix.p, p.post.pack_pos, p.front.u_plane);
  /* with a source crop the chroma upsampler still sees the frame's rows above / below the crop (do_unpack_lines :2966) */
  fp.crow_lo = -(p.rect.in_y >> 1);
  fp.crow_hi = ((p.rect.in_maxh + 1) >> 1) - 1 - (p.rect.in_y >> 1);
  if (rgb24)
    fast_params_rgb24 (fp, p.matrix.p, p.fout->pos, p.front.u_plane);
  return fp;
}

// rectangle complement of every destination plane <- the border pixel (convert_fill_border :7190-7290)
static hipError_t fill_borders (const VideoPlan &p, uint8_t *const planes[4], const int strides[4], hipStream_t stream)
{
  const FormatDesc *f = p.fout;
  const RectPlan &rc = p.rect;
  const int w = p.out_info.width, h = p.out_info.height;
  auto up = [](int v, int sub) { return -((-v) >> sub); };
  hipError_t e = hipSuccess;
  const int n_planes = f->kind == UNPACK_PLANAR_A ? 4 : f->kind == UNPACK_SEMI_LE32 || f->kind == UNPACK_SEMI_LE40 || f->kind == UNPACK_SEMI_TILED || f->kind == UNPACK_SEMI_LE40_TILED ? 2 : f->kind == UNPACK_SEMI_A || f->kind == UNPACK_PLANAR_H4 ? 3 : kind_has_planes (f->kind) ? (f->kind == UNPACK_SEMI ? 2 : 3) : 1;
  for (int i = 0; i < n_planes && e == hipSuccess; i++) {
    int es;
    uint32_t lo, hi;
    border_plane_value (f, rc.border, i, &es, &lo, &hi);
    /* the reference's fastpaths fill with convert_fill_border, whose per-plane values come from packing ONE border pixel (setup_borderline :2256: width
       1 << w_sub = 1 for multi-plane formats) - for NV61 that is pack_NV61's odd-width tail, which stores U before V: every border pair of the plane */
    const bool nv61_fastpath = i == 1 && f->format == GSTAMD_VIDEO_FORMAT_NV61 && !p.ref_fastpath.empty ();
    if (nv61_fastpath)
      lo = (lo >> 8) | ((lo & 0xffu) << 8);
    const bool pairs = f->kind == UNPACK_PACKED422 || f->kind == UNPACK_P422_16;          /* the plane's unit is the macropixel */
    const bool full = i == 0 || i == GSTAMD_KIND_ALPHA_PLANE (f->kind);      /* a plane of one sample per pixel */
    const int ws = !full || pairs ? f->w_sub : 0, hs = !full ? f->h_sub : 0;
    const int mw = up (rc.out_maxw, ws), mh = up (rc.out_maxh, hs);
    e = launch_fill_border (planes[i], strides[i], es, lo, hi, mw, mh, rc.out_x >> ws, rc.out_y >> hs, border_picture_positions (f, rc, w, ws), up (h, hs), stream);
    if (e == hipSuccess && i == 1 && f->format == GSTAMD_VIDEO_FORMAT_NV61 && (rc.out_maxw & 1) && !nv61_fastpath) {
      /* pack_NV61's odd-width tail (video-format.c:2005-2011) stores the last pair of every border line in NV16 order; the picture's rows
         are the packer's (tail_swap) when the rectangle reaches that column */
      const bool reaches = rc.out_x + w == rc.out_maxw;
      e = launch_fill_border (planes[1] + (size_t) (mw - 1) * 2, strides[1], 2, (lo >> 8) | ((lo & 0xff) << 8), 0, 1, mh, 0, rc.out_y >> hs, reaches ? 1 : 0,
          reaches ? up (h, hs) : 0, stream);
    }
    if (e == hipSuccess && f->format == GSTAMD_VIDEO_FORMAT_VYUY && (rc.out_maxw & 1) && p.ref_fastpath.empty () && !p.plane_mode) {
      /* pack_VYUY's odd-width tail (video-format.c:374-380) stores the frame line's last pixel in UYVY order: the last macropixel of every border
         line, and of the picture's lines too when the rectangle ends left of it */
      const bool reaches = rc.out_x + w == rc.out_maxw;
      e = launch_fill_border (planes[0] + (size_t) (mw - 1) * 4, strides[0], 4, ((lo >> 16) & 0xffffu) | (lo << 16), 0, 1, mh, 0, rc.out_y, reaches ? 1 : 0,
          reaches ? h : 0, stream);
    }
  }
  return e;
}

extern "C" {

const char *gstamd_last_error (void) { return g_last_error.c_str (); }
/* the audio and compositor translation units report through the same thread-local (not part of the drop-in surface) */
void gstamd_internal_set_error (const char *msg) { g_last_error = msg ? msg : ""; }

int gstamd_video_info_set_format (GstAmdVideoInfo *info, int format, int width, int height)
{
  int r = video_info_set_format (info, format, width, height);
  if (r != GSTAMD_OK)
    set_error (r, "unsupported format or bad size");
  return r;
}

void gstamd_video_converter_config_init (GstAmdVideoConverterConfig *config)
{
  if (config)
    converter_config_init (config);
}

/* the sub-conversions of a composite plan (gamma remap / the 16-bit part of the chain: GammaPlan) - made for `plan`, handed back
 * through sub_in / sub_out; the description and the divergence note of the plan take theirs on.  Used by _new and by _set_config,
 * which must not keep sub-converters planned with the old options (a new resampler method, rectangle or gamma mode changes them) */
static int build_sub_converters (VideoPlan &plan, GstAmdVideoConverter **sub_in, GstAmdVideoConverter **sub_out)
{
  *sub_in = *sub_out = nullptr;
  if (!plan.gamma.on || plan.gamma.planes_fast)
    return GSTAMD_OK;
  const GammaPlan &g = plan.gamma;
  int st = GSTAMD_OK;
  bool ok = true;
  plan_set_border_override (plan.rect.border);          /* the sub-conversion that fills the borders fills them with THIS plan's border pixel */
  if (!g.src16 && !g.src64) {
    if (g.lut_direct)
      plan_set_matrix_override (&g.to_rgb);
    ok = (*sub_in = gstamd_video_converter_new (&g.sub_in_info, &g.mid_in, &g.cfg_in, &st)) != nullptr;
    plan_set_matrix_override (nullptr);
  }
  if (ok && g.fused) {
    ok = (*sub_in)->plan.passes.empty () && !(*sub_in)->plan.out_planar && !(*sub_in)->plan.plane_mode && !(*sub_in)->plan.gamma.on;
    if (!ok) {
      st = GSTAMD_ERR_UNSUPPORTED;
      g_last_error = "the direct conversion of a fused gamma plan is not a one-kernel plan";
    }
  }
  if (ok && !g.pack16 && !g.store64 && !g.fused)
    ok = (*sub_out = gstamd_video_converter_new (&g.mid_out, &g.sub_out_info, &g.cfg_out, &st)) != nullptr;
  plan_set_border_override (nullptr);
  if (!ok) {
    const std::string why = g_last_error;
    gstamd_video_converter_free (*sub_in);
    gstamd_video_converter_free (*sub_out);
    *sub_in = *sub_out = nullptr;
    set_error (st, "16-bit chain: " + why);
    return st;
  }
  if (*sub_in) {
    plan.description += " <- " + (*sub_in)->plan.description;
    plan.divergence += (*sub_in)->plan.divergence;
  }
  if (*sub_out) {
    plan.description += " -> " + (*sub_out)->plan.description;
    plan.divergence += (*sub_out)->plan.divergence;
  }
  return GSTAMD_OK;
}

GstAmdVideoConverter *gstamd_video_converter_new (const GstAmdVideoInfo *in_info, const GstAmdVideoInfo *out_info,
    const GstAmdVideoConverterConfig *config, int *status)
{
  GstAmdVideoConverter *c = new GstAmdVideoConverter ();
  std::string err;
  int r = plan_video_converter (in_info, out_info, config, &c->plan, &err);
  if (r == GSTAMD_OK && c->plan.interlaced) {
    GstAmdVideoConverterConfig fcfg;
    (void) plan_field_config (in_info, out_info, config, &fcfg);          /* (true: the frame's plan exists) */
    for (int f = 0; f < 2 && r == GSTAMD_OK; f++) {
      GstAmdVideoInfo fin, fout;
      plan_field_infos (in_info, out_info, f, &fin, &fout);
      c->field[f] = gstamd_video_converter_new (&fin, &fout, &fcfg, &r);
    }
    if (r != GSTAMD_OK) {
      gstamd_video_converter_free (c->field[0]);
      gstamd_video_converter_free (c->field[1]);
      c->field[0] = c->field[1] = nullptr;
    }
  } else if (r == GSTAMD_OK)
    r = build_sub_converters (c->plan, &c->sub_in, &c->sub_out);
  else
    set_error (r, err);
  if (status)
    *status = r;
  if (r != GSTAMD_OK) {
    delete c;
    return nullptr;
  }
  return c;
}

static void release_tables (GstAmdVideoConverter *c);
static int build_tables (GstAmdVideoConverter *c);
static int alloc_scratch (GstAmdVideoConverter *c);

// device tables on first use; a failure part-way releases what was allocated (a retry starts from nothing)
static int ensure_tables (GstAmdVideoConverter *c)
{
  std::lock_guard<std::mutex> g (c->lock);
  if (c->tables_ready)
    return GSTAMD_OK;
  const int r = build_tables (c);
  if (r != GSTAMD_OK) {
    const std::string keep = g_last_error;
    release_tables (c);
    g_last_error = keep;
  }
  return r;
}

static int build_tables (GstAmdVideoConverter *c)
{
  hipError_t e;
  VideoPlan &p = c->plan;
  if ((e = hipGetDevice (&c->device)) != hipSuccess)
    return hip_fail (e, "hipGetDevice");
  if (p.gamma.on && !p.gamma.planes_fast) {
    const GammaPlan &g = p.gamma;
    if (!g.dec.empty () && ((e = hipMalloc ((void **) &c->gamma_dec_dev, 256 * sizeof (uint16_t))) != hipSuccess ||
            (e = hipMemcpy (c->gamma_dec_dev, g.dec.data (), 256 * sizeof (uint16_t), hipMemcpyHostToDevice)) != hipSuccess))
      return hip_fail (e, "decode table");
    if (!g.enc.empty () && ((e = hipMalloc ((void **) &c->gamma_enc_dev, 65536)) != hipSuccess ||
            (e = hipMemcpy (c->gamma_enc_dev, g.enc.data (), 65536, hipMemcpyHostToDevice)) != hipSuccess))
      return hip_fail (e, "encode table");
    if (!g.comp.empty () && ((e = hipMalloc ((void **) &c->gamma_comp_dev, 256)) != hipSuccess ||
            (e = hipMemcpy (c->gamma_comp_dev, g.comp.data (), 256, hipMemcpyHostToDevice)) != hipSuccess))
      return hip_fail (e, "composed gamma table");
    if (!g.dec16.empty () && ((e = hipMalloc ((void **) &c->gamma_dec16_dev, 65536 * sizeof (uint16_t))) != hipSuccess ||
            (e = hipMemcpy (c->gamma_dec16_dev, g.dec16.data (), 65536 * sizeof (uint16_t), hipMemcpyHostToDevice)) != hipSuccess))
      return hip_fail (e, "16-bit decode table");
    if (!g.enc16.empty () && ((e = hipMalloc ((void **) &c->gamma_enc16_dev, 65536 * sizeof (uint16_t))) != hipSuccess ||
            (e = hipMemcpy (c->gamma_enc16_dev, g.enc16.data (), 65536 * sizeof (uint16_t), hipMemcpyHostToDevice)) != hipSuccess))
      return hip_fail (e, "16-bit encode table");
  }
  if (!p.vpair.empty ()) {
    if ((e = hipMalloc ((void **) &c->vpair_dev, p.vpair.size () * sizeof (int32_t))) != hipSuccess)
      return hip_fail (e, "hipMalloc(vpair)");
    if ((e = hipMemcpy (c->vpair_dev, p.vpair.data (), p.vpair.size () * sizeof (int32_t), hipMemcpyHostToDevice)) != hipSuccess)
      return hip_fail (e, "hipMemcpy(vpair)");
  }
  for (size_t i = 0; i < p.passes.size () && i < 2; i++) {
    const ScalePass &sp = p.passes[i];
    if ((e = hipMalloc ((void **) &c->pass_dev[i].offset, sp.offset.size () * sizeof (uint32_t))) != hipSuccess)
      return hip_fail (e, "hipMalloc(offset)");
    if ((e = hipMemcpy (c->pass_dev[i].offset, sp.offset.data (), sp.offset.size () * sizeof (uint32_t), hipMemcpyHostToDevice)) != hipSuccess)
      return hip_fail (e, "hipMemcpy(offset)");
    if (!sp.taps.empty ()) {
      if ((e = hipMalloc ((void **) &c->pass_dev[i].taps, sp.taps.size () * sizeof (int16_t))) != hipSuccess)
        return hip_fail (e, "hipMalloc(taps)");
      if ((e = hipMemcpy (c->pass_dev[i].taps, sp.taps.data (), sp.taps.size () * sizeof (int16_t), hipMemcpyHostToDevice)) != hipSuccess)
        return hip_fail (e, "hipMemcpy(taps)");
    }
    if (sp.dot4_ok) {
      if ((e = hipMalloc ((void **) &c->pass_dev[i].tapw, sp.tapw.size () * sizeof (uint32_t))) != hipSuccess)
        return hip_fail (e, "hipMalloc(tapw)");
      if ((e = hipMemcpy (c->pass_dev[i].tapw, sp.tapw.data (), sp.tapw.size () * sizeof (uint32_t), hipMemcpyHostToDevice)) != hipSuccess)
        return hip_fail (e, "hipMemcpy(tapw)");
    }
  }
  for (size_t i = 0; i < p.passes.size (); i++)
    if (p.passes[i].horizontal)
      c->geom[i] = pass_tile_geom (p.passes[i]);
  {
    int lo, hi;
    if (!tuning_on ("GSTAMD_NO_COL") && col_plan_regular (p, &lo, &hi) &&
        col_choose (p.passes[0], p.passes[1], p.front.width, p.front.height, std::max (0, tuning_int ("GSTAMD_COL_OPL", 0)), tuning_int ("GSTAMD_COL_SHARE", 1) != 0,
            &c->col, &c->col_form, !tuning_on ("GSTAMD_COL_NO_REGWIN"))) {
      c->reg_lo = lo;
      c->reg_hi = hi;
      if (col_pick_waves (c)) {
        if ((e = hipMalloc ((void **) &c->col_tiles_dev, (c->col.tiles.size () + 8) * sizeof (int32_t))) != hipSuccess ||
            (e = hipMemset (c->col_tiles_dev, 0, (c->col.tiles.size () + 8) * sizeof (int32_t))) != hipSuccess ||
            (e = hipMemcpy (c->col_tiles_dev, c->col.tiles.data (), c->col.tiles.size () * sizeof (int32_t), hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMalloc ((void **) &c->col_hout_dev, c->col.hout.size () * sizeof (uint32_t))) != hipSuccess ||
            (e = hipMemcpy (c->col_hout_dev, c->col.hout.data (), c->col.hout.size () * sizeof (uint32_t), hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMalloc ((void **) &c->col_vrow_dev, c->col.vrow.size () * sizeof (uint32_t))) != hipSuccess ||
            (e = hipMemcpy (c->col_vrow_dev, c->col.vrow.data (), c->col.vrow.size () * sizeof (uint32_t), hipMemcpyHostToDevice)) != hipSuccess)
          return hip_fail (e, "column scaler tables");
        c->col_ok = true;
        if (tuning_on ("GSTAMD_COL_DEBUG"))
          fprintf (stderr, "k_scale_col: opl %d nw %d ngv %d wstep %d a8 %d (form %d %d %d %d) tiles %zu waves %d per_cu %d lds %zu pubn %d\n", c->col.opl, c->col.nw,
              c->col.ngv, c->col.wstep, c->col.a8, c->col_form.nw, c->col_form.ngv, c->col_form.wstep, c->col_form.a8, c->col.tiles.size () / 4, c->col_waves,
              c->col_per_cu, col_lds_bytes (c->col_form, c->col.pubn, c->col_waves), c->col.pubn);
      }
    }
  }
  if (p.passes.size () == 2 && p.passes[0].horizontal && p.passes[0].kind == SCALE_NTAP && p.passes[0].dot4_ok && c->geom[0].tile16_w > 0 &&
      p.front.chroma_v2 == 1 && kind_has_planes (p.front.kind) && p.front.w_sub == 1 && p.front.h_sub == 1 && !p.matrix_before_scale &&
      (int) p.vpair.size () >= 2 * p.front.height) {
    /* k_hscale420_reg: is the planner's simulated pair table the closed form (every line consumed in order)? */
    c->reg_lo = -(p.rect.in_y >> 1);
    c->reg_hi = ((p.rect.in_maxh + 1) >> 1) - 1 - (p.rect.in_y >> 1);
    bool regular = true;
    for (int y = 0; y < p.front.height && regular; y++) {
      int heavy, light;
      h420r_rows (c->reg_lo, c->reg_hi, y, &heavy, &light);
      const int e0 = p.vpair[2 * y], ta = vpair_row (e0), tb = p.vpair[2 * y + 1];
      const int th = vpair_role (e0) == 0 ? ta : tb, tl = vpair_role (e0) == 0 ? tb : ta;
      regular = th == heavy && tl == light;
    }
    c->reg420 = regular;
    if (regular && !p.passes[1].horizontal && p.passes[1].kind == SCALE_NTAP && p.passes[0].nw >= 3 && p.passes[0].nw <= 5 &&
        !tuning_on ("GSTAMD_NO_FUSED420") && make_fused420_tables (p.passes[1], p.front.height, &c->fused) &&
        fused_pick_geometry (c, (p.passes[0].out_size + c->geom[0].tile16_w - 1) / c->geom[0].tile16_w)) {
      if ((e = hipMalloc ((void **) &c->vgroup_dev, c->fused.vgroup.size () * sizeof (int32_t))) != hipSuccess ||
          (e = hipMemcpy (c->vgroup_dev, c->fused.vgroup.data (), c->fused.vgroup.size () * sizeof (int32_t), hipMemcpyHostToDevice)) != hipSuccess ||
          (e = hipMalloc ((void **) &c->vtapw_dev, c->fused.vtapw.size () * sizeof (uint32_t))) != hipSuccess ||
          (e = hipMemcpy (c->vtapw_dev, c->fused.vtapw.data (), c->fused.vtapw.size () * sizeof (uint32_t), hipMemcpyHostToDevice)) != hipSuccess)
        return hip_fail (e, "fused scaler tables");
      c->fused_ok = true;
    }
  }
  if (p.passes.size () == 2) {
    const ScalePass &s0 = p.passes[0];
    c->tmp_w = s0.horizontal ? s0.out_size : p.in_info.width;
    c->tmp_h = s0.horizontal ? p.in_info.height : s0.out_size;
    c->tmp_size = (size_t) c->tmp_w * 4 * (c->tmp_h + 1);      /* + the spare row k_hscale420_reg sends its out-of-picture lines to */
    /* allocated by the first frame that takes the two-pass form (ensure_tmp): the fused scaler needs no intermediate image */
  }
  {
    PlanePlan raw4;
    c->raw4_quad = plane_raw4_plan (p, &raw4);
    c->raw4_dstep = c->raw4_quad ? plane_quad_dstep (raw4) : 0;
    c->raw4_pack = plane_raw4_pack_plan (p, &raw4, &c->raw4_pack_enc420);
    c->raw4_pack_dstep = c->raw4_pack ? plane_quad_dstep (raw4) : 0;
  }
  if (p.plane_mode) {
    size_t tmp_bytes = 0;
    c->plane_dev.resize (p.planes.size ());
    for (size_t i = 0; i < p.planes.size (); i++) {
      const PlanePlan &pp = p.planes[i];
      c->plane_dev[i].resize (pp.passes.size ());
      for (size_t k = 0; k < pp.passes.size (); k++) {
        const ScalePass &sp = pp.passes[k];
        GstAmdVideoConverter::PlaneDev &pd = c->plane_dev[i][k];
        if ((e = hipMalloc ((void **) &pd.offset, sp.offset.size () * sizeof (uint32_t))) != hipSuccess ||
            (e = hipMemcpy (pd.offset, sp.offset.data (), sp.offset.size () * sizeof (uint32_t), hipMemcpyHostToDevice)) != hipSuccess)
          return hip_fail (e, "plane offsets");
        if (!sp.taps.empty () && ((e = hipMalloc ((void **) &pd.taps, sp.taps.size () * sizeof (int16_t))) != hipSuccess ||
            (e = hipMemcpy (pd.taps, sp.taps.data (), sp.taps.size () * sizeof (int16_t), hipMemcpyHostToDevice)) != hipSuccess))
          return hip_fail (e, "plane taps");
      }
      if (pp.passes.size () == 2) {
        const size_t tw = pp.passes[0].horizontal ? pp.ow : pp.iw, th = pp.passes[0].horizontal ? pp.ih : pp.oh;
        tmp_bytes = std::max (tmp_bytes, tw * th * (size_t) pp.n_elems);
      }
    }
    c->plane_tmp_bytes = tmp_bytes;
    c->plane_lds_bytes = 0;           /* of THIS plan (set_config re-plans in place: a size left over from the old plan is not this one's) */
    /* k_plane_frame: up to three planes, no merged packed-4:2:2 scaler, every tile's first pass inside its LDS */
    c->plane_frame_ok = p.planes.size () <= PLN_MAX_JOBS;
    c->plane_quad.clear ();
    c->plane_oct.clear ();
    c->plane_dstep.clear ();
    for (const PlanePlan &pp : p.planes) {
      c->plane_dstep.push_back (plane_quad_dstep (pp));
      c->plane_quad.push_back (plane_quad_ok (pp));
      c->plane_oct.push_back (plane_quad_ok (pp, 8));
    }
    for (const PlanePlan &pp : p.planes) {
      for (const ScalePass &sp : pp.passes)
        c->plane_frame_ok = c->plane_frame_ok && sp.merged == 0;
      c->plane_lds_bytes = std::max (c->plane_lds_bytes, plane_job_lds_bytes (pp));
      c->plane_frame_ok = c->plane_frame_ok && plane_job_lds_bytes (pp) <= PLN_LDS_BYTES;
    }
  }
  const int r = alloc_scratch (c);
  if (r != GSTAMD_OK)
    return r;
  c->tables_ready = true;
  return GSTAMD_OK;
}

// the images one stream's frames pass through on their way (the members are empty on entry); `tmp` and growth of deep_a / deep_b happen on
// first use (convert_two_pass, convert_deep_scaled)
static int alloc_scratch (GstAmdVideoConverter *c)
{
  hipError_t e;
  const VideoPlan &p = c->plan;
  if (p.gamma.on && !p.gamma.planes_fast) {
    const GammaPlan &g = p.gamma;
    const size_t in_px = (size_t) g.mid_in.width * g.mid_in.height, out_px = (size_t) g.mid_out.width * g.mid_out.height;
    if (!g.fused && !g.src16 && !g.src64 && (e = hipMalloc ((void **) &c->gamma_mid_a, in_px * 4)) != hipSuccess)
      return hip_fail (e, "hipMalloc(8-bit image)");
    if (!g.fused && !g.pack16 && !g.store64 && (e = hipMalloc ((void **) &c->gamma_mid_b, out_px * 4)) != hipSuccess)
      return hip_fail (e, "hipMalloc(8-bit image)");
    if (!g.fused && (!p.passes.empty () || g.pack16 || g.src16 || g.src64 || g.store64)) {
      size_t mid_px = 0;
      if (!p.passes.empty ()) {
        const ScalePass &s0 = p.passes[0];
        mid_px = (size_t) (s0.horizontal ? s0.out_size : g.mid_in.width) * (s0.horizontal ? g.mid_in.height : s0.out_size);
      }
      /* the passes alternate between the two images, starting with deep_a when the first image is the caller's own 16-bit frame (ARGB64 / AYUV64
         sources): either may hold the first pass's result, which a horizontal-first enlargement makes LARGER than both frames (26 x 20 between
         25 x 20 and 26 x 14: deep_a was 160 bytes short and the pass wrote into whatever the allocator had put behind it - the device fuzz's
         seed 863, wrong bytes in some processes only) */
      c->deep_a_size = std::max (std::max (in_px, out_px), mid_px) * 8;
      c->deep_b_size = std::max (std::max (in_px, out_px), mid_px) * 8;
      if ((e = hipMalloc ((void **) &c->deep_a, c->deep_a_size)) != hipSuccess ||
          (!p.passes.empty () && (e = hipMalloc ((void **) &c->deep_b, c->deep_b_size)) != hipSuccess))
        return hip_fail (e, "hipMalloc(16-bit scratch)");
    }
  }
  if (p.plane_mode)
    return GSTAMD_OK;           /* plane_tmp: by the first frame that takes the pass-by-pass form (k_plane_frame needs none) */
  if (p.out_planar) {
    if ((e = hipMalloc ((void **) &c->pk_img, (size_t) p.out_info.width * 4 * (p.out_info.height + 1))) != hipSuccess)      /* + the line past the picture */
      return hip_fail (e, "hipMalloc(pack image)");
    c->pk_img_frames = 1;
  }
  {
    const DitherParams &dp = p.gamma.on && (p.gamma.pack16 || p.gamma.store64) ? p.gamma.dither16 : (p.out_planar ? p.pack.dither : p.dither);
    if (dp.on && (dp.method == GSTAMD_DITHER_FLOYD_STEINBERG || dp.method == GSTAMD_DITHER_SIERRA_LITE) &&
        (e = hipMalloc (&c->ed_carry, (size_t) (p.out_info.width + 4) * 8)) != hipSuccess)
      return hip_fail (e, "hipMalloc(dither carry)");
  }
  return GSTAMD_OK;
}

// bytes of one pack image inside a list of them: the image (+ the line past the picture), rounded so that every image starts on the alignment of the first
static size_t pk_img_bytes (const VideoPlan &p) { return (((size_t) p.out_info.width * 4 * (size_t) (p.out_info.height + 1)) + 255) & ~(size_t) 255; }

static GstAmdVideoConverter::ScratchSet take_scratch (GstAmdVideoConverter *c)
{
  GstAmdVideoConverter::ScratchSet s = {c->tmp, c->plane_tmp, c->pk_img, c->deep_a, c->deep_b, c->gamma_mid_a, c->gamma_mid_b, c->pre_img, c->ed_carry,
    c->deep_a_size, c->deep_b_size, c->pre_img_size, c->pk_img_frames};
  c->pk_img_frames = 1;
  c->tmp = c->plane_tmp = c->pk_img = c->deep_a = c->deep_b = c->gamma_mid_a = c->gamma_mid_b = c->pre_img = nullptr;
  c->ed_carry = nullptr;
  c->deep_a_size = c->deep_b_size = c->pre_img_size = 0;
  return s;
}

static void put_scratch (GstAmdVideoConverter *c, const GstAmdVideoConverter::ScratchSet &s)
{
  c->tmp = s.tmp, c->plane_tmp = s.plane_tmp, c->pk_img = s.pk_img, c->deep_a = s.deep_a, c->deep_b = s.deep_b;
  c->gamma_mid_a = s.gamma_mid_a, c->gamma_mid_b = s.gamma_mid_b, c->ed_carry = s.ed_carry, c->pre_img = s.pre_img;
  c->deep_a_size = s.deep_a_size, c->deep_b_size = s.deep_b_size, c->pre_img_size = s.pre_img_size;
  c->pk_img_frames = s.pk_img_frames;
}

static void free_scratch (GstAmdVideoConverter::ScratchSet &s)
{
  void *all[] = {s.tmp, s.plane_tmp, s.pk_img, s.deep_a, s.deep_b, s.gamma_mid_a, s.gamma_mid_b, s.pre_img, s.ed_carry};
  for (void *q : all)
    if (q)
      (void) hipFree (q);
  memset (&s, 0, sizeof (s));
}

// The frame about to be enqueued goes to `stream`: make the scratch members that stream's set (tables are built by now).  Calls into one
// converter are the caller's to serialise (one streaming thread per element); what may overlap is the work on the streams.
static int bind_scratch (GstAmdVideoConverter *c, void *stream)
{
  std::lock_guard<std::mutex> g (c->lock);
  if (!c->bound) {
    c->bound = true;
    c->bound_stream = stream;
    return GSTAMD_OK;
  }
  if (c->bound_stream == stream)
    return GSTAMD_OK;
  c->parked.emplace_back (c->bound_stream, take_scratch (c));
  c->bound_stream = stream;
  for (size_t i = 0; i < c->parked.size (); i++)
    if (c->parked[i].first == stream) {
      put_scratch (c, c->parked[i].second);
      c->parked.erase (c->parked.begin () + (long) i);
      return GSTAMD_OK;
    }
  return alloc_scratch (c);
}

static int frame_planes_plan_order (GstAmdVideoConverter *c, const void *const src_planes[GSTAMD_VIDEO_MAX_PLANES],
    const int32_t src_stride[GSTAMD_VIDEO_MAX_PLANES], void *const dest_planes[GSTAMD_VIDEO_MAX_PLANES],
    const int32_t dest_stride[GSTAMD_VIDEO_MAX_PLANES], void *stream_);
// the chain up to a packed 4-byte image: unpack, chroma upsample, scale, matrix, alpha, byte order
static int convert_to_packed (GstAmdVideoConverter *c, const Planes &pl, uint8_t *dst, int dstride, hipStream_t stream);
static int convert_rect (GstAmdVideoConverter *c, const Planes &pl, void *const dest_planes[GSTAMD_VIDEO_MAX_PLANES],
    const int32_t dest_stride[GSTAMD_VIDEO_MAX_PLANES], hipStream_t stream);

// does this composite plan go through k_deep_scale_pack (video_deep_pack.h)?  *dp: everything but the pointers
static bool deep_pack_usable (const GstAmdVideoConverter *c, DeepPackParams *dp)
{
  return c->sub_out && !c->hook_on && !tuning_on ("GSTAMD_NO_DEEP_SCALE_PACK") && deep_scale_pack_plan_ok (c->plan, c->sub_out->plan, dp);
}

// gamma-mode = remap (GammaPlan, video_gamma.h): in -> 8-bit unpack-format image -> [decode | scalers on ARGB64 | primaries, alpha |
// encode] -> 8-bit unpack-format image -> out.  The images live in HBM (correctness and coverage first).
static int convert_gamma (GstAmdVideoConverter *c, const void *const src_planes[GSTAMD_VIDEO_MAX_PLANES], const int32_t src_stride[GSTAMD_VIDEO_MAX_PLANES],
    void *const dest_planes[GSTAMD_VIDEO_MAX_PLANES], const int32_t dest_stride[GSTAMD_VIDEO_MAX_PLANES], hipStream_t stream)
{
  const VideoPlan &p = c->plan;
  const GammaPlan &g = p.gamma;
  if (g.planes_fast) {
    DeepPlanesPtrs pp;
    memset (&pp, 0, sizeof (pp));
    for (int i = 0; i < p.in_info.n_planes && i < 3; i++) {
      pp.in[i] = (const uint8_t *) src_planes[i];
      pp.in_stride[i] = src_stride ? src_stride[i] : p.in_info.stride[i];
    }
    for (int i = 0; i < p.out_info.n_planes && i < 3; i++) {
      pp.out[i] = (uint8_t *) dest_planes[i];
      pp.out_stride[i] = dest_stride ? dest_stride[i] : p.out_info.stride[i];
      if (!pp.out[i])
        return set_error (GSTAMD_ERR_INVALID, "NULL destination plane");
    }
    for (int i = 0; i < p.in_info.n_planes && i < 3; i++)
      if (!pp.in[i])
        return set_error (GSTAMD_ERR_INVALID, "NULL source plane");
    pp.vec = 1;
    for (int i = 0; i < 3; i++)
      pp.vec = pp.vec && ((uintptr_t) pp.in[i] % 16) == 0 && (pp.in_stride[i] % 16) == 0 && ((uintptr_t) pp.out[i] % 16) == 0 && (pp.out_stride[i] % 16) == 0;
    const hipError_t pe = launch_deep_planes (g.planes, pp, stream);
    return pe == hipSuccess ? GSTAMD_OK : hip_fail (pe, "k_deep_planes");
  }
  const int in_w = g.mid_in.width, in_h = g.mid_in.height, out_w = g.mid_out.width, out_h = g.mid_out.height;
  GammaDev gd;
  gd.to_rgb = g.to_rgb;
  gd.to_yuv = g.to_yuv;
  gd.prim = g.prim;
  gd.alpha_kind = g.alpha_kind;
  gd.alpha_value = g.alpha_value;
  gd.dec = c->gamma_dec_dev;
  gd.enc = c->gamma_enc_dev;
  gd.comp = tuning_on ("GSTAMD_NO_GAMMA_COMP") ? nullptr : c->gamma_comp_dev;
  gd.to_rgb16 = g.to_rgb16;
  gd.to_yuv16 = g.to_yuv16;
  gd.dec16 = c->gamma_dec16_dev;
  gd.enc16 = c->gamma_enc16_dev;
  const bool dec16 = !g.dec16.empty (), enc16 = !g.enc16.empty ();          /* gamma remap with a 16-bit source / destination */
  if (g.fused && g.lut_direct) {
    /* the direct conversion with its own kernels, then the composed table over the converted rectangle */
    c->sub_in->hook_on = false;
    c->sub_in->post_lut = c->gamma_comp_dev;
    c->sub_in->post_lut_keep = p.fout->pos[0];
    c->sub_in->post_lut_done = false;
    const int dr = frame_planes_plan_order (c->sub_in, src_planes, src_stride, dest_planes, dest_stride, stream);
    if (dr != GSTAMD_OK || c->sub_in->post_lut_done)
      return dr;
    const int ds = dest_stride ? dest_stride[0] : p.out_info.stride[0];
    uint8_t *rect = (uint8_t *) dest_planes[0] + plane_origin (p.fout, 0, p.rect.out_x, p.rect.out_y, ds);
    const hipError_t le = launch_lut3 (rect, ds, p.out_info.width, p.out_info.height, c->gamma_comp_dev, p.fout->pos[0], stream);
    return le == hipSuccess ? GSTAMD_OK : hip_fail (le, "k_lut3");
  }
  if (g.fused) {
    c->sub_in->hook = gd;
    c->sub_in->hook_on = true;
    return frame_planes_plan_order (c->sub_in, src_planes, src_stride, dest_planes, dest_stride, stream);
  }
  Enc16Params ep16;
  if (enc16_params (p, &ep16) && !tuning_on ("GSTAMD_NO_ENCODE16") && src_planes[0]) {
    /* unscaled, 4-byte 8-bit source, deep planar destination: one kernel from the frame to the frame (k_encode16) where the rows sit on the
       words it moves */
    const int ss = src_stride ? src_stride[0] : p.in_info.stride[0];
    const uint8_t *sp = (const uint8_t *) src_planes[0] + plane_origin (p.fin, 0, p.rect.in_x, p.rect.in_y, ss);
    uint8_t *planes[3] = {nullptr, nullptr, nullptr};
    int strides[3] = {0, 0, 0};
    bool ok = ((uintptr_t) sp % 16) == 0 && (ss % 16) == 0;
    for (int i = 0; i < p.out_info.n_planes && i < 3; i++) {
      planes[i] = (uint8_t *) dest_planes[i];
      strides[i] = dest_stride ? dest_stride[i] : p.out_info.stride[i];
      if (!planes[i])
        return set_error (GSTAMD_ERR_INVALID, "NULL destination plane");
    }
    uint8_t *rect[3] = {nullptr, nullptr, nullptr};
    for (int i = 0; i < p.out_info.n_planes && i < 3; i++) {
      rect[i] = planes[i] + plane_origin (p.fout, i, p.rect.out_x, p.rect.out_y, strides[i]);
      ok = ok && ((uintptr_t) rect[i] % 8) == 0 && (strides[i] % 8) == 0;
    }
    if (ok) {
      hipError_t e16;
      if (p.rect.fill && (e16 = fill_borders (p, planes, strides, stream)) != hipSuccess)
        return hip_fail (e16, "k_fill_border");
      e16 = launch_encode16 (ep16, sp, ss, rect, strides, stream);
      return e16 == hipSuccess ? GSTAMD_OK : hip_fail (e16, "k_encode16");
    }
  }
  const bool has_mid = g.prim.has_matrix || g.alpha_kind != ALPHA_NONE;
  const size_t n = p.passes.size ();
  hipError_t e;
  int r;
  Deep16Image cur = {nullptr, 0, 0, 0};
  bool mid_done = !has_mid;
  bool cur_is_source = false;           /* an ARGB64 / AYUV64 source frame is the first image: stages must not run in place on it */
  size_t first_pass = 0;                /* 1: the first pass already ran, fused with the 16-bit front */
  if (g.src64) {
    if (!src_planes[0])
      return set_error (GSTAMD_ERR_INVALID, "NULL source plane");
    cur.stride = src_stride ? src_stride[0] : p.in_info.stride[0], cur.width = in_w, cur.height = in_h;
    cur.p = (const uint8_t *) src_planes[0] + plane_origin (p.fin, 0, p.rect.in_x, p.rect.in_y, cur.stride);          /* the source crop */
    cur_is_source = true;
  } else if (g.src16) {
    /* 10-bit source: the front of video_deep.h (unpack + chroma upsample) into an AYUV64 image */
    Planes pl;
    memset (&pl, 0, sizeof (pl));
    for (int i = 0; i < p.in_info.n_planes; i++) {
      pl.p[i] = (const uint8_t *) src_planes[i];
      pl.stride[i] = src_stride ? src_stride[i] : p.in_info.stride[i];
      if (!pl.p[i])
        return set_error (GSTAMD_ERR_INVALID, "NULL source plane");
      pl.p[i] += plane_origin (p.fin, i, p.rect.in_x, p.rect.in_y, pl.stride[i]);          /* the source crop (the vpair table knows the frame's rows) */
    }
    DeepPackParams dsp;
    if (!c->hook_on && !tuning_on ("GSTAMD_NO_DEEP_SCALE_PACK") && deep_scale_pack16_plan_ok (p, &dsp)) {
      /* ... into a 10 / 12 / 16-bit planar destination: front, both passes, 16-bit chroma downsamplers, dither and pack in one kernel (k_deep_scale_pack16);
         borders and the rectangle's origin as the tail of this function has them */
      uint8_t *bp4[4] = {nullptr, nullptr, nullptr, nullptr}, *rp[3] = {nullptr, nullptr, nullptr};
      int bs4[4] = {0, 0, 0, 0}, rs[3] = {0, 0, 0};
      bool all = true;
      for (int i = 0; i < p.out_info.n_planes && i < 3; i++) {
        bp4[i] = (uint8_t *) dest_planes[i];
        bs4[i] = rs[i] = dest_stride ? dest_stride[i] : p.out_info.stride[i];
        all = all && bp4[i];
        rp[i] = bp4[i] ? bp4[i] + plane_origin (p.fout, i, p.rect.out_x, p.rect.out_y, bs4[i]) : nullptr;
      }
      dsp.pl = pl;
      dsp.vpair = c->vpair_dev;
      dsp.sh.offset = c->pass_dev[0].offset, dsp.sh.taps = c->pass_dev[0].taps;
      dsp.sv.offset = c->pass_dev[1].offset, dsp.sv.taps = c->pass_dev[1].taps;
      if (all && deep_scale_pack16_usable (g.pack, dsp, rp, rs)) {
        if (p.rect.fill && (e = fill_borders (p, bp4, bs4, stream)) != hipSuccess)
          return hip_fail (e, "k_fill_border");
        e = launch_deep_scale_pack16 (g.pack, g.pack_hi_depth, g.dither16, dsp, rp, rs, stream);
        return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "k_deep_scale_pack16");
      }
    }
    bool dsp_ok = c->sub_out && deep_pack_usable (c, &dsp);
    if (dsp_ok) {               /* the frames' side of the question: the kernel's 16-byte loads and 4-byte stores */
      const VideoPlan &sp = c->sub_out->plan;
      uint8_t *rp[3] = {nullptr, nullptr, nullptr};
      int rs[3] = {0, 0, 0};
      for (int i = 0; i < sp.out_info.n_planes && i < 3; i++) {
        rs[i] = dest_stride ? dest_stride[i] : sp.out_info.stride[i];
        dsp_ok = dsp_ok && dest_planes[i];
        rp[i] = dest_planes[i] ? (uint8_t *) dest_planes[i] + plane_origin (sp.fout, i, sp.rect.out_x, sp.rect.out_y, rs[i]) : nullptr;
      }
      dsp.pl = pl;
      dsp_ok = dsp_ok && deep_scale_pack_usable (sp.pack, dsp, rp, rs);
    }
    if (dsp_ok) {
      /* front, both passes, narrowing and the sub-conversion's pack in one kernel (video_deep_pack.h): the sub-conversion does what it does around its
         pack (destination rectangle, borders) and takes the pixels from the hook instead of the narrowed image */
      dsp.pl = pl;
      dsp.vpair = c->vpair_dev;
      dsp.sh.offset = c->pass_dev[0].offset, dsp.sh.taps = c->pass_dev[0].taps;
      dsp.sv.offset = c->pass_dev[1].offset, dsp.sv.taps = c->pass_dev[1].taps;
      const void *hb[GSTAMD_VIDEO_MAX_PLANES] = {c->gamma_mid_b, nullptr, nullptr, nullptr};
      const int32_t hbs[GSTAMD_VIDEO_MAX_PLANES] = {out_w * 4, 0, 0, 0};
      c->sub_out->deep_hook = &dsp;
      r = frame_planes_plan_order (c->sub_out, hb, hbs, dest_planes, dest_stride, stream);
      c->sub_out->deep_hook = nullptr;
      return r;
    }
    if (n >= 1 && p.passes[0].horizontal && !dec16 && (mid_done || g.shrink) && front_hscale16_usable (p.front)) {
      /* nothing between the front and a first, horizontal pass: the front runs inside it (k_front_hscale16), no full-size AYUV64 image */
      ScaleDev sd;
      memset (&sd, 0, sizeof (sd));
      sd.kind = p.passes[0].kind;
      sd.n_taps = p.passes[0].n_taps;
      sd.inc = p.passes[0].inc;
      sd.offset = c->pass_dev[0].offset;
      sd.taps = c->pass_dev[0].taps;
      const int ow = p.passes[0].out_size;
      if ((e = launch_front_hscale16 (p.front, pl, c->vpair_dev, sd, c->deep_b, ow * 8, ow, stream)) != hipSuccess)
        return hip_fail (e, "k_front_hscale16");
      cur.p = c->deep_b, cur.stride = ow * 8, cur.width = ow, cur.height = in_h;
      first_pass = 1;
    } else {
      if ((e = launch_front16 (p.front, pl, c->vpair_dev, c->deep_a, in_w * 8, stream)) != hipSuccess)
        return hip_fail (e, "k_front16");
      cur.p = c->deep_a, cur.stride = in_w * 8, cur.width = in_w, cur.height = in_h;
    }
  } else {
    void *ma[GSTAMD_VIDEO_MAX_PLANES] = {c->gamma_mid_a, nullptr, nullptr, nullptr};
    const int32_t mas[GSTAMD_VIDEO_MAX_PLANES] = {in_w * 4, 0, 0, 0};
    if ((r = frame_planes_plan_order (c->sub_in, src_planes, src_stride, ma, mas, stream)) != GSTAMD_OK)
      return r;
    if (n == 0 && !g.pack16 && !g.store64) {
      /* nothing between the tables: one launch from image to image */
      if ((e = launch_gamma_stage (gd, GAMMA_STAGE_DEC | GAMMA_STAGE_MID | GAMMA_STAGE_ENC, c->gamma_mid_a, in_w * 4, c->gamma_mid_b, out_w * 4, out_w, out_h,
                  stream)) != hipSuccess)
        return hip_fail (e, "k_gamma_stage");
      cur.p = nullptr;
    } else {
      const bool mid_now = !mid_done && (n == 0 || !g.shrink);
      if ((e = launch_gamma_stage (gd, GAMMA_STAGE_DEC | (mid_now ? GAMMA_STAGE_MID : 0), c->gamma_mid_a, in_w * 4, c->deep_a, in_w * 8, in_w, in_h, stream)) !=
          hipSuccess)
        return hip_fail (e, "k_gamma_stage(decode)");
      mid_done = mid_done || mid_now;
      cur.p = c->deep_a, cur.stride = in_w * 8, cur.width = in_w, cur.height = in_h;
    }
  }
  if (cur.p && dec16) {
    /* 16-bit source: the matrix to R'G'B' + the 16 -> 16 decode table on the first image (never in place on the caller's frame), with the
       convert stage where it comes before the scalers */
    const bool mid_now = !mid_done && (n == 0 || !g.shrink);
    uint8_t *md = cur_is_source ? c->deep_a : (uint8_t *) cur.p;
    const int ms = cur_is_source ? cur.width * 8 : cur.stride;
    if ((e = launch_gamma_stage (gd, GAMMA_STAGE_DEC16 | (mid_now ? GAMMA_STAGE_MID : 0), cur.p, cur.stride, md, ms, cur.width, cur.height, stream)) != hipSuccess)
      return hip_fail (e, "k_gamma_stage(decode16)");
    cur.p = md, cur.stride = ms;
    cur_is_source = false;
    mid_done = mid_done || mid_now;
  }
  if (cur.p) {
    if (!mid_done && (n == 0 || !g.shrink)) {           /* the convert stage before the scalers (or no scalers): in place */
      uint8_t *md = cur_is_source ? c->deep_a : (uint8_t *) cur.p;
      const int ms = cur_is_source ? cur.width * 8 : cur.stride;
      if ((e = launch_gamma_stage (gd, GAMMA_STAGE_MID, cur.p, cur.stride, md, ms, cur.width, cur.height, stream)) != hipSuccess)
        return hip_fail (e, "k_gamma_stage(convert)");
      cur.p = md, cur.stride = ms;
      cur_is_source = false;
      mid_done = true;
    }
    for (size_t i = first_pass; i < n; i++) {
      ScaleDev sd;
      memset (&sd, 0, sizeof (sd));
      sd.kind = p.passes[i].kind;
      sd.n_taps = p.passes[i].n_taps;
      sd.inc = p.passes[i].inc;
      sd.offset = c->pass_dev[i].offset;
      sd.taps = c->pass_dev[i].taps;
      sd.tapw = c->pass_dev[i].tapw;
      sd.nw = p.passes[i].nw;
      sd.nw4 = p.passes[i].nw4;
      const bool hz = p.passes[i].horizontal;
      const int ow = hz ? p.passes[i].out_size : cur.width, oh = hz ? cur.height : p.passes[i].out_size;
      uint8_t *dst = cur.p == c->deep_a ? c->deep_b : c->deep_a;
      if ((size_t) ow * 8 * oh > (dst == c->deep_a ? c->deep_a_size : c->deep_b_size))
        return set_error (GSTAMD_ERR_INVALID, "16-bit scratch image smaller than a pass's result");
      if ((e = launch_scale16 (cur, sd, hz, dst, ow * 8, ow, oh, nullptr, nullptr, stream)) != hipSuccess)
        return hip_fail (e, "k_scale16");
      if (tuning_on ("GSTAMD_DEEP_DEBUG")) {          /* what each 16-bit pass read and wrote: sums of its tables, its source rows and its result */
        (void) hipStreamSynchronize (stream);
        std::vector<uint8_t> hb ((size_t) ow * 8 * oh), ht (p.passes[i].taps.size () * 2), ho (p.passes[i].offset.size () * 4);
        (void) hipMemcpy (hb.data (), dst, hb.size (), hipMemcpyDeviceToHost);
        (void) hipMemcpy (ht.data (), sd.taps, ht.size (), hipMemcpyDeviceToHost);
        (void) hipMemcpy (ho.data (), sd.offset, ho.size (), hipMemcpyDeviceToHost);
        unsigned long long sb = 0, st = 0, so = 0, stp = 0, sop = 0;
        for (size_t k = 0; k < hb.size (); k++) sb = sb * 31 + hb[k];
        for (size_t k = 0; k < ht.size (); k++) st = st * 31 + ht[k];
        for (size_t k = 0; k < ho.size (); k++) so = so * 31 + ho[k];
        const uint8_t *pt = (const uint8_t *) p.passes[i].taps.data (), *po = (const uint8_t *) p.passes[i].offset.data ();
        for (size_t k = 0; k < ht.size (); k++) stp = stp * 31 + pt[k];
        for (size_t k = 0; k < ho.size (); k++) sop = sop * 31 + po[k];
        fprintf (stderr, "deep pass %zu (%s, kind %d, %d taps) %dx%d <- %dx%d stride %d src %p%s: result %016llx, taps dev %016llx plan %016llx, offsets dev %016llx plan %016llx, dst %p taps %p offsets %p\n",
            i, hz ? "h" : "v", sd.kind, sd.n_taps, ow, oh, cur.width, cur.height, cur.stride, (const void *) cur.p, cur_is_source ? " (source)" : "", sb, st, stp, so, sop,
            (void *) dst, (const void *) sd.taps, (const void *) sd.offset);
      }
      cur.p = dst, cur.stride = ow * 8, cur.width = ow, cur.height = oh;
      cur_is_source = false;
    }
    if (g.pack16 || g.store64) {
      if (!mid_done || enc16) {
        uint8_t *md = cur_is_source ? c->deep_a : (uint8_t *) cur.p;
        const int ms = cur_is_source ? cur.width * 8 : cur.stride;
        if ((e = launch_gamma_stage (gd, (mid_done ? 0 : GAMMA_STAGE_MID) | (enc16 ? GAMMA_STAGE_ENC16 : 0), cur.p, cur.stride, md, ms, cur.width, cur.height,
                    stream)) != hipSuccess)
          return hip_fail (e, "k_gamma_stage(convert / encode16)");
        cur.p = md, cur.stride = ms;
      }
    } else if ((e = launch_gamma_stage (gd, (mid_done ? 0 : GAMMA_STAGE_MID) | GAMMA_STAGE_ENC, cur.p, cur.stride, c->gamma_mid_b, out_w * 4, out_w, out_h,
                    stream)) != hipSuccess) {
      return hip_fail (e, "k_gamma_stage(encode)");
    }
  }
  if (g.store64 || g.pack16) {
    /* a 16-bit destination frame: borders around the rectangle first (the border pixel packed like any other, border_plane_value) */
    uint8_t *planes[4] = {nullptr, nullptr, nullptr, nullptr};
    int strides[4] = {0, 0, 0, 0};
    for (int i = 0; i < p.out_info.n_planes && i < 4; i++) {
      planes[i] = (uint8_t *) dest_planes[i];
      strides[i] = dest_stride ? dest_stride[i] : p.out_info.stride[i];
      if (!planes[i])
        return set_error (GSTAMD_ERR_INVALID, "NULL destination plane");
    }
    /* (a v210 frame's borders are its packer's own: groups of six pixels hold border and picture - PackPlanarParams::frame_on) */
    const bool frame_pack = g.pack16 && g.pack.kind == UNPACK_V210 && g.pack.frame_on == 2;
    if (p.rect.fill && !frame_pack && (e = fill_borders (p, planes, strides, stream)) != hipSuccess)
      return hip_fail (e, "k_fill_border");
    for (int i = 0; i < p.out_info.n_planes && i < 3 && !frame_pack; i++)
      planes[i] += plane_origin (p.fout, i, p.rect.out_x, p.rect.out_y, strides[i]);
    if (g.store64) {
      /* the last 16-bit image is the picture (pack_ARGB64 / pack_AYUV64 are copies at native endianness) */
      e = hipMemcpy2DAsync (planes[0], (size_t) strides[0], cur.p, (size_t) cur.stride, (size_t) out_w * 8, (size_t) out_h, hipMemcpyDeviceToDevice, stream);
      if (e == hipSuccess && g.dither16.on)
        e = launch_dither16_any (g.dither16, planes[0], strides[0], out_w, out_h, stream, c->ed_carry);
      return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "copy of the 16-bit image");
    }
    if (dither_is_diffusion (g.dither16)) {
      /* error diffusion: chroma downsamplers, then the dither pass, in place on the image - never on the caller's frame */
      if (cur_is_source) {
        if ((size_t) out_w * 8 * out_h > c->deep_a_size)
          return set_error (GSTAMD_ERR_INVALID, "16-bit scratch image smaller than the picture");
        if ((e = hipMemcpy2DAsync (c->deep_a, (size_t) out_w * 8, cur.p, (size_t) cur.stride, (size_t) out_w * 8, (size_t) out_h, hipMemcpyDeviceToDevice, stream)) != hipSuccess)
          return hip_fail (e, "copy of the 16-bit source");
        cur.p = c->deep_a, cur.stride = out_w * 8;
        cur_is_source = false;
      }
      e = launch_pack16_ed (g.pack, g.pack_hi_depth, g.dither16, (uint8_t *) cur.p, cur.stride, planes, strides, stream, c->ed_carry);
      if (e == hipSuccess && p.fout->kind == UNPACK_PLANAR_A) {
        DitherParams none;
        memset (&none, 0, sizeof (none));           /* the error-diffusion pass ran over all four components of the image */
        const int sa = dest_stride ? dest_stride[3] : p.out_info.stride[3];
        e = launch_pack16_alpha_plane (g.pack, g.pack_hi_depth, none, cur.p, cur.stride, (uint8_t *) dest_planes[3] + plane_origin (p.fout, 3, p.rect.out_x, p.rect.out_y, sa), sa, stream);
      }
      return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "k_pack16 (error diffusion)");
    }
    e = launch_pack16 (g.pack, g.pack_hi_depth, g.dither16, cur.p, cur.stride, planes, strides, stream);
    if (e == hipSuccess && p.fout->kind == UNPACK_PLANAR_A) {
      const int sa = dest_stride ? dest_stride[3] : p.out_info.stride[3];
      e = launch_pack16_alpha_plane (g.pack, g.pack_hi_depth, g.dither16, cur.p, cur.stride, (uint8_t *) dest_planes[3] + plane_origin (p.fout, 3, p.rect.out_x, p.rect.out_y, sa), sa, stream);
    }
    return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "k_pack16");
  }
  const void *mb[GSTAMD_VIDEO_MAX_PLANES] = {c->gamma_mid_b, nullptr, nullptr, nullptr};
  const int32_t mbs[GSTAMD_VIDEO_MAX_PLANES] = {out_w * 4, 0, 0, 0};
  return frame_planes_plan_order (c->sub_out, mb, mbs, dest_planes, dest_stride, stream);
}

// The public entry takes the planes as a GstVideoFrame carries them in data[] / GstVideoMeta: GBR's are G, B, R.  Inside a plan they are
// R, G, B (format_plan_planes) - what _frame / _frames derive from the plan's own offsets and what every launcher expects.
int gstamd_video_converter_frame_planes (GstAmdVideoConverter *c, const void *const src_planes[GSTAMD_VIDEO_MAX_PLANES],
    const int32_t src_stride[GSTAMD_VIDEO_MAX_PLANES], void *const dest_planes[GSTAMD_VIDEO_MAX_PLANES],
    const int32_t dest_stride[GSTAMD_VIDEO_MAX_PLANES], void *stream)
{
  if (!c || !src_planes || !dest_planes)
    return set_error (GSTAMD_ERR_INVALID, "NULL converter or frame");
  int perm_i[4], perm_o[4];
  const bool in_perm = c->plan.fin && format_plane_perm (c->plan.fin->format, perm_i);
  const bool out_perm = c->plan.fout && format_plane_perm (c->plan.fout->format, perm_o);
  if (!in_perm && !out_perm)
    return frame_planes_plan_order (c, src_planes, src_stride, dest_planes, dest_stride, stream);
  const void *sp[GSTAMD_VIDEO_MAX_PLANES];
  void *dp[GSTAMD_VIDEO_MAX_PLANES];
  int32_t ss[GSTAMD_VIDEO_MAX_PLANES], ds[GSTAMD_VIDEO_MAX_PLANES];
  for (int i = 0; i < GSTAMD_VIDEO_MAX_PLANES; i++) {
    const int si = in_perm ? perm_i[i] : i, di = out_perm ? perm_o[i] : i;          /* plan plane i <- frame plane perm[i] */
    sp[i] = src_planes[si];
    dp[i] = dest_planes[di];
    if (src_stride)
      ss[i] = src_stride[si];
    if (dest_stride)
      ds[i] = dest_stride[di];
  }
  return frame_planes_plan_order (c, sp, src_stride ? ss : nullptr, dp, dest_stride ? ds : nullptr, stream);
}

static int frame_planes_plan_order (GstAmdVideoConverter *c, const void *const src_planes[GSTAMD_VIDEO_MAX_PLANES],
    const int32_t src_stride[GSTAMD_VIDEO_MAX_PLANES], void *const dest_planes[GSTAMD_VIDEO_MAX_PLANES],
    const int32_t dest_stride[GSTAMD_VIDEO_MAX_PLANES], void *stream_)
{
  if (!c || !src_planes || !dest_planes || !src_planes[0] || !dest_planes[0])
    return set_error (GSTAMD_ERR_INVALID, "NULL converter or frame");
  if (c->plan.interlaced) {
    /* an interleaved frame: field f is lines f, f + 2, ... of every plane (plan_field_infos) - except the SOURCE chroma planes of a field plan whose
       pair table names rows of the frame's chroma planes (VideoPlan::field_src_chroma_frame) */
    const VideoPlan &fp = c->plan;
    for (int f = 0; f < 2; f++) {
      GstAmdVideoConverter *fc = c->field[f];
      if (!fc)
        return set_error (GSTAMD_ERR_INVALID, "interlaced converter without its field conversions");
      const void *sp[GSTAMD_VIDEO_MAX_PLANES] = {nullptr, nullptr, nullptr, nullptr};
      void *dp[GSTAMD_VIDEO_MAX_PLANES] = {nullptr, nullptr, nullptr, nullptr};
      int32_t ss[GSTAMD_VIDEO_MAX_PLANES] = {0, 0, 0, 0}, ds[GSTAMD_VIDEO_MAX_PLANES] = {0, 0, 0, 0};
      const int alpha_plane = GSTAMD_KIND_ALPHA_PLANE (fp.fin->kind);
      for (int i = 0; i < fp.in_info.n_planes; i++) {
        const int st = src_stride ? src_stride[i] : fp.in_info.stride[i];
        const bool frame_rows = fc->plan.field_src_chroma_frame && i != 0 && i != alpha_plane;
        if (!src_planes[i])
          return set_error (GSTAMD_ERR_INVALID, "NULL source plane");
        sp[i] = frame_rows ? src_planes[i] : (const uint8_t *) src_planes[i] + (f ? st : 0);
        ss[i] = frame_rows ? st : 2 * st;
      }
      for (int i = 0; i < fp.out_info.n_planes; i++) {
        const int st = dest_stride ? dest_stride[i] : fp.out_info.stride[i];
        if (!dest_planes[i])
          return set_error (GSTAMD_ERR_INVALID, "NULL destination plane");
        dp[i] = (uint8_t *) dest_planes[i] + (f ? st : 0);
        ds[i] = 2 * st;
      }
      const int fr = frame_planes_plan_order (fc, sp, ss, dp, ds, stream_);
      if (fr != GSTAMD_OK)
        return fr;
    }
    return GSTAMD_OK;
  }
  int r = ensure_tables (c);
  if (r != GSTAMD_OK || (r = bind_scratch (c, stream_)) != GSTAMD_OK)
    return r;
  hipStream_t stream = (hipStream_t) stream_;
  const VideoPlan &p = c->plan;
  if (p.gamma.on)
    return convert_gamma (c, src_planes, src_stride, dest_planes, dest_stride, stream);
  if (p.v210_fast) {
    V210FastParams vp;
    memset ((void *) &vp, 0, sizeof (vp));
    const FormatDesc *f8 = p.fin->kind == UNPACK_V210 ? p.fout : p.fin;
    vp.to_v210 = p.fout->kind == UNPACK_V210;
    vp.kind = f8->kind, vp.h_sub = f8->h_sub, vp.u_plane = f8->u_plane, vp.v_plane = f8->v_plane;
    vp.bps = f8->hi_depth ? 2 : 1;
    memcpy (vp.pos, f8->pos, sizeof (vp.pos));
    vp.width = p.in_info.width, vp.height = p.in_info.height;
    for (int i = 0; i < p.in_info.n_planes && i < 3; i++) {
      if (!src_planes[i])
        return set_error (GSTAMD_ERR_INVALID, "NULL source plane");
      vp.s[i] = (const uint8_t *) src_planes[i], vp.sstride[i] = src_stride ? src_stride[i] : p.in_info.stride[i];
    }
    for (int i = 0; i < p.out_info.n_planes && i < 3; i++) {
      if (!dest_planes[i])
        return set_error (GSTAMD_ERR_INVALID, "NULL destination plane");
      vp.d[i] = (uint8_t *) dest_planes[i], vp.dstride[i] = dest_stride ? dest_stride[i] : p.out_info.stride[i];
    }
    const hipError_t ve = launch_v210_fast (vp, stream);
    return ve == hipSuccess ? GSTAMD_OK : hip_fail (ve, "k_v210_fast");
  }
  Planes pl;
  memset (&pl, 0, sizeof (pl));
  for (int i = 0; i < p.in_info.n_planes; i++) {
    pl.p[i] = (const uint8_t *) src_planes[i];
    pl.stride[i] = src_stride ? src_stride[i] : p.in_info.stride[i];
    if (!pl.p[i])
      return set_error (GSTAMD_ERR_INVALID, "NULL source plane");
    pl.p[i] += plane_origin (p.fin, i, p.rect.in_x, p.rect.in_y, pl.stride[i]);
  }
  /* destination rectangle: the picture is written at (out_x, out_y), the rest of the frame gets the border */
  void *dest_rect[GSTAMD_VIDEO_MAX_PLANES] = {nullptr, nullptr, nullptr, nullptr};
  int32_t dest_rect_stride[GSTAMD_VIDEO_MAX_PLANES] = {0, 0, 0, 0};
  for (int i = 0; i < p.out_info.n_planes; i++) {
    dest_rect_stride[i] = dest_stride ? dest_stride[i] : p.out_info.stride[i];
    if (!dest_planes[i])
      return set_error (GSTAMD_ERR_INVALID, "NULL destination plane");
    dest_rect[i] = (uint8_t *) dest_planes[i] + plane_origin (p.fout, i, p.rect.out_x, p.rect.out_y, dest_rect_stride[i]);
  }
  if (p.rect.fill) {
    uint8_t *bp[4] = {(uint8_t *) dest_planes[0], (uint8_t *) dest_planes[1], (uint8_t *) dest_planes[2], (uint8_t *) dest_planes[3]};
    hipError_t be = fill_borders (p, bp, dest_rect_stride, stream);
    if (be != hipSuccess)
      return hip_fail (be, "k_fill_border");
  }
  r = convert_rect (c, pl, dest_rect, dest_rect_stride, stream);
  if (r == GSTAMD_OK && p.dither.on) {
    /* the dither stage (video_dither.h) over the converted rectangle of the packed destination */
    hipError_t de = launch_dither4 (p.dither, (uint8_t *) dest_rect[0], dest_rect_stride[0], p.out_info.width, p.out_info.height, stream, c->ed_carry);
    if (de != hipSuccess)
      return hip_fail (de, "k_dither4");
  }
  return r;
}

// the conversion proper: source planes already at the crop origin, destination planes at the rectangle origin
static int convert_rect (GstAmdVideoConverter *c, const Planes &pl, void *const dest_planes[GSTAMD_VIDEO_MAX_PLANES],
    const int32_t dest_stride[GSTAMD_VIDEO_MAX_PLANES], hipStream_t stream)
{
  const VideoPlan &p = c->plan;
  int r = GSTAMD_OK;
  if (p.plane_mode && c->plane_frame_ok && !tuning_on ("GSTAMD_NO_PLANE_FRAME")) {
    /* convert_scale_planes in one launch: the tiles of every plane, the first pass of a tile in LDS (video_planes.h) */
    PlaneJobs jobs;
    memset ((void *) &jobs, 0, sizeof (jobs));
    int tiles = 0;
    /* planes with two passes first: their tiles are the long ones, the pass-free planes' tiles fill in around them */
    size_t order[PLN_MAX_JOBS], n_order = 0;
    for (int two = 1; two >= 0; two--)
      for (size_t i = 0; i < p.planes.size (); i++)
        if ((p.planes[i].kind == PLANE_SCALE && p.planes[i].passes.size () == 2) == (two == 1))
          order[n_order++] = i;
    for (size_t k = 0; k < n_order; k++) {
      const size_t i = order[k];
      const PlanePlan &pp = p.planes[i];
      PlaneJob &J = jobs.job[k];
      J.kind = pp.kind;
      J.s.p = pl.p[pp.src_plane], J.s.stride = pl.stride[pp.src_plane], J.s.n = pp.n_elems;
      J.s.pairs = pp.n_elems == 2 && ((uintptr_t) J.s.p % 2) == 0 && (J.s.stride % 2) == 0;
      J.d.p = (uint8_t *) dest_planes[pp.dst_plane], J.d.stride = dest_stride ? dest_stride[pp.dst_plane] : p.out_info.stride[pp.dst_plane], J.d.n = pp.n_elems;
      if (!J.s.p || !J.d.p)
        return set_error (GSTAMD_ERR_INVALID, "NULL plane");
      J.iw = pp.iw, J.ih = pp.ih, J.ow = pp.ow, J.oh = pp.oh;
      J.n_pass = (int) pp.passes.size ();
      J.h_first = J.n_pass ? pp.passes[0].horizontal : 0;
      for (size_t q = 0; q < pp.passes.size (); q++) {
        J.pass[q].kind = pp.passes[q].kind;
        J.pass[q].n_taps = pp.passes[q].n_taps;
        J.pass[q].inc = pp.passes[q].inc;
        J.pass[q].offset = c->plane_dev[i][q].offset;
        J.pass[q].taps = c->plane_dev[i][q].taps;
      }
      const int unit = 4 * pp.n_elems;
      J.wide = pp.n_elems <= 2 && ((uintptr_t) J.d.p % unit) == 0 && (J.d.stride % unit) == 0;
      J.wide_src = ((uintptr_t) J.s.p % 8) == 0 && (J.s.stride % 8) == 0;
      J.dstep = i < c->plane_dstep.size () && !tuning_on ("GSTAMD_PLANE_QUAD_NO_DSTEP") ? c->plane_dstep[i] : 0;
      J.quad = plane_job_quad (J, i < c->plane_quad.size () && c->plane_quad[i], i < c->plane_oct.size () && c->plane_oct[i], tuning_int ("GSTAMD_PLANE_QUAD_MODE", 2));
      J.tile0 = tiles;
      J.tiles_x = (pp.ow + PLN_TW - 1) / PLN_TW;
      tiles += J.tiles_x * ((pp.oh + PLN_TH - 1) / PLN_TH);
    }
    jobs.n = (int) p.planes.size ();
    hipError_t e = launch_plane_frame (jobs, c->plane_lds_bytes, stream);
    return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "k_plane_frame");
  }
  if (p.plane_mode) {
    /* convert_scale_planes: every destination plane from one source plane */
    for (size_t i = 0; i < p.planes.size (); i++) {
      const PlanePlan &pp = p.planes[i];
      const uint8_t *sp = pl.p[pp.src_plane];
      const int ss = pl.stride[pp.src_plane];
      uint8_t *dp = (uint8_t *) dest_planes[pp.dst_plane];
      const int ds = dest_stride ? dest_stride[pp.dst_plane] : p.out_info.stride[pp.dst_plane];
      if (!sp || !dp)
        return set_error (GSTAMD_ERR_INVALID, "NULL plane");
      hipError_t e;
      if (pp.kind != PLANE_SCALE) {
        e = launch_plane_simple (pp.kind, sp, ss, dp, ds, pp.n_elems, pp.ow, pp.oh, stream);
      } else {
        ScaleDev sd[2];
        for (size_t k = 0; k < pp.passes.size (); k++) {
          memset (&sd[k], 0, sizeof (sd[k]));
          sd[k].kind = pp.passes[k].kind;
          sd[k].n_taps = pp.passes[k].n_taps;
          sd[k].inc = pp.passes[k].inc;
          sd[k].offset = c->plane_dev[i][k].offset;
          sd[k].taps = c->plane_dev[i][k].taps;
          sd[k].merged = pp.passes[k].merged;
        }
        if (pp.passes.size () == 1) {
          e = launch_plane_pass (pp.passes[0].horizontal, sd[0], sp, ss, dp, ds, pp.n_elems, pp.ow, pp.oh, stream);
        } else {
          const int tw = pp.passes[0].horizontal ? pp.ow : pp.iw, th = pp.passes[0].horizontal ? pp.ih : pp.oh;
          if (!c->plane_tmp && (e = hipMalloc ((void **) &c->plane_tmp, c->plane_tmp_bytes)) != hipSuccess)
            return hip_fail (e, "hipMalloc(plane tmp)");
          e = launch_plane_pass (pp.passes[0].horizontal, sd[0], sp, ss, c->plane_tmp, tw * pp.n_elems, pp.n_elems, tw, th, stream);
          if (e == hipSuccess)
            e = launch_plane_pass (pp.passes[1].horizontal, sd[1], c->plane_tmp, tw * pp.n_elems, dp, ds, pp.n_elems, pp.ow, pp.oh, stream);
        }
      }
      if (e != hipSuccess)
        return hip_fail (e, "plane scaler");
    }
    return GSTAMD_OK;
  }
  if (p.out_planar) {
    /* chain -> AYUV image in HBM, then chroma downsample + pack into the destination planes */
    uint8_t *planes[3] = {nullptr, nullptr, nullptr};
    int strides[3] = {0, 0, 0};
    for (int i = 0; i < p.out_info.n_planes && i < 3; i++) {
      planes[i] = (uint8_t *) dest_planes[i];
      strides[i] = dest_stride ? dest_stride[i] : p.out_info.stride[i];
      if (!planes[i])
        return set_error (GSTAMD_ERR_INVALID, "NULL destination plane");
    }
    if (c->deep_hook) {
      hipError_t he = launch_deep_scale_pack (p.pack, *c->deep_hook, planes, strides, stream);
      return he == hipSuccess ? GSTAMD_OK : hip_fail (he, "k_deep_scale_pack");
    }
    if (p.fout->kind == UNPACK_PACKED3 && fast_pair_usable (p, pl, planes[0], strides[0], 4)) {
      /* NV12 / NV21 -> RGB / BGR, same size: the line-pair kernel stores the 3-byte pixels itself (12 bytes per lane and line) */
      const FastParams fp = make_fast_params (p, true);
      const uint8_t *y = pl.p[0], *uv = pl.p[1];
      hipError_t e = launch_convert_pair (fp, p.front.chroma_h, 1, &y, &uv, &planes[0], pl.stride[0], pl.stride[1], strides[0], stream);
      return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "k_convert_pair(rgb24)");
    }
    if (p.fast_enc420 && ((uintptr_t) pl.p[0] % 16) == 0 && (pl.stride[0] % 16) == 0 && ((uintptr_t) planes[0] % 4) == 0 && (strides[0] % 4) == 0 &&
        ((uintptr_t) planes[1] % 4) == 0 && (strides[1] % 4) == 0 && (p.fout->kind == UNPACK_SEMI || (((uintptr_t) planes[2] % 4) == 0 && (strides[2] % 4) == 0))) {
      hipError_t e = launch_encode420 (make_enc420_params (p), p.fout->kind == UNPACK_SEMI, pl.p[0], pl.stride[0], planes, strides, stream);
      return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "k_encode420");
    }
    if (p.fout->kind == UNPACK_PACKED3 && p.passes.empty () && !p.deep16 && !c->hook_on && !p.pack.dither.on && p.matrix.kind == MATRIX_NONE &&
        p.post.alpha_kind == ALPHA_NONE && p.front.hi_depth == 0 && (p.front.kind == UNPACK_PACKED4 || p.front.kind == UNPACK_PACKED3) &&
        p.post.pack_pos[0] == 0 && p.post.pack_pos[1] == 1 && p.post.pack_pos[2] == 2 && p.post.pack_pos[3] == 3 && !tuning_on ("GSTAMD_NO_SWIZZLE34")) {
      Swz34Params sp;
      const int sb = p.front.kind == UNPACK_PACKED4 ? 4 : 3;
      if (swizzle34_setup (sb, p.front.pos, 3, p.pack.pos, pl.p[0], pl.stride[0], planes[0], strides[0], p.out_info.width, &sp)) {
        hipError_t se = launch_swizzle34 (sp, sb, 3, p.out_info.height, stream);
        return se == hipSuccess ? GSTAMD_OK : hip_fail (se, "k_swizzle34");
      }
    }
    if (p.relayout && !tuning_on ("GSTAMD_NO_RELAYOUT")) {
      RelayoutParams rp;
      memset ((void *) &rp, 0, sizeof (rp));
      rp.width = p.out_info.width, rp.height = p.out_info.height;
      rp.cw = (rp.width + (1 << p.fout->w_sub) - 1) >> p.fout->w_sub, rp.ch = (rp.height + (1 << p.fout->h_sub) - 1) >> p.fout->h_sub;
      rp.in_semi = p.fin->kind == UNPACK_SEMI, rp.out_semi = p.fout->kind == UNPACK_SEMI;
      rp.in_u = p.fin->u_plane, rp.in_v = p.fin->v_plane, rp.out_u = p.fout->u_plane, rp.out_v = p.fout->v_plane;
      for (int i = 0; i < p.in_info.n_planes && i < 3; i++)
        rp.in[i] = pl.p[i], rp.in_stride[i] = pl.stride[i];
      for (int i = 0; i < p.out_info.n_planes && i < 3; i++)
        rp.out[i] = planes[i], rp.out_stride[i] = strides[i];
      if (relayout_usable (rp)) {
        hipError_t re = launch_planes_relayout (rp, stream);
        return re == hipSuccess ? GSTAMD_OK : hip_fail (re, "k_planes_relayout");
      }
    }
    const bool diffusion_first = p.pack.dither.on && p.pack.dither.method != GSTAMD_DITHER_NONE && p.pack.dither.method != GSTAMD_DITHER_BAYER;
    ColorParams color;
    color.matrix = p.matrix;
    color.alpha_kind = p.post.alpha_kind;
    color.alpha_value = p.post.alpha_value;
    const int alpha_idx = GSTAMD_KIND_ALPHA_PLANE (p.fout->kind);
    const bool alpha_plane = alpha_idx >= 0;      /* A420: the image path below, then the fourth plane from the image */
    if (!alpha_plane && p.passes.empty () && !p.deep16 && !c->hook_on && !diffusion_first && p.post.pack_pos[0] == 0 && p.post.pack_pos[1] == 1 &&
        p.post.pack_pos[2] == 2 && p.post.pack_pos[3] == 3 && convert_pack_usable (p.front, pl, color) && !tuning_on ("GSTAMD_NO_CONVERT_PACK")) {
      /* an unscaled 8-bit chain with a cheap pixel source: the packer takes its pixels from the chain itself, nothing goes through HBM
         in between */
      hipError_t ce = launch_convert_pack (p.pack, p.front, pl, c->vpair_dev, color, planes, strides, stream);
      return ce == hipSuccess ? GSTAMD_OK : hip_fail (ce, "k_convert_pack");
    }
    if (c->raw4_pack && !alpha_plane && pl.p[0] && !c->hook_on && !tuning_on ("GSTAMD_NO_PLANE_QUAD")) {
      /* plane_raw4_pack_plan: the scaler on the raw 4-byte pixels into the image, then the unscaled block kernels from it */
      const int ow = p.out_info.width, oh = p.out_info.height;
      PlaneJobs jobs;
      memset ((void *) &jobs, 0, sizeof (jobs));
      PlaneJob &J = jobs.job[0];
      J.kind = PLANE_SCALE;
      J.s.p = pl.p[0], J.s.stride = pl.stride[0], J.s.n = 4;
      J.d.p = c->pk_img, J.d.stride = ow * 4, J.d.n = 4;
      J.iw = p.front.width, J.ih = p.front.height, J.ow = ow, J.oh = oh;
      J.n_pass = 2;
      J.h_first = p.passes[0].horizontal ? 1 : 0;
      for (int q = 0; q < 2; q++) {
        J.pass[q].kind = p.passes[q].kind;
        J.pass[q].n_taps = p.passes[q].n_taps;
        J.pass[q].inc = p.passes[q].inc;
        J.pass[q].offset = c->pass_dev[q].offset;
        J.pass[q].taps = c->pass_dev[q].taps;
      }
      J.dstep = tuning_on ("GSTAMD_PLANE_QUAD_NO_DSTEP") ? 0 : c->raw4_pack_dstep;
      J.quad = 1 + QUAD_8;
      J.tiles_x = 1;
      jobs.n = 1;
      hipError_t qe = launch_plane_frame (jobs, 0, stream);
      if (qe != hipSuccess)
        return hip_fail (qe, "k_plane_quad(4-byte pixels)");
      bool enc = c->raw4_pack_enc420 && ((uintptr_t) planes[0] % 4) == 0 && (strides[0] % 4) == 0 && ((uintptr_t) planes[1] % 4) == 0 && (strides[1] % 4) == 0 &&
          (p.fout->kind == UNPACK_SEMI || (((uintptr_t) planes[2] % 4) == 0 && (strides[2] % 4) == 0));
      if (enc) {
        qe = launch_encode420 (make_enc420_params (p), p.fout->kind == UNPACK_SEMI, c->pk_img, ow * 4, planes, strides, stream);
        return qe == hipSuccess ? GSTAMD_OK : hip_fail (qe, "k_encode420");
      }
      FrontParams f2 = p.front;
      f2.width = ow, f2.height = oh;
      Planes pl2;
      memset ((void *) &pl2, 0, sizeof (pl2));
      pl2.p[0] = c->pk_img, pl2.stride[0] = ow * 4;
      qe = launch_convert_pack (p.pack, f2, pl2, nullptr, color, planes, strides, stream);
      return qe == hipSuccess ? GSTAMD_OK : hip_fail (qe, "k_convert_pack");
    }
    r = convert_to_packed (c, pl, c->pk_img, p.out_info.width * 4, stream);
    if (r != GSTAMD_OK)
      return r;
    const bool diffusion = p.pack.dither.on && p.pack.dither.method != GSTAMD_DITHER_NONE && p.pack.dither.method != GSTAMD_DITHER_BAYER;
    hipError_t e = diffusion ? launch_pack_planar_ed (p.pack, c->pk_img, p.out_info.width * 4, planes, strides, stream, c->ed_carry)
        : launch_pack_planar (p.pack, c->pk_img, p.out_info.width * 4, planes, strides, stream);
    if (e == hipSuccess && alpha_plane) {
      if (!dest_planes[alpha_idx])
        return set_error (GSTAMD_ERR_INVALID, "NULL destination plane");
      PackPlanarParams pa = p.pack;
      if (diffusion)
        memset (&pa.dither, 0, sizeof (pa.dither));         /* the error-diffusion pass ran over all four components of the image already */
      e = launch_pack_alpha_plane (pa, c->pk_img, p.out_info.width * 4, (uint8_t *) dest_planes[alpha_idx], dest_stride ? dest_stride[alpha_idx] : p.out_info.stride[alpha_idx], stream);
    }
    return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "k_pack_planar");
  }
  return convert_to_packed (c, pl, (uint8_t *) dest_planes[0], dest_stride ? dest_stride[0] : p.out_info.stride[0], stream);
}

// Does the plan take the direct 4:2:0 bilinear kernels (video_bilinear_fast.h / video_bilinear_rows.h)?  Fills everything of
// BilParams that does not depend on a particular frame's pointers.
// k_scale_col applies to these planes: the plan's tables exist, no colour step ahead of the scaler, U and V planes of one pitch
static bool col_usable (const GstAmdVideoConverter *c, const Planes &pl, const ColorParams &pre)
{
  const VideoPlan &p = c->plan;
  return c->col_ok && pre.matrix.kind == MATRIX_NONE && pre.alpha_kind == ALPHA_NONE &&
      (p.front.kind == UNPACK_SEMI || pl.stride[p.front.u_plane] == pl.stride[p.front.v_plane]);
}

// n frames (source plane pointers src[f][0 .. 2], destinations dst[f]) of the strides of `pl`: launches of up to GSTAMD_COL_MAX_FRAMES frames
static hipError_t col_launch (GstAmdVideoConverter *c, int n, const void *const (*src)[3], uint8_t *const *dst, const Planes &pl, int dstride, const ColorParams &post,
    const PostFast &pf, hipStream_t stream)
{
  const VideoPlan &p = c->plan;
  const bool semi = p.front.kind == UNPACK_SEMI;
  ColParams q;
  memset ((void *) &q, 0, sizeof (q));
  q.ystride = pl.stride[0];
  q.cstride = semi ? pl.stride[1] : pl.stride[p.front.u_plane];
  q.width = p.front.width;
  q.height = p.front.height;
  q.u_first = p.front.u_plane != 0;
  q.crow_lo = c->reg_lo;
  q.crow_hi = c->reg_hi;
  q.tiles = c->col_tiles_dev;
  q.hout = c->col_hout_dev;
  q.vrow = c->col_vrow_dev;
  q.out_w = p.passes[0].out_size;
  q.out_h = p.passes[1].out_size;
  q.dstride = dstride;
  for (int base = 0; base < n; base += GSTAMD_COL_MAX_FRAMES) {
    const int nb = std::min (n - base, GSTAMD_COL_MAX_FRAMES);
    col_geometry (c, nb, &q);
    ColFrames fr;
    memset ((void *) &fr, 0, sizeof (fr));
    for (int f = 0; f < nb; f++) {
      fr.y[f] = (const uint8_t *) src[base + f][0];
      fr.c0[f] = (const uint8_t *) (semi ? src[base + f][1] : src[base + f][p.front.u_plane]);
      fr.c1[f] = (const uint8_t *) (semi ? src[base + f][1] : src[base + f][p.front.v_plane]);
      fr.dst[f] = dst[base + f];
    }
#ifdef GSTAMD_COL_TRACE
    /* profiling builds: cycles per phase of the walk, averaged over the waves of the launch (stderr) */
    const size_t trace_n = (size_t) q.n_tiles * q.n_chunks * nb * GSTAMD_COL_MAX_WAVES * 8;
    q.trace = nullptr;
    static int trace_left = 3;
    if (trace_left > 0 && hipMalloc ((void **) &q.trace, trace_n * 8) == hipSuccess)
      (void) hipMemsetAsync (q.trace, 0, trace_n * 8, stream);
#endif
    const hipError_t e = launch_scale_col (q, c->col_form, p.front.chroma_h, semi ? 1 : 0, c->col_waves, fr, nb, dstride, post, p.post.pack_pos, pf, stream);
#ifdef GSTAMD_COL_TRACE
    if (q.trace) {
      std::vector<unsigned long long> h (trace_n);
      (void) hipStreamSynchronize (stream);
      (void) hipMemcpy (h.data (), q.trace, trace_n * 8, hipMemcpyDeviceToHost);
      (void) hipFree (q.trace);
      double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      size_t waves = 0;
      for (size_t i = 0; i < trace_n; i += 8)
        if (h[i + 7]) {
          waves++;
          for (int k = 0; k < 7; k++)
            sum[k] += (double) h[i + k];
        }
      if (waves)
        fprintf (stderr, "k_scale_col trace: %zu waves, %.1f groups each; cycles per wave: prologue %.0f, per group: load wait %.0f stage %.0f horizontal %.0f rows %.0f; hand-over %.0f\n",
            waves, sum[6] / waves, sum[0] / waves, sum[1] / sum[6], sum[2] / sum[6], sum[3] / sum[6], sum[4] / sum[6], sum[5] / waves);
      trace_left--;
    }
#endif
    if (e != hipSuccess)
      return e;           /* hipErrorNotSupported can only come from the first launch (alignment): nothing has run yet */
  }
  return hipSuccess;
}

static bool bilinear420_params (GstAmdVideoConverter *c, BilParams *out)
{
  const VideoPlan &p = c->plan;
  const auto small_kind = [](int k) { return k == SCALE_NEAREST || k == SCALE_2TAP; };
  /* no colour stage at all (YUV -> YUV of one colorimetry: the pack image of a planar / semi-planar destination, an AYUV frame): the same kernels
     with the layout that stores A Y U V (GSTAMD_LAYOUT_AYUV) - NV12 4K -> I420 1080p took 65 us through the generic wave-tile scaler */
  const bool ayuv = bilinear420_ayuv_plan (p) && !tuning_on ("GSTAMD_NO_BILINEAR_AYUV");
  if ((p.out_planar && !ayuv) || p.passes.size () != 2 || !small_kind (p.passes[0].kind) || !small_kind (p.passes[1].kind))
    return false;
  /* semi-planar / planar 4:2:0 source, horizontal-first 2-tap x 2-tap, fast matrix */
  if (!(p.passes[0].horizontal && p.passes[0].kind == SCALE_2TAP && p.passes[1].kind == SCALE_2TAP && kind_has_planes (p.front.kind) && p.front.w_sub == 1 &&
        p.front.h_sub == 1 && !p.matrix_before_scale && (p.fast_post || ayuv) && p.front.chroma_v2 != 2 && !tuning_on ("GSTAMD_NO_BILINEAR420")))
    return false;
  const int out_w = p.out_info.width, out_h = p.out_info.height;
  BilParams bp;
  memset (&bp, 0, sizeof (bp));
  bp.tile_w = bil_pick_tile (out_w, p.passes[0].inc, &bp.ylen);
  if (tuning_on ("GSTAMD_BIL_TILE")) {      /* tuning knob for profiling sessions */
    bp.tile_w = tuning_int ("GSTAMD_BIL_TILE", 0);
    bp.ylen = bil_ylen (out_w, p.passes[0].inc, bp.tile_w);
  }
  if (bp.tile_w <= 0 || bp.ylen <= 0)
    return false;
  bp.fp = make_fast_params (p);
  bp.fp.ayuv = ayuv ? (p.matrix.kind == MATRIX_NONE ? 1 : 2) : 0;
  bp.fp.m8 = p.matrix;
  bp.out_w = out_w;
  bp.out_h = out_h;
  bp.inc = p.passes[0].inc;
  bp.voffset = c->pass_dev[1].offset;
  bp.vtaps = c->pass_dev[1].taps;
  bp.vpair = p.front.chroma_v2 ? c->vpair_dev : nullptr;
  bp.regular_pairs = 0;
  bp.planar = p.front.kind == UNPACK_PLANAR;
  bp.u_plane = p.front.u_plane;
  bp.v_plane = p.front.v_plane;
  if (p.front.chroma_v2 && !tuning_on ("GSTAMD_BIL_TABLE")) {
    /* are the pairs of every source line the kernel will touch the closed form of bil_rows? */
    bool regular = true;
    BilParams probe = bp;
    probe.regular_pairs = 1;
    for (int y = 0; y < out_h && regular; y++)
      for (int l = 0; l < 2 && regular; l++) {
        const int line = (int) p.passes[1].offset[y] + l;
        int ra, rb, role;
        bil_rows (probe, line, &ra, &rb, &role);
        const int e0 = p.vpair[2 * line], ta = vpair_row (e0), trole = vpair_role (e0), tb = p.vpair[2 * line + 1];
        regular = ta == ra && tb == rb && (ra == rb || trole == role);
      }
    bp.regular_pairs = regular ? 1 : 0;
  }
  /* rows per wave of k_bilinear420_rows: every source line pair has to sit in the three-row window of video_bilinear_rows.h */
  bp.rows = 0;
  if (bp.regular_pairs && (p.front.width % 16) == 0 && !tuning_on ("GSTAMD_NO_BILINEAR_ROWS")) {
    bool fits = true;
    for (int y = 0; y < out_h && fits; y++)
      fits = bilr_window_matches (bp, (int) p.passes[1].offset[y]);
    int rows_ylen = 0;
    bp.rows_tile_w = bilr_pick_tile (out_w, p.passes[0].inc, &rows_ylen);
#ifdef GSTAMD_TUNING
    if (tuning_on ("GSTAMD_BIL_ROWS_TILE")) {
      bp.rows_tile_w = tuning_int ("GSTAMD_BIL_ROWS_TILE", 0);
      rows_ylen = bil_ylen (out_w, p.passes[0].inc, bp.rows_tile_w);
    }
#endif
    fits = fits && bp.rows_tile_w > 0 && rows_ylen > 0;
    if (fits)
      bp.rows = -1;
#ifdef GSTAMD_TUNING
    if (fits && tuning_on ("GSTAMD_BIL_ROWS"))
      bp.rows = tuning_int ("GSTAMD_BIL_ROWS", 0);
#endif
  }
  bp.half = !tuning_on ("GSTAMD_NO_BILINEAR_HALF") && bilh_plan_ok (bp, p.passes[1].offset.data (), p.passes[1].taps.data ());
  *out = bp;
  return true;
}

// 10-bit source with scaling (video_deep.h).  The picture shrinks: front -> AYUV64 image, the passes on 16-bit lines, the last one
// fused with the convert stage.  It grows: the convert stage first (8-bit unpack-order image), then the 8-bit scalers.  The images
// between the stages live in HBM (two scratch images per converter).
static int convert_deep_scaled (GstAmdVideoConverter *c, const Planes &pl, uint8_t *dst, int dstride, hipStream_t stream)
{
  const VideoPlan &p = c->plan;
  const int in_w = p.front.width, in_h = p.front.height, out_w = p.out_info.width, out_h = p.out_info.height;
  const size_t n = p.passes.size ();
  const size_t bpp = p.matrix_before_scale ? 4 : 8;
  DeepPackParams ds4;
  if (!c->hook_on && !tuning_on ("GSTAMD_NO_DEEP_SCALE_PACK") && deep_scale4_plan_ok (p, &ds4)) {
    /* a picture that halves (2-tap both ways): front, both passes, convert stage and pack in one kernel, no 16-bit image (video_deep_pack.h) */
    ds4.pl = pl;
    ds4.vpair = c->vpair_dev;
    ds4.sh.offset = c->pass_dev[0].offset, ds4.sh.taps = c->pass_dev[0].taps;
    ds4.sv.offset = c->pass_dev[1].offset, ds4.sv.taps = c->pass_dev[1].taps;
    if (deep_scale4_usable (ds4, dst, dstride)) {
      const hipError_t fe = launch_deep_scale4 (ds4, p.deep, p.post, dst, dstride, stream);
      return fe == hipSuccess ? GSTAMD_OK : hip_fail (fe, "k_deep_scale4");
    }
  }
  const int mid_w = p.passes[0].horizontal ? p.passes[0].out_size : in_w, mid_h = p.passes[0].horizontal ? in_h : p.passes[0].out_size;
  const size_t need_a = (size_t) in_w * in_h * bpp, need_b = n == 2 ? (size_t) mid_w * mid_h * bpp : 0;
  hipError_t e;
  if (c->deep_a_size < need_a) {
    if (c->deep_a)
      (void) hipFree (c->deep_a);
    c->deep_a = nullptr;
    if ((e = hipMalloc ((void **) &c->deep_a, need_a)) != hipSuccess)
      return hip_fail (e, "hipMalloc(16-bit scratch)");
    c->deep_a_size = need_a;
  }
  if (c->deep_b_size < need_b) {
    if (c->deep_b)
      (void) hipFree (c->deep_b);
    c->deep_b = nullptr;
    if ((e = hipMalloc ((void **) &c->deep_b, need_b)) != hipSuccess)
      return hip_fail (e, "hipMalloc(16-bit scratch)");
    c->deep_b_size = need_b;
  }
  ScaleDev sd[2];
  for (size_t i = 0; i < n; i++) {
    memset (&sd[i], 0, sizeof (sd[i]));
    sd[i].kind = p.passes[i].kind;
    sd[i].n_taps = p.passes[i].n_taps;
    sd[i].inc = p.passes[i].inc;
    sd[i].offset = c->pass_dev[i].offset;
    sd[i].taps = c->pass_dev[i].taps;
    sd[i].tapw = c->pass_dev[i].tapw;
    sd[i].nw = p.passes[i].nw;
    sd[i].nw4 = p.passes[i].nw4;
  }
  if (!p.matrix_before_scale) {
    Deep16Image cur = {c->deep_a, in_w * 8, in_w, in_h};
    size_t first = 0;
    if (n == 2 && p.passes[0].horizontal && front_hscale16_usable (p.front)) {
      /* the front inside the first, horizontal pass: no full-size AYUV64 image */
      if ((e = launch_front_hscale16 (p.front, pl, c->vpair_dev, sd[0], c->deep_b, mid_w * 8, mid_w, stream)) != hipSuccess)
        return hip_fail (e, "k_front_hscale16");
      cur.p = c->deep_b, cur.stride = mid_w * 8, cur.width = mid_w, cur.height = in_h;
      first = 1;
    } else if ((e = launch_front16 (p.front, pl, c->vpair_dev, c->deep_a, in_w * 8, stream)) != hipSuccess) {
      return hip_fail (e, "k_front16");
    }
    for (size_t i = first; i < n; i++) {
      const bool hz = p.passes[i].horizontal, last = i + 1 == n;
      const int ow = hz ? p.passes[i].out_size : cur.width, oh = hz ? cur.height : p.passes[i].out_size;
      if (last)
        e = launch_scale16 (cur, sd[i], hz, dst, dstride, ow, oh, &p.deep, &p.post, stream);
      else
        e = launch_scale16 (cur, sd[i], hz, c->deep_b, ow * 8, ow, oh, nullptr, nullptr, stream);
      if (e != hipSuccess)
        return hip_fail (e, "k_scale16");
      cur.p = c->deep_b, cur.stride = ow * 8, cur.width = ow, cur.height = oh;
    }
    return GSTAMD_OK;
  }
  PostParams mid = p.post;
  for (int i = 0; i < 4; i++)
    mid.pack_pos[i] = i;
  if ((e = launch_convert16 (p.front, pl, c->vpair_dev, p.deep, mid, c->deep_a, in_w * 4, stream)) != hipSuccess)
    return hip_fail (e, "k_convert16");
  ColorParams none;
  memset (&none, 0, sizeof (none));
  PostFast pf_none;
  memset (&pf_none, 0, sizeof (pf_none));
  const uint8_t *src = c->deep_a;
  int sw = in_w, sh = in_h;
  for (size_t i = 0; i < n; i++) {
    const bool hz = p.passes[i].horizontal, last = i + 1 == n;
    const int ow = hz ? p.passes[i].out_size : sw, oh = hz ? sh : p.passes[i].out_size;
    e = launch_scale_from_image (hz, src, sw * 4, sd[i], last ? dst : c->deep_b, last ? dstride : ow * 4, last, none, p.post.pack_pos, ow, oh,
        p.passes[i].max_span, sw, c->geom[i], pf_none, stream);
    if (e != hipSuccess)
      return hip_fail (e, "scale pass (8-bit lines of a 10-bit source)");
    src = c->deep_b, sw = ow, sh = oh;
  }
  (void) out_w;
  (void) out_h;
  return GSTAMD_OK;
}

static int convert_to_packed (GstAmdVideoConverter *c, const Planes &pl, uint8_t *dst, int dstride, hipStream_t stream)
{
  const VideoPlan &p = c->plan;
  ColorParams color, none;
  memset (&none, 0, sizeof (none));
  color.matrix = p.matrix;
  color.alpha_kind = p.post.alpha_kind;
  color.alpha_value = p.post.alpha_value;
  hipError_t e;
  if (fast_pair_usable (p, pl, dst, dstride)) {
    FastParams fp = make_fast_params (p);
    if (c->post_lut) {
      fp.lut = c->post_lut;
      fp.lut_keep = c->post_lut_keep;
      c->post_lut_done = true;
    }
    const uint8_t *y = pl.p[0], *uv = pl.p[1];
    e = launch_convert_pair (fp, p.front.chroma_h, 1, &y, &uv, &dst, pl.stride[0], pl.stride[1], dstride, stream);
    return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "k_convert_pair");
  }
  if (p.fast_420p && ((uintptr_t) pl.p[0] % 8) == 0 && (pl.stride[0] % 8) == 0 && ((uintptr_t) pl.p[1] % 4) == 0 && ((uintptr_t) pl.p[2] % 4) == 0 &&
      (pl.stride[1] % 4) == 0 && pl.stride[1] == pl.stride[2] && ((uintptr_t) dst % 16) == 0 && (dstride % 16) == 0 &&
      !tuning_on ("GSTAMD_NO_FAST420P")) {
    Fast420pParams q;
    q.fp = make_fast_params (p);
    q.y = pl.p[0];
    q.u = pl.p[p.front.u_plane];
    q.v = pl.p[p.front.v_plane];
    q.ystride = pl.stride[0];
    q.cstride = pl.stride[1];
    e = launch_convert420p (q, dst, dstride, stream);
    return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "k_convert420p");
  }
  if (p.fast_422 && ((uintptr_t) pl.p[0] % 16) == 0 && (pl.stride[0] % 16) == 0 && ((uintptr_t) dst % 16) == 0 && (dstride % 16) == 0 &&
      !tuning_on ("GSTAMD_NO_FAST422")) {
    Fast422Params q;
    q.fp = make_fast_params (p);
    q.chroma_h = p.front.chroma_h;
    fast422_selectors (p.front.pos[1], p.front.pos[2], p.front.pos[3], &q);
    e = launch_convert422 (q, pl.p[0], pl.stride[0], dst, dstride, stream);
    return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "k_convert422");
  }
  if (p.fast_422_ayuv && !c->hook_on && ((uintptr_t) pl.p[0] % 16) == 0 && (pl.stride[0] % 16) == 0 && ((uintptr_t) dst % 16) == 0 && (dstride % 16) == 0 &&
      !tuning_on ("GSTAMD_NO_FAST422")) {
    Fast422Params q;
    memset ((void *) &q, 0, sizeof (q));
    q.fp.width = p.front.width;
    q.fp.height = p.front.height;
    q.chroma_h = p.front.chroma_h;
    fast422_selectors (p.front.pos[1], p.front.pos[2], p.front.pos[3], &q);
    e = launch_convert422 (q, pl.p[0], pl.stride[0], dst, dstride, stream, true);
    return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "k_convert422_ayuv");
  }
  if (p.deep16) {
    /* (the byte-stream kinds - NV12_10LE40 & co, UYVP - are read byte by byte: rows of five-byte groups have no alignment; the three-samples-per-word
       kinds are read as 32-bit words) */
    const bool bytes_in = p.front.kind == UNPACK_SEMI_LE40 || p.front.kind == UNPACK_P422_UYVP || p.front.kind == UNPACK_SEMI_LE40_TILED;
    const bool words_in = p.front.kind == UNPACK_SEMI_LE32 || p.front.kind == UNPACK_GRAY_LE32;
    if (words_in && (((uintptr_t) pl.p[0] % 4) != 0 || (pl.stride[0] % 4) != 0 || (p.front.kind == UNPACK_SEMI_LE32 && (((uintptr_t) pl.p[1] % 4) != 0 || (pl.stride[1] % 4) != 0))))
      return set_error (GSTAMD_ERR_UNSUPPORTED, "frames with three 10-bit samples per 32-bit word need 4-byte aligned planes and pitches");
    if (((uintptr_t) dst % 4) != 0 || (dstride % 4) != 0)
      return set_error (GSTAMD_ERR_UNSUPPORTED, "10-bit frames need 2-byte aligned planes and pitches, the destination 4-byte aligned ones");
    if (!bytes_in && (((uintptr_t) pl.p[0] % 2) != 0 || (pl.stride[0] % 2) != 0 || ((uintptr_t) pl.p[1] % 2) != 0 ||
        (pl.stride[1] % 2) != 0 || (p.front.kind == UNPACK_PLANAR && (((uintptr_t) pl.p[2] % 2) != 0 || (pl.stride[2] % 2) != 0))))
      return set_error (GSTAMD_ERR_UNSUPPORTED, "10-bit frames need 2-byte aligned planes and pitches, the destination 4-byte aligned ones");
    if (p.passes.empty ()) {
      e = launch_convert16 (p.front, pl, c->vpair_dev, p.deep, p.post, dst, dstride, stream);
      return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "k_convert16");
    }
    return convert_deep_scaled (c, pl, dst, dstride, stream);
  }
  if (p.passes.empty ()) {
    if (c->hook_on) {
      e = launch_convert_gamma (p.front, pl, c->vpair_dev, color, p.post.pack_pos, dst, dstride, c->hook, stream);
      return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "k_convert_gamma");
    }
    if (p.front.kind == UNPACK_PACKED3 && p.front.hi_depth == 0 && color.matrix.kind == MATRIX_NONE && color.alpha_kind == ALPHA_NONE &&
        !tuning_on ("GSTAMD_NO_SWIZZLE34")) {
      Swz34Params sp;
      if (swizzle34_setup (3, p.front.pos, 4, p.post.pack_pos, pl.p[0], pl.stride[0], dst, dstride, p.front.width, &sp)) {
        e = launch_swizzle34 (sp, 3, 4, p.front.height, stream);
        return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "k_swizzle34");
      }
    }
    if (swizzle4_usable (p.front, pl, color, dst, dstride)) {
      e = launch_swizzle4 (p.front, pl, p.post.pack_pos, dst, dstride, stream);
      return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "k_swizzle4");
    }
    e = launch_convert (p.front, pl, c->vpair_dev, color, p.post.pack_pos, dst, dstride, stream, p.out_planar && dst == c->pk_img ? p.pack.virtual_line : 0);
    return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "k_convert");
  }
  /* downscale: scale, then matrix+alpha in the post stage; upscale: matrix+alpha first */
  const ColorParams &pre = p.matrix_before_scale ? color : none;
  const ColorParams &post = p.matrix_before_scale ? none : color;
  ScaleDev sd[2];
  for (size_t i = 0; i < p.passes.size (); i++) {
    sd[i].kind = p.passes[i].kind;
    sd[i].n_taps = p.passes[i].n_taps;
    sd[i].inc = p.passes[i].inc;
    sd[i].offset = c->pass_dev[i].offset;
    sd[i].taps = c->pass_dev[i].taps;
    sd[i].tapw = c->pass_dev[i].tapw;
    sd[i].nw = p.passes[i].nw;
    sd[i].nw4 = p.passes[i].nw4;
  }
  const int out_w = p.out_info.width, out_h = p.out_info.height;
  PostFast pf, pf_none;
  memset (&pf_none, 0, sizeof (pf_none));
  pf.use = p.fast_post ? 1 : 0;
  pf.fp = make_fast_params (p);
  const auto small_kind = [](int k) { return k == SCALE_NEAREST || k == SCALE_2TAP; };
  if (p.passes.size () == 2 && small_kind (p.passes[0].kind) && small_kind (p.passes[1].kind)) {
    /* nearest / 2-tap in both directions ("bilinear"): one fused kernel, no intermediate image */
    const bool h_first = p.passes[0].horizontal;
    BilParams bp;
    if (((uintptr_t) dst % 4) == 0 && (dstride % 4) == 0 && bilinear420_params (c, &bp)) {
      e = launch_bilinear420 (bp, p.front.chroma_h, pl, dst, dstride, stream);
      if (e != hipErrorNotSupported)
        return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "k_bilinear420");
    }
    const ScaleDev &sh = h_first ? sd[0] : sd[1], &sv = h_first ? sd[1] : sd[0];
    if (c->raw4_quad && pl.p[0] && !tuning_on ("GSTAMD_NO_PLANE_QUAD")) {
      /* the plane scaler on 4-byte pixels, both passes short: k_plane_quad with a pixel of four one-byte components (video_planes.h) */
      PlaneJobs jobs;
      memset ((void *) &jobs, 0, sizeof (jobs));
      PlaneJob &J = jobs.job[0];
      J.kind = PLANE_SCALE;
      J.s.p = pl.p[0], J.s.stride = pl.stride[0], J.s.n = 4;
      J.d.p = dst, J.d.stride = dstride, J.d.n = 4;
      J.iw = p.front.width, J.ih = p.front.height, J.ow = out_w, J.oh = out_h;
      J.n_pass = 2;
      J.h_first = h_first ? 1 : 0;
      J.pass[0] = sd[0], J.pass[1] = sd[1];
      J.dstep = tuning_on ("GSTAMD_PLANE_QUAD_NO_DSTEP") ? 0 : c->raw4_dstep;
      J.quad = 1 + QUAD_8;
      J.tiles_x = 1;
      jobs.n = 1;
      e = launch_plane_frame (jobs, 0, stream);
      return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "k_plane_quad(4-byte pixels)");
    }
    if (p.matrix_before_scale && p.front.kind != UNPACK_PACKED4 && p.front.hi_depth == 0 && ((uintptr_t) dst % 4) == 0 && (dstride % 4) == 0 &&
        !tuning_on ("GSTAMD_NO_BILINEAR4")) {
      /* enlarging (the matrix runs on the source's pixels, chain_convert ahead of chain_scale): the source frame through the front and the
         colour stage into an A, c1, c2, c3 image of ITS size - one of the unscaled kernels, a quarter of the destination's pixels at 1080p
         -> 4K - and the four-outputs-per-lane scaler from that image; the scaler works per byte, so this is the chain's own order */
      const int in_w = p.front.width, in_h = p.front.height;
      const size_t need = (size_t) in_w * 4 * in_h;
      if (c->pre_img_size < need) {
        if (c->pre_img)
          (void) hipFree (c->pre_img);
        c->pre_img = nullptr;
        c->pre_img_size = 0;
        if ((e = hipMalloc ((void **) &c->pre_img, need)) != hipSuccess)
          return hip_fail (e, "hipMalloc(converted source)");
        c->pre_img_size = need;
      }
      const int ident[4] = {0, 1, 2, 3};
      if (p.fast_pre && ((uintptr_t) pl.p[0] % 4) == 0 && (pl.stride[0] % 4) == 0 && ((uintptr_t) pl.p[1] % 4) == 0 && (pl.stride[1] % 4) == 0 &&
          !tuning_on ("GSTAMD_NO_FAST_PRE")) {
        /* NV12 / NV21: that image is the unscaled conversion into A, R, G, B bytes - the line-pair kernel (11 MB at 1080p: 13 us through the per-pixel
           kernel, round 6) */
        FastParams fp;
        fp.width = in_w, fp.height = in_h;
        fast_params_finish (fp, p.matrix.p, ident, p.front.u_plane);
        fp.crow_lo = -(p.rect.in_y >> 1);
        fp.crow_hi = ((p.rect.in_maxh + 1) >> 1) - 1 - (p.rect.in_y >> 1);
        const uint8_t *y = pl.p[0], *uv = pl.p[1];
        uint8_t *img = c->pre_img;
        if ((e = launch_convert_pair (fp, p.front.chroma_h, 1, &y, &uv, &img, pl.stride[0], pl.stride[1], in_w * 4, stream)) != hipSuccess)
          return hip_fail (e, "k_convert_pair(source size)");
      } else
      if ((e = launch_convert (p.front, pl, c->vpair_dev, pre, ident, c->pre_img, in_w * 4, stream, 0)) != hipSuccess)
        return hip_fail (e, "k_convert(source size)");
      FrontParams f4;
      memset ((void *) &f4, 0, sizeof (f4));
      f4.kind = UNPACK_PACKED4;
      f4.width = in_w, f4.height = in_h;
      for (int i = 0; i < 4; i++)
        f4.pos[i] = i;
      f4.swap_k = -1;
      Planes p4;
      memset ((void *) &p4, 0, sizeof (p4));
      p4.p[0] = c->pre_img, p4.stride[0] = in_w * 4;
      e = launch_scale2x2_from_front (f4, p4, nullptr, none, sh, sv, h_first, dst, dstride, post, p.post.pack_pos, out_w, out_h,
          p.passes[h_first ? 0 : 1].max_span, c->geom[h_first ? 0 : 1], pf_none, stream);
      return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "k_bilinear4_rows");
    }
    e = launch_scale2x2_from_front (p.front, pl, c->vpair_dev, pre, sh, sv, h_first, dst, dstride, post, p.post.pack_pos,
        out_w, out_h, p.passes[h_first ? 0 : 1].max_span, c->geom[h_first ? 0 : 1], pf, stream);
    return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "k_scale2x2");
  }
  /* a 4-byte packed source whose unpack is the identity (ARGB, AYUV, and every 4-byte format in plane scaling, where the bytes go through
     raw) with no colour step before the scaler IS an image in the scalers' own layout: the image kernels (wave tiles, k_vscale_pk) take it
     directly instead of the one-lane-per-pixel front kernels */
  const bool raw4 = p.front.kind == UNPACK_PACKED4 && p.front.pos[0] == 0 && p.front.pos[1] == 1 && p.front.pos[2] == 2 && p.front.pos[3] == 3 &&
      pre.matrix.kind == MATRIX_NONE && pre.alpha_kind == ALPHA_NONE && ((uintptr_t) pl.p[0] % 4) == 0 && (pl.stride[0] % 4) == 0;
  if (p.passes.size () == 1) {
    if (raw4)
      e = launch_scale_from_image (p.passes[0].horizontal, pl.p[0], pl.stride[0], sd[0], dst, dstride, true, post, p.post.pack_pos, out_w, out_h,
          p.passes[0].max_span, p.front.width, c->geom[0], pf, stream);
    else
    e = launch_scale_from_front (p.passes[0].horizontal, p.front, pl, c->vpair_dev, pre, sd[0], dst, dstride, true,
        post, p.post.pack_pos, out_w, out_h, p.passes[0].max_span, c->geom[0], pf, stream);
    return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "scale pass");
  }
  if (col_usable (c, pl, pre)) {
    const void *sp[3] = {pl.p[0], pl.p[1], pl.p[2]};
    e = col_launch (c, 1, &sp, &dst, pl, dstride, post, pf, stream);
    if (e == hipSuccess)
      return GSTAMD_OK;
    if (e != hipErrorNotSupported)
      return hip_fail (e, "k_scale_col");
  }
  e = hipErrorNotSupported;
  const bool reg_usable = c->reg420 && pre.matrix.kind == MATRIX_NONE && pre.alpha_kind == ALPHA_NONE &&
      (p.front.kind == UNPACK_SEMI || pl.stride[p.front.u_plane] == pl.stride[p.front.v_plane]);
  if (reg_usable && c->fused_ok) {
    Fused420Params fq;
    memset (&fq, 0, sizeof (fq));
    H420RegParams &hp = fq.h;
    hp.y = pl.p[0];
    hp.ystride = pl.stride[0];
    hp.semi = p.front.kind == UNPACK_SEMI;
    hp.u_first = p.front.u_plane != 0;
    hp.c0 = hp.semi ? pl.p[1] : pl.p[p.front.u_plane];
    hp.c1 = hp.semi ? pl.p[1] : pl.p[p.front.v_plane];
    hp.cstride = hp.semi ? pl.stride[1] : pl.stride[p.front.u_plane];
    hp.width = p.front.width;
    hp.height = p.front.height;
    hp.crow_lo = c->reg_lo;
    hp.crow_hi = c->reg_hi;
    hp.offset = sd[0].offset;
    hp.tapw = sd[0].tapw;
    hp.nw4 = sd[0].nw4;
    hp.out_w = p.passes[0].out_size;
    hp.tile_w = c->geom[0].tile16_w;
    fq.n_taps_h = sd[0].n_taps;
    fq.vgroup = c->vgroup_dev;
    fq.vtapw = c->vtapw_dev;
    fq.ngv = c->fused.ngv;
    fq.out_h = out_h;
    fq.rows_per_chunk = c->fused_rpc;
    fq.first_rows = c->fused_first;
    fq.ring = c->fused_ring;
    fq.sched = c->fused_sched;
    fq.n_groups = c->fused.n_groups;
#ifdef GSTAMD_TUNING
    /* profiling builds: GSTAMD_FUSED_TRACE=<file> dumps the per-wave stage stamps of every launch (the last one stays) */
    const char *trace_path = tuning_text ("GSTAMD_FUSED_TRACE");
    const size_t trace_n = (size_t) ((hp.out_w + hp.tile_w - 1) / hp.tile_w) * ((out_h + fq.rows_per_chunk - 1) / fq.rows_per_chunk) * 16 * 32;
    if (trace_path && hipMalloc ((void **) &fq.trace, trace_n * 8) == hipSuccess)
      (void) hipMemset (fq.trace, 0, trace_n * 8);
#endif
    e = hipErrorNotSupported;
    e = launch_scale420_fused (fq, p.front.chroma_h, sd[0].nw, c->fused_waves, dst, dstride, post, p.post.pack_pos, pf, stream);
#ifdef GSTAMD_TUNING
    if (fq.trace) {
      std::vector<unsigned long long> h (trace_n);
      (void) hipStreamSynchronize (stream);
      (void) hipMemcpy (h.data (), fq.trace, trace_n * 8, hipMemcpyDeviceToHost);
      (void) hipFree (fq.trace);
      if (FILE *f = fopen (trace_path, "wb")) {
        fwrite (h.data (), 8, trace_n, f);
        fclose (f);
      }
    }
#endif
    if (e == hipSuccess)
      return GSTAMD_OK;
    if (e != hipErrorNotSupported)
      return hip_fail (e, "k_scale420_fused");
  }
  if (!c->tmp && (e = hipMalloc ((void **) &c->tmp, c->tmp_size)) != hipSuccess)
    return hip_fail (e, "hipMalloc(tmp)");
  e = hipErrorNotSupported;
  if (reg_usable) {
    H420RegParams hp;
    memset (&hp, 0, sizeof (hp));
    hp.y = pl.p[0];
    hp.ystride = pl.stride[0];
    hp.semi = p.front.kind == UNPACK_SEMI;
    hp.u_first = p.front.u_plane != 0;
    hp.c0 = hp.semi ? pl.p[1] : pl.p[p.front.u_plane];
    hp.c1 = hp.semi ? pl.p[1] : pl.p[p.front.v_plane];
    hp.cstride = hp.semi ? pl.stride[1] : pl.stride[p.front.u_plane];
    hp.width = p.front.width;
    hp.height = p.front.height;
    hp.crow_lo = c->reg_lo;
    hp.crow_hi = c->reg_hi;
    hp.offset = sd[0].offset;
    hp.tapw = sd[0].tapw;
    hp.nw4 = sd[0].nw4;
    hp.dst = c->tmp;
    hp.dstride = c->tmp_w * 4;
    hp.out_w = c->tmp_w;
    hp.tile_w = c->geom[0].tile16_w;
    e = launch_hscale420_reg (hp, p.front.chroma_h, sd[0].nw, sd[0].n_taps, stream);
    if (e != hipSuccess && e != hipErrorNotSupported)
      return hip_fail (e, "k_hscale420_reg");
  }
  if (e == hipErrorNotSupported && raw4)
    e = launch_scale_from_image (p.passes[0].horizontal, pl.p[0], pl.stride[0], sd[0], c->tmp, c->tmp_w * 4, false, none, p.post.pack_pos, c->tmp_w,
        c->tmp_h, p.passes[0].max_span, p.front.width, c->geom[0], pf_none, stream);
  else if (e == hipErrorNotSupported)
  e = launch_scale_from_front (p.passes[0].horizontal, p.front, pl, c->vpair_dev, pre, sd[0], c->tmp, c->tmp_w * 4,
      false, none, p.post.pack_pos, c->tmp_w, c->tmp_h, p.passes[0].max_span, c->geom[0], pf_none, stream);
  if (e != hipSuccess)
    return hip_fail (e, "scale pass 1");
  e = launch_scale_from_image (p.passes[1].horizontal, c->tmp, c->tmp_w * 4, sd[1], dst, dstride, true, post,
      p.post.pack_pos, out_w, out_h, p.passes[1].max_span, c->tmp_w, c->geom[1], pf, stream);
  return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "scale pass 2");
}

int gstamd_video_converter_frame (GstAmdVideoConverter *c, const void *src, void *dest, void *stream)
{
  if (!c || !src || !dest)
    return set_error (GSTAMD_ERR_INVALID, "NULL converter or frame");
  const void *sp[GSTAMD_VIDEO_MAX_PLANES] = {nullptr, nullptr, nullptr, nullptr};
  void *dp[GSTAMD_VIDEO_MAX_PLANES] = {nullptr, nullptr, nullptr, nullptr};
  for (int i = 0; i < c->plan.in_info.n_planes; i++)
    sp[i] = (const uint8_t *) src + c->plan.in_info.offset[i];
  for (int i = 0; i < c->plan.out_info.n_planes; i++)
    dp[i] = (uint8_t *) dest + c->plan.out_info.offset[i];
  return frame_planes_plan_order (c, sp, nullptr, dp, nullptr, stream);
}

// frames [0, n) of a list one by one on the caller's stream.  (Round 5 tried fanning them out over four internal streams, forked from and joined
// into the caller's with events, a scratch set per stream: lists of 8 got SLOWER - BGRA 4K -> NV12 1080p bilinear 16.4 -> 23.5 us per frame, I420 4K ->
// RGBA 720p 7.9 -> 15.4 - the cross-queue event waits cost more than the overlap of the short launches gives; profiles/r05/survey_fan_out_4_streams.log.)
static int frames_one_by_one (GstAmdVideoConverter *c, int n_frames, const void *const *src, void *const *dest, void *stream_)
{
  for (int i = 0; i < n_frames; i++) {
    const int r = gstamd_video_converter_frame (c, src[i], dest[i], stream_);
    if (r != GSTAMD_OK)
      return r;
  }
  return GSTAMD_OK;
}

int gstamd_video_converter_frames (GstAmdVideoConverter *c, int n_frames, const void *const *src, void *const *dest, void *stream_)
{
  if (!c || n_frames < 0 || (n_frames > 0 && (!src || !dest)))
    return set_error (GSTAMD_ERR_INVALID, "NULL converter or frame list");
  if (n_frames == 0)
    return GSTAMD_OK;
  const VideoPlan &p = c->plan;
  c->list_launches = 0;
  if (p.interlaced)
    return frames_one_by_one (c, n_frames, src, dest, stream_);
  int r = ensure_tables (c);
  if (r != GSTAMD_OK || (r = bind_scratch (c, stream_)) != GSTAMD_OK)
    return r;
  if (p.gamma.on && p.gamma.fused && p.gamma.lut_direct && !p.rect.fill && c->sub_in && n_frames > 1) {
    /* gamma remap collapsed into the direct conversion + one composed table: the direct conversion's frame list, the table inside its
       kernel where that kernel applies it (the line-pair kernel) */
    c->sub_in->hook_on = false;
    c->sub_in->post_lut = c->gamma_comp_dev;
    c->sub_in->post_lut_keep = p.fout->pos[0];
    c->sub_in->post_lut_done = false;
    r = gstamd_video_converter_frames (c->sub_in, n_frames, src, dest, stream_);
    c->list_launches = c->sub_in->post_lut_done ? c->sub_in->list_launches : 0;
    if (r != GSTAMD_OK || c->sub_in->post_lut_done)
      return r;
    for (int i = 0; i < n_frames; i++) {
      const int ds = p.out_info.stride[0];
      uint8_t *rect = (uint8_t *) dest[i] + p.out_info.offset[0] + plane_origin (p.fout, 0, p.rect.out_x, p.rect.out_y, ds);
      const hipError_t le = launch_lut3 (rect, ds, p.out_info.width, p.out_info.height, c->gamma_comp_dev, p.fout->pos[0], (hipStream_t) stream_);
      if (le != hipSuccess)
        return hip_fail (le, "k_lut3");
    }
    return GSTAMD_OK;
  }
  Enc16Params ep16_list;
  const bool enc16_list = p.gamma.on && enc16_params (p, &ep16_list) && !tuning_on ("GSTAMD_NO_ENCODE16");         /* k_encode16: one kernel, takes lists */
  DeepPackParams dsp_list;
  const bool deep_pack_list = p.gamma.on && !c->hook_on && !tuning_on ("GSTAMD_NO_DEEP_SCALE_PACK") &&
      ((deep_pack_usable (c, &dsp_list) && !c->sub_out->plan.rect.fill) || deep_scale_pack16_plan_ok (p, &dsp_list));        /* k_deep_scale_pack / _pack16: one kernel, takes lists */
  if (p.gamma.on && !p.gamma.planes_fast && !enc16_list && !deep_pack_list)
    return frames_one_by_one (c, n_frames, src, dest, stream_);
  /* one launch for the whole list when the line-pair kernel applies to every frame */
  bool all_fast = true;
  std::vector<const uint8_t *> y (n_frames), uv (n_frames);
  std::vector<uint8_t *> d (n_frames);
  for (int i = 0; i < n_frames && all_fast; i++) {
    if (!src[i] || !dest[i])
      return set_error (GSTAMD_ERR_INVALID, "NULL frame in list");
    Planes pl;
    memset (&pl, 0, sizeof (pl));
    pl.p[0] = (const uint8_t *) src[i] + p.in_info.offset[0];
    pl.p[1] = (const uint8_t *) src[i] + p.in_info.offset[1];
    pl.stride[0] = p.in_info.stride[0];
    pl.stride[1] = p.in_info.stride[1];
    d[i] = (uint8_t *) dest[i] + p.out_info.offset[0];
    all_fast = (p.fout->kind == UNPACK_PACKED4 || p.fout->kind == UNPACK_PACKED3) &&
        fast_pair_usable (p, pl, d[i], p.out_info.stride[0], p.fout->kind == UNPACK_PACKED3 ? 4 : 16) && p.rect.in_x == 0 && p.rect.in_y == 0 && p.rect.out_x == 0 &&
        p.rect.out_y == 0 && !p.rect.fill;
    y[i] = pl.p[0];
    uv[i] = pl.p[1];
  }
  if (all_fast) {
    FastParams fp = make_fast_params (p, p.fout->kind == UNPACK_PACKED3);
    if (c->post_lut) {          /* this converter is the direct conversion of a collapsed gamma remap (see above) */
      fp.lut = c->post_lut;
      fp.lut_keep = c->post_lut_keep;
      c->post_lut_done = true;
    }
    hipError_t e = launch_convert_pair (fp, p.front.chroma_h, n_frames, y.data (), uv.data (), d.data (), p.in_info.stride[0],
        p.in_info.stride[1], p.out_info.stride[0], (hipStream_t) stream_);
    for (int i = 0; i < n_frames && e == hipSuccess && p.dither.on; i++)
      e = launch_dither4 (p.dither, d[i], p.out_info.stride[0], p.out_info.width, p.out_info.height, (hipStream_t) stream_, c->ed_carry);
    c->list_launches = p.dither.on ? 0 : (n_frames + 31) / 32;
    return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "k_convert_pair(batch)");
  }
  /* one launch for the whole list through the bilinear 4:2:0 kernel as well (k_bilinear420_rows takes the frames as one grid) */
  BilParams bp;
  if (!p.rect.fill && p.rect.in_x == 0 && p.rect.in_y == 0 && p.rect.out_x == 0 && p.rect.out_y == 0 && bilinear420_params (c, &bp) && bp.rows != 0) {
    std::vector<Planes> pls (n_frames);
    bool ok = true;
    for (int i = 0; i < n_frames && ok; i++) {
      if (!src[i] || !dest[i])
        return set_error (GSTAMD_ERR_INVALID, "NULL frame in list");
      memset (&pls[i], 0, sizeof (Planes));
      for (int k = 0; k < p.in_info.n_planes && k < 3; k++) {
        pls[i].p[k] = (const uint8_t *) src[i] + p.in_info.offset[k];
        pls[i].stride[k] = p.in_info.stride[k];
      }
      d[i] = (uint8_t *) dest[i] + p.out_info.offset[0];
      ok = ((uintptr_t) d[i] % 4) == 0;
    }
    if (ok && (p.out_info.stride[0] % 4) == 0) {
      hipError_t e = launch_bilinear420_frames (bp, p.front.chroma_h, n_frames, pls.data (), d.data (), p.out_info.stride[0], (hipStream_t) stream_);
      for (int i = 0; i < n_frames && e == hipSuccess && p.dither.on; i++)
        e = launch_dither4 (p.dither, d[i], p.out_info.stride[0], p.out_info.width, p.out_info.height, (hipStream_t) stream_, c->ed_carry);
      if (e != hipErrorNotSupported) {
        c->list_launches = p.dither.on ? 0 : 1;
        return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "k_bilinear420(batch)");
      }
    }
  }
  /* ... and through the column-walk scaler (k_scale_col takes the frames as the grid's third dimension) */
  if (c->col_ok && !p.rect.fill && p.rect.in_x == 0 && p.rect.in_y == 0 && p.rect.out_x == 0 && p.rect.out_y == 0 && !p.out_planar && !p.plane_mode &&
      p.fout->kind == UNPACK_PACKED4 && p.fout->hi_depth == 0 && !p.matrix_before_scale) {
    Planes pl0;
    memset (&pl0, 0, sizeof (pl0));
    for (int k = 0; k < p.in_info.n_planes && k < 3; k++)
      pl0.stride[k] = p.in_info.stride[k];
    ColorParams none_c, color;
    memset (&none_c, 0, sizeof (none_c));
    color.matrix = p.matrix;
    color.alpha_kind = p.post.alpha_kind;
    color.alpha_value = p.post.alpha_value;
    if (col_usable (c, pl0, none_c)) {
      std::vector<std::array<const void *, 3>> sp (n_frames);
      for (int i = 0; i < n_frames; i++) {
        if (!src[i] || !dest[i])
          return set_error (GSTAMD_ERR_INVALID, "NULL frame in list");
        for (int k = 0; k < 3; k++)
          sp[i][k] = k < p.in_info.n_planes ? (const uint8_t *) src[i] + p.in_info.offset[k] : nullptr;
        d[i] = (uint8_t *) dest[i] + p.out_info.offset[0];
      }
      PostFast pf;
      pf.use = p.fast_post ? 1 : 0;
      pf.fp = make_fast_params (p);
      hipError_t e = col_launch (c, n_frames, (const void *const (*)[3]) sp.data (), d.data (), pl0, p.out_info.stride[0], color, pf, (hipStream_t) stream_);
      for (int i = 0; i < n_frames && e == hipSuccess && p.dither.on; i++)
        e = launch_dither4 (p.dither, d[i], p.out_info.stride[0], p.out_info.width, p.out_info.height, (hipStream_t) stream_, c->ed_carry);
      if (e != hipErrorNotSupported) {
        c->list_launches = p.dither.on ? 0 : (n_frames + GSTAMD_COL_MAX_FRAMES - 1) / GSTAMD_COL_MAX_FRAMES;
        return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "k_scale_col(batch)");
      }
    }
  }
  /* plans that convert a frame in ONE launch of a kernel that takes frame lists (video_kernels.hip: the list is the grid's third
     dimension): frame `base` is converted with the list of up to 32 frames behind it armed; the launcher says whether it took the list */
  int base = 0;
  if (!p.rect.fill && !p.dither.on && (!p.gamma.on || p.gamma.planes_fast || enc16_list || deep_pack_list)) {
    /* a plan that scales into the pack image and packs from it with list-taking kernels (k_plane_quad + k_encode420 / k_convert_pack: raw4_pack) gets
       one image per frame of the list - up to 1 GiB of them - so that both launches take the list */
    if (c->raw4_pack && c->pk_img && !c->hook_on) {
      const size_t per = pk_img_bytes (p);
      const int want = (int) std::min<size_t> ((size_t) std::min (n_frames, 32), std::max<size_t> (1, ((size_t) 1 << 30) / per));
      if (want > c->pk_img_frames) {
        uint8_t *bigger = nullptr;
        if (hipMalloc ((void **) &bigger, per * (size_t) want) == hipSuccess) {
          (void) hipStreamSynchronize ((hipStream_t) stream_);          /* nothing of this stream reads the old image any more */
          (void) hipFree (c->pk_img);
          c->pk_img = bigger;
          c->pk_img_frames = want;
        } else
          (void) hipGetLastError ();
      }
    }
    while (n_frames - base >= 2) {
      int nb = std::min (n_frames - base, 32);
      if (c->raw4_pack && c->pk_img && c->pk_img_frames > 1)
        nb = std::min (nb, c->pk_img_frames);
      for (int i = 0; i < nb; i++)
        if (!src[base + i] || !dest[base + i])
          return set_error (GSTAMD_ERR_INVALID, "NULL frame in list");
      video_frame_list_begin (nb, src + base, dest + base, p.in_info.size, p.out_info.size);
      if (c->pk_img && c->pk_img_frames >= nb)
        video_frame_list_scratch (c->pk_img, pk_img_bytes (p));
      r = gstamd_video_converter_frame (c, src[base], dest[base], stream_);
      const int used = video_frame_list_end ();
      if (r != GSTAMD_OK)
        return r;
      if (!used) {
        base++;                 /* this plan's kernels do not take lists: frame by frame from here */
        break;
      }
      c->list_launches += used;
      base += nb;
    }
  }
  return frames_one_by_one (c, n_frames - base, src + base, dest + base, stream_);
}

int gstamd_video_converter_list_launches (GstAmdVideoConverter *c)
{
  return c ? c->list_launches : 0;
}

// every device table / scratch image of the converter; the plan stays
static void release_tables (GstAmdVideoConverter *c)
{
  for (auto &ps : c->parked)
    free_scratch (ps.second);
  c->parked.clear ();
  c->bound = false;
  c->bound_stream = nullptr;
  if (c->vpair_dev)
    (void) hipFree (c->vpair_dev);
  c->vpair_dev = nullptr;
  for (auto &pd : c->pass_dev) {
    if (pd.offset)
      (void) hipFree (pd.offset);
    if (pd.taps)
      (void) hipFree (pd.taps);
    if (pd.tapw)
      (void) hipFree (pd.tapw);
    pd = GstAmdVideoConverter::PassDev ();
  }
  if (c->tmp)
    (void) hipFree (c->tmp);
  if (c->deep_a)
    (void) hipFree (c->deep_a);
  if (c->deep_b)
    (void) hipFree (c->deep_b);
  if (c->gamma_dec16_dev)
    (void) hipFree (c->gamma_dec16_dev);
  if (c->gamma_enc16_dev)
    (void) hipFree (c->gamma_enc16_dev);
  c->gamma_dec16_dev = c->gamma_enc16_dev = nullptr;
  if (c->gamma_dec_dev)
    (void) hipFree (c->gamma_dec_dev);
  if (c->gamma_enc_dev)
    (void) hipFree (c->gamma_enc_dev);
  if (c->gamma_comp_dev)
    (void) hipFree (c->gamma_comp_dev);
  c->gamma_comp_dev = nullptr;
  if (c->gamma_mid_a)
    (void) hipFree (c->gamma_mid_a);
  if (c->gamma_mid_b)
    (void) hipFree (c->gamma_mid_b);
  c->gamma_dec_dev = nullptr;
  c->gamma_enc_dev = nullptr;
  c->gamma_mid_a = c->gamma_mid_b = nullptr;
  c->deep_a = c->deep_b = nullptr;
  c->deep_a_size = c->deep_b_size = 0;
  if (c->vgroup_dev)
    (void) hipFree (c->vgroup_dev);
  if (c->vtapw_dev)
    (void) hipFree (c->vtapw_dev);
  if (c->pk_img)
    (void) hipFree (c->pk_img);
  if (c->ed_carry)
    (void) hipFree (c->ed_carry);
  c->ed_carry = nullptr;
  if (c->pre_img)
    (void) hipFree (c->pre_img);
  c->pre_img = nullptr;
  c->pre_img_size = 0;
  if (c->plane_tmp)
    (void) hipFree (c->plane_tmp);
  c->tmp = c->pk_img = c->plane_tmp = nullptr;
  c->vgroup_dev = nullptr;
  c->vtapw_dev = nullptr;
  for (auto &v : c->plane_dev)
    for (auto &pd : v) {
      if (pd.offset)
        (void) hipFree (pd.offset);
      if (pd.taps)
        (void) hipFree (pd.taps);
    }
  c->plane_dev.clear ();
  c->fused_ok = false;
  if (c->walk_vt_dev)
    (void) hipFree (c->walk_vt_dev);
  if (c->walk_ht_dev)
    (void) hipFree (c->walk_ht_dev);
  c->walk_vt_dev = c->walk_ht_dev = nullptr;
  c->walk_state = 0;
  if (c->col_tiles_dev)
    (void) hipFree (c->col_tiles_dev);
  if (c->col_hout_dev)
    (void) hipFree (c->col_hout_dev);
  if (c->col_vrow_dev)
    (void) hipFree (c->col_vrow_dev);
  c->col_tiles_dev = nullptr;
  c->col_hout_dev = c->col_vrow_dev = nullptr;
  c->col_ok = false;
  c->reg420 = false;
  c->tables_ready = false;
}

void gstamd_video_converter_free (GstAmdVideoConverter *c)
{
  if (!c)
    return;
  release_tables (c);
  gstamd_video_converter_free (c->sub_in);
  gstamd_video_converter_free (c->sub_out);
  gstamd_video_converter_free (c->field[0]);
  gstamd_video_converter_free (c->field[1]);
  delete c;
}

int gstamd_video_converter_set_config (GstAmdVideoConverter *c, const GstAmdVideoConverterConfig *config)
{
  if (!c || !config)
    return set_error (GSTAMD_ERR_INVALID, "NULL converter or config");
  if (c->plan.interlaced) {
    /* both field conversions take the new options, or neither does (the first one's failure leaves everything as it was; the second cannot fail where
       the first did not - the two plans differ in their field's tables only) */
    GstAmdVideoConverterConfig fcfg;
    if (!plan_field_config (&c->plan.orig_in, &c->plan.orig_out, config, &fcfg))
      return set_error (GSTAMD_ERR_UNSUPPORTED, "interlaced frames with a source crop or a destination rectangle are not implemented on the GPU path");
    int fr = gstamd_video_converter_set_config (c->field[0], &fcfg);
    if (fr == GSTAMD_OK)
      fr = gstamd_video_converter_set_config (c->field[1], &fcfg);
    if (fr == GSTAMD_OK) {
      VideoPlan top;
      std::string terr;
      const GstAmdVideoInfo tin = c->plan.orig_in, tout = c->plan.orig_out;
      if (plan_video_converter (&tin, &tout, config, &top, &terr) == GSTAMD_OK)
        c->plan = std::move (top);
    }
    return fr;
  }
  VideoPlan np;
  std::string err;
  const GstAmdVideoInfo in = c->plan.orig_in, out = c->plan.orig_out;
  int r = plan_video_converter (&in, &out, config, &np, &err);
  if (r != GSTAMD_OK)
    return set_error (r, err);          /* the old plan stays, like a failed gst_video_converter_set_config leaves the converter usable */
  /* the new plan's sub-conversions first; on failure the old plan and the old sub-converters stay */
  GstAmdVideoConverter *nin = nullptr, *nout = nullptr;
  if ((r = build_sub_converters (np, &nin, &nout)) != GSTAMD_OK)
    return r;
  GstAmdVideoConverter *oin, *oout;
  {
    std::lock_guard<std::mutex> g (c->lock);
    (void) hipDeviceSynchronize ();       /* frames in flight still read the old tables */
    release_tables (c);
    c->plan = std::move (np);
    oin = c->sub_in;
    oout = c->sub_out;
    c->sub_in = nin;
    c->sub_out = nout;
    c->hook_on = false;
  }
  gstamd_video_converter_free (oin);
  gstamd_video_converter_free (oout);
  return GSTAMD_OK;
}

int gstamd_internal_pad_scaler_tile_rows (GstAmdVideoConverter *c)
{
  return c ? scaled_tile_rows_for (c->plan) : 16;
}

/* internal (compositor_kernels.hip): the scaler passes of a pad converter as device tables; 0 when the plan is more than that */
int gstamd_internal_pad_scaler (GstAmdVideoConverter *c, gstamd::ScaleDev *sh, gstamd::ScaleDev *sv, int *h_first, int *in_w, int *in_h,
    int *out_w, int *out_h, int *format)
{
  int hi = -1, vi = -1;
  if (!c || !plan_is_pad_scaler (c->plan, &hi, &vi) || ensure_tables (c) != GSTAMD_OK)
    return 0;
  const VideoPlan &p = c->plan;
  ScaleDev *out[2] = {sh, sv};
  const int idx[2] = {hi, vi};
  for (int k = 0; k < 2; k++) {
    ScaleDev sd;
    if (idx[k] >= 0) {
      const ScalePass &sp = p.passes[idx[k]];
      sd.kind = sp.kind;
      sd.n_taps = sp.n_taps;
      sd.inc = sp.inc;
      sd.offset = c->pass_dev[idx[k]].offset;
      sd.taps = c->pass_dev[idx[k]].taps;
    }
    *out[k] = sd;
  }
  *h_first = hi >= 0 && vi >= 0 && hi < vi;
  *in_w = p.in_info.width;
  *in_h = p.in_info.height;
  *out_w = p.out_info.width;
  *out_h = p.out_info.height;
  *format = p.in_info.format;
  return 1;
}

// One 8-tap pass onto a ring that advances two source positions per output: output i reads positions 2 i + base .. 2 i + base + 7.
// words: four per output (two s16 taps per word, even ring position in the low half).  false: some non-zero tap falls outside the ring.
static bool walk_remap (const ScalePass &sp, std::vector<uint32_t> *words, int *base)
{
  if (sp.kind != SCALE_NTAP || sp.n_taps != 8 || sp.merged || sp.out_size < 1 || (int) sp.offset.size () < sp.out_size ||
      (int) sp.taps.size () < sp.out_size * 8)
    return false;
  const int mid = sp.out_size / 2;
  const int b = (int) sp.offset[mid] - 2 * mid;
  words->assign ((size_t) sp.out_size * 4, 0u);
  for (int i = 0; i < sp.out_size; i++) {
    int16_t ring[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < 8; k++) {
      const int16_t t = sp.taps[(size_t) i * 8 + k];
      if (!t)
        continue;
      const int pos = (int) sp.offset[i] + k - (2 * i + b);
      if (pos < 0 || pos > 7 || (int) sp.offset[i] + k >= sp.in_size)
        return false;
      ring[pos] = (int16_t) (ring[pos] + t);
    }
    for (int k = 0; k < 4; k++)
      (*words)[(size_t) i * 4 + k] = (uint32_t) (uint16_t) ring[2 * k] | ((uint32_t) (uint16_t) ring[2 * k + 1] << 16);
  }
  *base = b;
  return true;
}

/* internal (compositor_kernels.hip): the walk tables of a pad converter (compositor_walk.h); 0 when the plan does not fit - a pad
 * scaler whose vertical 8-tap pass runs first and whose passes both halve */
int gstamd_internal_pad_walk (GstAmdVideoConverter *c, const uint32_t **vt, const uint32_t **ht, int *vbase, int *hbase)
{
  int hi = -1, vi = -1;
  if (!c || !plan_is_pad_scaler (c->plan, &hi, &vi) || hi < 0 || vi < 0 || vi > hi)
    return 0;
  std::lock_guard<std::mutex> guard (c->lock);
  if (c->walk_state == 0) {
    c->walk_state = -1;
    std::vector<uint32_t> wv, wh;
    int bv = 0, bh = 0;
    if (walk_remap (c->plan.passes[vi], &wv, &bv) && walk_remap (c->plan.passes[hi], &wh, &bh)) {
      if (hipMalloc ((void **) &c->walk_vt_dev, wv.size () * 4) == hipSuccess && hipMalloc ((void **) &c->walk_ht_dev, wh.size () * 4) == hipSuccess &&
          hipMemcpy (c->walk_vt_dev, wv.data (), wv.size () * 4, hipMemcpyHostToDevice) == hipSuccess &&
          hipMemcpy (c->walk_ht_dev, wh.data (), wh.size () * 4, hipMemcpyHostToDevice) == hipSuccess) {
        c->walk_vbase = bv;
        c->walk_hbase = bh;
        c->walk_state = 1;
      } else {
        (void) hipGetLastError ();
      }
    }
  }
  if (c->walk_state != 1)
    return 0;
  *vt = c->walk_vt_dev;
  *ht = c->walk_ht_dev;
  *vbase = c->walk_vbase;
  *hbase = c->walk_hbase;
  return 1;
}

const char *gstamd_video_converter_describe (const GstAmdVideoConverter *c)
{
  return c ? c->plan.description.c_str () : "";
}

const char *gstamd_video_converter_divergence (const GstAmdVideoConverter *c)
{
  return c ? c->plan.divergence.c_str () : "";
}

uint64_t gstamd_video_converter_algorithmic_bytes (const GstAmdVideoConverter *c)
{
  return c ? c->plan.algorithmic_bytes : 0;
}

/* test introspection (not part of the drop-in surface): copies planner tables out.
 * what: 0 = matrix {kind, p[5], im[3][4]} (18 ints), 1 = vpair (2*in_height ints),
 *       10+i = offsets of pass i (uint32), 20+i = taps of pass i (int16 widened to int32),
 *       30+i = {kind, horizontal, n_taps, inc, in_size, out_size} of pass i.  Returns #ints or <0. */
int gstamd_video_converter_debug_get (const GstAmdVideoConverter *c, int what, int32_t *out, int max_out)
{
  if (!c)
    return -1;
  const VideoPlan &p = c->plan;
  std::vector<int32_t> v;
  if (what == 0) {
    v.push_back (p.matrix.kind);
    for (int i = 0; i < 5; i++)
      v.push_back (p.matrix.p[i]);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 4; j++)
        v.push_back (p.matrix.im[i][j]);
  } else if (what == 1) {
    v.assign (p.vpair.begin (), p.vpair.end ());
  } else if (what >= 10 && what < 40) {
    size_t i = (size_t) (what % 10);
    if (i >= p.passes.size ())
      return -1;
    const ScalePass &sp = p.passes[i];
    if (what < 20)
      v.assign (sp.offset.begin (), sp.offset.end ());
    else if (what < 30)
      v.assign (sp.taps.begin (), sp.taps.end ());
    else
      v = {sp.kind, sp.horizontal ? 1 : 0, sp.n_taps, sp.inc, sp.in_size, sp.out_size};
  } else {
    return -1;
  }
  if (out) {
    int n = (int) v.size () < max_out ? (int) v.size () : max_out;
    memcpy (out, v.data (), (size_t) n * sizeof (int32_t));
  }
  return (int) v.size ();
}

/* ---- device helpers ------------------------------------------------------------------- */
void *gstamd_device_alloc (size_t size)
{
  void *p = nullptr;
  hipError_t e = hipMalloc (&p, size);
  if (e != hipSuccess) {
    hip_fail (e, "hipMalloc");
    return nullptr;
  }
  return p;
}

void gstamd_device_free (void *ptr)
{
  if (ptr)
    (void) hipFree (ptr);
}

int gstamd_device_upload (void *dst_device, const void *src_host, size_t size, void *stream)
{
  hipError_t e = hipMemcpyAsync (dst_device, src_host, size, hipMemcpyHostToDevice, (hipStream_t) stream);
  return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "upload");
}

int gstamd_device_download (void *dst_host, const void *src_device, size_t size, void *stream)
{
  hipError_t e = hipMemcpyAsync (dst_host, src_device, size, hipMemcpyDeviceToHost, (hipStream_t) stream);
  if (e == hipSuccess)
    e = hipStreamSynchronize ((hipStream_t) stream);
  return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "download");
}

int gstamd_stream_synchronize (void *stream)
{
  hipError_t e = hipStreamSynchronize ((hipStream_t) stream);
  return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "hipStreamSynchronize");
}

void *gstamd_stream_new (void)
{
  hipStream_t st = nullptr;
  hipError_t e = hipStreamCreateWithFlags (&st, hipStreamNonBlocking);
  if (e != hipSuccess) {
    hip_fail (e, "hipStreamCreate");
    return nullptr;
  }
  return st;
}

void gstamd_stream_free (void *stream)
{
  if (stream)
    (void) hipStreamDestroy ((hipStream_t) stream);
}

void *gstamd_event_new (void)
{
  hipEvent_t ev = nullptr;
  hipError_t e = hipEventCreateWithFlags (&ev, hipEventDisableTiming);
  if (e != hipSuccess) {
    hip_fail (e, "hipEventCreate");
    return nullptr;
  }
  return ev;
}

void gstamd_event_free (void *event)
{
  if (event)
    (void) hipEventDestroy ((hipEvent_t) event);
}

int gstamd_event_record (void *event, void *stream)
{
  hipError_t e = hipEventRecord ((hipEvent_t) event, (hipStream_t) stream);
  return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "hipEventRecord");
}

int gstamd_stream_wait_event (void *stream, void *event)
{
  hipError_t e = hipStreamWaitEvent ((hipStream_t) stream, (hipEvent_t) event, 0);
  return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "hipStreamWaitEvent");
}

int gstamd_event_synchronize (void *event)
{
  hipError_t e = hipEventSynchronize ((hipEvent_t) event);
  return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "hipEventSynchronize");
}

int gstamd_event_query (void *event)
{
  hipError_t e = hipEventQuery ((hipEvent_t) event);
  if (e == hipSuccess)
    return 1;
  if (e == hipErrorNotReady)
    return 0;
  return hip_fail (e, "hipEventQuery");
}

void *gstamd_host_alloc (size_t size)
{
  void *p = nullptr;
  hipError_t e = hipHostMalloc (&p, size, hipHostMallocDefault);
  if (e != hipSuccess) {
    hip_fail (e, "hipHostMalloc");
    return nullptr;
  }
  return p;
}

void gstamd_host_free (void *ptr)
{
  if (ptr)
    (void) hipHostFree (ptr);
}

int gstamd_host_is_pinned (const void *ptr)
{
  if (!ptr)
    return 0;
  hipPointerAttribute_t attr;
  memset (&attr, 0, sizeof (attr));
  const hipError_t e = hipPointerGetAttributes (&attr, ptr);
  if (e != hipSuccess) {
    (void) hipGetLastError ();          /* pageable memory: "invalid value" is the answer, not a failure */
    return 0;
  }
  return attr.type == hipMemoryTypeHost ? 1 : 0;
}

int gstamd_device_copy (void *dst_device, const void *src_device, size_t size, void *stream)
{
  hipError_t e = hipMemcpyAsync (dst_device, src_device, size, hipMemcpyDeviceToDevice, (hipStream_t) stream);
  return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "device copy");
}

int gstamd_device_upload_async (void *dst_device, const void *src_host, size_t size, void *stream)
{
  hipError_t e = hipMemcpyAsync (dst_device, src_host, size, hipMemcpyHostToDevice, (hipStream_t) stream);
  return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "upload");
}

int gstamd_device_download_async (void *dst_host, const void *src_device, size_t size, void *stream)
{
  hipError_t e = hipMemcpyAsync (dst_host, src_device, size, hipMemcpyDeviceToHost, (hipStream_t) stream);
  return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "download");
}

/* pitched copies (one plane of a frame whose host rows and device rows have different strides): hipMemcpy2DAsync */
int gstamd_device_upload_2d_async (void *dst_device, size_t dst_pitch, const void *src_host, size_t src_pitch, size_t row_bytes, size_t rows, void *stream)
{
  if (!row_bytes || !rows)
    return GSTAMD_OK;
  hipError_t e = hipMemcpy2DAsync (dst_device, dst_pitch, src_host, src_pitch, row_bytes, rows, hipMemcpyHostToDevice, (hipStream_t) stream);
  return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "2-D upload");
}

int gstamd_device_download_2d_async (void *dst_host, size_t dst_pitch, const void *src_device, size_t src_pitch, size_t row_bytes, size_t rows, void *stream)
{
  if (!row_bytes || !rows)
    return GSTAMD_OK;
  hipError_t e = hipMemcpy2DAsync (dst_host, dst_pitch, src_device, src_pitch, row_bytes, rows, hipMemcpyDeviceToHost, (hipStream_t) stream);
  return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "2-D download");
}

int gstamd_get_device (void)
{
  int d = -1;
  if (hipGetDevice (&d) != hipSuccess)
    return -1;
  return d;
}

int gstamd_video_converter_is_reentrant (GstAmdVideoConverter *c)
{
  /* every stream a frame is sent on has a scratch set of its own (bind_scratch): plans with intermediate images - gamma remap, plane
   * scalers, planar packers, two-pass and 16-bit scalers - overlap across streams like the single-kernel ones */
  if (c && c->plan.interlaced)
    return gstamd_video_converter_is_reentrant (c->field[0]) && gstamd_video_converter_is_reentrant (c->field[1]) ? 1 : 0;
  return c && ensure_tables (c) == GSTAMD_OK ? 1 : 0;
}

int gstamd_video_converter_get_config (const GstAmdVideoConverter *c, GstAmdVideoConverterConfig *config)
{
  if (!c || !config)
    return set_error (GSTAMD_ERR_INVALID, "NULL converter or config");
  *config = c->plan.config;
  return GSTAMD_OK;
}

int gstamd_video_converter_frame_finish (GstAmdVideoConverter *c, void *stream)
{
  if (!c)
    return set_error (GSTAMD_ERR_INVALID, "NULL converter");
  return gstamd_stream_synchronize (stream);
}

int gstamd_device_count (void)
{
  int n = 0;
  if (hipGetDeviceCount (&n) != hipSuccess)
    return 0;
  return n;
}

int gstamd_set_device (int device)
{
  hipError_t e = hipSetDevice (device);
  return e == hipSuccess ? GSTAMD_OK : hip_fail (e, "hipSetDevice");
}

}  // extern "C"
