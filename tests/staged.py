"""TEST INFRASTRUCTURE: the reference (oracle/_ref) run STAGE BY STAGE.

For some conversions the reference's own one-step output is undefined: it reads temporary lines it has not converted, converts a line that
the nearest vertical scaler hands out twice once per hand-out, or converts only MIN (in_width, out_width) pixels of a line.  The plan says so
(gstamd_video_converter_divergence) and computes what the chain's stages MEAN.  What they mean is pinned here by the reference itself: the same
chain (gst_video_converter_new, video-converter.c:2517-2535) split into the separate conversions it consists of - each of them well defined -
and run one after the other through gst_video_converter_frame:

    A  unpack + chroma upsampling      in format (source crop)        -> its unpack format, crop size
    S  the scaler passes, one conversion per pass, in chain_scale's order (video-converter.c:1685-1714), before C when the picture shrinks
    C  colour matrix / bit depth / alpha stage (chain_convert :1719, chain_alpha :1921)  unpack format of the source -> that of the destination
    S  ... after C when the picture grows (chain_scale force = TRUE, :2529)
    P  chroma downsampling + dither + pack + borders   -> out format, destination rectangle

Every step is forced onto the reference's generic chain (never one of its fused fastpaths, whose arithmetic differs): a converter with
dither-quantization != 1 skips video_converter_lookup_fastpath (:8921) and with dither-method = none adds no dither stage (:2044); the last
step, which must keep the conversion's own dither settings, starts from a format no fastpath begins with.

staged_expected () returns None for draws this split cannot express (gamma-mode = remap: the stages between the transfer tables work on
linear light, which no video format carries; GRAY sources / destinations: chain_convert keys on the GRAY flag of the REAL formats) - the
callers count those per class instead of comparing them.  The split itself is validated on every draw whose one-step reference IS defined:
tests/test_video_fuzz.py compares staged_expected () with the one-step reference there (test_staged_reference_equals_the_one_step_reference).
"""
import numpy as np

import cases

NO_FAST = dict(dither_method="none", dither_quantization=2)
SCALER_KEYS = ("resampler_method", "max_taps", "envelope", "sharpness", "sharpen")
# a 4-component, 4:4:4 format per (yuv?, bits) that NO fastpath starts from (video-converter.c:8413-8905 has AYUV -> ..., ARGB -> ARGB, AYUV64 -> AYUV64,
# ARGB64 -> ARGB64); its unpack format is the chain's line format, so unpacking it is a byte swizzle
SAFE_SOURCE = {(True, 8): "VUYA", (False, 8): "ABGR", (True, 16): "A444_16LE", (False, 16): "RGBA64_LE"}


def chain_passes(iw, ih, ow, oh):
    """chain_scale's decision (video-converter.c:1685-1714): (before or after the convert stage, [(direction, in size, out size) ...])"""
    first = ow * oh <= iw * ih
    order = ["h", "v"] if ow * ih <= iw * oh else ["v", "h"]
    passes = [d for d in order if (d == "h" and iw != ow) or (d == "v" and ih != oh)]
    return first, passes


def effective_alpha(ref, ifmt, ofmt, cfg):
    """convert_get_alpha_mode (video-converter.c:2264-2294) -> the config a conversion between two formats WITH alpha needs for the same stage"""
    mode, value = cfg.get("alpha_mode", "copy"), cfg.get("alpha_value", 1.0)
    pi, po = ref.format_props(ifmt), ref.format_props(ofmt)
    if not po["alpha"]:
        return {}
    if pi["alpha"]:
        if mode == "copy":
            return {}
        if mode == "mult":
            return {} if value == 1.0 else dict(alpha_mode="mult", alpha_value=value)
    if value == 1.0:
        return {}
    return dict(alpha_mode="set", alpha_value=value)


def _sizes(case):
    ifmt, w, h, ofmt, OW, OH, cfg, col, site = case
    sx, sy = cfg.get("src_x", 0), cfg.get("src_y", 0)
    dx, dy = cfg.get("dest_x", 0), cfg.get("dest_y", 0)
    return (cfg.get("src_width", w - sx), cfg.get("src_height", h - sy), cfg.get("dest_width", OW - dx), cfg.get("dest_height", OH - dy))


def front_case(ref, case):
    """unpack + chroma upsampling + the scaler passes of a conversion that shrinks the picture, as ONE conversion into a 4:4:4 format of the source's
    colour space and depth: the front half of the same chain, lines requested in the same order"""
    ifmt, w, h, ofmt, OW, OH, cfg, col, site = case
    pi = ref.format_props(ifmt)
    iw, ih, ow, oh = _sizes(case)
    keep = {k: cfg[k] for k in ("src_x", "src_y", "src_width", "src_height", "chroma_mode") + SCALER_KEYS if k in cfg}
    return (ifmt, w, h, SAFE_SOURCE[(pi["yuv"], pi["bits"])], ow, oh, dict(NO_FAST, **keep), col, site)


def _requests_in_frame_order(ref, case):
    """the windows of the vertical scaler the chain makes (the reference's own gst_video_scaler_new under the same config), then do_upsample_lines'
    grouping (video-converter.c:2991-3047: the first line asked for that is not cached starts a group of two; line 0 starts at -1)"""
    ifmt, w, h, ofmt, OW, OH, cfg, col, site = case
    iw, ih, ow, oh = _sizes(case)
    offs, taps = ref.scaler_windows(cases.ref_config_string(ref, {k: cfg[k] for k in SCALER_KEYS if k in cfg}), ih, oh)
    cached_to = -1          # the last line the upsampler has produced
    for j in range(oh):
        for line in range(offs[j], min(offs[j] + taps, ih)):
            if line <= cached_to:
                continue
            if line != 0 and line % 2 == 0:
                return False
            cached_to = line + 1 if line else 0
    return True


def stageable(ref, case, diverges=None, canonical_pairs=False):
    """-> (True, "" | "front") or (False, why).  diverges (case) -> bool: does the plan of that (sub-)conversion announce a divergence?"""
    ifmt, w, h, ofmt, OW, OH, cfg, col, site = case
    if cfg.get("gamma_mode") == "remap":
        return False, "gamma-remap"
    pi, po = ref.format_props(ifmt), ref.format_props(ofmt)
    if pi["gray"] or po["gray"]:
        return False, "gray"          # chain_convert decides on the GRAY flag of the real formats (tried: a split through AYUV differs on draws whose one-step reference is defined)
    if "VYUY" in (ifmt, ofmt):
        return False, "VYUY"          # unpack_VYUY / pack_VYUY's fallback loops depend on the alignment of the line they work on (video-format.c:337-373)
    # a vertically subsampled source under a vertical scaler: the reference's chroma upsampler pairs the lines in the ORDER the scaler asks for them
    # (do_upsample_lines :2991 makes the requested line the first of a group) - defined, reproduced by the product's plans, and not what a separate
    # in -> unpack-format conversion (lines in order) computes.  Where the scalers come before the convert stage the chain's whole front half is taken
    # as one conversion instead (front_case) - provided that one is itself defined
    iw, ih, ow, oh = _sizes(case)
    if pi["h_sub"] and ih != oh and cfg.get("chroma_mode", "full") in ("full", "upsample-only"):
        # the class whose one-step reference has no defined result to follow (its unpack ring is one line short): the plan still pairs the lines as the
        # scaler asks for them; that IS the frame order (pairs 2k-1, 2k) as long as every line the scaler asks for and the upsampler has not made yet
        # is an odd one - true whenever consecutive filter windows touch; a window that starts after a gap on an even line opens a pair (2k, 2k+1)
        if canonical_pairs and _requests_in_frame_order(ref, case):
            return True, ""
        first, _passes = chain_passes(iw, ih, ow, oh)
        # (not into a vertically subsampled destination: its chroma downsampler asks the scaler for lines in pairs, another request order)
        if first and not po["h_sub"] and diverges is not None and not diverges(front_case(ref, case)):
            return True, "front"
        return False, "4:2:0 source under a vertical scaler"
    return True, ""


def staged_steps(ref, case, front=False):
    """-> [(in fmt, w, h, in colorimetry, in chroma-site, out fmt, w, h, out colorimetry, out chroma-site, config) ...]"""
    return _steps(ref, case, front)[0]


def _steps(ref, case, front=False):
    """-> (steps, index of step C)"""
    ifmt, w, h, ofmt, OW, OH, cfg, col, site = case
    pi, po = ref.format_props(ifmt), ref.format_props(ofmt)
    in_col = col or ref.video_info(ifmt, w, h)["colorimetry"]
    if ifmt.startswith("RGBA_F"):
        # gst_video_info_from_caps forces the 0_1 range on float formats (video-info.c:580-586): the stages behind the unpack step must see the range the
        # one-step conversion works with (0_1 = full-range code values; on an integer format the same rule turns it into 0_255)
        from gstreamer_amd import video as V
        if in_col in V.COLORIMETRY:
            rng, mtx, trc, prim = V.COLORIMETRY[in_col]
            parts = [V.COLOR_RANGE[rng], V.COLOR_MATRIX[mtx], V.TRANSFER[trc], V.PRIMARIES[prim]]
        else:
            parts = [int(v) for v in in_col.split(":")]
        if parts[0] != 0:
            parts[0] = 3
        in_col = ":".join(str(v) for v in parts)
    out_col = ref.video_info(ofmt, OW, OH)["colorimetry"]
    sx, sy = cfg.get("src_x", 0), cfg.get("src_y", 0)
    iw, ih = cfg.get("src_width", w - sx), cfg.get("src_height", h - sy)
    dx, dy = cfg.get("dest_x", 0), cfg.get("dest_y", 0)
    ow, oh = cfg.get("dest_width", OW - dx), cfg.get("dest_height", OH - dy)
    scaler = {k: cfg[k] for k in SCALER_KEYS if k in cfg}
    # video_converter_compute_resample (:2850-2895): no chroma resampler at all unless subsampling, siting or the FRAME sizes differ
    in_site = site or ref.video_info(ifmt, w, h)["chroma_site"]
    out_site = ref.video_info(ofmt, OW, OH)["chroma_site"]
    resample = (pi["w_sub"], pi["h_sub"]) != (po["w_sub"], po["h_sub"]) or in_site != out_site or (w, h) != (OW, OH)
    chroma = {"chroma_mode": cfg["chroma_mode"]} if resample and "chroma_mode" in cfg else ({} if resample else {"chroma_mode": "none"})
    # every intermediate frame in a format that is neither a fastpath's source nor its own unpack format: a destination in its unpack format lends its
    # rows to the chain's last stages (identity_pack :2103, get_dest_line), which is where several of the announced classes live
    uin, uout = SAFE_SOURCE[(pi["yuv"], pi["bits"])], SAFE_SOURCE[(po["yuv"], po["bits"])]
    steps = []
    # A: unpack + chroma upsampling
    a_cfg = dict(NO_FAST, **{k: cfg[k] for k in ("src_x", "src_y", "src_width", "src_height") if k in cfg}, **chroma)
    first, passes = chain_passes(iw, ih, ow, oh)
    cw, ch = iw, ih
    if front:          # A and the scaler passes in one conversion (front_case)
        f = front_case(ref, case)
        steps.append((ifmt, w, h, in_col, site, uin, ow, oh, in_col, None, f[6]))
        cw, ch, passes = ow, oh, []
    else:
        steps.append((ifmt, w, h, in_col, site, uin, iw, ih, in_col, None, a_cfg))

    def scale(fmt, colr):
        nonlocal cw, ch
        for d in passes:
            nw, nh = (ow, ch) if d == "h" else (cw, oh)
            steps.append((fmt, cw, ch, colr, None, fmt, nw, nh, colr, None, dict(NO_FAST, **scaler)))
            cw, ch = nw, nh
    if first:
        scale(uin, in_col)
    c_cfg = dict(NO_FAST, **{k: cfg[k] for k in ("matrix_mode", "primaries_mode") if k in cfg})
    c_cfg.update(effective_alpha(ref, ifmt, ofmt, cfg))
    c_index = len(steps)
    steps.append((uin, cw, ch, in_col, None, uout, cw, ch, out_col, None, c_cfg))
    if not first:
        scale(uout, out_col)
    # P: from a format no fastpath starts with, under the conversion's own dither / chroma / border settings
    safe = uout
    p_cfg = dict({k: cfg[k] for k in ("dest_x", "dest_y", "dest_width", "dest_height", "border_argb", "fill_border", "dither_method",
                                       "dither_quantization") if k in cfg}, **chroma)
    steps.append((safe, cw, ch, out_col, None, ofmt, OW, OH, out_col, None, p_cfg))
    return steps, c_index


def run_steps(ref, steps, src, last_cfg=None):
    cur = src
    for k, (sf, sw, sh, scol, ssite, df, dw, dh, dcol, dsite, scfg) in enumerate(steps):
        if last_cfg is not None and k == len(steps) - 1:
            scfg = last_cfg
        if (sf, sw, sh, scol) == (df, dw, dh, dcol) and scfg == NO_FAST:
            continue          # nothing to do (a source already in its unpack format, no crop)
        cur = ref.VideoConverter(sf, sw, sh, df, dw, dh, in_colorimetry=scol, in_chroma_site=ssite, out_colorimetry=dcol, out_chroma_site=dsite,
                                 config=cases.ref_config_string(ref, scfg)).frame(cur)
    return cur


def staged_expected(ref, case, src, diverges=None, canonical_pairs=False):
    """-> (expected frame, compare mask or None), or None when the draw cannot be split (stageable ())"""
    ok, how = stageable(ref, case, diverges, canonical_pairs)
    if not ok:
        return None
    ifmt, w, h, ofmt, OW, OH, cfg, col, site = case
    steps = staged_steps(ref, case, front=(how == "front"))
    want = run_steps(ref, steps, src)
    mask = None
    # (matrix-mode = input-only converts the BORDER colour with the source's matrix - setup_borderline through compute_matrix_to_YUV (force) - which
    # the last step, whose source is already in the destination's colour space, cannot know: compared inside the rectangle as well)
    if "dest_x" in cfg and (cfg.get("fill_border", 1) == 0 or cfg.get("matrix_mode") == "input-only"):
        # without a border the reference's generic chain packs whatever its line buffers held beside the rectangle: compare what the picture alone
        # decides = the bytes three border colours agree on (scripts/fuzz_video.py matches_reference does the same for the one-step reference)
        p_cfg = steps[-1][-1]
        ab = [run_steps(ref, steps, src, dict(p_cfg, fill_border=1, border_argb=b)) for b in (0x00000000, 0xffffffff, 0x80aa5533)]
        mask = (ab[0] == ab[1]) & (ab[1] == ab[2])
    return want, mask


def picture_bytes(ref, fmt, w, h, size):
    """bool mask of the frame's bytes that belong to the picture (stride padding and the unused second luma slot of an odd-width packed 4:2:2 line
    excepted): an index image sent through cases.visible_bytes"""
    oi = ref.video_info(fmt, w, h)
    pic = np.zeros(size, bool)
    try:
        idx = cases.visible_bytes(fmt, w, h, list(oi["stride"]), list(oi["offset"]), np.arange(1, size + 1, dtype=np.int64))
        pic[idx[idx > 0] - 1] = True
    except Exception:
        pic[:] = True
    return pic


def compare(ref, case, got, want, mask):
    """bytes of the picture (stride padding excepted) -> (equal, text)"""
    ifmt, w, h, ofmt, OW, OH, cfg, col, site = case
    if mask is not None:
        mask = mask & picture_bytes(ref, ofmt, OW, OH, got.size)
        bad = int((got[mask] != want[mask]).sum())
        return bad == 0, "" if bad == 0 else ": %d of %d bytes inside the rectangle differ from the staged reference" % (bad, int(mask.sum()))
    if (got == want).all():
        return True, ""
    oi = ref.video_info(ofmt, OW, OH)
    vb = lambda b: cases.visible_bytes(ofmt, OW, OH, list(oi["stride"]), list(oi["offset"]), b)
    try:
        a, b = vb(got), vb(want)
        if (a == b).all():
            return True, ""
        return False, ": %d of %d picture bytes differ from the staged reference" % (int((a != b).sum()), a.size)
    except Exception:
        return False, ": %d of %d bytes differ from the staged reference" % (int((got != want).sum()), got.size)


# ---- second opinion for draws the split cannot express: the ONE-STEP reference on the bytes its undefined part cannot reach -------------------
# Two of the announced classes leave most of the reference's frame well defined: "MIN (in_width, out_width) pixels converted" leaves the columns
# right of that width unconverted, "a repeated line is processed once per repetition" leaves every repetition after the first processed twice.
# Which BYTES of the destination that reaches is asked of the reference itself: the tail of the chain (the steps after the point where the
# undefined pixels appear) is run on images that differ only in those pixels; a byte that comes out the same for all of them does not depend on
# them, and there the product must equal the one-step reference.  (Chroma downsampling and error diffusion spread the difference - the probe shows
# it; formats, rectangles and borders need no model.)

def _tail_probe(ref, case, dirty_after_c, dirty):
    """dirty (h, w) -> bool array of the image after step C (dirty_after_c) or after the scaler passes: pixels that are undefined.  -> bool mask of
    destination bytes that do not depend on them, or None"""
    ifmt, w, h, ofmt, OW, OH, cfg, col, site = case
    steps, k = _steps(ref, case)
    start = k + 1 if dirty_after_c else len(steps) - 1
    tail = steps[start:]
    sf, sw, sh = tail[0][0], tail[0][1], tail[0][2]
    d = dirty(sh, sw)
    if d is None or d.all():
        return None
    size = ref.video_info(sf, sw, sh)["size"]
    po = ref.format_props(sf)
    planar16 = sf == "A444_16LE"
    outs = []
    for v in range(7):          # the undefined pixels as 0x00, 0xff, alternating columns of both, and four random fills (small filter taps show on the extremes)
        img = cases.frame_bytes(size, "random", 4242, sw).copy()
        alt = cases.frame_bytes(size, "random", 777 + v, sw).copy()
        if v < 2:
            alt[:] = 0xff * v
        elif v == 2:
            alt[:] = np.where((np.arange(size) // (8 if po["bits"] == 16 else 4)) % 2, 0xff, 0x00)
        if planar16:          # four planes of (sh, stride) 16-bit samples
            st = ref.video_info(sf, sw, sh)["stride"][0]
            a, b = img.reshape(4, sh, st), alt.reshape(4, sh, st)
            for p in range(4):
                m = np.repeat(d, 2, axis=1)
                a[p, :, :sw * 2][m] = b[p, :, :sw * 2][m]
        else:
            bpp = 8 if po["bits"] == 16 else 4
            a, b = img.reshape(sh, sw * bpp), alt.reshape(sh, sw * bpp)
            m = np.repeat(d, bpp, axis=1)
            a[m] = b[m]
        outs.append(run_steps(ref, tail, img))
    same = np.ones(outs[0].size, bool)
    for o in outs[1:]:
        same &= (o == outs[0])
    # a sample is more than a byte (16-bit components, 10-bit components in 32-bit words): when one byte of an aligned 4-byte word depends on the undefined
    # pixels, the word does (the high byte of a sample rarely moves in seven probes)
    n4 = same.size // 4 * 4
    same[:n4] = np.repeat(same[:n4].reshape(-1, 4).all(axis=1), 4)
    return same


def masked_check(ref, case, src, got, divergence):
    """-> (checked?, equal?, text): compares the product's frame with the one-step reference where that is defined; checked = False when this
    divergence class (or this draw) has no such region"""
    ifmt, w, h, ofmt, OW, OH, cfg, col, site = case
    if cfg.get("fill_border", 1) == 0 and "dest_x" in cfg:
        return False, True, ""
    if cfg.get("dither_method") in ("verterr", "floyd-steinberg", "sierra-lite"):
        return False, True, ""          # error diffusion carries the undefined pixels' share to every pixel after them; seven probes do not show all of it
    iw, ih, ow, oh = _sizes(case)
    first, passes = chain_passes(iw, ih, ow, oh)
    if ref.format_props(ifmt)["h_sub"] and ih != oh and cfg.get("chroma_mode", "full") in ("full", "upsample-only"):
        # stageable ()'s "4:2:0 source under a vertical scaler": the reference's unpack ring is a line short there and whole source lines come out aliased
        # (round 6, device seed 81273: I420_10LE 4 x 4 -> GRAY10_LE16 4 x 32 nearest - rows 16 .. 23 of the reference repeat source line 1), first
        # hand-outs included; no region of such a frame is the reference's word
        return False, True, ""
    notes = [d for d in divergence.split(". ") if d.strip()]
    only = lambda text: all(n.strip().startswith(text) for n in notes)
    po = ref.format_props(ofmt)
    mask = None
    try:
        if only("the reference converts only MIN"):
            # the unconverted columns at the convert stage: x >= MIN (in_width, out_width).  With the horizontal pass behind that stage they spread
            # through its filter windows - taken from the reference's own scaler (small taps barely show in a probe) -, the rest of the chain is probed
            lim = min(iw, ow)
            dirty_cols = np.arange(ow) >= lim
            if not first and iw != ow:
                offs, taps = ref.scaler_windows(cases.ref_config_string(ref, {k: cfg[k] for k in SCALER_KEYS if k in cfg}), iw, ow)
                dirty_cols = np.array([offs[x] + taps > lim for x in range(ow)])
            mask = _tail_probe(ref, case, False, lambda hh, ww: (dirty_cols[None, :] & np.ones((hh, 1), bool)) if ww == ow else None)
        elif only("nearest vertical enlargement ahead of a stage") and not po["h_sub"] and cfg.get("resampler_method") == "nearest":
            # rows of the destination rectangle that are NOT the first hand-out of their source line - asked of the reference's own vertical pass: a frame
            # whose row r is filled with the byte r, through that step alone
            steps, _k = _steps(ref, case)
            vstep = [st for st in steps[1:-1] if st[0] == st[5] and st[2] != st[7]][0]
            fsize, fh = ref.video_info(vstep[0], vstep[1], vstep[2])["size"], vstep[2]
            rows = fsize // fh if vstep[0] != "A444_16LE" else fsize // (4 * fh)
            ramp = np.repeat(np.arange(fh, dtype=np.uint8), rows)
            ramp = np.tile(ramp, 4) if vstep[0] == "A444_16LE" else ramp
            got_rows = run_steps(ref, [vstep], ramp)
            orows = got_rows.size // oh if vstep[0] != "A444_16LE" else got_rows.size // (4 * oh)
            srow = got_rows[:orows * oh].reshape(oh, orows)[:, 0].astype(int)
            rep = np.array([j > 0 and srow[j] == srow[j - 1] for j in range(oh)])
            after_c = first          # scalers before the convert stage: the repeated line is converted twice; after it: the late stages see it twice
            mask = _tail_probe(ref, case, after_c, lambda hh, ww: (rep[:hh, None] & np.ones((1, ww), bool)) if hh == oh else None)
    except ValueError:
        mask = None
    if mask is None or not mask.any():
        return False, True, ""
    one = ref.VideoConverter(ifmt, w, h, ofmt, OW, OH, in_colorimetry=col, in_chroma_site=site, config=cases.ref_config_string(ref, cfg)).frame(src)
    mask &= picture_bytes(ref, ofmt, OW, OH, got.size)
    bad = int((got[mask] != one[mask]).sum())
    return True, bad == 0, "" if bad == 0 else ": %d of %d bytes the reference's undefined part cannot reach differ from the one-step reference" % (bad, int(mask.sum()))


# ---- interlaced frames --------------------------------------------------------------------------------------------------------------------------------
# The reference's generic chain is not a usable reference for a VERTICAL scaler pass on interlaced frames (planner.cpp, plan_video_converter: its line
# cache keeps 2 * taps lines of backlog over shorter temporary-line rings - the windows read other lines' pixels).  Its plane scaler is: the same
# gst_video_scaler_new (..., GST_VIDEO_SCALER_FLAG_INTERLACED, ...) object, the same 8-bit taps, applied by gst_video_scaler_2d straight from the frame
# (convert_scale_planes on a 4-byte format scales whole pixels as 4 x u8: setup_scale :8016-8087).  So the split above with every stage run on
# interlace-mode=interleaved infos and every scaler pass run as <4-byte unpack format> -> <same format> through that fastpath.
PLANE_SCALED = {"VUYA": "AYUV", "ABGR": "ABGR"}          # 4-byte formats convert_scale_planes serves onto themselves (video-converter.c:8547-8904); VUYA is none


def staged_expected_interlaced(ref, case, src):
    """-> expected frame of an 8-bit interlaced conversion, or None (formats / options the split cannot express)"""
    ifmt, w, h, ofmt, OW, OH, cfg, col, site = case
    pi, po = ref.format_props(ifmt), ref.format_props(ofmt)
    if pi["bits"] != 8 or po["bits"] != 8 or pi["gray"] or po["gray"] or cfg.get("gamma_mode") == "remap" or any(k in cfg for k in ("src_x", "dest_x")):
        return None
    if "VYUY" in (ifmt, ofmt):
        return None
    if (w, h) == (OW, OH) and pi["h_sub"] and ofmt in ("ARGB", "AYUV"):
        return None          # the chain works in the destination's own rows there: the edge groups of video_chroma_up_vi2 stay unfiltered (planner.cpp simulate_vpairs_field), which no split shows
    steps, _c = _steps(ref, case)
    cur = src
    for (sf, sw, sh, scol, ssite, df, dw, dh, dcol, dsite, scfg) in steps:
        if (sf, sw, sh, scol) == (df, dw, dh, dcol) and scfg == NO_FAST:
            continue
        if sf == df and (sw, sh) != (dw, dh):
            # a scaler pass: through the plane scaler, in a format it serves (a byte swizzle either side where needed; those are per-pixel and exact)
            pf = PLANE_SCALED[sf]
            scaler = {k: v for k, v in scfg.items() if k not in NO_FAST}
            if "resampler_taps" in cfg:
                scaler["resampler_taps"] = cfg["resampler_taps"]
            if pf != sf:
                cur = ref.VideoConverter(sf, sw, sh, pf, sw, sh, in_colorimetry=scol, out_colorimetry=scol, config=cases.ref_config_string(ref, NO_FAST), interlaced=True).frame(cur)
            cur = ref.VideoConverter(pf, sw, sh, pf, dw, dh, in_colorimetry=scol, out_colorimetry=scol, config=cases.ref_config_string(ref, dict(scaler, threads=1)),
                                     interlaced=True).frame(cur)
            if pf != sf:
                cur = ref.VideoConverter(pf, dw, dh, sf, dw, dh, in_colorimetry=scol, out_colorimetry=scol, config=cases.ref_config_string(ref, NO_FAST), interlaced=True).frame(cur)
            continue
        cur = ref.VideoConverter(sf, sw, sh, df, dw, dh, in_colorimetry=scol, in_chroma_site=ssite, out_colorimetry=dcol, out_chroma_site=dsite,
                                 config=cases.ref_config_string(ref, scfg), interlaced=True).frame(cur)
    return cur
