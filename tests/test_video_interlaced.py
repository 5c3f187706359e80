"""Interlaced frames (GstVideoInfo interlace-mode=interleaved: every frame carries GST_VIDEO_FRAME_FLAG_INTERLACED) against the reference run on
interlaced infos (oracle/_ref, ref.VideoConverter (..., interlaced=True)): video_converter_generic :3303-3312 (upsample_i / v_scaler_i),
GET_LINE_OFFSETS :3383, chain_vscale :1651-1660, setup_scale :7977, GET_UV_420 (video-format.c:1045), video_chroma_up_vi2 (video-chroma.c:347),
GST_VIDEO_SCALER_FLAG_INTERLACED (video-scaler.c:229-249).

What a conversion is checked against:
  * plans without a divergence note: the reference's one-step output, byte for byte;
  * plans whose note says the reference's interlaced chain scaler reads aliased lines: the reference run STAGE BY STAGE with every vertical pass
    through its plane scaler (tests/staged.py staged_expected_interlaced) - the split is itself pinned here against the one-step reference on
    the conversions where that IS defined (test_interlaced_staged_reference_equals_the_one_step_reference).
Host: the kernel bodies on the emulator (tests/emu).  Device (-m gpu): the same checker through the C ABI."""
import ctypes as C
import os
import random
import sys

import numpy as np
import pytest

import cases
import staged
from gstreamer_amd import video as V

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FORMATS_420 = ["I420", "YV12", "NV12", "NV21", "A420"]
FORMATS = FORMATS_420 + ["YUY2", "UYVY", "YVYU", "AYUV", "VUYA", "Y42B", "Y444", "NV16", "NV24", "Y41B", "BGRA", "RGBx", "ARGB", "xBGR", "RGB", "BGR", "v308", "GRAY8", "GBR"]


def _emu_run(emu, case, src):
    ifmt, w, h, ofmt, ow, oh, cfg, col, site = case
    emu.emu_video_convert.argtypes = [C.POINTER(V.VideoInfo), C.POINTER(V.VideoInfo), C.POINTER(V.ConverterConfig), C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_int]
    emu.emu_video_last_divergence.restype = C.c_char_p
    ii, oi = V.video_info(ifmt, w, h, colorimetry=col, chroma_site=site), V.video_info(ofmt, ow, oh)
    ii.interlace_mode = oi.interlace_mode = 1
    dst = np.zeros(int(oi.size), np.uint8)
    desc = C.create_string_buffer(1024)
    r = emu.emu_video_convert(C.byref(ii), C.byref(oi), C.byref(V.converter_config(**cfg)), src.ctypes.data, dst.ctypes.data, 1, desc, 1024)
    if r != 0:
        assert r == V.ERR_UNSUPPORTED, (case, desc.value)
        return None, desc.value.decode(), ""
    return dst, desc.value.decode(), emu.emu_video_last_divergence().decode()


def _dev_run(gpu, case, src):
    import torch
    ifmt, w, h, ofmt, ow, oh, cfg, col, site = case
    ii, oi = V.video_info(ifmt, w, h, colorimetry=col, chroma_site=site), V.video_info(ofmt, ow, oh)
    ii.interlace_mode = oi.interlace_mode = 1
    try:
        conv = V.VideoConverter(ii, oi, V.converter_config(**cfg))
    except V.GstAmdError as e:
        assert e.code == V.ERR_UNSUPPORTED, (case, str(e))
        return None, str(e), ""
    d_src = torch.from_numpy(src).to(gpu)
    d_dst = torch.zeros(int(oi.size), dtype=torch.uint8, device=gpu)
    conv.frame(d_src, d_dst)
    torch.cuda.synchronize()
    out = d_dst.cpu().numpy()
    desc, div = conv.describe(), conv.divergence()
    conv.free()
    return out, desc, div


def check(run, ref, case, seed=1):
    """-> "refused" | "ok" (one-step reference) | "staged" | "unchecked"; raises on a mismatch"""
    ifmt, w, h, ofmt, ow, oh, cfg, col, site = case
    src = cases.frame_bytes(int(V.video_info(ifmt, w, h).size), "random", seed, w)
    got, desc, div = run(case, src)
    if got is None:
        return "refused"
    assert desc.startswith("interlaced{"), desc
    if div and "scale_planes" in desc:
        return "unchecked"          # (an announced class of the plane scaler: the split models the chain)
    if div:
        want = staged.staged_expected_interlaced(ref, case, src)
        if want is None:
            return "unchecked"
        how = "staged"
    else:
        want = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, in_colorimetry=col, in_chroma_site=site, config=cases.ref_config_string(ref, dict(cfg, threads=1)),
                                  interlaced=True).frame(src)
        how = "ok"
    ri = ref.video_info(ofmt, ow, oh)
    a = cases.visible_bytes(ofmt, ow, oh, list(ri["stride"]), list(ri["offset"]), got)
    b = cases.visible_bytes(ofmt, ow, oh, list(ri["stride"]), list(ri["offset"]), want)
    bad = int((a != b).sum())
    assert bad == 0, "%s: %d of %d bytes differ from the %s reference | %s | %s" % (case, bad, a.size, how, desc, div[:80])
    return how


def random_case(rnd):
    """format pair, sizes (4:2:0 frames: heights on multiples of four - the reference addresses rows past the chroma planes otherwise), scaler and chroma options"""
    ifmt, ofmt = rnd.choice(FORMATS), rnd.choice(FORMATS)
    kind = rnd.choice(["same", "same", "h", "v", "hv", "hv"])

    def height(f):
        return 4 * rnd.randint(1, 14) if f in FORMATS_420 else rnd.randint(2, 56)
    w = rnd.choice([rnd.randint(2, 70), 2 * rnd.randint(1, 35), 16 * rnd.randint(1, 5)])
    h = height(ifmt)
    ow, oh = w, h
    if kind in ("h", "hv"):
        ow = rnd.choice([rnd.randint(2, 70), 2 * rnd.randint(1, 35)])
    if kind in ("v", "hv"):
        oh = height(ofmt)
    if ofmt in FORMATS_420 and oh % 4:
        oh = max(4, oh & ~3)
    if oh == h and ifmt in FORMATS_420 and ofmt not in FORMATS_420:
        pass
    cfg = {}
    if (w, h) != (ow, oh):
        m = rnd.choice(["linear", "cubic", "lanczos", "sinc", "linear", None])
        if m:
            cfg["resampler_method"] = m
        if rnd.random() < 0.2:
            cfg["resampler_taps"] = rnd.choice([2, 3, 4, 6])
    if rnd.random() < 0.25:
        cfg["chroma_mode"] = rnd.choice(["full", "upsample-only", "downsample-only", "none"])
    if rnd.random() < 0.15:
        cfg["alpha_mode"], cfg["alpha_value"] = rnd.choice([("set", 0.5), ("mult", 0.75), ("copy", 1.0)])
    if rnd.random() < 0.15:
        cfg["matrix_mode"] = rnd.choice(["full", "none", "input-only", "output-only"])
    site = rnd.choice([None, None, "mpeg2", "jpeg", "cosited", "v-cosited"]) if ifmt in FORMATS_420 + ["YUY2", "UYVY", "Y42B", "NV16"] else None
    col = rnd.choice([None, None, "bt709", "bt601"]) if ifmt not in ("BGRA", "RGBx", "ARGB", "xBGR", "RGB", "BGR", "GBR") else None
    return (ifmt, w, h, ofmt, ow, oh, cfg, col, site)


UNSCALED = [(a, b) for a in ["I420", "NV12", "YV12", "A420", "YUY2", "AYUV", "Y42B", "Y444", "BGRA", "RGB", "GRAY8", "NV16"]
            for b in ["I420", "NV21", "UYVY", "AYUV", "ARGB", "Y444", "BGRA", "xBGR", "BGR", "NV24", "Y41B", "GBR"]]


@pytest.mark.parametrize("pair", UNSCALED, ids=lambda p: "%s-%s" % p)
def test_interlaced_unscaled_conversions_match_the_reference_on_host(native_lib, emu_lib, ref, pair):
    """same size: the reference's own one-step output is defined for every pair - the field-aware 4:2:0 rows (GET_UV_420), video_chroma_up_vi2's
    weights and edge groups (a destination in its unpack format keeps them unfiltered), the keeps_interlaced fastpaths (GET_LINE_OFFSETS), the
    interlaced plane scaler (no copy even at equal sizes: its fields are shifted by half a line)"""
    a, b = pair
    for (w, h) in ((32, 16), (38, 24), (17, 8)):
        r = check(lambda c, s: _emu_run(emu_lib, c, s), ref, (a, w, h, b, w, h, {}, None, None), 40 + w)
        assert r in ("ok", "refused"), r



@pytest.mark.parametrize("case", [
    ("I420", 64, 32, "I420", 48, 20, {"resampler_method": "linear"}), ("NV12", 64, 32, "NV12", 80, 48, {}), ("I420", 32, 16, "YV12", 32, 24, {"resampler_method": "lanczos"}),
    ("NV12", 32, 16, "NV16", 40, 30, {"resampler_method": "linear"}), ("NV24", 30, 18, "NV12", 30, 12, {}), ("BGRA", 33, 17, "BGRA", 20, 31, {"resampler_method": "cubic"}),
    ("YUY2", 34, 20, "YUY2", 50, 14, {"resampler_method": "linear"}), ("RGB", 31, 15, "RGB", 31, 22, {}), ("Y444", 20, 11, "Y444", 9, 7, {"resampler_method": "sinc"}),
    ("GRAY8", 40, 20, "GRAY8", 40, 30, {"resampler_method": "nearest"}), ("A420", 32, 16, "A420", 16, 8, {"resampler_method": "linear"}),
    ("AYUV", 32, 32, "BGRA", 20, 32, {"resampler_method": "linear"}), ("I420", 32, 32, "BGRA", 48, 32, {}), ("UYVY", 32, 16, "Y444", 10, 16, {"resampler_method": "lanczos"}),
], ids=lambda c: "%s_%dx%d_%s_%dx%d" % c[:6])
def test_interlaced_plane_scaler_and_horizontal_chain_match_the_reference_on_host(native_lib, emu_lib, ref, case):
    """convert_scale_planes with GST_VIDEO_SCALER_FLAG_INTERLACED scalers (setup_scale :8075, 8236: the field's own resampler over the field's lines, the 2-D
    scaler's pass order from the zipped offsets) and the generic chain with a horizontal pass only: one-step reference, byte for byte"""
    r = check(lambda c, s: _emu_run(emu_lib, c, s), ref, case + (None, None), 77)
    assert r == "ok", r


@pytest.mark.parametrize("case", [
    ("Y444", 22, 12, "BGRA", 22, 20, {"resampler_method": "linear"}), ("Y444", 32, 32, "AYUV", 32, 24, {"resampler_method": "linear"}), ("I420", 32, 16, "BGRA", 24, 8, {}),
    ("I420", 32, 32, "BGRA", 48, 40, {"resampler_method": "linear"}), ("NV12", 64, 32, "RGB", 32, 16, {"resampler_method": "lanczos"}),
    ("BGRA", 32, 32, "I420", 24, 16, {"resampler_method": "linear"}), ("YUY2", 32, 32, "NV12", 32, 16, {}), ("I420", 32, 16, "NV12", 32, 8, {"resampler_method": "linear"}),
    ("NV12", 64, 64, "BGRA", 64, 8, {"resampler_method": "linear"}), ("I420", 48, 40, "Y42B", 20, 60, {"resampler_method": "cubic"}),
], ids=lambda c: "%s_%dx%d_%s_%dx%d" % c[:6])
def test_interlaced_chain_with_a_vertical_scaler_matches_the_reference_stage_by_stage_on_host(native_lib, emu_lib, ref, case):
    """the generic chain's vertical scaler on interlaced frames: the reference's one-step output is line-aliased (the plan says so); the plan's result is the
    chain run stage by stage with the vertical pass through the reference's plane scaler"""
    r = check(lambda c, s: _emu_run(emu_lib, c, s), ref, case + (None, None), 91)
    assert r == "staged", r


@pytest.mark.parametrize("pair", [("UYVY", "v210"), ("v210", "UYVY"), ("I420", "v210"), ("v210", "I420"), ("YUY2", "v210"), ("v210", "Y42B"), ("I422_10LE", "v210"),
                                  ("v210", "I422_10LE"), ("v210", "I420_10LE"), ("I420_10LE", "v210"), ("P010_10LE", "BGRA"), ("I422_10LE", "RGBx"), ("v210", "BGRA"),
                                  ("Y210", "ARGB"), ("I420_10LE", "AYUV")], ids=lambda p: "%s-%s" % p)
def test_interlaced_v210_fastpaths_and_10_bit_sources_match_the_reference_on_host(native_lib, emu_lib, ref, pair):
    """broadcast capture: the reference's own v210 fastpaths keep interlaced frames (GET_LINE_OFFSETS pairs lines l and l + 2 over one chroma row of their
    field) and have no dither stage; 10-bit sources into 4-byte 8-bit formats run the 16-bit chain with video_chroma_up_vi2 on 16-bit lines"""
    a, b = pair
    for (w, h) in ((48, 16), (50, 24), (12, 8)):
        assert check(lambda c, s: _emu_run(emu_lib, c, s), ref, (a, w, h, b, w, h, {}, None, None), 7 + w) == "ok"


def test_interlaced_reference_chain_scaler_is_line_aliased(ref):
    """the evidence behind the note: the reference's chain and its own plane scaler (same gst_video_scaler_new object, same taps) disagree on LUMA for an
    interlaced enlargement - the chain's top-field rows come from source lines further down"""
    w, h, oh = 22, 12, 20
    cfg = cases.ref_config_string(ref, dict(resampler_method="linear", threads=1))
    src = cases.frame_bytes(ref.video_info("Y444", w, h)["size"], "random", 5, w)
    chain = ref.VideoConverter("Y444", w, h, "AYUV", w, oh, config=cfg, interlaced=True).frame(src).reshape(oh, w, 4)[:, :, 1]
    st = ref.video_info("Y444", w, oh)["stride"][0]
    plane = ref.VideoConverter("Y444", w, h, "Y444", w, oh, config=cfg, interlaced=True).frame(src)[:st * oh].reshape(oh, st)[:, :w]
    rows = [int(r) for r in np.nonzero((chain != plane).any(axis=1))[0]]
    assert rows and all(r % 2 == 0 for r in rows), rows          # top-field rows only
    sst = ref.video_info("Y444", w, h)["stride"][0]
    ys = src[:sst * h].reshape(h, sst)[:, :w]
    assert (plane[2] == ys[0]).all() and (chain[2] == ys[4]).all()          # output row 2: the plane scaler copies line 0 (its taps say so), the chain delivers line 4


@pytest.mark.parametrize("seed", [11, 22, 33])
def test_interlaced_staged_reference_equals_the_one_step_reference(native_lib, emu_lib, ref, seed):
    """the checker checked: wherever the interlaced one-step reference IS defined and runs its generic chain (no vertical scaler), the split gives the same frame"""
    rnd = random.Random(seed)
    same = 0
    for it in range(140):
        case = random_case(rnd)
        ifmt, w, h, ofmt, ow, oh, cfg, col, site = case
        if h != oh:
            continue
        src = cases.frame_bytes(int(V.video_info(ifmt, w, h).size), "random", seed * 100 + it, w)
        got, desc, div = _emu_run(emu_lib, case, src)
        if got is None or div or any(k in desc for k in ("scale_planes", "{as convert_", "v210_fast")):
            continue
        want = staged.staged_expected_interlaced(ref, case, src)
        if want is None:
            continue
        one = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, in_colorimetry=col, in_chroma_site=site, config=cases.ref_config_string(ref, dict(cfg, threads=1)),
                                 interlaced=True).frame(src)
        ri = ref.video_info(ofmt, ow, oh)
        a = cases.visible_bytes(ofmt, ow, oh, list(ri["stride"]), list(ri["offset"]), want)
        b = cases.visible_bytes(ofmt, ow, oh, list(ri["stride"]), list(ri["offset"]), one)
        assert (a == b).all(), (case, desc, int((a != b).sum()))
        same += 1
    assert same >= 15, same


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_interlaced_random_conversions_match_reference_or_are_refused_on_host(native_lib, emu_lib, ref, seed):
    rnd = random.Random(9000 + seed)
    tally = {}
    for it in range(90):
        case = random_case(rnd)
        r = check(lambda c, s: _emu_run(emu_lib, c, s), ref, case, seed * 1000 + it)
        tally[r] = tally.get(r, 0) + 1
    assert tally.get("ok", 0) + tally.get("staged", 0) >= 45, tally
    assert tally.get("unchecked", 0) <= 12, tally


def test_interlaced_refusals_and_modes(native_lib):
    """what is not built is refused, not converted as if progressive: different modes on the two infos (the reference refuses that too), fields / alternate,
    4:2:0 heights off the multiple of four, crops, dither stages, gamma remap"""
    def conv(ifmt, w, h, ofmt, ow, oh, mi=1, mo=1, **cfg):
        ii, oi = V.video_info(ifmt, w, h), V.video_info(ofmt, ow, oh)
        ii.interlace_mode, oi.interlace_mode = mi, mo
        return V.VideoConverter(ii, oi, V.converter_config(**cfg))
    with pytest.raises(V.GstAmdError) as e:
        conv("I420", 32, 16, "BGRA", 32, 16, 1, 0)
    assert e.value.code == V.ERR_INVALID
    for bad in (dict(mi=4, mo=4), dict(mi=3, mo=3)):
        with pytest.raises(V.GstAmdError) as e:
            conv("I420", 32, 16, "BGRA", 32, 16, **bad)
        assert e.value.code == V.ERR_UNSUPPORTED
    for args, cfg in ((("I420", 32, 18, "BGRA", 32, 18), {}), (("BGRA", 32, 16, "NV12", 32, 10), {}), (("I420", 32, 16, "BGRA", 32, 16), dict(src_x=2, src_width=16)),
                      (("BGRA", 32, 16, "RGB16", 32, 16), {}), (("I420", 32, 16, "BGRA", 32, 16), dict(gamma_mode="remap")),
                      (("Y444", 32, 16, "BGRA", 32, 16), dict(dither_quantization=8))):
        with pytest.raises(V.GstAmdError) as e:
            conv(*args, **cfg)
        assert e.value.code == V.ERR_UNSUPPORTED, (args, cfg)
    c = conv("NV12", 64, 32, "BGRA", 64, 32, 2, 2)          # mixed: the caller's flagged frames
    assert c.describe().startswith("interlaced{") and c.divergence() == ""
    c.free()


# ---- device ---------------------------------------------------------------------------------------------------------------------------------------
def _seeds(default):
    """GSTAMD_ILACE_SEEDS=a-b: a longer run (scripts/gpu_r06_interlaced.sh); one tally line per seed into GSTAMD_FUZZ_TALLY"""
    e = os.environ.get("GSTAMD_ILACE_SEEDS")
    if not e:
        return default
    a, b = e.split("-")
    return list(range(int(a), int(b) + 1))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", _seeds([1, 2, 3, 4, 5, 6, 7, 8]))
def test_hip_interlaced_random_conversions_match_reference_or_are_refused(native_lib, gpu, ref, seed):
    rnd = random.Random(9000 + seed)
    tally = {}
    for it in range(90):
        case = random_case(rnd)
        r = check(lambda c, s: _dev_run(gpu, c, s), ref, case, seed * 1000 + it)
        tally[r] = tally.get(r, 0) + 1
    if os.environ.get("GSTAMD_FUZZ_TALLY"):
        import json
        with open(os.environ["GSTAMD_FUZZ_TALLY"], "a") as f:
            f.write(json.dumps(dict(tally, seed=seed)) + "\n")
    assert tally.get("ok", 0) + tally.get("staged", 0) >= 45, tally


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    ("NV12", 1920, 1080, "BGRA", 1920, 1080, {}), ("I420", 1920, 1080, "UYVY", 1920, 1080, {}), ("UYVY", 1920, 1080, "I420", 1920, 1080, {}),
    ("NV12", 1920, 1080, "NV12", 720, 576, {"resampler_method": "linear"}), ("I420", 720, 576, "I420", 1920, 1080, {"resampler_method": "lanczos"}),
    ("YUY2", 720, 480, "BGRA", 720, 480, {}), ("BGRA", 1920, 1080, "NV12", 1920, 1080, {}), ("NV12", 1920, 1080, "BGRA", 1280, 720, {"resampler_method": "linear"}),
], ids=lambda c: "%s_%dx%d_%s_%dx%d" % c[:6])
def test_hip_interlaced_broadcast_sizes_match_reference(native_lib, gpu, ref, case):
    """1080i / 576i / 480i: decoder output to display and capture formats, the interlaced plane scaler between broadcast sizes"""
    r = check(lambda c, s: _dev_run(gpu, c, s), ref, case + ("bt709", None), 4242)
    assert r in ("ok", "staged"), r
