"""CPU-side tests (run with -m "not gpu"): the oracle against the golden vectors, the planner's host
logic, and the kernel bodies run by the host emulator (tests/emu) against the same goldens."""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import cases
from gstreamer_amd import video as V

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "video_golden.json")))


def test_c_abi_exports_every_declared_symbol(native_lib):
    """Every function declared in include/*.h must be exported by the shared library."""
    declared = set()
    for hdr in os.listdir(os.path.join(ROOT, "include")):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b(gstamd_[a-z0-9_]+)\s*\(", text))
    assert len(declared) > 15
    out = subprocess.check_output(["nm", "-D", "--defined-only", native_lib], text=True)
    exported = set(line.split()[-1] for line in out.splitlines() if line.strip())
    missing = sorted(declared - exported)
    assert not missing, "declared in include/ but not exported: %s" % missing
    lib = C.CDLL(native_lib)
    for name in declared:
        assert hasattr(lib, name)


def test_video_info_matches_reference_layout(native_lib, ref):
    for fmt in V.FORMATS:
        for (w, h) in ((2, 2), (3, 3), (322, 241), (1919, 1079), (1920, 1080), (3840, 2160), (320, 576), (320, 577)):
            mine = V.video_info(fmt, w, h)
            r = ref.video_info(fmt, w, h)
            n = r["n_planes"]
            assert mine.n_planes == n
            assert list(mine.stride)[:n] == r["stride"][:n], (fmt, w, h)
            assert list(mine.offset)[:n] == r["offset"][:n], (fmt, w, h)
            assert mine.size == r["size"]
            if r["colorimetry"] in V.COLORIMETRY:
                rng, mtx, trc, prim = V.COLORIMETRY[r["colorimetry"]]
                want = (V.COLOR_RANGE[rng], V.COLOR_MATRIX[mtx], V.TRANSFER[trc], V.PRIMARIES[prim])
            else:           # no name for it (GRAY8's 0..255 / bt601 / unknown / unknown): gst_video_colorimetry_to_string prints range:matrix:transfer:primaries
                want = tuple(int(v) for v in r["colorimetry"].split(":"))
            assert (mine.color_range, mine.color_matrix, mine.color_transfer, mine.color_primaries) == want, (fmt, w, h)
            assert mine.chroma_site == V.CHROMA_SITE[r["chroma_site"] or "unknown"]


def test_matrix_coefficients_pinned(native_lib):
    """bt709/bt601 limited-range YUV -> full-range RGB integer matrices."""
    for col, exp in (("bt709", [298, 459, 541, -55, -136]), ("bt601", [298, 409, 516, -100, -208])):
        c = V.VideoConverter(V.video_info("NV12", 64, 64, colorimetry=col), V.video_info("BGRA", 64, 64))
        m = c.debug_get(0)
        assert m[0] == 1 and m[1:6] == exp, (col, m)
        assert "fused_convert" in c.describe()
        c.free()


def test_c2_plan_and_algorithmic_bytes(native_lib):
    c = V.VideoConverter(V.video_info("NV12", 3840, 2160), V.video_info("BGRA", 3840, 2160))
    assert c.describe() == "fused_convert_pair[NV12->BGRA,h2cs,v2,matrix=ayuv_argb]"
    assert c.algorithmic_bytes() == 45619200
    vp = c.debug_get(1)
    # regular pairing (2k-1, 2k): line 0 and the last line unpaired
    assert (vp[0] & 0x3fffffff, vp[1]) == (0, 0)
    assert (vp[2] & 0x3fffffff, vp[2] >> 30, vp[3]) == (0, 0, 1) and (vp[4] & 0x3fffffff, vp[4] >> 30, vp[5]) == (0, 1, 1)
    assert (vp[2 * 2159] & 0x3fffffff, vp[2 * 2159 + 1]) == (1079, 1079)
    c.free()


def test_c3_plan(native_lib):
    cfg = V.converter_config(resampler_method="lanczos")
    c = V.VideoConverter(V.video_info("I420", 7680, 4320), V.video_info("RGBA", 1920, 1080), cfg)
    assert c.describe() == "scale[I420->RGBA,h2cs,v2,H16b,V16,matrix=ayuv_argb]"
    assert c.algorithmic_bytes() == 58060800
    info = c.debug_get(30)
    assert info[:3] == [3, 1, 16]            # N-tap, horizontal first, 16 taps
    taps = np.array(c.debug_get(20)).reshape(1920, 16)
    assert (taps.sum(axis=1) == 64).all()    # 6-bit DC-normalised (video-scaler.c:339-388)
    c.free()


@pytest.mark.parametrize("case", cases.VIDEO_REFUSED, ids=lambda c: "%s_%dx%d_%s_%dx%d" % tuple(c[:6]))
def test_unsupported_paths_are_refused_not_approximated(native_lib, case):
    ifmt, w, h, ofmt, ow, oh, cfg = case[:7]
    col, site = (case[7], case[8]) if len(case) > 7 else (None, None)
    try:
        oi = V.video_info(ofmt, ow, oh)
    except KeyError:
        return
    with pytest.raises(V.GstAmdError) as e:
        V.VideoConverter(V.video_info(ifmt, w, h, colorimetry=col, chroma_site=site), oi, V.converter_config(**cfg))
    assert e.value.code == V.ERR_UNSUPPORTED


def _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src, vec=1):
    emu_lib.emu_video_convert.argtypes = [C.POINTER(V.VideoInfo), C.POINTER(V.VideoInfo), C.POINTER(V.ConverterConfig),
                                          C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_int]
    col, ocol = cases.split_colorimetry(col)
    ii = V.video_info(ifmt, w, h, colorimetry=col, chroma_site=site)
    oi = V.video_info(ofmt, ow, oh, colorimetry=ocol)
    c = V.converter_config(**cfg)
    dst = np.zeros(oi.size, np.uint8)
    desc = C.create_string_buffer(256)
    r = emu_lib.emu_video_convert(C.byref(ii), C.byref(oi), C.byref(c), src.ctypes.data, dst.ctypes.data, vec, desc, 256)
    assert r == 0, desc.value
    return dst


@pytest.mark.parametrize("case", cases.VIDEO_DEFINED, ids=lambda c: c[0])
def test_reference_undefined_plans_compute_the_stage_by_stage_result_on_host(native_lib, emu_lib, ref, case):
    """Where the reference's ONE-step output is undefined (see cases.VIDEO_DEFINED) the plan is no longer refused: it announces the
    divergence and its bytes equal the reference run as the separate, well-defined conversions the chain consists of.  The one-step
    reference output is shown to differ - if it stops differing the class has left the undefined territory and belongs in the goldens."""
    name, (ifmt, w, h, ofmt, ow, oh, cfg), steps, mask = case
    conv = V.VideoConverter(V.video_info(ifmt, w, h), V.video_info(ofmt, ow, oh), V.converter_config(**cfg))
    assert conv.divergence() != "", name
    conv.free()
    src, exp, keep = cases.video_defined_expected(ref, case)
    got = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, None, None, src)
    emu_lib.emu_video_last_divergence.restype = C.c_char_p
    assert emu_lib.emu_video_last_divergence() != b""
    if keep is not None:
        assert (got[keep] == exp[keep]).all()
    else:
        assert (got == exp).all(), int((got != exp).sum())
    one_step = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, config=cases.ref_config_string(ref, cfg)).frame(src)
    assert (one_step != got).any()


@pytest.mark.parametrize("pair", [("BGRA", "RGBA"), ("ARGB", "BGRx"), ("RGBx", "xBGR"), ("ABGR", "ARGB"), ("AYUV", "VUYA"), ("BGRA", "BGRA")])
@pytest.mark.parametrize("w", [100, 64, 3])
def test_byte_permutations_take_the_copy_shaped_kernel_on_host(native_lib, emu_lib, ref, pair, w):
    """4-byte packed -> 4-byte packed with no colour step: k_swizzle4's body (one v_perm_b32 selector) against the reference"""
    a, b = pair
    if a == b:
        return          # the same format is the reference's plane copy, not this kernel
    h = 7
    src = cases.frame_bytes(w * h * 4, "random", 99, w)
    before = emu_lib.emu_swizzle4_runs()
    got = _emu_convert(emu_lib, a, w, h, b, w, h, {}, None, None, src)
    assert emu_lib.emu_swizzle4_runs() == before + (1 if (4 * w) % 16 == 0 else 0)       # rows that are not 16-byte aligned: the generic kernel
    exp = ref.VideoConverter(a, w, h, b, w, h).frame(src)
    assert (got == exp).all()


def test_plans_that_reproduce_the_reference_carry_no_divergence_note(native_lib):
    for (ifmt, w, h, ofmt, ow, oh, cfg) in (("NV12", 3840, 2160, "BGRA", 3840, 2160, {}), ("I420", 7680, 4320, "RGBA", 1920, 1080, cases.LAN),
                                            ("AYUV", 58, 18, "ARGB", 30, 20, cases.LIN), ("P010_10LE", 64, 48, "NV12", 64, 48, {})):
        conv = V.VideoConverter(V.video_info(ifmt, w, h), V.video_info(ofmt, ow, oh), V.converter_config(**cfg))
        assert conv.divergence() == "", (ifmt, ofmt, conv.divergence())
        conv.free()


@pytest.mark.parametrize("ow,oh,inside", [(28, 16, False), (30, 16, False), (16, 18, False), (18, 18, False), (20, 18, True), (30, 18, True), (83, 28, True),
                                          (40, 40, False), (60, 40, True)])
def test_nearest_enlargement_aliasing_class_has_its_boundary_where_the_reference_has_it(emu_lib, ref, ow, oh, inside):
    """the nearest scaler both ways on a 4 x 4 crop of NV12: exact while no source line is handed out more than four times or the horizontal pass runs
    first (out_width <= out_height for a square crop), announced as reference line aliasing otherwise - plans on both sides of both edges of the class"""
    import ctypes
    w, h = 16, 6
    cfg = dict(resampler_method="nearest", src_x=6, src_y=2, src_width=4, src_height=4)
    src = cases.frame_bytes(int(V.video_info("NV12", w, h).size), "random", 99, w)
    dst = _emu_convert(emu_lib, "NV12", w, h, "VUYA", ow, oh, cfg, None, None, src)
    emu_lib.emu_video_last_divergence.restype = ctypes.c_char_p
    note = emu_lib.emu_video_last_divergence().decode()
    exp = ref.VideoConverter("NV12", w, h, "VUYA", ow, oh, config=cases.ref_config_string(ref, cfg)).frame(src)
    if inside:
        assert "handed out more than four times" in note, note
        assert (dst != exp).any()           # the reference really differs there (every second source line's rows)
    else:
        assert note == "" and (dst == exp).all()


@pytest.mark.parametrize("cfg,ow,oh,inside", [
    (dict(resampler_method="nearest", gamma_mode="remap", primaries_mode="fast"), 3, 4, True),
    (dict(resampler_method="nearest", gamma_mode="remap", primaries_mode="fast"), 3, 3, True),
    (dict(resampler_method="nearest", gamma_mode="remap", alpha_mode="mult", alpha_value=0.5), 3, 4, True),
    (dict(resampler_method="linear", gamma_mode="remap", primaries_mode="fast"), 3, 4, False),          # two taps, two source lines: exact
    (dict(resampler_method="nearest", gamma_mode="remap", primaries_mode="fast"), 3, 2, False),         # no line handed out twice
    (dict(resampler_method="nearest", gamma_mode="remap"), 3, 4, False),                                  # nothing in place between the tables
    (dict(resampler_method="nearest", gamma_mode="remap", primaries_mode="fast"), 30, 4, False),        # enlarged overall: scaled after the encode table
    (dict(resampler_method="nearest", primaries_mode="fast"), 3, 4, False)])                              # the 8-bit chain converts a fresh copy
def test_gamma_chain_in_place_stage_after_a_repeating_vertical_scaler_is_announced(emu_lib, ref, cfg, ow, oh, inside):
    """gamma-mode = remap, scaling first, a source line handed out twice, and the primaries matrix / alpha multiply between the decode and encode
    tables: the reference converts the repeated line again (the two scalers share a one-line allocator there; found by the device fuzz of round 5,
    seed 7596) - announced, every row computed once from the source; the neighbouring plans are exact"""
    w, h = 20, 2
    ifmt = "ARGB" if "alpha_mode" in cfg else "xRGB"
    src = cases.frame_bytes(int(V.video_info(ifmt, w, h).size), "random", 5, w)
    dst = _emu_convert(emu_lib, ifmt, w, h, ifmt, ow, oh, cfg, "bt601", None, src)
    emu_lib.emu_video_last_divergence.restype = C.c_char_p
    note = emu_lib.emu_video_last_divergence().decode()
    exp = ref.VideoConverter(ifmt, w, h, ifmt, ow, oh, in_colorimetry="bt601", config=cases.ref_config_string(ref, cfg)).frame(src)
    if inside:
        assert "once more per repetition" in note, note
        got, want = dst.reshape(oh, -1), exp.reshape(oh, -1)
        assert (got != want).any()
        assert (got[0] == want[0]).all()            # a line's first hand-out is converted once by both
        if oh == 2 * h:
            assert (got[1] == got[0]).all()         # and its repetition equals it here
    else:
        assert note == "" and (dst == exp).all()


def test_set_config_replans_the_sub_converters(native_lib):
    """gst_video_converter_set_config (video-converter.c:2759): a converter re-configured in place ends up with the plan - and, for the
    composite plans, the sub-conversions - of a converter created with the new options (round 2 kept sub-converters planned with the old
    ones and dereferenced a missing one when gamma-mode = remap was switched on afterwards)"""
    for (ifmt, ofmt, w, h, ow, oh) in (("NV12", "BGRA", 64, 48, 64, 48), ("P010_10LE", "NV12", 64, 48, 32, 24), ("NV12", "I420", 64, 48, 96, 64)):
        for new in (dict(gamma_mode="remap"), dict(resampler_method="lanczos"), dict(gamma_mode="remap", resampler_method="nearest")):
            ii, oi = V.video_info(ifmt, w, h), V.video_info(ofmt, ow, oh)
            try:
                fresh = V.VideoConverter(ii, oi, V.converter_config(**new))
            except V.GstAmdError:
                fresh = None
            c = V.VideoConverter(ii, oi)
            before = c.describe()
            if fresh is None:
                with pytest.raises(V.GstAmdError):
                    c.set_config(V.converter_config(**new))
                assert c.describe() == before            # a refused config leaves the converter as it was
            else:
                c.set_config(V.converter_config(**new))
                assert c.describe() == fresh.describe(), (ifmt, ofmt, new)
                c.set_config(V.converter_config())
                assert c.describe() == before
                fresh.free()
            c.free()


SMALL = [c for c in enumerate(cases.VIDEO_CASES) if c[1][2] * c[1][3] <= 1280 * 720]


@pytest.mark.parametrize("idx_case", SMALL, ids=lambda c: c[1][0])
def test_kernel_bodies_on_host_match_golden(native_lib, emu_lib, idx_case):
    """Kernel bodies (same source as the HIP kernels) on the host CPU vs the reference's golden hashes."""
    i, (name, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern) = idx_case
    ii = V.video_info(ifmt, w, h)
    src = cases.frame_bytes(ii.size, pattern, cases.case_seed(name), w)
    assert cases.sha(src) == GOLDEN[name]["in_sha256"]
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"], (name, list(dst[:16]), GOLDEN[name]["head"][:16])


H420 = [c for c in SMALL if "_h420_" in c[1][0] or c[1][0] in ("nv12_bgra_quarter_lanczos", "i420_rgba_quarter_lanczos")]


@pytest.mark.parametrize("rows", [1, 3, 4, 7])
@pytest.mark.parametrize("idx_case", H420, ids=lambda c: c[1][0])
def test_hscale420_bodies_any_rows_per_wave(native_lib, emu_lib, idx_case, rows, monkeypatch):
    """k_hscale420_dot4 (video_hscale420.h): the lanes' chroma-row cache gives the same bytes wherever a wave starts and however many
    lines it walks; the path must actually be the one taken."""
    i, (name, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern) = idx_case
    monkeypatch.setenv("GSTAMD_H420_ROWS", str(rows))
    monkeypatch.setenv("GSTAMD_NO_H420_REG", "1")
    monkeypatch.setenv("GSTAMD_NO_COL", "1")
    src = cases.frame_bytes(V.video_info(ifmt, w, h).size, pattern, cases.case_seed(name), w)
    before = emu_lib.emu_h420_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert emu_lib.emu_h420_runs() == before + 1
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]


H420_REG = [c for c in H420 if c[1][1] in ("I420", "YV12", "NV12", "NV21")]


@pytest.mark.parametrize("rows", [4, 6, 10, 20])
@pytest.mark.parametrize("idx_case", H420_REG, ids=lambda c: c[1][0])
def test_hscale420_reg_bodies_any_lines_per_wave(native_lib, emu_lib, idx_case, rows, monkeypatch):
    """k_hscale420_reg (line pairs, closed-form chroma pairing, fixed register roles): same bytes for any block height, crop and
    chroma site included; the path must be the one taken for these 4:2:0 cases."""
    i, (name, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern) = idx_case
    monkeypatch.setenv("GSTAMD_H420_ROWS", str(rows))
    monkeypatch.setenv("GSTAMD_NO_FUSED420", "1")           # the two-pass form (the fused scalers are tested below)
    monkeypatch.setenv("GSTAMD_NO_COL", "1")
    src = cases.frame_bytes(V.video_info(ifmt, w, h).size, pattern, cases.case_seed(name), w)
    before = emu_lib.emu_h420_reg_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert emu_lib.emu_h420_reg_runs() == before + 1
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]


@pytest.mark.parametrize("geom", [(8, 17), (4, 4), (8, 8), (2, 5), (16, 1000), (3, 7)], ids=lambda g: "waves%d_rows%d" % g)
@pytest.mark.parametrize("idx_case", H420_REG, ids=lambda c: c[1][0])
def test_scale420_fused_bodies_any_geometry(native_lib, emu_lib, idx_case, geom, monkeypatch):
    """k_scale420_fused (video_scale420_fused.h: horizontal pass into an LDS ring of four-line groups, vertical pass as byte dot
    products down the ring) gives the reference's bytes for any waves-per-workgroup / rows-per-workgroup split (partial rounds,
    one chunk, tiny chunks), crop and chroma sites included; where the vertical pass is N-tap it must be the path taken."""
    i, (name, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern) = idx_case
    monkeypatch.setenv("GSTAMD_FUSED_WAVES", str(geom[0]))
    monkeypatch.setenv("GSTAMD_FUSED_ROWS", str(geom[1]))
    monkeypatch.setenv("GSTAMD_NO_COL", "1")
    src = cases.frame_bytes(V.video_info(ifmt, w, h).size, pattern, cases.case_seed(name), w)
    emu_lib.emu_fused_runs.restype = C.c_int
    before, before_reg = emu_lib.emu_fused_runs(), emu_lib.emu_h420_reg_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert emu_lib.emu_fused_runs() + emu_lib.emu_h420_reg_runs() == before + before_reg + 1
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]


def test_scale420_fused_is_taken_for_the_c3_shape(native_lib, emu_lib, monkeypatch):
    name = "i420_rgba_quarter_lanczos"
    monkeypatch.setenv("GSTAMD_NO_COL", "1")
    _, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern = [c for c in cases.VIDEO_CASES if c[0] == name][0]
    src = cases.frame_bytes(V.video_info(ifmt, w, h).size, pattern, cases.case_seed(name), w)
    before = emu_lib.emu_fused_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert emu_lib.emu_fused_runs() == before + 1
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]


# (outputs per lane, shared windows, waves per workgroup, rows per wave): one-wave workgroups (no hand-over), the smallest runs the
# hand-over allows, long runs (one workgroup for the picture)
COL_GEOMS = [(0, 1, 3, 5), (1, 1, 1, 4), (1, 1, 8, 1), (2, 1, 4, 3), (2, 0, 2, 7), (2, 1, 1, 1000), (1, 1, 5, 1000), (2, 1, 8, 2)]


@pytest.mark.parametrize("geom", COL_GEOMS, ids=lambda g: "opl%d_share%d_waves%d_rows%d" % g)
@pytest.mark.parametrize("idx_case", H420_REG, ids=lambda c: c[1][0])
def test_scale_col_bodies_any_geometry(native_lib, emu_lib, idx_case, geom, monkeypatch):
    """k_scale_col (video_scale_col.h: a wave per column tile walks down the source in groups of four lines, horizontal pass from byte
    planes in LDS, vertical pass down the lane's own ring words, the groups at the seam of two waves handed over through LDS) gives the
    reference's bytes for either number of outputs per lane, shared and private windows, any waves-per-workgroup / rows-per-wave split,
    crop and chroma sites included; where both passes are N-tap it must be the path taken."""
    i, (name, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern) = idx_case
    monkeypatch.setenv("GSTAMD_COL_OPL", str(geom[0]))
    monkeypatch.setenv("GSTAMD_COL_SHARE", str(geom[1]))
    monkeypatch.setenv("GSTAMD_COL_WAVES", str(geom[2]))
    monkeypatch.setenv("GSTAMD_COL_ROWS", str(geom[3]))
    src = cases.frame_bytes(V.video_info(ifmt, w, h).size, pattern, cases.case_seed(name), w)
    emu_lib.emu_col_runs.restype = C.c_int
    before = emu_lib.emu_col_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    ran = emu_lib.emu_col_runs() - before
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]
    if name in ("nv12_bgra_quarter_lanczos", "i420_rgba_quarter_lanczos", "nv12_bgra_h420_cubic_down"):
        assert ran == 1, name


def test_scale_col_is_taken_for_the_c3_shape(native_lib, emu_lib):
    name = "i420_rgba_quarter_lanczos"
    _, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern = [c for c in cases.VIDEO_CASES if c[0] == name][0]
    src = cases.frame_bytes(V.video_info(ifmt, w, h).size, pattern, cases.case_seed(name), w)
    before = emu_lib.emu_col_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert emu_lib.emu_col_runs() == before + 1
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]


def test_hscale420_switch_off_gives_the_same_bytes(native_lib, emu_lib, monkeypatch):
    name, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern = [c for c in cases.VIDEO_CASES if c[0] == "nv21_bgra_h420_jpeg_lanczos"][0]
    monkeypatch.setenv("GSTAMD_H420_ROWS", "0")
    monkeypatch.setenv("GSTAMD_NO_H420_REG", "1")
    src = cases.frame_bytes(V.video_info(ifmt, w, h).size, pattern, cases.case_seed(name), w)
    before = emu_lib.emu_h420_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert emu_lib.emu_h420_runs() == before
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]


FAST422 = [c for c in SMALL if "_fast422_" in c[1][0] or c[1][0] == "yuy2_bgra_640x360"]


@pytest.mark.parametrize("idx_case", FAST422, ids=lambda c: c[1][0])
def test_convert422_body_is_the_path_taken(native_lib, emu_lib, idx_case, monkeypatch):
    """k_convert422 (video_422_fast.h) serves unscaled packed 4:2:2 -> RGB with whole 8-pixel groups; switched off, the generic kernel
    gives the same bytes."""
    i, (name, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern) = idx_case
    src = cases.frame_bytes(V.video_info(ifmt, w, h).size, pattern, cases.case_seed(name), w)
    before = emu_lib.emu_fast422_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert emu_lib.emu_fast422_runs() == before + 1
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]
    monkeypatch.setenv("GSTAMD_NO_FAST422", "1")
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert emu_lib.emu_fast422_runs() == before + 1
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]


FAST420P = [c for c in SMALL if ("_fast420p_" in c[1][0] and "crop" not in c[1][0]) or c[1][0] in ("yv12_xrgb_640x360_fastpath", "i420_rgba_1280x720_fastpath_bt601")]


@pytest.mark.parametrize("idx_case", FAST420P, ids=lambda c: c[1][0])
def test_convert420p_body_is_the_path_taken(native_lib, emu_lib, idx_case, monkeypatch):
    """k_convert420p serves the reference's I420 / YV12 -> RGB same-size fastpaths (nearest chroma) with whole 8-pixel groups; switched
    off, the generic kernel gives the same bytes."""
    i, (name, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern) = idx_case
    src = cases.frame_bytes(V.video_info(ifmt, w, h).size, pattern, cases.case_seed(name), w)
    before = emu_lib.emu_fast420p_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert emu_lib.emu_fast420p_runs() == before + 1
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]
    monkeypatch.setenv("GSTAMD_NO_FAST420P", "1")
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert emu_lib.emu_fast420p_runs() == before + 1
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]


@pytest.mark.parametrize("idx_case", [c for c in SMALL if "_bil420_" in c[1][0]], ids=lambda c: c[1][0])
def test_bilinear420_serves_planar_sources(native_lib, emu_lib, idx_case):
    i, (name, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern) = idx_case
    src = cases.frame_bytes(V.video_info(ifmt, w, h).size, pattern, cases.case_seed(name), w)
    before = emu_lib.emu_bil_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert emu_lib.emu_bil_runs() == before + 1
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]


WIDE = [c for c in SMALL if c[1][1] in ("NV12", "NV21") and c[1][2] >= 512 and (c[1][2], c[1][3]) == (c[1][5], c[1][6])
        and c[1][4] not in ("AYUV",)]


@pytest.mark.parametrize("k", [101, 102, 105, 201, 203, 205])
@pytest.mark.parametrize("idx_case", WIDE, ids=lambda c: c[1][0])
def test_fast_kernel_bodies_any_pairs_per_wave(native_lib, emu_lib, idx_case, k):
    """Strip (100 + K) and wide (200 + K) kernel bodies: the strip length (line pairs per wave) is a tuning knob,
    results must not depend on it."""
    i, (name, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern) = idx_case
    src = cases.frame_bytes(V.video_info(ifmt, w, h).size, pattern, cases.case_seed(name), w)
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src, vec=k)
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"], name


BIL = [c for c in SMALL if c[1][1] in ("NV12", "NV21") and c[1][7] == cases.LIN and c[1][0] != "nv12_bgra_up2_bilinear"]


@pytest.mark.parametrize("variant", [300, 464, 528, 656])
@pytest.mark.parametrize("idx_case", BIL, ids=lambda c: c[1][0])
def test_bilinear_kernel_bodies_agree(native_lib, emu_lib, idx_case, variant):
    """Bilinear plans from semi-planar 4:2:0: the generic wave-tile kernel (300) and the fused kernel with tiles of
    64 / 128 / 256 outputs per wave (400 + w) all reproduce the reference."""
    i, (name, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern) = idx_case
    src = cases.frame_bytes(V.video_info(ifmt, w, h).size, pattern, cases.case_seed(name), w)
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src, vec=variant)
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"], name


BILR = [c for c in SMALL if c[1][1] in ("NV12", "NV21", "I420", "YV12") and c[1][7] == cases.LIN and c[1][2] % 16 == 0
        and c[1][4] in ("BGRA", "RGBA", "ARGB", "ABGR", "BGRx", "RGBx", "xRGB", "xBGR") and "letterbox" not in c[1][0]]


@pytest.mark.parametrize("rows,tile", [(1, 0), (3, 0), (4, 0), (16, 0), (4, 256), (5, 128), (-100, 0), (-7, 0), (100, 0)])
@pytest.mark.parametrize("idx_case", BILR, ids=lambda c: c[1][0])
def test_bilinear_rows_kernel_body_any_rows_per_wave(native_lib, emu_lib, idx_case, rows, tile, monkeypatch):
    """k_bilinear420_rows (video_bilinear_rows.h): chroma upsampled once per source pixel in byte lanes, a wave walking `rows`
    output rows with the filtered chroma rows it carries.  The rows per wave are a tuning knob (negative: balanced strips for a device with that many wave
    slots, as the launcher makes them; 100: more rows than the 64 a strip holds); where the three-row window form
    does not apply (no vertical chroma upsampling, vertical-first plans) the older kernels serve the plan."""
    i, (name, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern) = idx_case
    src = cases.frame_bytes(V.video_info(ifmt, w, h).size, pattern, cases.case_seed(name), w)
    monkeypatch.setenv("EMU_BIL_ROWS", str(rows))
    monkeypatch.setenv("EMU_NO_BILINEAR_HALF", "1")           # exact halvings have their own kernel (test below)
    if tile:
        monkeypatch.setenv("EMU_BIL_ROWS_TILE", str(tile))     # default: 384 outputs per wave where the span fits, else 256 ...
    emu_lib.emu_bilr_runs.restype = C.c_int
    before = emu_lib.emu_bilr_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    took = emu_lib.emu_bilr_runs() - before
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"], name
    if rows == 4 and not tile:
        print(name, "rows kernel" if took else "older kernel")
    if name in ("nv12_bgra_2to1_bilinear_1280x720", "i420_bgra_bil420_half", "nv12_bgra_half_bilinear"):
        assert took == 1, name


HALF = [c for c in SMALL if c[1][0].startswith("half_") or c[1][0] in ("nv12_bgra_2to1_bilinear_1280x720", "i420_bgra_bil420_half", "nv12_bgra_half_bilinear")]


@pytest.mark.parametrize("rows", [5, 1, 4, 64, -100, -7, 100])
@pytest.mark.parametrize("idx_case", HALF, ids=lambda c: c[1][0])
def test_bilinear_half_kernel_body_any_rows_per_wave(native_lib, emu_lib, idx_case, rows, monkeypatch):
    """k_bilinear420_half (video_bilinear_half.h): pictures that shrink by exactly two - a lane turns the 16 source pixels of one load into eight
    consecutive outputs, both passes as v_dot2 on {even | odd} pixel pairs, no LDS.  Rows per wave are a tuning knob (negative: balanced strips);
    a width that is no multiple of 16 stays with the rows kernel's predecessors."""
    i, (name, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern) = idx_case
    src = cases.frame_bytes(V.video_info(ifmt, w, h).size, pattern, cases.case_seed(name), w)
    monkeypatch.setenv("EMU_BIL_HALF_ROWS", str(rows))
    emu_lib.emu_bilh_runs.restype = C.c_int
    before = emu_lib.emu_bilh_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    took = emu_lib.emu_bilh_runs() - before
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"], name
    assert took == (0 if name == "half_nv12_bgra_width_not_16" else 1), name


UP4 = [c for c in SMALL if c[1][0].startswith("up4_") or c[1][0] in ("bgrx_bgrx_up_bilinear_planes", "nv12_bgra_up2_bilinear", "i420_vuya_up_bilinear",
                                                                     "quad4_argb_up_bilinear_odd")]


@pytest.mark.parametrize("rows", [5, 1, 2, 64])
@pytest.mark.parametrize("idx_case", UP4, ids=lambda c: c[1][0])
def test_bilinear4_up_kernel_body_any_rows_per_wave(native_lib, emu_lib, idx_case, rows, monkeypatch):
    """k_bilinear4_up (bilinear4_up_lane): enlargements of 4-byte pixels with the filtered source lines carried down a strip of rows; the strip
    height is a tuning knob.  Plans it does not take (vertical first, a colour stage ahead of the scaler) stay with the older kernels."""
    i, (name, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern) = idx_case
    src = cases.frame_bytes(V.video_info(ifmt, w, h).size, pattern, cases.case_seed(name), w)
    monkeypatch.setenv("EMU_BIL4_UP_ROWS", str(rows))
    emu_lib.emu_bil4_up_runs.restype = C.c_int
    before = emu_lib.emu_bil4_up_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    took = emu_lib.emu_bil4_up_runs() - before
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"], name
    if rows == 5:
        print(name, "up kernel" if took else "older kernel")
    assert took == (0 if name in ("up4_ayuv_bgra_mixed", "up4_xrgb_bgrx_tiny_source") else 1), name
    monkeypatch.setenv("EMU_NO_BILINEAR4_UP", "1")
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"], name


@pytest.mark.parametrize("idx_case", [c for c in SMALL if c[1][0].startswith("pack422up_")], ids=lambda c: c[1][0])
def test_convert_pack_422up_is_the_body_that_runs(native_lib, emu_lib, idx_case, monkeypatch):
    """packed 4:2:2 -> semi-planar 4:2:0 / 4:4:4 through k_convert_pack_422up (the chain's horizontal chroma upsampler inside the packer's row source),
    through the AYUV image + k_pack_planar with it switched off: the reference's bytes either way"""
    i, (name, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern) = idx_case
    src = cases.frame_bytes(V.video_info(ifmt, w, h).size, pattern, cases.case_seed(name), w)
    emu_lib.emu_pack422up_runs.restype = C.c_int
    before = emu_lib.emu_pack422up_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    took = emu_lib.emu_pack422up_runs() - before
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"], name
    assert (took > 0) == (w >= 8), (name, took)         # whole blocks of pictures at least eight pixels wide
    monkeypatch.setenv("GSTAMD_NO_CONVERT_PACK_422UP", "1")
    before = emu_lib.emu_pack422up_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert emu_lib.emu_pack422up_runs() == before
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"], name


@pytest.mark.parametrize("idx_case", SMALL[::4], ids=lambda c: c[1][0])
def test_golden_vectors_are_the_references_output(ref, idx_case):
    """Pins the committed golden hashes to the reference implementation itself (oracle/_ref)."""
    i, (name, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern) = idx_case
    src = cases.frame_bytes(ref.video_info(ifmt, w, h)["size"], pattern, cases.case_seed(name), w)
    col, ocol = cases.split_colorimetry(col)
    rc = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, in_colorimetry=col, in_chroma_site=site, out_colorimetry=ocol,
                            config=cases.ref_config_string(ref, cfg))
    assert cases.video_digest(name, rc.frame(src)) == GOLDEN[name]["sha256"]


def test_reference_output_depends_on_thread_slicing_for_420(ref):
    """Documents a reference property the parity target has to be pinned against: for 4:2:0 input the
    generic path's chroma line pairing restarts at every thread slice (video-converter.c:2991-3021 with
    the slices of :3346-3363), so n-threads > 1 changes pixels from the second slice on (the
    reference's own invariance test, tests/check/libs/video.c:3189, only covers ARGB->BGRx).
    Parity is therefore defined against n-threads=1, the element default (gstvideoconvertscale.c:144)."""
    src = cases.frame_bytes(ref.video_info("NV12", 640, 800)["size"], "random", 77)
    outs = []
    for t in (1, 4):
        rc = ref.VideoConverter("NV12", 640, 800, "BGRA", 640, 800, config=ref.config_string(GstVideoConverter__threads=t))
        outs.append(rc.frame(src).reshape(800, 640 * 4))
    assert (outs[0][:200] == outs[1][:200]).all()          # first slice: same pairing
    assert (outs[0][200:] != outs[1][200:]).any()          # later slices: pairing phase flipped
    # no chroma subsampling -> slice independent, as the reference's own test pins
    src = cases.frame_bytes(ref.video_info("ARGB", 640, 800)["size"], "random", 78)
    outs = []
    for t in (1, 4):
        rc = ref.VideoConverter("ARGB", 640, 800, "AYUV", 640, 800, config=ref.config_string(GstVideoConverter__threads=t))
        outs.append(rc.frame(src))
    assert (outs[0] == outs[1]).all()


@pytest.mark.parametrize("name", [c[0] for c in cases.VIDEO_CASES if c[0].startswith("planes16_")])
def test_deep_planes16_body_is_the_one_that_runs(native_lib, emu_lib, name, monkeypatch):
    """the planes16_* cases go through deep_planes16_body (sixteen samples per lane) - all but the one whose rows are no multiple of 16
    samples - and through deep_planes_body with GSTAMD_NO_DEEP_PLANES16, with the reference's bytes either way"""
    _, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern = [c for c in cases.VIDEO_CASES if c[0] == name][0]
    src = cases.frame_bytes(V.video_info(ifmt, w, h).size, pattern, cases.case_seed(name), w)
    emu_lib.emu_deep16_runs.restype = C.c_int
    before = emu_lib.emu_deep16_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert emu_lib.emu_deep16_runs() - before == (0 if name.endswith("row_not_16") else 1)
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]
    monkeypatch.setenv("GSTAMD_NO_DEEP_PLANES16", "1")
    before = emu_lib.emu_deep16_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert emu_lib.emu_deep16_runs() == before
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]


@pytest.mark.parametrize("name", [c[0] for c in cases.VIDEO_CASES if c[0].startswith("quad_") and c[2] * c[3] <= 1280 * 720])
def test_plane_quad_body_is_the_one_that_runs(native_lib, emu_lib, name, monkeypatch):
    """the quad_* cases go through plane_quad_body (four output bytes per lane from two 8-byte windows) - all but the one that shrinks by
    more than 2:1 - and through the per-pixel bodies with GSTAMD_NO_PLANE_QUAD, with the reference's bytes either way"""
    _, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern = [c for c in cases.VIDEO_CASES if c[0] == name][0]
    src = cases.frame_bytes(V.video_info(ifmt, w, h).size, pattern, cases.case_seed(name), w)
    emu_lib.emu_quad_runs.restype = C.c_int
    before = emu_lib.emu_quad_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    ran = emu_lib.emu_quad_runs() - before
    assert (ran == 0) == ("too_steep" in name), ran
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]
    monkeypatch.setenv("GSTAMD_NO_PLANE_QUAD", "1")
    before = emu_lib.emu_quad_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert emu_lib.emu_quad_runs() == before
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]


@pytest.mark.parametrize("counts", [(540, 540, 0), (540, 0, 0), (1, 1, 1), (7, 1000, 3), (1000, 7, 0), (0, 5, 9), (136, 34, 34), (3, 0, 8), (4097, 129, 77)])
def test_plane_quad_grid_mapping_is_a_bijection(emu_lib, counts):
    """k_plane_quad's workgroup -> (plane, workgroup of the plane) mapping: every pair exactly once"""
    assert emu_lib.emu_quad_grid_check(counts[0], counts[1], counts[2]) == 1, counts


@pytest.mark.parametrize("name", [c[0] for c in cases.VIDEO_CASES if c[0].startswith("enc16_") and c[2] * c[3] <= 1280 * 720])
def test_encode16_body_is_the_one_that_runs(native_lib, emu_lib, name, monkeypatch):
    """the enc16_* cases go through enc16_block (frame to frame in one kernel) - all but the one whose width is no multiple of 4 - and through
    the three-stage composite with GSTAMD_NO_ENCODE16, with the reference's bytes either way"""
    _, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern = [c for c in cases.VIDEO_CASES if c[0] == name][0]
    src = cases.frame_bytes(V.video_info(ifmt, w, h).size, pattern, cases.case_seed(name), w)
    emu_lib.emu_enc16_runs.restype = C.c_int
    before = emu_lib.emu_enc16_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert emu_lib.emu_enc16_runs() - before == (0 if "width_not_4" in name else 1)
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]
    monkeypatch.setenv("GSTAMD_NO_ENCODE16", "1")
    before = emu_lib.emu_enc16_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert emu_lib.emu_enc16_runs() == before
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]


@pytest.mark.parametrize("name", [c[0] for c in cases.VIDEO_CASES if c[0].startswith("quad4_") and c[2] * c[3] <= 1280 * 720])
def test_plane_quad_body_takes_the_4_byte_plane_scaler(native_lib, emu_lib, name, monkeypatch):
    """convert_scale_planes on a packed 4-byte format with two short passes is one plane of four-byte pixels to plane_quad_body (all quad4_*
    cases but the one that shrinks by more than 2:1 and the enlargement); k_bilinear4_rows with GSTAMD_NO_PLANE_QUAD; the reference's bytes either way"""
    _, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern = [c for c in cases.VIDEO_CASES if c[0] == name][0]
    src = cases.frame_bytes(V.video_info(ifmt, w, h).size, pattern, cases.case_seed(name), w)
    emu_lib.emu_quad_runs.restype = C.c_int
    before = emu_lib.emu_quad_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert (emu_lib.emu_quad_runs() - before == 0) == ("too_steep" in name or "_up_" in name)
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]
    monkeypatch.setenv("GSTAMD_NO_PLANE_QUAD", "1")
    before = emu_lib.emu_quad_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert emu_lib.emu_quad_runs() == before
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]


@pytest.mark.parametrize("case", cases.DEEP_NOFILL, ids=lambda c: "%s_%s_w%d_q%d" % (c[0], c[3], c[1], c[6].get("dither_quantization", 1)))
def test_deep_plane_copies_without_border_fill_on_host(native_lib, emu_lib, ref, case):
    """10 / 12 / 16-bit plane copies into a destination rectangle with fill-border = FALSE (round 5; the plans that ignored the rectangle
    in round 4): the bytes the picture decides equal the reference's, the border lines above and below stay untouched"""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import fuzz_video
    ifmt, w, h, ofmt, ow, oh, cfg = case
    src = cases.frame_bytes(V.video_info(ifmt, w, h).size, "random", 4242 + w, w)
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, None, None, src)
    ok, text = fuzz_video.matches_reference(ref, (ifmt, w, h, ofmt, ow, oh, cfg, None, None), src, dst, V.video_info(ofmt, ow, oh))
    assert ok, text
    stride = V.video_info(ofmt, ow, oh).stride[0]
    assert not dst[:cfg["dest_y"] * stride].any()


@pytest.mark.parametrize("name", [c[0] for c in cases.VIDEO_CASES if c[0].startswith("dsp_") or c[0].startswith("dsp4_") or c[0].startswith("dsp16_") or c[0].startswith("dspm_")])
def test_deep_scale_pack_body_is_the_one_that_runs(native_lib, emu_lib, name, monkeypatch):
    """the dsp_* cases (a 10-bit planar / semi-planar source that halves into an 8-bit planar / semi-planar destination) go through
    k_deep_scale_pack's lane function (video_deep_pack.h: front, both u16 passes, narrowing and pack per block), the dsp4_* ones (a 4-byte
    destination) through k_deep_scale4's, the dsp16_* ones (a 10 / 12 / 16-bit planar destination) through k_deep_scale_pack16's - the *_not_* ones (other filters, ratios, widths, a dither stage ahead of a planar pack) do not - and through
    the multi-launch forms with GSTAMD_NO_DEEP_SCALE_PACK, with the reference's bytes either way"""
    _, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern = [c for c in cases.VIDEO_CASES if c[0] == name][0]
    src = cases.frame_bytes(V.video_info(ifmt, w, h).size, pattern, cases.case_seed(name), w)
    emu_lib.emu_deep_pack_runs.restype = C.c_int
    before = emu_lib.emu_deep_pack_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert emu_lib.emu_deep_pack_runs() - before == (0 if "_not_" in name else 1)
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]
    monkeypatch.setenv("GSTAMD_NO_DEEP_SCALE_PACK", "1")
    before = emu_lib.emu_deep_pack_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert emu_lib.emu_deep_pack_runs() == before
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]


def test_packed_chroma_identities_of_deep_scale_pack():
    """video_deep_pack.h runs the u16 chroma upsamplers on {c1 | c2 << 16} pairs through (a | b) - ((a ^ b) >> 1), (a & b) + ((a ^ b) >> 1) and
    (3 a + b + 2) >> 2 = avg_ceil (a, avg_floor (a, b)): checked here on the edge values and a random sample of 16-bit pairs"""
    import numpy as np
    rng = np.random.default_rng(5)
    a = np.concatenate([np.array([0, 0, 1, 65535, 65535, 65534, 1, 2, 3, 32768], np.int64), rng.integers(0, 65536, 200000)])
    b = np.concatenate([np.array([0, 65535, 0, 65535, 0, 65535, 2, 1, 65535, 32767], np.int64), rng.integers(0, 65536, 200000)])
    avgc = (a | b) - ((a ^ b) >> 1)
    avgf = (a & b) + ((a ^ b) >> 1)
    assert (avgc == (a + b + 1) >> 1).all() and (avgf == (a + b) >> 1).all()
    f31 = (a | avgf) - ((a ^ avgf) >> 1)
    assert (f31 == (3 * a + b + 2) >> 2).all()
    assert (f31 == (6 * a + 2 * b + 4) >> 3).all()


@pytest.mark.parametrize("name", [c[0] for c in cases.VIDEO_CASES if (c[0].startswith("bay_") or c[0].startswith("baym_")) and c[2] * c[3] <= 1280 * 720])
def test_bilinear420_ayuv_layout_is_the_path_taken(native_lib, emu_lib, name, monkeypatch):
    """the bay_* cases (a scaled 8-bit 4:2:0 -> YUV conversion without a colour stage) go through the bilinear 4:2:0 kernels with the layout that
    stores A Y U V, the baym_* ones with the 8-bit convert stage behind it - bay_not_* (three-plane sources of other widths) does not - and through the generic scalers with GSTAMD_NO_BILINEAR_AYUV, with
    the reference's bytes either way"""
    _, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern = [c for c in cases.VIDEO_CASES if c[0] == name][0]
    src = cases.frame_bytes(V.video_info(ifmt, w, h).size, pattern, cases.case_seed(name), w)
    emu_lib.emu_bil_ayuv_runs.restype = C.c_int
    before = emu_lib.emu_bil_ayuv_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert emu_lib.emu_bil_ayuv_runs() - before == (0 if "_not_" in name else 1)
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]
    monkeypatch.setenv("GSTAMD_NO_BILINEAR_AYUV", "1")
    before = emu_lib.emu_bil_ayuv_runs()
    dst = _emu_convert(emu_lib, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert emu_lib.emu_bil_ayuv_runs() == before
    assert cases.video_digest(name, dst) == GOLDEN[name]["sha256"]
