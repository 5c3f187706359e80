"""Compositor blend path: kernel bodies on the host emulator (CPU tests) and the HIP kernel (-m gpu)
against the reference's BlendFunction / fill functions (oracle/_ref: blend.c + compositororc-dist.c)."""
import ctypes as C

import numpy as np
import pytest

import cases
from gstreamer_amd import video as V

FAM = {"BGRA": "bgra", "RGBA": "bgra", "ARGB": "argb", "ABGR": "argb", "AYUV": "argb", "VUYA": "bgra"}       # blend.h:55-65
CHECKER_FN = {"AYUV": "ayuv", "VUYA": "vuya"}

BLEND_CASES = []
for fmt in ("BGRA", "ARGB", "RGBA", "AYUV", "VUYA"):
    for mode in (0, 1, 2):
        for alpha in (1.0, 0.5, 0.3, 0.004, 0.0):
            BLEND_CASES.append((fmt, 0, 37, 21, 64, 48, 5, 7, alpha, mode, 0, 48))
            BLEND_CASES.append((fmt, 1, 37, 21, 64, 48, -9, -4, alpha, mode, 0, 48))
BLEND_CASES += [("BGRA", 0, 100, 80, 64, 48, 40, 30, 0.7, 1, 10, 40), ("BGRA", 1, 100, 80, 64, 48, -20, -30, 0.7, 1, 10, 40),
                ("BGRA", 0, 10, 10, 64, 48, 70, 30, 0.7, 1, 0, 48), ("ARGB", 0, 64, 48, 64, 48, 0, 0, 1.0, 0, 0, 48)]


class PadDev(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_int), ("height", C.c_int), ("stride", C.c_int), ("xpos", C.c_int),
                ("ypos", C.c_int), ("s_alpha", C.c_int), ("mode", C.c_int)]


class AggParams(C.Structure):
    _fields_ = [("ashift", C.c_int), ("overlay", C.c_int), ("bg_kind", C.c_int), ("checker_yuv", C.c_int),
                ("bg_word", C.c_uint32), ("n_pads", C.c_int), ("fast", C.c_int), ("pads", PadDev * 32)]


def ref_blend(ref, case, src, dst):
    fmt, overlay, sw, sh, dw, dh, xpos, ypos, alpha, mode, y0, y1 = case
    func = ("overlay_" if overlay else "blend_") + FAM[fmt]
    return ref.compositor_blend(func, fmt, src, sw, sh, xpos, ypos, alpha, dst, dw, dh, y0, y1, mode)


def make_blend_inputs(case, i):
    fmt, overlay, sw, sh, dw, dh = case[:6]
    return cases.frame_bytes(sw * sh * 4, "random", 500 + i), cases.frame_bytes(dw * dh * 4, "random", 900 + i)


@pytest.mark.parametrize("i_case", list(enumerate(BLEND_CASES))[::3], ids=lambda c: "%s_%d_%s_m%d" % (c[1][0], c[1][1], c[1][8], c[1][9]))
def test_blend_body_on_host_matches_reference(emu_lib, ref, i_case):
    i, case = i_case
    fmt, overlay, sw, sh, dw, dh, xpos, ypos, alpha, mode, y0, y1 = case
    emu_lib.emu_compositor_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    assert C.sizeof(AggParams) == emu_lib.emu_sizeof_params()
    src, dst = make_blend_inputs(case, i)
    exp = ref_blend(ref, case, src, dst.copy())
    got = dst.copy()
    sa = max(0, min(255, int(alpha * 255)))
    if sa:
        p = AggParams()
        p.ashift = 0 if FAM[fmt] == "argb" else 24
        p.overlay, p.bg_kind, p.n_pads = overlay, 2, 1
        pd = p.pads[0]
        pd.data, pd.width, pd.height, pd.stride, pd.xpos, pd.ypos, pd.s_alpha, pd.mode = src.ctypes.data, sw, sh, sw * 4, xpos, ypos, sa, mode
        yy1 = min(y1, dh)
        x0, r0, x1, r1 = max(xpos, 0), max(ypos, y0), min(xpos + sw, dw), min(ypos + sh, yy1)
        if x1 > x0 and r1 > r0:
            emu_lib.emu_compositor_run(C.byref(p), got.ctypes.data, dw * 4, x0, r0, x1 - x0, r1 - r0)
    assert (exp == got).all()


@pytest.mark.gpu
@pytest.mark.parametrize("i_case", list(enumerate(BLEND_CASES)), ids=lambda c: "%s_%d_%s_m%d" % (c[1][0], c[1][1], c[1][8], c[1][9]))
def test_hip_blend_matches_reference(native_lib, gpu, ref, i_case):
    import torch
    i, case = i_case
    fmt, overlay, sw, sh, dw, dh, xpos, ypos, alpha, mode, y0, y1 = case
    src, dst = make_blend_inputs(case, i)
    exp = ref_blend(ref, case, src, dst.copy())
    d_src, d_dst = torch.from_numpy(src).to(gpu), torch.from_numpy(dst).to(gpu)
    r = V.lib().gstamd_compositor_blend(V.FORMATS[fmt], overlay, d_src.data_ptr(), sw, sh, sw * 4, xpos, ypos, alpha,
                                        d_dst.data_ptr(), dw, dh, dw * 4, y0, y1, mode, None)
    assert r == 0
    torch.cuda.synchronize()
    assert (d_dst.cpu().numpy() == exp).all()


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["BGRA", "ARGB", "RGBA", "ABGR", "AYUV", "VUYA"])
def test_hip_fills_match_reference(native_lib, gpu, ref, fmt):
    import torch
    w, h = 70, 33
    exp = np.zeros(w * h * 4, np.uint8)
    checker_fn = CHECKER_FN.get(fmt, FAM[fmt])
    d = torch.zeros(w * h * 4, dtype=torch.uint8, device=gpu)
    if checker_fn:
        ref.compositor_fill(0, checker_fn, fmt, exp, w, h, 3, 30)
        assert V.lib().gstamd_compositor_fill_checker(V.FORMATS[fmt], d.data_ptr(), w, h, w * 4, 3, 30, None) == 0
        torch.cuda.synchronize()
        assert (d.cpu().numpy() == exp).all()
    if True:
        exp[:] = 0
        d.zero_()
        ref.compositor_fill(1, fmt.lower(), fmt, exp, w, h, 2, 31, 10, 200, 77)
        assert V.lib().gstamd_compositor_fill_color(V.FORMATS[fmt], d.data_ptr(), w, h, w * 4, 2, 31, 10, 200, 77, None) == 0
        torch.cuda.synchronize()
        assert (d.cpu().numpy() == exp).all()


def _black_white(fmt, background):
    """compositor.c:1131-1149: the colours of gst_video_color_range_offsets for the format's default range"""
    if fmt in CHECKER_FN:
        return (16, 128, 128) if background == 1 else (235, 128, 128)
    return (0, 0, 0) if background == 1 else (255, 255, 255)


def _aggregate_expected(ref, fmt, background, geo, pads_np, pw, ph, dw, dh):
    fam = FAM[fmt]
    exp = np.zeros(dw * dh * 4, np.uint8)
    if background == 0:
        ref.compositor_fill(0, CHECKER_FN.get(fmt, fam), fmt, exp, dw, dh, 0, dh)
    elif background in (1, 2):
        ref.compositor_fill(1, fmt.lower(), fmt, exp, dw, dh, 0, dh, *_black_white(fmt, background))
    func = ("overlay_" if background == 3 else "blend_") + fam
    for i, (xpos, ypos, alpha, mode) in enumerate(geo):
        ref.compositor_blend(func, fmt, pads_np[i], pw, ph, xpos, ypos, alpha, exp, dw, dh, 0, dh, mode)
    return exp


def _over_geometry(n_pads):
    """all-OVER / ADD pads at odd offsets, partly outside the canvas: the packed 4-pixel blend path"""
    return [((i % 4) * 43 - 11, (i // 4) * 29 - 7, min(1.0, 0.25 + 0.05 * i), 1 if i % 3 else 2) for i in range(n_pads)]


@pytest.mark.parametrize("direct,dw,n_pads", [(1, 203, 12), (1, 204, 32), (1, 517, 29), (0, 203, 12)])
@pytest.mark.parametrize("fmt,background", [("BGRA", 0), ("ARGB", 1), ("AYUV", 0), ("RGBA", 2), ("VUYA", 0), ("VUYA", 1)])
def test_aggregate_packed_path_on_host_matches_reference(emu_lib, ref, fmt, background, direct, dw, n_pads, monkeypatch):
    """the opaque-blend path: k_aggregate_direct's body (mask walk over the hits, 12 request slots per round: 29 / 32 pads piled on
    each other need three rounds; a width of 203 moves the last lane back onto the last four pixels) and k_aggregate's (hit list),
    which still serves rectangles narrower than four pixels"""
    emu_lib.emu_compositor_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    if not direct:
        monkeypatch.setenv("EMU_AGG_NO_DIRECT", "1")
    dh, pw, ph = 97, 90, 41
    pads_np = [cases.frame_bytes(pw * ph * 4, "random", 7100 + i) for i in range(n_pads)]
    geo = _over_geometry(n_pads)
    if n_pads == 29:            # piled up: more than 12 hits under one strip and row
        geo = [(3 * i - 5, i - 3, min(1.0, 0.25 + 0.02 * i), 1 if i % 3 else 2) for i in range(n_pads)]
    exp = _aggregate_expected(ref, fmt, background, geo, pads_np, pw, ph, dw, dh)
    p = AggParams()
    p.ashift = 0 if FAM[fmt] == "argb" else 24
    p.overlay, p.bg_kind, p.checker_yuv = 0, (0 if background == 0 else 1), int(fmt in CHECKER_FN)
    if background:
        word = np.zeros(4, np.uint8)
        ref.compositor_fill(1, fmt.lower(), fmt, word, 1, 1, 0, 1, *_black_white(fmt, background))
        p.bg_word = int(word.view(np.uint32)[0])
    k = 0
    for i, (xpos, ypos, alpha, mode) in enumerate(geo):
        sa = max(0, min(255, int(alpha * 255)))
        if not sa:
            continue
        pd = p.pads[k]
        pd.data, pd.width, pd.height, pd.stride, pd.xpos, pd.ypos, pd.s_alpha, pd.mode = pads_np[i].ctypes.data, pw, ph, pw * 4, xpos, ypos, sa, mode
        k += 1
    p.n_pads = k
    got = np.zeros(dw * dh * 4, np.uint8)
    emu_lib.emu_compositor_strip_runs.restype = emu_lib.emu_compositor_rows_runs.restype = C.c_int
    emu_lib.emu_compositor_direct_runs.restype = C.c_int
    before = (emu_lib.emu_compositor_strip_runs(), emu_lib.emu_compositor_rows_runs(), emu_lib.emu_compositor_direct_runs())
    emu_lib.emu_compositor_run(C.byref(p), got.ctypes.data, dw * 4, 0, 0, dw, dh)
    after = (emu_lib.emu_compositor_strip_runs(), emu_lib.emu_compositor_rows_runs(), emu_lib.emu_compositor_direct_runs())
    assert tuple(a - b for a, b in zip(after, before)) == (0, 0, direct)      # the strip / rows forms are tuning-build variants
    assert (exp == got).all()


def test_aggregate_continuation_chunk_on_host_matches_reference(emu_lib, ref):
    """35 pads = two launches: the first 32 (with SOURCE pads that lower the canvas alpha: the general per-pixel path), then three OVER
    pads that continue on the canvas (bg_kind 2) through k_aggregate_direct's body - which may force alpha only where its pads are"""
    emu_lib.emu_compositor_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    dw, dh, pw, ph, n_pads = 480, 270, 240, 136, 35
    pads_np = [cases.frame_bytes(pw * ph * 4, "random", 7000 + i) for i in range(n_pads)]
    exp = np.zeros(dw * dh * 4, np.uint8)
    ref.compositor_fill(0, "bgra", "BGRA", exp, dw, dh, 0, dh)
    geo = []
    for i in range(n_pads):
        xpos, ypos, alpha = (i % 4) * 80 - 10, (i // 4) * 45 - 5, min(1.0, 0.25 + 0.05 * i)
        mode = 1 if i % 5 else (0 if i % 2 else 2)
        geo.append((xpos, ypos, alpha, mode))
        ref.compositor_blend("blend_bgra", "BGRA", pads_np[i], pw, ph, xpos, ypos, alpha, exp, dw, dh, 0, dh, mode)
    got = np.zeros(dw * dh * 4, np.uint8)
    emu_lib.emu_compositor_direct_runs.restype = C.c_int
    runs = []
    for lo, hi, bg in ((0, 32, 0), (32, 35, 2)):
        p = AggParams()
        p.ashift, p.overlay, p.bg_kind, p.n_pads = 24, 0, bg, hi - lo
        for k in range(lo, hi):
            pd = p.pads[k - lo]
            xpos, ypos, alpha, mode = geo[k]
            pd.data, pd.width, pd.height, pd.stride, pd.xpos, pd.ypos, pd.s_alpha, pd.mode = pads_np[k].ctypes.data, pw, ph, pw * 4, xpos, ypos, int(alpha * 255), mode
        before = emu_lib.emu_compositor_direct_runs()
        emu_lib.emu_compositor_run(C.byref(p), got.ctypes.data, dw * 4, 0, 0, dw, dh)
        runs.append(emu_lib.emu_compositor_direct_runs() - before)
    assert runs == [0, 1]
    assert (got == exp).all()


@pytest.mark.parametrize("kind", ["strip", "strip8", "rows"])
@pytest.mark.parametrize("rows,n_pads,dw", [(1, 5, 300), (7, 30, 530), (16, 9, 257), (2, 32, 1030)])
def test_aggregate_rows_per_wave_list_on_host_matches_reference(emu_lib, ref, rows, n_pads, dw, kind, monkeypatch):
    """k_aggregate_rows (a measured variant kept in tuning builds, compositor_kernels.hip) - its entry list: more x-hits than one pass holds (30 pads x 7 rows > AGG_LIST_MAX -> several passes), rows no
    pad touches (skip entries), several strips of 256 columns with a ragged last lane, rows per wave not dividing the height."""
    emu_lib.emu_compositor_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    # measured variants kept in tuning builds: "strip" k_aggregate_strip (pad walk on the scalar unit), "strip8" the same with 8
    # pixels per lane, "rows" k_aggregate_rows (entry list in LDS)
    monkeypatch.setenv("EMU_AGG_ROWS" if kind == "rows" else "EMU_AGG_STRIP_ROWS", str(rows))
    if kind == "strip8":
        monkeypatch.setenv("EMU_AGG_STRIP_PX", "8")            # eight pixels per lane, 512 columns per wave
    fmt, dh, pw, ph = "BGRA", 61, 170, 23
    pads_np = [cases.frame_bytes(pw * ph * 4, "random", 7300 + i) for i in range(n_pads)]
    geo = [((i * 37) % (dw - 20) - 9, (i * 5) % 30 + (8 if i % 2 else -6), min(1.0, 0.2 + 0.03 * i), 1) for i in range(n_pads)]
    exp = _aggregate_expected(ref, fmt, 0, geo, pads_np, pw, ph, dw, dh)
    p = AggParams()
    p.ashift, p.overlay, p.bg_kind, p.checker_yuv = 24, 0, 0, 0
    for k, (xpos, ypos, alpha, mode) in enumerate(geo):
        pd = p.pads[k]
        pd.data, pd.width, pd.height, pd.stride, pd.xpos, pd.ypos, pd.s_alpha, pd.mode = pads_np[k].ctypes.data, pw, ph, pw * 4, xpos, ypos, int(alpha * 255), mode
    p.n_pads = n_pads
    got = np.zeros(dw * dh * 4, np.uint8)
    emu_lib.emu_compositor_strip_runs.restype = emu_lib.emu_compositor_rows_runs.restype = C.c_int
    before = (emu_lib.emu_compositor_strip_runs(), emu_lib.emu_compositor_rows_runs())
    emu_lib.emu_compositor_run(C.byref(p), got.ctypes.data, dw * 4, 0, 0, dw, dh)
    after = (emu_lib.emu_compositor_strip_runs(), emu_lib.emu_compositor_rows_runs())
    assert (after[0] - before[0], after[1] - before[1]) == ((0, 1) if kind == "rows" else (1, 0))
    assert (exp == got).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dw,n_pads", [(203, 12), (204, 32), (517, 29), (1300, 40)])
@pytest.mark.parametrize("fmt,background", [("BGRA", 0), ("ARGB", 1), ("AYUV", 0), ("RGBA", 2), ("VUYA", 0), ("VUYA", 1)])
def test_hip_aggregate_packed_path_matches_reference(native_lib, gpu, ref, fmt, background, dw, n_pads):
    """k_aggregate_direct: several strips per row, more hits than request slots (29 piled pads), more pads than one launch takes
    (40: a continuation chunk on the canvas), a last lane moved back (203)"""
    import torch
    dh, pw, ph = 97, 90, 41
    pads_np = [cases.frame_bytes(pw * ph * 4, "random", 7100 + i) for i in range(n_pads)]
    geo = _over_geometry(n_pads)
    if n_pads == 29:
        geo = [(3 * i - 5, i - 3, min(1.0, 0.25 + 0.02 * i), 1 if i % 3 else 2) for i in range(n_pads)]
    if n_pads == 40:
        geo = [((i % 10) * 120 - 11, (i // 10) * 19 - 7, min(1.0, 0.25 + 0.015 * i), 1 if i % 3 else 2) for i in range(n_pads)]
    exp = _aggregate_expected(ref, fmt, background, geo, pads_np, pw, ph, dw, dh)
    d_pads = [torch.from_numpy(p).to(gpu) for p in pads_np]
    arr = (V.CompositorPad * n_pads)()
    for i, (xpos, ypos, alpha, mode) in enumerate(geo):
        arr[i].data, arr[i].width, arr[i].height, arr[i].stride = d_pads[i].data_ptr(), pw, ph, pw * 4
        arr[i].xpos, arr[i].ypos, arr[i].alpha, arr[i].blend_mode = xpos, ypos, alpha, mode
    d_out = torch.zeros(dw * dh * 4, dtype=torch.uint8, device=gpu)
    assert V.lib().gstamd_compositor_aggregate(V.FORMATS[fmt], background, arr, n_pads, d_out.data_ptr(), dw, dh, dw * 4, None) == 0
    torch.cuda.synchronize()
    assert (d_out.cpu().numpy() == exp).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dw,n_pads", [(800, 24), (264, 32), (1366, 30), (40, 12)])
@pytest.mark.parametrize("fmt,background", [("BGRA", 0), ("RGBA", 2)])
def test_hip_aggregate_high_index_pads_in_a_short_last_strip(native_lib, gpu, ref, fmt, background, dw, n_pads):
    """k_aggregate_direct tests pad k on lane k: in a short last strip (800 = 3 x 256 + 32: 8 lanes' worth of pixels) every lane still has to
    take part in the hit test, or the pads with an index past the strip's lanes are never blended there (round-3 advisor finding)"""
    import torch
    dh, pw, ph = 61, 70, 33
    pads_np = [cases.frame_bytes(pw * ph * 4, "random", 7300 + i) for i in range(n_pads)]
    geo = [(dw - 50 + (i % 5) * 6 - 9, (i // 5) * 9 - 4, min(1.0, 0.3 + 0.02 * i), 1 if i % 3 else 2) for i in range(n_pads)]
    exp = _aggregate_expected(ref, fmt, background, geo, pads_np, pw, ph, dw, dh)
    d_pads = [torch.from_numpy(p).to(gpu) for p in pads_np]
    arr = (V.CompositorPad * n_pads)()
    for i, (xpos, ypos, alpha, mode) in enumerate(geo):
        arr[i].data, arr[i].width, arr[i].height, arr[i].stride = d_pads[i].data_ptr(), pw, ph, pw * 4
        arr[i].xpos, arr[i].ypos, arr[i].alpha, arr[i].blend_mode = xpos, ypos, alpha, mode
    d_out = torch.zeros(dw * dh * 4, dtype=torch.uint8, device=gpu)
    assert V.lib().gstamd_compositor_aggregate(V.FORMATS[fmt], background, arr, n_pads, d_out.data_ptr(), dw, dh, dw * 4, None) == 0
    torch.cuda.synchronize()
    assert (d_out.cpu().numpy() == exp).all()


@pytest.mark.gpu
@pytest.mark.parametrize("background", [0, 1, 2, 3])
def test_hip_aggregate_equals_reference_pad_by_pad(native_lib, gpu, ref, background):
    """Fused aggregate (one pass) == the reference's _draw_background + blend_pads loop
    (compositor.c:1619-1697): C4 layout scaled down (16 overlapping pads, alpha 0.25+0.05 i), 35 pads too
    to cross the 32-pad launch chunk."""
    import torch
    dw, dh, pw, ph = 480, 270, 240, 136
    for n_pads in (16, 35):
        pads_np = [cases.frame_bytes(pw * ph * 4, "random", 7000 + i) for i in range(n_pads)]
        exp = np.zeros(dw * dh * 4, np.uint8)
        if background == 0:
            ref.compositor_fill(0, "bgra", "BGRA", exp, dw, dh, 0, dh)
        elif background == 1:
            ref.compositor_fill(1, "bgra", "BGRA", exp, dw, dh, 0, dh, 0, 0, 0)
        elif background == 2:
            ref.compositor_fill(1, "bgra", "BGRA", exp, dw, dh, 0, dh, 255, 255, 255)
        func = "overlay_bgra" if background == 3 else "blend_bgra"
        geo = []
        for i in range(n_pads):
            xpos, ypos, alpha = (i % 4) * 80 - 10, (i // 4) * 45 - 5, min(1.0, 0.25 + 0.05 * i)
            mode = 1 if i % 5 else (0 if i % 2 else 2)
            geo.append((xpos, ypos, alpha, mode))
            ref.compositor_blend(func, "BGRA", pads_np[i], pw, ph, xpos, ypos, alpha, exp, dw, dh, 0, dh, mode)
        d_pads = [torch.from_numpy(p).to(gpu) for p in pads_np]
        arr = (V.CompositorPad * n_pads)()
        for i, (xpos, ypos, alpha, mode) in enumerate(geo):
            arr[i].data, arr[i].width, arr[i].height, arr[i].stride = d_pads[i].data_ptr(), pw, ph, pw * 4
            arr[i].xpos, arr[i].ypos, arr[i].alpha, arr[i].blend_mode = xpos, ypos, alpha, mode
        d_out = torch.zeros(dw * dh * 4, dtype=torch.uint8, device=gpu)
        assert V.lib().gstamd_compositor_aggregate(V.FORMATS["BGRA"], background, arr, n_pads, d_out.data_ptr(), dw, dh, dw * 4, None) == 0
        torch.cuda.synchronize()
        assert (d_out.cpu().numpy() == exp).all(), (background, n_pads)


@pytest.mark.gpu
def test_hip_aggregate_c4_full_size_properties(native_lib, gpu):
    """BASELINE config 4 at full size (16 x 1080p onto 4K): fused aggregate == pad-by-pad gstamd_compositor_blend
    on the GPU itself (the latter is pinned to the reference above), output alpha is opaque."""
    import torch
    dw, dh, pw, ph = 3840, 2160, 1920, 1080
    base = torch.from_numpy(cases.frame_bytes(pw * ph * 4, "random", 31)).to(gpu)
    pads = [torch.roll(base, shifts=i * 4099) for i in range(16)]
    arr = (V.CompositorPad * 16)()
    a = torch.zeros(dw * dh * 4, dtype=torch.uint8, device=gpu)
    b = torch.zeros_like(a)
    L = V.lib()
    assert L.gstamd_compositor_fill_checker(V.FORMATS["BGRA"], b.data_ptr(), dw, dh, dw * 4, 0, dh, None) == 0
    for i in range(16):
        xpos, ypos, alpha = (i % 4) * 640, (i // 4) * 360, 0.25 + 0.05 * i
        arr[i].data, arr[i].width, arr[i].height, arr[i].stride = pads[i].data_ptr(), pw, ph, pw * 4
        arr[i].xpos, arr[i].ypos, arr[i].alpha, arr[i].blend_mode = xpos, ypos, alpha, 1
        assert L.gstamd_compositor_blend(V.FORMATS["BGRA"], 0, pads[i].data_ptr(), pw, ph, pw * 4, xpos, ypos, alpha,
                                         b.data_ptr(), dw, dh, dw * 4, 0, dh, 1, None) == 0
    assert L.gstamd_compositor_aggregate(V.FORMATS["BGRA"], 0, arr, 16, a.data_ptr(), dw, dh, dw * 4, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert bool((a.view(-1, 4)[:, 3] == 255).all())


@pytest.mark.gpu
def test_hip_aggregate_c4_full_size_matches_reference(native_lib, gpu, ref):
    """BASELINE config 4 at its own size (16 x 1080p BGRA pads onto a 4K checker canvas, SURVEY 8d's layout), fused aggregate
    memcmp'd against the REFERENCE: fill_checker + 16 x blend_bgra of compositor/blend.c run on this host (compositor.c:1678-1697)."""
    import torch
    dw, dh, pw, ph = 3840, 2160, 1920, 1080
    base_np = cases.frame_bytes(pw * ph * 4, "random", 31)
    pads_np = [np.roll(base_np, i * 4099) for i in range(16)]
    exp = np.zeros(dw * dh * 4, np.uint8)
    ref.compositor_fill(0, "bgra", "BGRA", exp, dw, dh, 0, dh)
    pads = [torch.from_numpy(p).to(gpu) for p in pads_np]
    arr = (V.CompositorPad * 16)()
    for i in range(16):
        xpos, ypos, alpha = (i % 4) * 640, (i // 4) * 360, 0.25 + 0.05 * i
        ref.compositor_blend("blend_bgra", "BGRA", pads_np[i], pw, ph, xpos, ypos, alpha, exp, dw, dh, 0, dh, 1)
        arr[i].data, arr[i].width, arr[i].height, arr[i].stride = pads[i].data_ptr(), pw, ph, pw * 4
        arr[i].xpos, arr[i].ypos, arr[i].alpha, arr[i].blend_mode = xpos, ypos, alpha, 1
    a = torch.zeros(dw * dh * 4, dtype=torch.uint8, device=gpu)
    assert V.lib().gstamd_compositor_aggregate(V.FORMATS["BGRA"], 0, arr, 16, a.data_ptr(), dw, dh, dw * 4, None) == 0
    torch.cuda.synchronize()
    out = a.cpu().numpy()
    assert (out == exp).all(), int((out != exp).sum())


def test_div255_identity_used_by_packed_blend():
    """compositor_device.h pk16_div255: (x*0x8081)>>23 == (x+1+((x+1)>>8))>>8 for every reachable x (<= 255*255)."""
    x = np.arange(0, 255 * 255 + 1, dtype=np.int64)
    assert np.array_equal((x * 0x8081) >> 23, (x + 1 + ((x + 1) >> 8)) >> 8)


# ---- outputs without per-pixel alpha: plane-by-plane aggregation (compositor_planes.h) ---------------------------------------
FRAME_FMTS = ["I420", "YV12", "Y42B", "Y444", "NV12", "NV21", "RGB", "BGR",
              # planar canvases of 10 / 12 / 16 bits (blend.c:609-697: compositor_orc_blend_u10 / u12 / u16, PLANAR_YUV_HIGH_FILL_*)
              "I420_10LE", "I420_12LE", "I422_10LE", "I422_12LE", "Y444_10LE", "Y444_12LE", "Y444_16LE",
              # 32-bit RGB without alpha (RGB_BLEND with bpp 4) and packed 4:2:2 (PACKED_422_BLEND): one plane of bytes, compared WHOLE
              # (the reference's fills write whole macropixels / leave the x byte, its transparent memset stops at 2 * width)
              "xRGB", "xBGR", "RGBx", "BGRx", "YUY2", "UYVY", "YVYU"]
FRAME_RGB = ("RGB", "BGR", "xRGB", "xBGR", "RGBx", "BGRx")
FRAME_WHOLE = ("xRGB", "xBGR", "RGBx", "BGRx", "YUY2", "UYVY", "YVYU")


def _frame_depth_shift(fmt):
    return 2 if "_10" in fmt else 4 if "_12" in fmt else 8 if "_16" in fmt else 0


def _frame_layout(fmt, w, h):
    """(strides, offsets): cases.default_layout for the 8-bit formats, the library's video_info (pinned against the reference's
    gst_video_info_set_format elsewhere) for the deep planar ones"""
    if not _frame_depth_shift(fmt):
        return cases.default_layout(fmt, w, h)
    vi = V.video_info(fmt, w, h)
    return [int(vi.stride[i]) for i in range(3)], [int(vi.offset[i]) for i in range(3)]


def _frame_visible(fmt, w, h, buf):
    if fmt in FRAME_WHOLE:
        return np.asarray(buf)
    if not _frame_depth_shift(fmt):
        strides, offsets = cases.default_layout(fmt, w, h)
        return cases.visible_bytes(fmt, w, h, strides, offsets, buf)
    strides, offsets = _frame_layout(fmt, w, h)
    ws = 1 if ("I420" in fmt or "I422" in fmt) else 0
    hs = 1 if "I420" in fmt else 0
    out = []
    for i in range(3):
        cw, ch = (w, h) if i == 0 else (-((-w) >> ws), -((-h) >> hs))
        plane = buf[offsets[i]:offsets[i] + strides[i] * ch].reshape(ch, strides[i])
        out.append(plane[:, :2 * cw].reshape(-1))
    return np.concatenate(out)
# (width, height, xpos, ypos, alpha, mode) per pad: odd positions (rounded up to even where the format subsamples), negative
# offsets, a pad hanging over the right / bottom edge, opaque, `source`, transparent and tiny-alpha pads
FRAME_PADS = [(64, 48, 5, 7, 0.5, 1), (37, 21, -9, -3, 0.3, 1), (40, 30, 70, 40, 1.0, 1), (33, 17, 20, 10, 0.7, 0),
              (16, 16, 1, 30, 0.0, 1), (21, 9, 50, 3, 0.004, 1), (64, 48, 30, 20, 0.996, 1)]
FDW, FDH = 101, 67


class EmuFramePad(C.Structure):
    _fields_ = [("data", C.c_void_p * 3), ("stride", C.c_int * 3), ("width", C.c_int), ("height", C.c_int), ("xpos", C.c_int),
                ("ypos", C.c_int), ("alpha", C.c_double), ("mode", C.c_int)]


def _frame_size(fmt, w, h):
    return int(V.video_info(fmt, w, h).size)


def _frame_inputs(fmt):
    return [cases.frame_bytes(_frame_size(fmt, w, h), "random", 7000 + k) for k, (w, h, *_r) in enumerate(FRAME_PADS)]


def _frame_expected(ref, fmt, background):
    """_draw_background + the reference's BlendFunction pad by pad (compositor.c:1619-1697)."""
    dst = cases.frame_bytes(_frame_size(fmt, FDW, FDH), "random", 6999)           # stale canvas: every byte must be rewritten
    low = fmt.lower()
    if background == 0:
        ref.compositor_fill(0, low, fmt, dst, FDW, FDH, 0, FDH)
    elif background == 3 and _frame_depth_shift(fmt):
        dst[:] = 0          # (padding is not compared)
    elif background == 3 and fmt in FRAME_WHOLE:
        (stride,), _o = cases.default_layout(fmt, FDW, FDH)
        dst.reshape(FDH, stride)[:, :FDW * (2 if "Y" in fmt else 4)] = 0          # compositor.c:1657: comp width x pixel stride
    elif background == 3:
        strides, offsets = cases.default_layout(fmt, FDW, FDH)
        for i, (rb, rows) in enumerate(cases.visible_planes(fmt, FDW, FDH)):
            plane = dst[offsets[i]:offsets[i] + strides[i] * rows].reshape(rows, strides[i])
            plane[:, :rb] = 0
    else:
        yuv = fmt not in FRAME_RGB
        c = ((16, 128, 128) if yuv else (0, 0, 0)) if background == 1 else ((235, 128, 128) if yuv else (255, 255, 255))
        c = tuple(v << _frame_depth_shift(fmt) for v in c)
        ref.compositor_fill(1, low, fmt, dst, FDW, FDH, 0, FDH, *c)
    func = {"YV12": "blend_i420", "BGR": "blend_rgb", "xBGR": "blend_xrgb", "RGBx": "blend_xrgb", "BGRx": "blend_xrgb", "UYVY": "blend_yuy2",
            "YVYU": "blend_yuy2"}.get(fmt, "blend_" + low)
    for src, (w, h, x, y, alpha, mode) in zip(_frame_inputs(fmt), FRAME_PADS):
        ref.compositor_blend(func, fmt, src, w, h, x, y, alpha, dst, FDW, FDH, 0, FDH, mode)
    return dst


def _visible(fmt, buf):
    return _frame_visible(fmt, FDW, FDH, buf)


@pytest.mark.parametrize("background", [0, 1, 2, 3])
@pytest.mark.parametrize("fmt", FRAME_FMTS)
def test_aggregate_frame_on_host_matches_reference(emu_lib, ref, fmt, background):
    srcs = _frame_inputs(fmt)
    pads = (EmuFramePad * len(FRAME_PADS))()
    for k, (w, h, x, y, alpha, mode) in enumerate(FRAME_PADS):
        strides, offsets = _frame_layout(fmt, w, h)
        for i in range(len(strides)):
            pads[k].data[i] = srcs[k].ctypes.data + offsets[i]
            pads[k].stride[i] = strides[i]
        pads[k].width, pads[k].height, pads[k].xpos, pads[k].ypos, pads[k].alpha, pads[k].mode = w, h, x, y, alpha, mode
    dst = cases.frame_bytes(_frame_size(fmt, FDW, FDH), "random", 6999)
    strides, offsets = _frame_layout(fmt, FDW, FDH)
    dp = (C.c_void_p * 3)(*[dst.ctypes.data + o for o in offsets] + [None] * (3 - len(offsets)))
    ds = (C.c_int * 3)(*strides + [0] * (3 - len(strides)))
    yuv = fmt not in FRAME_RGB
    sh = _frame_depth_shift(fmt)
    black = (C.c_int * 3)(*[v << sh for v in ((16, 128, 128) if yuv else (0, 0, 0))])
    white = (C.c_int * 3)(*[v << sh for v in ((235, 128, 128) if yuv else (255, 255, 255))])
    emu_lib.emu_compositor_aggregate_frame.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                       C.c_int, C.c_int]
    assert emu_lib.emu_compositor_aggregate_frame(V.FORMATS[fmt], background, black, white, pads, len(FRAME_PADS), dp, ds, FDW, FDH) == 0
    exp = _frame_expected(ref, fmt, background)
    assert (_visible(fmt, dst) == _visible(fmt, exp)).all()


@pytest.mark.parametrize("fmt", FRAME_WHOLE)
def test_aggregate_frame_background_colours_land_where_the_reference_puts_them(emu_lib, ref, fmt):
    """three different colour values: MEMSET_XRGB's shifts for xRGB / xBGR (R G 0 B / B G 0 R), U and V of the packed 4:2:2 word"""
    col = (10, 200, 77)
    dst = cases.frame_bytes(_frame_size(fmt, FDW, FDH), "random", 6999)
    exp = dst.copy()
    ref.compositor_fill(1, fmt.lower(), fmt, exp, FDW, FDH, 0, FDH, *col)
    strides, offsets = _frame_layout(fmt, FDW, FDH)
    dp = (C.c_void_p * 3)(dst.ctypes.data, None, None)
    ds = (C.c_int * 3)(strides[0], 0, 0)
    c3 = (C.c_int * 3)(*col)
    emu_lib.emu_compositor_aggregate_frame.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                       C.c_int, C.c_int]
    assert emu_lib.emu_compositor_aggregate_frame(V.FORMATS[fmt], 1, c3, c3, None, 0, dp, ds, FDW, FDH) == 0
    assert (dst == exp).all()


@pytest.mark.gpu
@pytest.mark.parametrize("background", [0, 1, 2, 3])
@pytest.mark.parametrize("fmt", FRAME_FMTS)
def test_hip_aggregate_frame_matches_reference(native_lib, gpu, ref, fmt, background):
    import torch
    srcs = [torch.from_numpy(s).to(gpu) for s in _frame_inputs(fmt)]
    pads = (V.CompositorFramePad * len(FRAME_PADS))()
    for k, (w, h, x, y, alpha, mode) in enumerate(FRAME_PADS):
        strides, offsets = _frame_layout(fmt, w, h)
        for i in range(len(strides)):
            pads[k].data[i] = srcs[k].data_ptr() + offsets[i]
            pads[k].stride[i] = strides[i]
        pads[k].width, pads[k].height, pads[k].xpos, pads[k].ypos, pads[k].alpha, pads[k].blend_mode = w, h, x, y, alpha, mode
    d = torch.from_numpy(cases.frame_bytes(_frame_size(fmt, FDW, FDH), "random", 6999)).to(gpu)
    strides, offsets = _frame_layout(fmt, FDW, FDH)
    dp = (C.c_void_p * 3)(*[d.data_ptr() + o for o in offsets] + [None] * (3 - len(offsets)))
    ds = (C.c_int32 * 3)(*strides + [0] * (3 - len(strides)))
    V._check(V.lib().gstamd_compositor_aggregate_frame(V.FORMATS[fmt], background, None, None, pads, len(FRAME_PADS), dp, ds, FDW, FDH, None))
    torch.cuda.synchronize()
    exp = _frame_expected(ref, fmt, background)
    assert (_visible(fmt, d.cpu().numpy()) == _visible(fmt, exp)).all()


# ---- ARGB64 / AYUV64 canvases (compositor_wide.h) against blend_argb64 / overlay_argb64 / fill_*_argb64 ---------------------------------
class Wide64Params(C.Structure):
    _fields_ = [("overlay", C.c_int), ("bg_kind", C.c_int), ("checker_yuv", C.c_int), ("bg_px", C.c_uint64), ("n_pads", C.c_int),
                ("pads", PadDev * 32)]


WIDE_CASES = []
for fmt in ("ARGB64", "AYUV64"):
    for overlay in (0, 1):
        for mode in (0, 1, 2):
            for alpha in (1.0, 0.5, 0.3, 0.00002, 0.0):
                WIDE_CASES.append((fmt, overlay, 37, 21, 64, 48, 5 if overlay else -9, 7 if overlay else -4, alpha, mode, 0, 48))
WIDE_CASES += [("ARGB64", 0, 100, 80, 64, 48, 40, 30, 0.7, 1, 10, 40), ("ARGB64", 1, 100, 80, 64, 48, -20, -30, 0.7, 2, 10, 40),
               ("AYUV64", 0, 10, 10, 64, 48, 70, 30, 0.7, 1, 0, 48), ("AYUV64", 0, 64, 48, 64, 48, 0, 0, 1.0, 0, 0, 60)]


def wide_inputs(case, i):
    fmt, overlay, sw, sh, dw, dh = case[:6]
    src, dst = cases.frame_bytes(sw * sh * 8, "random", 1500 + i), cases.frame_bytes(dw * dh * 8, "random", 1900 + i)
    if i % 4 == 1:      # fully transparent / opaque alphas: the "final alpha 0" branch of the overlay functions
        src.view(np.uint16)[::4][::3] = 0
        dst.view(np.uint16)[::4][::2] = 0
        src.view(np.uint16)[::4][1::3] = 65535
    return src, dst


def ref_wide_blend(ref, case, src, dst):
    fmt, overlay, sw, sh, dw, dh, xpos, ypos, alpha, mode, y0, y1 = case
    return ref.compositor_blend("overlay_argb64" if overlay else "blend_argb64", fmt, src, sw, sh, xpos, ypos, alpha, dst, dw, dh, y0, y1, mode)


def wide_id(c):
    return "%s_%d_%s_m%d" % (c[1][0], c[1][1], c[1][8], c[1][9])


@pytest.mark.parametrize("i_case", list(enumerate(WIDE_CASES)), ids=wide_id)
def test_wide64_body_on_host_matches_reference(emu_lib, ref, i_case):
    i, case = i_case
    fmt, overlay, sw, sh, dw, dh, xpos, ypos, alpha, mode, y0, y1 = case
    emu_lib.emu_compositor_wide64.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    assert C.sizeof(Wide64Params) == emu_lib.emu_sizeof_wide64()
    src, dst = wide_inputs(case, i)
    exp = ref_wide_blend(ref, case, src, dst.copy())
    got = dst.copy()
    sa = max(0, min(65535, int(alpha * 65535)))
    if sa:
        p = Wide64Params()
        p.overlay, p.bg_kind, p.n_pads = overlay, 2, 1
        pd = p.pads[0]
        pd.data, pd.width, pd.height, pd.stride, pd.xpos, pd.ypos, pd.s_alpha, pd.mode = src.ctypes.data, sw, sh, sw * 8, xpos, ypos, sa, mode
        x0, r0, x1, r1 = max(xpos, 0), max(ypos, y0), min(xpos + sw, dw), min(ypos + sh, min(y1, dh))
        if x1 > x0 and r1 > r0:
            emu_lib.emu_compositor_wide64(C.byref(p), got.ctypes.data, dw * 8, x0, r0, x1 - x0, r1 - r0)
    assert (exp == got).all()


def test_wide64_fills_on_host_match_reference(emu_lib, ref):
    emu_lib.emu_compositor_wide64.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    w, h = 70, 33
    for fmt in ("ARGB64", "AYUV64"):
        exp, got = np.zeros(w * h * 8, np.uint8), np.zeros(w * h * 8, np.uint8)
        ref.compositor_fill(0, fmt.lower(), fmt, exp, w, h, 3, 30)
        p = Wide64Params()
        p.bg_kind, p.checker_yuv = 0, int(fmt == "AYUV64")
        emu_lib.emu_compositor_wide64(C.byref(p), got.ctypes.data, w * 8, 0, 3, w, 27)
        assert (exp == got).all()
        exp[:] = 0
        got[:] = 0
        ref.compositor_fill(1, "argb64", fmt, exp, w, h, 2, 31, 4096, 51234, 777)
        p = Wide64Params()
        p.bg_kind, p.bg_px = 1, 0xffff | (4096 << 16) | (51234 << 32) | (777 << 48)
        emu_lib.emu_compositor_wide64(C.byref(p), got.ctypes.data, w * 8, 0, 2, w, 29)
        assert (exp == got).all()


@pytest.mark.gpu
@pytest.mark.parametrize("i_case", list(enumerate(WIDE_CASES)), ids=wide_id)
def test_hip_wide64_blend_matches_reference(native_lib, gpu, ref, i_case):
    import torch
    i, case = i_case
    fmt, overlay, sw, sh, dw, dh, xpos, ypos, alpha, mode, y0, y1 = case
    src, dst = wide_inputs(case, i)
    exp = ref_wide_blend(ref, case, src, dst.copy())
    d_src, d_dst = torch.from_numpy(src).to(gpu), torch.from_numpy(dst).to(gpu)
    r = V.lib().gstamd_compositor_blend(V.FORMATS[fmt], overlay, d_src.data_ptr(), sw, sh, sw * 8, xpos, ypos, alpha,
                                        d_dst.data_ptr(), dw, dh, dw * 8, y0, y1, mode, None)
    assert r == 0
    torch.cuda.synchronize()
    assert (d_dst.cpu().numpy() == exp).all()


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["ARGB64", "AYUV64"])
def test_hip_wide64_fills_match_reference(native_lib, gpu, ref, fmt):
    import torch
    w, h = 70, 33
    exp = np.zeros(w * h * 8, np.uint8)
    d = torch.zeros(w * h * 8, dtype=torch.uint8, device=gpu)
    ref.compositor_fill(0, fmt.lower(), fmt, exp, w, h, 3, 30)
    assert V.lib().gstamd_compositor_fill_checker(V.FORMATS[fmt], d.data_ptr(), w, h, w * 8, 3, 30, None) == 0
    torch.cuda.synchronize()
    assert (d.cpu().numpy() == exp).all()
    exp[:] = 0
    d.zero_()
    ref.compositor_fill(1, "argb64", fmt, exp, w, h, 2, 31, 4096, 51234, 777)
    assert V.lib().gstamd_compositor_fill_color(V.FORMATS[fmt], d.data_ptr(), w, h, w * 8, 2, 31, 4096, 51234, 777, None) == 0
    torch.cuda.synchronize()
    assert (d.cpu().numpy() == exp).all()


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,background", [("ARGB64", 0), ("AYUV64", 0), ("ARGB64", 1), ("AYUV64", 2), ("ARGB64", 3), ("AYUV64", 3)])
def test_hip_wide64_aggregate_matches_reference_loop(native_lib, gpu, ref, fmt, background):
    """the whole blend_pads loop (compositor.c:1641-1697) on a 64-bit canvas: background, then 40 pads in order (two chunks of the pad table)"""
    import torch
    dw, dh, pw, ph, n_pads = 200, 120, 48, 30, 40
    yuv = fmt == "AYUV64"
    geo = [((i % 7) * 29 - 11, (i // 7) * 21 - 7, min(1.0, 0.2 + 0.03 * i), i % 3) for i in range(n_pads)]
    pads_np = [cases.frame_bytes(pw * ph * 8, "random", 3100 + i) for i in range(n_pads)]
    exp = np.zeros(dw * dh * 8, np.uint8)
    if background == 0:
        ref.compositor_fill(0, fmt.lower(), fmt, exp, dw, dh, 0, dh)
    elif background == 1:
        ref.compositor_fill(1, "argb64", fmt, exp, dw, dh, 0, dh, *((4096, 32768, 32768) if yuv else (0, 0, 0)))
    elif background == 2:
        ref.compositor_fill(1, "argb64", fmt, exp, dw, dh, 0, dh, *((60160, 32768, 32768) if yuv else (65535, 65535, 65535)))
    func = "overlay_argb64" if background == 3 else "blend_argb64"
    for i, (xpos, ypos, alpha, mode) in enumerate(geo):
        ref.compositor_blend(func, fmt, pads_np[i], pw, ph, xpos, ypos, alpha, exp, dw, dh, 0, dh, mode)
    d_pads = [torch.from_numpy(p).to(gpu) for p in pads_np]
    arr = (V.CompositorPad * n_pads)()
    for i, (xpos, ypos, alpha, mode) in enumerate(geo):
        arr[i].data, arr[i].width, arr[i].height, arr[i].stride = d_pads[i].data_ptr(), pw, ph, pw * 8
        arr[i].xpos, arr[i].ypos, arr[i].alpha, arr[i].blend_mode = xpos, ypos, alpha, mode
    d = torch.empty(dw * dh * 8, dtype=torch.uint8, device=gpu)
    assert V.lib().gstamd_compositor_aggregate(V.FORMATS[fmt], background, arr, n_pads, d.data_ptr(), dw, dh, dw * 8, None) == 0
    torch.cuda.synchronize()
    assert (d.cpu().numpy() == exp).all()


# ---- pads scaled inside the blend pass (compositor_scaled.h) against the reference's per-pad converter + blend -------------------------
class EmuScaledPad(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_int), ("height", C.c_int), ("stride", C.c_int), ("xpos", C.c_int), ("ypos", C.c_int),
                ("alpha", C.c_double), ("mode", C.c_int), ("out_w", C.c_int), ("out_h", C.c_int), ("method", C.c_int)]


# (w, h, out_w, out_h, method, xpos, ypos, alpha, mode); out_w 0: not scaled
SCALED_LAYOUTS = {
    "cubic_down": [(96, 54, 48, 27, "cubic", 3, 2, 1.0, 1), (96, 54, 48, 27, "cubic", 40, 20, 0.6, 1), (64, 40, 37, 23, "cubic", -9, 30, 0.8, 1)],
    "mixed_methods": [(60, 40, 90, 60, "cubic", 0, 0, 1.0, 1), (60, 40, 31, 40, "lanczos", 50, 10, 0.7, 1), (60, 40, 60, 17, "sinc", 20, 40, 0.9, 2),
                      (60, 40, 0, 0, "cubic", 70, 30, 0.5, 1), (33, 21, 64, 50, "linear", 5, 25, 0.75, 0), (33, 21, 70, 9, "nearest", 30, 5, 0.4, 1)],
    "extreme": [(400, 60, 16, 40, "cubic", 10, 5, 0.9, 1), (24, 200, 60, 14, "linear", 40, 30, 0.8, 1), (16, 12, 110, 70, "lanczos", 3, 4, 0.5, 1)],
    "many": [(40 + (i % 3) * 8, 30, 20 + (i % 5) * 9, 12 + (i % 4) * 7, ("cubic", "linear", "lanczos")[i % 3], (i % 6) * 17 - 5, (i // 6) * 19 - 4,
              min(1.0, 0.3 + 0.04 * i), i % 3) for i in range(20)],
}
METHODS = {"nearest": 0, "linear": 1, "cubic": 2, "sinc": 3, "lanczos": 4}


def scaled_expected(ref, fmt, background, layout, frames, dw, dh):
    fam = FAM[fmt]
    exp = np.zeros(dw * dh * 4, np.uint8)
    yuv = fmt in CHECKER_FN
    if background == 0:
        ref.compositor_fill(0, CHECKER_FN.get(fmt, fam), fmt, exp, dw, dh, 0, dh)
    elif background == 1:
        ref.compositor_fill(1, fmt.lower(), fmt, exp, dw, dh, 0, dh, *((16, 128, 128) if yuv else (0, 0, 0)))
    elif background == 2:
        ref.compositor_fill(1, fmt.lower(), fmt, exp, dw, dh, 0, dh, *((235, 128, 128) if yuv else (255, 255, 255)))
    func = ("overlay_" if background == 3 else "blend_") + fam
    for (w, h, ow, oh, method, xpos, ypos, alpha, mode), frame in zip(layout, frames):
        if ow:
            rc = ref.VideoConverter(fmt, w, h, fmt, ow, oh, config=cases.ref_config_string(ref, dict(resampler_method=method)))
            frame, w, h = rc.frame(frame), ow, oh
        ref.compositor_blend(func, fmt, frame, w, h, xpos, ypos, alpha, exp, dw, dh, 0, dh, mode)
    return exp


def scaled_frames(layout, seed):
    return [cases.frame_bytes(p[0] * p[1] * 4, "random", seed + i) for i, p in enumerate(layout)]


SCALED_CASES = [("BGRA", 0, "cubic_down"), ("BGRA", 1, "extreme"), ("ARGB", 1, "mixed_methods"), ("AYUV", 2, "mixed_methods"), ("RGBA", 3, "mixed_methods"),
                ("BGRA", 3, "many"), ("ABGR", 0, "many")]


@pytest.mark.parametrize("tile_rows", [0, 13])
@pytest.mark.parametrize("fmt,background,name", SCALED_CASES)
def test_scaled_pads_on_host_match_reference(emu_lib, ref, fmt, background, name, tile_rows):
    layout = SCALED_LAYOUTS[name]
    dw, dh = 120, 80
    frames = scaled_frames(layout, 4200)
    exp = scaled_expected(ref, fmt, background, layout, frames, dw, dh)
    arr = (EmuScaledPad * len(layout))()
    for i, (w, h, ow, oh, method, xpos, ypos, alpha, mode) in enumerate(layout):
        arr[i].data, arr[i].width, arr[i].height, arr[i].stride = frames[i].ctypes.data, w, h, w * 4
        arr[i].xpos, arr[i].ypos, arr[i].alpha, arr[i].mode = xpos, ypos, alpha, mode
        arr[i].out_w, arr[i].out_h, arr[i].method = ow, oh, METHODS[method]
    yuv = fmt in CHECKER_FN
    words = []
    for c in ((16, 128, 128) if yuv else (0, 0, 0)), ((235, 128, 128) if yuv else (255, 255, 255)):
        one = np.zeros(4, np.uint8)
        ref.compositor_fill(1, fmt.lower(), fmt, one, 1, 1, 0, 1, *c)
        words.append(int(one.view(np.uint32)[0]))
    got = np.zeros(dw * dh * 4, np.uint8)
    emu_lib.emu_compositor_aggregate_scaled.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                        C.c_uint32, C.c_uint32, C.c_int]
    r = emu_lib.emu_compositor_aggregate_scaled(V.FORMATS[fmt], 0 if FAM[fmt] == "argb" else 24, background, arr, len(layout),
                                                got.ctypes.data, dw, dh, dw * 4, words[0], words[1], tile_rows)
    assert r == 0
    assert (exp == got).all(), int((exp != got).sum())
    assert emu_lib.emu_scaled_tile_stages() > 0           # the LDS form ran (and the per-pixel form for what does not fit: "extreme")


def hip_scaled(gpu, fmt, background, layout, frames, dw, dh):
    import torch
    d_frames = [torch.from_numpy(f).to(gpu) for f in frames]
    arr = (V.CompositorScaledPad * len(layout))()
    convs = []
    for i, (w, h, ow, oh, method, xpos, ypos, alpha, mode) in enumerate(layout):
        arr[i].data, arr[i].width, arr[i].height, arr[i].stride = d_frames[i].data_ptr(), w, h, w * 4
        arr[i].xpos, arr[i].ypos, arr[i].alpha, arr[i].blend_mode = xpos, ypos, alpha, mode
        if ow:
            c = V.VideoConverter(V.video_info(fmt, w, h), V.video_info(fmt, ow, oh), V.converter_config(resampler_method=method))
            assert V.lib().gstamd_compositor_pad_scaler_usable(c._h) == 1
            convs.append(c)
            arr[i].scaler = c._h
    d = torch.empty(dw * dh * 4, dtype=torch.uint8, device=gpu)
    assert V.lib().gstamd_compositor_aggregate_scaled(V.FORMATS[fmt], background, arr, len(layout), d.data_ptr(), dw, dh, dw * 4, None) == 0
    torch.cuda.synchronize()
    return d.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,background,name", SCALED_CASES)
def test_hip_scaled_pads_match_reference(native_lib, gpu, ref, fmt, background, name):
    layout = SCALED_LAYOUTS[name]
    dw, dh = 120, 80
    frames = scaled_frames(layout, 4200)
    exp = scaled_expected(ref, fmt, background, layout, frames, dw, dh)
    got = hip_scaled(gpu, fmt, background, layout, frames, dw, dh)
    assert (exp == got).all(), int((exp != got).sum())


@pytest.mark.gpu
def test_hip_scaled_pads_c4_variant_a_full_size(native_lib, gpu, ref):
    """BASELINE C4 variant A at its own size: 16 x 1080p BGRA pads, each cubic-downscaled to 960 x 540 by the pad's converter, in a
    4 x 4 grid on the 4K canvas with pad alphas - against the reference's converter + blend per pad."""
    dw, dh = 3840, 2160
    layout = [(1920, 1080, 960, 540, "cubic", (i % 4) * 960, (i // 4) * 540, 1.0 if i % 2 else 0.8, 1) for i in range(16)]
    frames = [cases.frame_bytes(1920 * 1080 * 4, "random", 5100 + i) for i in range(16)]
    exp = scaled_expected(ref, "BGRA", 1, layout, frames, dw, dh)
    got = hip_scaled(gpu, "BGRA", 1, layout, frames, dw, dh)
    assert (exp == got).all(), int((exp != got).sum())


@pytest.mark.gpu
def test_hip_pad_scaler_usable_only_for_plain_scalers(native_lib, gpu):
    usable = V.lib().gstamd_compositor_pad_scaler_usable
    assert usable(V.VideoConverter(V.video_info("BGRA", 64, 48), V.video_info("BGRA", 32, 24))._h) == 1
    assert usable(V.VideoConverter(V.video_info("NV12", 64, 48), V.video_info("BGRA", 32, 24))._h) == 0
    assert usable(V.VideoConverter(V.video_info("BGRA", 64, 48), V.video_info("RGBA", 32, 24))._h) == 0
    assert usable(V.VideoConverter(V.video_info("BGRA", 64, 48), V.video_info("BGRA", 64, 48))._h) == 0
    assert usable(V.VideoConverter(V.video_info("BGRA", 64, 48), V.video_info("BGRA", 32, 24), V.converter_config(dest_x=4, dest_width=20))._h) == 0


# ---- the column walk (compositor_walk.h): exact halvings with 8-tap passes, scaled pads that do not overlap each other ----------------------
# (w, h, ow, oh, method, xpos, ypos, alpha, mode)
WALK_LAYOUTS = {
    # a 2 x 2 mosaic that covers the canvas: no filler
    "mosaic": [(128, 96, 64, 48, "cubic", 0, 0, 1.0, 1), (128, 96, 64, 48, "cubic", 64, 0, 0.8, 1), (128, 96, 64, 48, "cubic", 0, 48, 0.5, 1),
               (128, 96, 64, 48, "cubic", 64, 48, 1.0, 1)],
    # strips wider than one wave's 61 outputs, an odd number of rows, gaps the filler has to write
    "wide": [(400, 66, 200, 33, "cubic", 10, 3, 0.9, 1), (300, 70, 150, 35, "cubic", 40, 40, 1.0, 1)],
    # pads hanging over every canvas edge
    "clipped": [(128, 96, 64, 48, "cubic", -20, -10, 1.0, 1), (128, 96, 64, 48, "cubic", 150, 50, 0.7, 1), (256, 64, 128, 32, "cubic", 0, 60, 1.0, 1)],
    # unscaled pads under, between and over the scaled ones (z order = list order)
    "layers": [(200, 80, 0, 0, None, 0, 0, 1.0, 1), (128, 96, 64, 48, "cubic", 8, 4, 0.8, 1), (60, 40, 0, 0, None, 50, 30, 0.6, 1),
               (128, 96, 64, 48, "cubic", 100, 20, 1.0, 1), (30, 70, 0, 0, None, 90, 5, 0.5, 1)],
    # the SOURCE and ADD operators
    "operators": [(128, 96, 64, 48, "cubic", 0, 0, 0.7, 0), (128, 96, 64, 48, "cubic", 70, 10, 0.9, 2), (40, 40, 0, 0, None, 40, 30, 0.5, 0)],
    # a small pad: fewer rows than the ring is deep
    "tiny": [(32, 24, 16, 12, "cubic", 3, 3, 1.0, 1), (24, 32, 12, 16, "cubic", 30, 3, 1.0, 1)],
    # the other 8-tap filters of a halving
    "filters": [(128, 96, 64, 48, "lanczos", 0, 0, 1.0, 1), (128, 96, 64, 48, "sinc", 70, 10, 0.9, 1)],
}
WALK_CASES = [("BGRA", 0, "mosaic"), ("ARGB", 1, "mosaic"), ("BGRA", 1, "wide"), ("AYUV", 0, "wide"), ("RGBA", 2, "clipped"), ("BGRA", 3, "clipped"),
              ("BGRA", 0, "layers"), ("ABGR", 3, "layers"), ("BGRA", 1, "operators"), ("ARGB", 3, "operators"), ("BGRA", 0, "tiny"), ("BGRA", 1, "filters")]


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,background,name", WALK_CASES)
def test_hip_column_walk_of_halved_pads_matches_reference(native_lib, gpu, ref, fmt, background, name, monkeypatch):
    """k_aggregate_walk against the reference's converter + blend per pad, and against k_aggregate_scaled on the same pads"""
    layout = WALK_LAYOUTS[name]
    dw, dh = 200, 100
    frames = scaled_frames(layout, 7300)
    exp = scaled_expected(ref, fmt, background, layout, frames, dw, dh)
    got = hip_scaled(gpu, fmt, background, layout, frames, dw, dh)
    V.lib().gstamd_internal_last_scaled_kernel.restype = C.c_int
    assert V.lib().gstamd_internal_last_scaled_kernel() == 1, "the pad set did not take the column walk"
    assert (exp == got).all(), (int((exp != got).sum()), np.flatnonzero(exp != got)[:8])


@pytest.mark.gpu
def test_hip_column_walk_is_not_taken_by_pad_sets_it_cannot_serve(native_lib, gpu, ref):
    """overlapping scaled pads, other ratios and other filters stay with k_aggregate_scaled - same bytes as the reference either way"""
    V.lib().gstamd_internal_last_scaled_kernel.restype = C.c_int
    for layout in ([(128, 96, 64, 48, "cubic", 0, 0, 1.0, 1), (128, 96, 64, 48, "cubic", 30, 20, 0.5, 1)],          # overlap
                   [(150, 96, 50, 48, "cubic", 0, 0, 1.0, 1)],                                                          # 3 : 1 horizontally
                   [(128, 96, 64, 48, "linear", 0, 0, 1.0, 1)]):
        frames = scaled_frames(layout, 7400)
        exp = scaled_expected(ref, "BGRA", 0, layout, frames, 120, 80)
        got = hip_scaled(gpu, "BGRA", 0, layout, frames, 120, 80)
        assert V.lib().gstamd_internal_last_scaled_kernel() == 2, layout
        assert (exp == got).all(), layout


# ---- opaque culling (gstamd_compositor_aggregate_opaque) ---------------------------------------------------------------------------
CULL_PW, CULL_PH = 330, 41


def _cull_pads(fmt, n_pads, seed):
    """pad frames with known opacity: i % 4 == 0 alpha 255 everywhere (flagged all_opaque), 1: alpha 255 in the columns [64, 256) only, 2: alpha 255 but
    for one pixel of row 7, 3: random alpha.  Returns the frames and what a caller may claim about each: "all", "map", "map", None"""
    ash = 0 if FAM[fmt] == "argb" else 3
    pads, kinds = [], []
    for i in range(n_pads):
        f = cases.frame_bytes(CULL_PW * CULL_PH * 4, "random", seed + i).copy().reshape(CULL_PH, CULL_PW, 4)
        if i % 4 == 0:
            f[:, :, ash] = 255
        elif i % 4 == 1:
            f[:, 64:256, ash] = 255
            f[:, 63, ash] = 254
        elif i % 4 == 2:
            f[:, :, ash] = 255
            f[7, 200, ash] = 3
        pads.append(f.reshape(-1))
        kinds.append(("all", "map", "map", None)[i % 4])
    return pads, kinds


def _cull_geometry(n_pads, dw):
    """pads of 330 x 41 that cover whole 256-pixel strips of some rows, piled on each other; most at pad alpha 1.0"""
    return [((i * 97) % max(1, dw - 200) - 40, (i * 13) % 50 - 6, 0.5 if i % 7 == 3 else 1.0, 1 if i % 5 else 2) for i in range(n_pads)]


def _opacity_words(frame, w, h, ash):
    a = frame.reshape(h, w, 4)[:, :, ash]
    words = np.zeros(h, np.uint64)
    for b in range((w + 63) // 64):
        words |= (a[:, 64 * b:64 * b + 64] == 255).all(axis=1).astype(np.uint64) << np.uint64(b)
    return words


def test_opacity_map_body_on_host(emu_lib):
    emu_lib.emu_compositor_opacity_map.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    for fmt in ("BGRA", "ARGB"):
        pads, _ = _cull_pads(fmt, 4, 8100)
        for f in pads:
            got = np.zeros(CULL_PH, np.uint64)
            emu_lib.emu_compositor_opacity_map(f.ctypes.data, CULL_PW, CULL_PH, CULL_PW * 4, 0 if FAM[fmt] == "argb" else 24, got.ctypes.data)
            assert (got == _opacity_words(f, CULL_PW, CULL_PH, 0 if FAM[fmt] == "argb" else 3)).all()
    assert int(_opacity_words(pads[1], CULL_PW, CULL_PH, 0)[0]) == 0b001110      # columns [64, 256): blocks 1, 2, 3; the last block is the partial one (330 - 320)


@pytest.mark.parametrize("dw,n_pads", [(600, 12), (512, 31), (259, 9)])
@pytest.mark.parametrize("fmt,background", [("BGRA", 0), ("ARGB", 1), ("AYUV", 0), ("RGBA", 2)])
def test_aggregate_opaque_culling_on_host_matches_reference(emu_lib, ref, fmt, background, dw, n_pads):
    """k_aggregate_direct_cull's body: the pads under a strip that an opaque pad covers are left out of the walk, and the canvas is what blend_pads makes
    of ALL the pads (compositor.c:1678-1697).  Opaque everywhere, opaque in some 64-pixel blocks, opaque but for one pixel, pad alpha below 1, a short last
    strip (600 = 2 x 256 + 88; 259: the last lane moved back)"""
    emu_lib.emu_compositor_run_cull.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    emu_lib.emu_compositor_culled.restype = C.c_long
    dh = 97
    pads_np, kinds = _cull_pads(fmt, n_pads, 8200)
    geo = _cull_geometry(n_pads, dw)
    exp = _aggregate_expected(ref, fmt, background, geo, pads_np, CULL_PW, CULL_PH, dw, dh)
    ash = 0 if FAM[fmt] == "argb" else 3
    p = AggParams()
    p.ashift = 0 if FAM[fmt] == "argb" else 24
    p.overlay, p.bg_kind, p.checker_yuv = 0, (0 if background == 0 else 1), int(fmt in CHECKER_FN)
    if background:
        word = np.zeros(4, np.uint8)
        ref.compositor_fill(1, fmt.lower(), fmt, word, 1, 1, 0, 1, *_black_white(fmt, background))
        p.bg_word = int(word.view(np.uint32)[0])
    maps, keep, all_bits = (C.c_void_p * 32)(), [], 0
    for i, (xpos, ypos, alpha, mode) in enumerate(geo):
        pd = p.pads[i]
        pd.data, pd.width, pd.height, pd.stride = pads_np[i].ctypes.data, CULL_PW, CULL_PH, CULL_PW * 4
        pd.xpos, pd.ypos, pd.s_alpha, pd.mode = xpos, ypos, int(alpha * 255), mode
        if kinds[i] == "all":
            all_bits |= 1 << i
        elif kinds[i] == "map":
            keep.append(_opacity_words(pads_np[i], CULL_PW, CULL_PH, ash))
            maps[i] = keep[-1].ctypes.data
    p.n_pads = n_pads
    got = np.zeros(dw * dh * 4, np.uint8)
    before = emu_lib.emu_compositor_culled()
    emu_lib.emu_compositor_run_cull(C.byref(p), maps, all_bits, got.ctypes.data, dw * 4, 0, 0, dw, dh)
    assert emu_lib.emu_compositor_culled() > before          # something was left out
    assert (exp == got).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dw,n_pads", [(600, 12), (512, 31), (259, 9), (800, 40)])
@pytest.mark.parametrize("fmt,background", [("BGRA", 0), ("ARGB", 1), ("AYUV", 0), ("RGBA", 2), ("BGRA", 3)])
def test_hip_aggregate_opaque_matches_reference(native_lib, gpu, ref, fmt, background, dw, n_pads):
    """gstamd_compositor_aggregate_opaque: the maps from gstamd_compositor_pad_opacity_map (k_opacity_map against numpy), the culled canvas against
    blend_pads over all the pads; 40 pads: a continuation chunk on the canvas; background 3 (transparent: the overlay functions) ignores the hints"""
    import torch
    dh = 97
    pads_np, kinds = _cull_pads(fmt, n_pads, 8200)
    geo = _cull_geometry(n_pads, dw)
    exp = _aggregate_expected(ref, fmt, background, geo, pads_np, CULL_PW, CULL_PH, dw, dh)
    ash = 0 if FAM[fmt] == "argb" else 3
    d_pads = [torch.from_numpy(p).to(gpu) for p in pads_np]
    arr = (V.CompositorPad * n_pads)()
    opa = (V.CompositorPadOpacity * n_pads)()
    d_maps = []
    for i, (xpos, ypos, alpha, mode) in enumerate(geo):
        arr[i].data, arr[i].width, arr[i].height, arr[i].stride = d_pads[i].data_ptr(), CULL_PW, CULL_PH, CULL_PW * 4
        arr[i].xpos, arr[i].ypos, arr[i].alpha, arr[i].blend_mode = xpos, ypos, alpha, mode
        if kinds[i] == "all":
            opa[i].all_opaque = 1
        elif kinds[i] == "map":
            m = torch.full((CULL_PH,), -1, dtype=torch.int64, device=gpu)
            V._check(V.lib().gstamd_compositor_pad_opacity_map(V.FORMATS[fmt], d_pads[i].data_ptr(), CULL_PW, CULL_PH, CULL_PW * 4, m.data_ptr(), None))
            torch.cuda.synchronize()
            assert (m.cpu().numpy().view(np.uint64) == _opacity_words(pads_np[i], CULL_PW, CULL_PH, ash)).all()
            d_maps.append(m)
            opa[i].map = m.data_ptr()
    d_out = torch.zeros(dw * dh * 4, dtype=torch.uint8, device=gpu)
    V._check(V.lib().gstamd_compositor_aggregate_opaque(V.FORMATS[fmt], background, arr, opa, n_pads, d_out.data_ptr(), dw, dh, dw * 4, None))
    torch.cuda.synchronize()
    assert (d_out.cpu().numpy() == exp).all()
    d_out2 = torch.zeros(dw * dh * 4, dtype=torch.uint8, device=gpu)
    V._check(V.lib().gstamd_compositor_aggregate_opaque(V.FORMATS[fmt], background, arr, None, n_pads, d_out2.data_ptr(), dw, dh, dw * 4, None))
    torch.cuda.synchronize()
    assert (d_out2.cpu().numpy() == exp).all()


@pytest.mark.gpu
def test_hip_opacity_map_wide_pad_and_refusals(native_lib, gpu):
    """a 4096-pixel pad fills all 64 bits of a row's word; wider pads have no map"""
    import torch
    w, h = 4096, 3
    f = np.full((h, w, 4), 255, np.uint8)
    f[1, 4095, 3] = 0
    f[2, 0, 3] = 0
    d = torch.from_numpy(f.reshape(-1)).to(gpu)
    m = torch.zeros(h, dtype=torch.int64, device=gpu)
    V._check(V.lib().gstamd_compositor_pad_opacity_map(V.FORMATS["BGRA"], d.data_ptr(), w, h, w * 4, m.data_ptr(), None))
    torch.cuda.synchronize()
    got = m.cpu().numpy().view(np.uint64)
    assert [int(x) for x in got] == [2 ** 64 - 1, 2 ** 63 - 1, 2 ** 64 - 2]
    assert V.lib().gstamd_compositor_pad_opacity_map(V.FORMATS["BGRA"], d.data_ptr(), 4097, 1, 4097 * 4, m.data_ptr(), None) != 0
    assert V.lib().gstamd_compositor_pad_opacity_map(V.FORMATS["I420"], d.data_ptr(), 64, 1, 256, m.data_ptr(), None) != 0


def _cull_scene(seed):
    """a random scene for the culling fuzz: 3..20 pads of random sizes (some wider than a 256-pixel strip) and positions, pad alpha mostly 1.0, OVER / ADD, pixel alpha
    255 everywhere / in runs of 64-pixel blocks / random; what a caller may claim about each pad ("all", "map", None)"""
    rng = np.random.default_rng(seed)
    dw, dh = int(rng.integers(260, 900)), int(rng.integers(20, 70))
    scene = []
    for i in range(int(rng.integers(3, 21))):
        pw, ph = int(rng.integers(4, 700)), int(rng.integers(3, 50))
        f = rng.integers(0, 256, (ph, pw, 4), dtype=np.uint8)
        kind = [None, "all", "map", "map"][int(rng.integers(0, 4))]
        scene.append(dict(w=pw, h=ph, x=int(rng.integers(-pw // 2, dw)), y=int(rng.integers(-ph // 2, dh)), alpha=1.0 if rng.random() < 0.75 else float(rng.random()),
                          mode=1 if rng.random() < 0.8 else 2, frame=f, kind=kind, rng_blocks=rng.random((ph, (pw + 63) // 64)) < 0.7))
    return dw, dh, scene


def _cull_scene_alpha(scene, ash):
    for s in scene:
        f = s["frame"]
        if s["kind"] == "all":
            f[:, :, ash] = 255
        elif s["kind"] == "map":
            for b in range((s["w"] + 63) // 64):
                rows = s["rng_blocks"][:, b]
                f[rows, 64 * b:64 * b + 64, ash] = 255
        s["bytes"] = np.ascontiguousarray(f).reshape(-1)


@pytest.mark.parametrize("seed", range(9100, 9140))
def test_aggregate_opaque_culling_fuzz_on_host(emu_lib, seed):
    """random scenes through k_aggregate_direct_cull's body and through the plain kernel's: the same canvas, byte for byte (the plain one is pinned to blend_pads elsewhere)"""
    emu_lib.emu_compositor_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    emu_lib.emu_compositor_run_cull.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    dw, dh, scene = _cull_scene(seed)
    fam_argb = seed % 2 == 0
    ash = 0 if fam_argb else 3
    _cull_scene_alpha(scene, ash)
    p = AggParams()
    p.ashift, p.overlay, p.bg_kind, p.checker_yuv = (0 if fam_argb else 24), 0, 0, 0
    maps, keep, all_bits, k = (C.c_void_p * 32)(), [], 0, 0
    for s in scene:
        sa = max(0, min(255, int(s["alpha"] * 255)))
        if not sa:
            continue
        pd = p.pads[k]
        pd.data, pd.width, pd.height, pd.stride = s["bytes"].ctypes.data, s["w"], s["h"], s["w"] * 4
        pd.xpos, pd.ypos, pd.s_alpha, pd.mode = s["x"], s["y"], sa, s["mode"]
        if sa == 255 and s["kind"] == "all":
            all_bits |= 1 << k
        elif sa == 255 and s["kind"] == "map":
            keep.append(_opacity_words(s["bytes"], s["w"], s["h"], ash))
            maps[k] = keep[-1].ctypes.data
        k += 1
    p.n_pads = k
    plain, culled = np.zeros(dw * dh * 4, np.uint8), np.zeros(dw * dh * 4, np.uint8)
    emu_lib.emu_compositor_run(C.byref(p), plain.ctypes.data, dw * 4, 0, 0, dw, dh)
    emu_lib.emu_compositor_run_cull(C.byref(p), maps, all_bits, culled.ctypes.data, dw * 4, 0, 0, dw, dh)
    assert (plain == culled).all()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(9200, 9260))
def test_hip_aggregate_opaque_fuzz(native_lib, gpu, seed):
    """random scenes on the device: gstamd_compositor_aggregate_opaque with maps from k_opacity_map / all_opaque flags equals gstamd_compositor_aggregate"""
    import torch
    dw, dh, scene = _cull_scene(seed)
    fmt = ["BGRA", "ARGB", "AYUV", "RGBA"][seed % 4]
    ash = 0 if FAM[fmt] == "argb" else 3
    _cull_scene_alpha(scene, ash)
    n = len(scene)
    arr, opa, hold = (V.CompositorPad * n)(), (V.CompositorPadOpacity * n)(), []
    for i, s in enumerate(scene):
        d = torch.from_numpy(s["bytes"]).to(gpu)
        hold.append(d)
        arr[i].data, arr[i].width, arr[i].height, arr[i].stride = d.data_ptr(), s["w"], s["h"], s["w"] * 4
        arr[i].xpos, arr[i].ypos, arr[i].alpha, arr[i].blend_mode = s["x"], s["y"], s["alpha"], s["mode"]
        if s["kind"] == "all":
            opa[i].all_opaque = 1
        elif s["kind"] == "map":
            m = torch.zeros(s["h"], dtype=torch.int64, device=gpu)
            V._check(V.lib().gstamd_compositor_pad_opacity_map(V.FORMATS[fmt], d.data_ptr(), s["w"], s["h"], s["w"] * 4, m.data_ptr(), None))
            hold.append(m)
            opa[i].map = m.data_ptr()
    background = seed % 3
    a = torch.zeros(dw * dh * 4, dtype=torch.uint8, device=gpu)
    b = torch.zeros(dw * dh * 4, dtype=torch.uint8, device=gpu)
    V._check(V.lib().gstamd_compositor_aggregate(V.FORMATS[fmt], background, arr, n, a.data_ptr(), dw, dh, dw * 4, None))
    V._check(V.lib().gstamd_compositor_aggregate_opaque(V.FORMATS[fmt], background, arr, opa, n, b.data_ptr(), dw, dh, dw * 4, None))
    torch.cuda.synchronize()
    assert torch.equal(a, b)
