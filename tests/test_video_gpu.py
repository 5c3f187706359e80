"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the reference's golden
hashes (tests/golden, generated from oracle/_ref) and - when loadable on the box - against the
reference itself byte for byte.  Bit-exact is the bar for this integer path."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import cases
from gstreamer_amd import video as V

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "video_golden.json")))


def gpu_convert(gpu, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src, in_stride=None, in_offset=None, src_pad=0):
    import torch
    col, ocol = cases.split_colorimetry(col)
    ii = V.video_info(ifmt, w, h, colorimetry=col, chroma_site=site, stride=in_stride, offset=in_offset)
    oi = V.video_info(ofmt, ow, oh, colorimetry=ocol)
    conv = V.VideoConverter(ii, oi, V.converter_config(**cfg))
    d_src = torch.from_numpy(src).to(gpu)
    d_dst = torch.zeros(int(oi.size), dtype=torch.uint8, device=gpu)
    conv.frame(d_src.data_ptr() + src_pad, d_dst)
    torch.cuda.synchronize()
    out = d_dst.cpu().numpy()
    conv.free()
    return out


@pytest.mark.parametrize("idx_case", list(enumerate(cases.VIDEO_CASES)), ids=lambda c: c[1][0])
def test_hip_matches_reference_golden(native_lib, gpu, idx_case):
    i, (name, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern) = idx_case
    ii = V.video_info(ifmt, w, h)
    src = cases.frame_bytes(int(ii.size), pattern, cases.case_seed(name), w)
    out = gpu_convert(gpu, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert cases.video_digest(name, out) == GOLDEN[name]["sha256"], (name, list(out[:16]), GOLDEN[name]["head"][:16])


@pytest.mark.parametrize("case", cases.VIDEO_DEFINED, ids=lambda c: c[0])
def test_hip_reference_undefined_plans_compute_the_stage_by_stage_result(native_lib, gpu, ref, case):
    """the conversions whose one-step reference output is undefined (cases.VIDEO_DEFINED): the HIP path against the reference run as the
    separate conversions the chain consists of"""
    name, (ifmt, w, h, ofmt, ow, oh, cfg), steps, mask = case
    src, exp, keep = cases.video_defined_expected(ref, case)
    got = gpu_convert(gpu, ifmt, w, h, ofmt, ow, oh, cfg, None, None, src)
    if keep is not None:
        assert (got[keep] == exp[keep]).all()
    else:
        assert (got == exp).all(), int((got != exp).sum())


@pytest.mark.parametrize("pair", [("BGRA", "RGBA"), ("ARGB", "BGRx"), ("RGBx", "xBGR"), ("ABGR", "ARGB"), ("AYUV", "VUYA")])
@pytest.mark.parametrize("size", [(3840, 2160), (1022, 33), (3, 5)])
def test_hip_byte_permutations_match_reference(native_lib, gpu, ref, pair, size):
    """k_swizzle4 (4-byte packed -> 4-byte packed, no colour step) at 4K, with a scalar tail, and narrower than one lane"""
    (a, b), (w, h) = pair, size
    src = cases.frame_bytes(w * h * 4, "random", 98, w)
    got = gpu_convert(gpu, a, w, h, b, w, h, {}, None, None, src)
    exp = ref.VideoConverter(a, w, h, b, w, h).frame(src)
    assert (got == exp).all()


def test_hip_set_config_gives_the_frames_of_a_fresh_converter(native_lib, gpu, ref):
    """gst_video_converter_set_config on a live converter, incl. the composite plans whose sub-conversions have to be re-planned"""
    import torch
    for (ifmt, ofmt, w, h, ow, oh) in (("NV12", "BGRA", 64, 48, 64, 48), ("P010_10LE", "NV12", 64, 48, 32, 24), ("NV12", "BGRA", 64, 48, 100, 70)):
        ii, oi = V.video_info(ifmt, w, h), V.video_info(ofmt, ow, oh)
        src = cases.frame_bytes(int(ii.size), "random", 77, w)
        d_src = torch.from_numpy(src).to(gpu)
        c = V.VideoConverter(ii, oi)
        for new in (dict(gamma_mode="remap"), dict(resampler_method="lanczos"), {}, dict(gamma_mode="remap", resampler_method="nearest")):
            try:
                c.set_config(V.converter_config(**new))
            except V.GstAmdError as e:
                assert e.code == V.ERR_UNSUPPORTED
                continue
            if c.divergence():
                continue
            d_dst = torch.zeros(int(oi.size), dtype=torch.uint8, device=gpu)
            c.frame(d_src, d_dst)
            torch.cuda.synchronize()
            exp = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, config=cases.ref_config_string(ref, new)).frame(src)
            assert (d_dst.cpu().numpy() == exp).all(), (ifmt, ofmt, new)
        c.free()


HALF_CASES = [c for c in enumerate(cases.VIDEO_CASES) if c[1][0].startswith("half_") or c[1][0] in ("nv12_bgra_2to1_bilinear_1280x720", "i420_bgra_bil420_half",
                                                                                                   "nv12_bgra_half_bilinear")]


@pytest.mark.parametrize("idx_case", HALF_CASES, ids=lambda c: c[1][0])
def test_hip_bilinear_half_kernel_matches_golden(native_lib, gpu, idx_case):
    """The exact halvings through k_bilinear420_half: single frames of this size stay with the rows kernel by default (GSTAMD_BIL_HALF_SMALL lifts
    that), in a list they take it - both against the golden vectors"""
    import torch
    i, (name, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern) = idx_case
    src = cases.frame_bytes(int(V.video_info(ifmt, w, h).size), pattern, cases.case_seed(name), w)
    with V.tuning(GSTAMD_BIL_HALF_SMALL=1):
        out = gpu_convert(gpu, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert cases.video_digest(name, out) == GOLDEN[name]["sha256"]
    col_i, col_o = cases.split_colorimetry(col)
    ii, oi = V.video_info(ifmt, w, h, colorimetry=col_i, chroma_site=site), V.video_info(ofmt, ow, oh, colorimetry=col_o)
    conv = V.VideoConverter(ii, oi, V.converter_config(**cfg))
    d_src = torch.from_numpy(src).to(gpu)
    outs = [torch.zeros(int(oi.size), dtype=torch.uint8, device=gpu) for _ in range(3)]
    conv.frames([d_src] * 3, outs)
    torch.cuda.synchronize()
    for o in outs:
        assert cases.video_digest(name, o.cpu().numpy()) == GOLDEN[name]["sha256"]
    conv.free()


H420_GENERAL = [c for c in enumerate(cases.VIDEO_CASES) if "_h420_" in c[1][0] or c[1][0] in ("nv12_bgra_quarter_lanczos", "i420_rgba_1080p_to_270p_lanczos")]


@pytest.mark.parametrize("idx_case", H420_GENERAL, ids=lambda c: c[1][0])
def test_hip_general_hscale420_kernel_matches_golden(native_lib, gpu, idx_case):
    """The 4:2:0 cases again with k_hscale420_reg switched off: k_hscale420_dot4 (pair table, chroma-row cache) serves them."""
    i, (name, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern) = idx_case
    src = cases.frame_bytes(int(V.video_info(ifmt, w, h).size), pattern, cases.case_seed(name), w)
    with V.tuning(GSTAMD_NO_H420_REG=1):
        out = gpu_convert(gpu, ifmt, w, h, ofmt, ow, oh, cfg, col, site, src)
    assert cases.video_digest(name, out) == GOLDEN[name]["sha256"]


@pytest.mark.parametrize("size", [(3840, 2160), (1920, 1080), (1918, 1078), (4095, 31)])
def test_hip_matches_reference_bytewise_c2(native_lib, gpu, ref, size):
    """BASELINE config 2 (and neighbours) memcmp'd against the reference run on this host."""
    w, h = size
    src = cases.frame_bytes(ref.video_info("NV12", w, h)["size"], "random", 4242 + w)
    exp = ref.VideoConverter("NV12", w, h, "BGRA", w, h).frame(src)
    out = gpu_convert(gpu, "NV12", w, h, "BGRA", w, h, {}, None, None, src)
    assert out.size == exp.size and (out == exp).all(), int((out != exp).sum())


def test_hip_matches_reference_bytewise_c3_shape(native_lib, gpu, ref):
    """BASELINE config 3 shape at a size the CPU reference finishes quickly: 4:1 Lanczos I420 -> RGBA."""
    w, h, ow, oh = 3840, 2160, 960, 540
    src = cases.frame_bytes(ref.video_info("I420", w, h)["size"], "random", 555)
    exp = ref.VideoConverter("I420", w, h, "RGBA", ow, oh, config=cases.ref_config_string(ref, cases.LAN)).frame(src)
    out = gpu_convert(gpu, "I420", w, h, "RGBA", ow, oh, cases.LAN, None, None, src)
    assert (out == exp).all(), int((out != exp).sum())


def test_hip_matches_reference_bytewise_c3_full_size(native_lib, gpu, ref):
    """BASELINE config 3 at its own size: 7680x4320 I420 -> 1920x1080 RGBA Lanczos, memcmp'd against the reference (n-threads=1)
    run on this host - the benched shape with its own tile counts, lines-per-wave and 8K strides."""
    w, h, ow, oh = 7680, 4320, 1920, 1080
    src = cases.frame_bytes(ref.video_info("I420", w, h)["size"], "random", 7680)
    exp = ref.VideoConverter("I420", w, h, "RGBA", ow, oh, config=cases.ref_config_string(ref, cases.LAN)).frame(src)
    for _ in range(2):
        out = gpu_convert(gpu, "I420", w, h, "RGBA", ow, oh, cases.LAN, None, None, src)
        assert (out == exp).all(), int((out != exp).sum())


def test_hip_matches_reference_bytewise_c5_full_size(native_lib, gpu, ref):
    """BASELINE config 5 (per GPU) at its own size: 7680x4320 NV12 -> 3840x2160 BGRA, bilinear (the element's default method)."""
    w, h, ow, oh = 7680, 4320, 3840, 2160
    src = cases.frame_bytes(ref.video_info("NV12", w, h)["size"], "random", 4320)
    exp = ref.VideoConverter("NV12", w, h, "BGRA", ow, oh, config=cases.ref_config_string(ref, cases.LIN)).frame(src)
    out = gpu_convert(gpu, "NV12", w, h, "BGRA", ow, oh, cases.LIN, None, None, src)
    assert (out == exp).all(), int((out != exp).sum())


@pytest.mark.parametrize("shape", [("P010_10LE", 3840, 2160, "NV12"), ("I420_10LE", 3840, 2160, "I420"), ("P010_10LE", 1280, 2400, "NV21"), ("P010_10LE", 2016, 1208, "NV12"),
                                   ("P010_10LE", 3840, 2160, "BGRA"), ("I420_10LE", 2016, 1208, "RGBA"), ("P010_10LE", 3840, 2160, "P010_10LE"), ("P010_10LE", 2016, 1208, "I420_10LE")],
                         ids=lambda s: "%s_%dx%d_%s" % s)
def test_hip_deep_scale_pack_matches_reference_bytewise(native_lib, gpu, ref, shape):
    """k_deep_scale_pack (video_deep_pack.h): a 10-bit source that halves into an 8-bit planar / semi-planar destination - the decoder-to-encoder
    shape of an HDR transcode at its own size (8 workgroups a line; the cosited downsampler's chroma traded between lanes and, at the workgroups'
    seams, made twice), memcmp'd against the reference; one launch (the converter says so for a frame list)"""
    import torch
    ifmt, w, h, ofmt = shape
    ow, oh = w // 2, h // 2
    src = cases.frame_bytes(ref.video_info(ifmt, w, h)["size"], "random", w + h)
    exp = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, config=cases.ref_config_string(ref, cases.LIN)).frame(src)
    out = gpu_convert(gpu, ifmt, w, h, ofmt, ow, oh, cases.LIN, None, None, src)
    assert out.size == exp.size and (out == exp).all(), int((out != exp).sum())
    ii, oi = V.video_info(ifmt, w, h), V.video_info(ofmt, ow, oh)
    conv = V.VideoConverter(ii, oi, V.converter_config(**cases.LIN))
    srcs = [torch.from_numpy(np.roll(src, i * 4099)).to(gpu) for i in range(3)]
    outs = [torch.zeros(int(oi.size), dtype=torch.uint8, device=gpu) for _ in range(3)]
    conv.frames(srcs, outs)
    torch.cuda.synchronize()
    assert conv.list_launches() == 1
    assert (outs[0].cpu().numpy() == exp).all()
    rc = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, config=cases.ref_config_string(ref, cases.LIN))
    assert (outs[2].cpu().numpy() == rc.frame(np.roll(src, 2 * 4099))).all()
    conv.free()


def test_hip_frame_list_32x4k_matches_reference(native_lib, gpu, ref):
    """The launch bench.py times - 32 frames of 3840x2160 NV12 -> BGRA in ONE gstamd_video_converter_frames call, 32 distinct
    output buffers - against 32 reference frames (sha256 per frame)."""
    import torch
    w, h, n = 3840, 2160, 32
    ii, oi = V.video_info("NV12", w, h), V.video_info("BGRA", w, h)
    base = cases.frame_bytes(int(ii.size), "random", 3232)
    rc = ref.VideoConverter("NV12", w, h, "BGRA", w, h)
    srcs_np = [np.roll(base, i * 4099) for i in range(n)]
    exp = [cases.sha(rc.frame(s_)) for s_ in srcs_np]
    conv = V.VideoConverter(ii, oi)
    srcs = [torch.from_numpy(s_).to(gpu) for s_ in srcs_np]
    outs = [torch.zeros(int(oi.size), dtype=torch.uint8, device=gpu) for _ in range(n)]
    conv.frames(srcs, outs)
    torch.cuda.synchronize()
    got = [cases.sha(o.cpu().numpy()) for o in outs]
    assert got == exp, [i for i in range(n) if got[i] != exp[i]]
    conv.free()


COL_SHAPES = [("I420", 7680, 4320, "RGBA", 1920, 1080), ("NV12", 1280, 720, "BGRA", 320, 180), ("NV21", 1920, 1080, "RGBA", 480, 270),
              ("YV12", 640, 359, "BGRA", 160, 90), ("I420", 2048, 856, "ARGB", 512, 214), ("NV12", 3840, 2160, "BGRA", 1920, 1080),
              ("NV12", 3840, 2160, "BGRA", 1280, 720), ("I420", 1920, 1080, "BGRx", 1280, 720)]
# (outputs per lane, shared windows, waves per workgroup, workgroups down the frame): 0 = the library's own choice
COL_KNOBS = [(0, 1, 0, 0), (1, 1, 8, 0), (1, 1, 1, 0), (2, 1, 4, 0), (2, 0, 2, 3), (2, 1, 8, 1)]


@pytest.mark.parametrize("knobs", COL_KNOBS, ids=lambda k: "opl%d_share%d_waves%d_chunks%d" % k)
@pytest.mark.parametrize("shape", COL_SHAPES, ids=lambda s: "%s_%dx%d_%dx%d" % (s[0], s[1], s[2], s[4], s[5]))
def test_hip_scale_col_matches_reference_bytewise(native_lib, gpu, ref, shape, knobs):
    """k_scale_col (video_scale_col.h) memcmp'd against the reference: C3 at its own size, both plane layouts, odd height, several tiles,
    2:1 / 3:1 / 1.5:1, either number of outputs per lane, shared and private windows, one-wave and eight-wave workgroups (the hand-over
    of line groups between the waves of a workgroup through LDS flags is what only the device can show), one workgroup per column."""
    ifmt, w, h, ofmt, ow, oh = shape
    src = cases.frame_bytes(ref.video_info(ifmt, w, h)["size"], "random", 31 + w)
    exp = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, config=cases.ref_config_string(ref, cases.LAN)).frame(src)
    kw = {}
    if knobs[0]:
        kw["GSTAMD_COL_OPL"] = knobs[0]
    kw["GSTAMD_COL_SHARE"] = knobs[1]
    if knobs[2]:
        kw["GSTAMD_COL_WAVES"] = knobs[2]
    if knobs[3]:
        kw["GSTAMD_COL_CHUNKS"] = knobs[3]
    with V.tuning(**kw):
        out = gpu_convert(gpu, ifmt, w, h, ofmt, ow, oh, cases.LAN, None, None, src)
    assert (out == exp).all(), int((out != exp).sum())


def test_hip_scale_col_frame_list_is_one_grid(native_lib, gpu, ref):
    """a list of frames through k_scale_col (the frames are the grid's third dimension; 18 frames: two launches) == the reference frame by frame"""
    import torch
    ifmt, w, h, ofmt, ow, oh = "NV12", 1920, 1080, "BGRA", 960, 540
    n = 18
    ii, oi = V.video_info(ifmt, w, h), V.video_info(ofmt, ow, oh)
    conv = V.VideoConverter(ii, oi, V.converter_config(**cases.LAN))
    rc = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, config=cases.ref_config_string(ref, cases.LAN))
    srcs, outs, exp = [], [], []
    for i in range(n):
        b = cases.frame_bytes(ii.size, "random", 900 + i)
        exp.append(cases.sha(rc.frame(b)))
        srcs.append(torch.from_numpy(b).to(gpu))
        outs.append(torch.zeros(oi.size, dtype=torch.uint8, device=gpu))
    conv.frames(srcs, outs)
    torch.cuda.synchronize()
    got = [cases.sha(o.cpu().numpy()) for o in outs]
    assert got == exp, [i for i in range(n) if got[i] != exp[i]]
    conv.free()


LIST_PLANS = [      # (id, in, w, h, out, ow, oh, config, list launches expected per chunk: 0 = the plan goes frame by frame)
    ("plane_direct", "NV12", 1920, 1080, "NV12", 1280, 720, cases.LIN, 1),            # (a 540-line output would be bt601: a matrix, not the plane scaler)
    ("plane_tiles", "I420", 1920, 1080, "I420", 1280, 720, cases.LAN, 1),
    ("plane_quad", "I420", 640, 480, "I420", 480, 360, cases.LIN, 1),
    ("pack_422", "YUY2", 1920, 1080, "I420", 1920, 1080, {}, 1),
    ("pack_422_nv12", "UYVY", 1280, 720, "NV12", 1280, 720, {}, -1),
    ("convert_pack", "BGRA", 1280, 720, "I420", 1280, 720, {}, 1),
    ("encode420", "BGRA", 1920, 1080, "NV12", 1920, 1080, {}, 1),
    ("swizzle4", "BGRA", 1920, 1080, "RGBA", 1920, 1080, {}, 1),
    ("swizzle34", "RGB", 1280, 720, "BGRA", 1280, 720, {}, 1),
    ("swizzle43", "BGRx", 1280, 720, "BGR", 1280, 720, {}, 1),
    ("relayout", "I420", 1280, 720, "NV12", 1280, 720, {}, 1),
    ("convert422", "YUY2", 1280, 720, "BGRA", 1280, 720, {}, 1),
    ("convert422_ayuv", "UYVY", 1280, 720, "AYUV", 1280, 720, {}, 1),
    ("convert420p", "I420", 1280, 720, "RGB", 1280, 720, {}, -1),
    ("p010_out", "NV12", 1920, 1080, "P010_10LE", 1920, 1080, {}, 1),
    ("p010_in", "P010_10LE", 1920, 1080, "NV12", 1920, 1080, {}, 1),
    ("i420_10_in", "I420_10LE", 1280, 720, "I420", 1280, 720, {}, 1),
    ("encode16", "BGRA", 1280, 720, "P010_10LE", 1280, 720, {}, 1),
    ("gamma_remap", "NV12", 1920, 1080, "BGRA", 1920, 1080, dict(gamma_mode="remap"), 1),
    ("odd_size", "YUY2", 322, 242, "I420", 322, 242, {}, -1),
    ("two_pass_lanczos", "BGRA", 640, 360, "RGBA", 500, 300, cases.LAN, 0),
    ("dithered", "BGRA", 640, 360, "NV12", 640, 360, dict(dither_quantization=8), -1),               # (ordered dither inside the packer: one kernel)
    ("dithered_floyd", "BGRA", 640, 360, "NV12", 640, 360, dict(dither_quantization=8, dither_method="floyd-steinberg"), 0),
    ("gamma_remap_scaled", "NV12", 640, 360, "BGRA", 480, 270, dict(gamma_mode="remap"), 0),
    ("borders", "BGRA", 640, 360, "RGBA", 640, 360, dict(dest_x=16, dest_y=8, dest_width=600, dest_height=340), 0),
]


@pytest.mark.parametrize("plan", LIST_PLANS, ids=lambda p: p[0])
def test_hip_single_kernel_plans_take_frame_lists(native_lib, gpu, ref, plan):
    """gstamd_video_converter_frames on the single-kernel plans: the list is the grid's third dimension (one launch per chunk of 32
    frames and kernel), every frame memcmp'd against the reference; plans of several kernels go frame by frame with the same bytes.
    35 frames: a chunk of 32 and one of 3; frames of the list lie in scattered allocations (different distances between them)."""
    import torch
    _, ifmt, w, h, ofmt, ow, oh, cfg, expect = plan
    n = 35 if w * h <= 1280 * 720 else 5
    ii, oi = V.video_info(ifmt, w, h), V.video_info(ofmt, ow, oh)
    conv = V.VideoConverter(ii, oi, V.converter_config(**cfg))
    rc = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, config=cases.ref_config_string(ref, cfg))
    srcs, outs, exp, pad = [], [], [], []
    for i in range(n):
        b = cases.frame_bytes(ii.size, "random", 7100 + i)
        exp.append(rc.frame(b))
        pad.append(torch.empty(4096 * (1 + (i * 7) % 5), dtype=torch.uint8, device=gpu))         # uneven gaps between the frames
        srcs.append(torch.from_numpy(b).to(gpu))
        outs.append(torch.zeros(int(oi.size), dtype=torch.uint8, device=gpu))          # (row padding stays as the reference's calloc leaves it)
    order = list(range(n))
    order[1], order[-1] = order[-1], order[1]                                                     # and a list that is not in address order
    conv.frames([srcs[i] for i in order], [outs[i] for i in order])
    torch.cuda.synchronize()
    launches = conv.list_launches()
    bad = [i for i in range(n) if not (outs[i].cpu().numpy()[: len(exp[i])] == exp[i]).all()]
    assert not bad, (bad, launches)
    chunks = (n + 31) // 32
    if expect > 0:
        assert chunks * expect <= launches <= chunks * 3, launches
    elif expect == 0:
        assert launches == 0, launches
    # ... and frame by frame again on the same converter (the list context must not outlive the call)
    conv.frame(srcs[0], outs[1])
    torch.cuda.synchronize()
    assert (outs[1].cpu().numpy()[: len(exp[0])] == exp[0]).all()
    conv.free()


BILR_SHAPES = [("NV12", 3840, 2160, "BGRA", 1920, 1080, None), ("NV21", 7680, 4320, "xRGB", 3840, 2160, "jpeg"), ("YV12", 1040, 362, "RGBA", 520, 181, "none"), ("NV12", 1920, 1080, "RGBA", 1280, 720, None), ("I420", 2048, 858, "ARGB", 1024, 429, "jpeg"),
               ("NV21", 1280, 720, "BGRx", 1000, 562, None), ("YV12", 640, 480, "BGRA", 1280, 960, None), ("NV12", 4096, 2160, "BGRA", 2730, 1440, "mpeg2")]


@pytest.mark.parametrize("shape", BILR_SHAPES, ids=lambda s: "%s_%dx%d_%dx%d" % (s[0], s[1], s[2], s[4], s[5]))
def test_hip_bilinear420_rows_shapes_match_reference_bytewise(native_lib, gpu, ref, shape):
    """k_bilinear420_rows (chroma once per source pixel in byte lanes, balanced row strips): 2:1, 1.5:1, non-integer and upscaling
    ratios, both plane layouts, tiles of 384 and fewer outputs, memcmp'd against the reference run on this host - and the older
    k_bilinear420 on the same frames (GSTAMD_NO_BILINEAR_ROWS), which stays the path for sources the rows kernel does not take."""
    ifmt, w, h, ofmt, ow, oh, site = shape
    src = cases.frame_bytes(ref.video_info(ifmt, w, h)["size"], "random", 4243 + w + oh)
    exp = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, config=cases.ref_config_string(ref, cases.LIN), in_chroma_site=site).frame(src)
    for _ in range(2):
        out = gpu_convert(gpu, ifmt, w, h, ofmt, ow, oh, cases.LIN, None, site, src)
        assert (out == exp).all(), int((out != exp).sum())
    with V.tuning(GSTAMD_NO_BILINEAR_ROWS=1):
        out = gpu_convert(gpu, ifmt, w, h, ofmt, ow, oh, cases.LIN, None, site, src)
    assert (out == exp).all(), int((out != exp).sum())
    if (w, h) == (2 * ow, 2 * oh):
        # exact halvings go through k_bilinear420_half: the rows kernel on the same frames, and the direct stores instead of the trade through LDS
        for knobs in (dict(GSTAMD_NO_BILINEAR_HALF=1), dict(GSTAMD_BIL_HALF_SMALL=1), dict(GSTAMD_BIL_HALF_SMALL=1, GSTAMD_BIL_HALF_STORE=2),
                      dict(GSTAMD_BIL_HALF_SMALL=1, GSTAMD_BIL_HALF_STORE=3), dict(GSTAMD_BIL_HALF_SMALL=1, GSTAMD_BIL_HALF_ROWS=3)):
            with V.tuning(**knobs):
                out = gpu_convert(gpu, ifmt, w, h, ofmt, ow, oh, cases.LIN, None, site, src)
            assert (out == exp).all(), (knobs, int((out != exp).sum()))


@pytest.mark.parametrize("shape", [("NV12", 1920, 1080, "BGRA", 960, 540, 5), ("I420", 1280, 720, "RGBA", 854, 480, 3), ("NV12", 3840, 2160, "BGRA", 1920, 1080, 34)],
                         ids=lambda s: "%s_%dx%d_x%d" % (s[0], s[1], s[2], s[5]))
def test_hip_bilinear_frame_list_is_one_launch_and_matches_reference(native_lib, gpu, ref, shape):
    """gstamd_video_converter_frames on a bilinear 4:2:0 plan: the list goes to the GPU as one grid (k_bilinear420_rows_frames; lists
    longer than 32 frames in chunks) and every frame equals the reference's."""
    import torch
    ifmt, w, h, ofmt, ow, oh, n = shape
    ii, oi = V.video_info(ifmt, w, h), V.video_info(ofmt, ow, oh)
    conv = V.VideoConverter(ii, oi, V.converter_config(**cases.LIN))
    rc = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, config=cases.ref_config_string(ref, cases.LIN))
    srcs = [cases.frame_bytes(int(ii.size), "random", 9100 + 7 * i + w) for i in range(min(n, 4))]
    d_src = [torch.from_numpy(s).to(gpu) for s in srcs]
    d_dst = [torch.zeros(int(oi.size), dtype=torch.uint8, device=gpu) for _ in range(n)]
    conv.frames([d_src[i % len(d_src)] for i in range(n)], d_dst)
    torch.cuda.synchronize()
    exp = [rc.frame(s) for s in srcs]
    for i in range(n):
        out = d_dst[i].cpu().numpy()
        assert (out == exp[i % len(exp)]).all(), (i, int((out != exp[i % len(exp)]).sum()))
    conv.free()


H420_SHAPES = [("NV12", 1280, 720, "BGRA", 320, 180, "lanczos"), ("NV21", 1920, 1080, "RGBA", 480, 270, "lanczos"),
               ("I420", 2048, 856, "ARGB", 512, 214, "lanczos"), ("YV12", 640, 359, "BGRA", 160, 90, "lanczos"),
               ("NV12", 1920, 1080, "BGRA", 640, 360, "lanczos"), ("I420", 1280, 720, "RGBA", 640, 360, "lanczos"),
               ("NV12", 3840, 2160, "BGRA", 1280, 720, "cubic"), ("NV12", 1024, 2050, "xRGB", 300, 700, "sinc")]


@pytest.mark.parametrize("shape", H420_SHAPES, ids=lambda s: "%s_%dx%d_%dx%d_%s" % (s[0], s[1], s[2], s[4], s[5], s[6]))
def test_hip_hscale420_shapes_match_reference_bytewise(native_lib, gpu, ref, shape):
    """k_hscale420_reg over ratios (4:1, 3:1, 2:1, non-integer), window widths (3 / 4 / 5 tap words), both plane layouts, odd heights and
    several tiles per row, memcmp'd against the reference run on this host."""
    ifmt, w, h, ofmt, ow, oh, method = shape
    cfg = dict(resampler_method=method)
    src = cases.frame_bytes(ref.video_info(ifmt, w, h)["size"], "random", 977 + w + h)
    exp = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, config=cases.ref_config_string(ref, cfg)).frame(src)
    for _ in range(3):                       # the same bytes every time (no dependence on timing)
        out = gpu_convert(gpu, ifmt, w, h, ofmt, ow, oh, cfg, None, None, src)
        assert (out == exp).all(), int((out != exp).sum())


def test_unaligned_pitch_and_base_take_the_scalar_path(native_lib, gpu, ref):
    """Pitch/offset the 16-byte fast path cannot use (GstVideoMeta strides must be honoured)."""
    w, h = 322, 240
    stride = [326, 326, 0, 0]
    offset = [0, 326 * 240 + 2, 0, 0]
    size = offset[1] + 326 * 120
    src = cases.frame_bytes(size + 1, "random", 99)
    # reference with the same custom layout
    import ctypes as C
    L = ref.lib()
    st = (C.c_int * 4)(*stride)
    of = (C.c_size_t * 4)(*offset)
    hnd = L.ref_video_converter_new(b"NV12", w, h, None, None, st, of, b"BGRA", w, h, None, None, None, None, None)
    exp = np.zeros(w * h * 4, np.uint8)
    s1 = np.ascontiguousarray(src[1:])
    assert L.ref_video_converter_frame(hnd, s1.ctypes.data, s1.size, exp.ctypes.data, exp.size) == 0
    out = gpu_convert(gpu, "NV12", w, h, "BGRA", w, h, {}, None, None, src, in_stride=stride, in_offset=offset, src_pad=1)
    assert (out == exp).all(), int((out != exp).sum())


def test_full_size_properties_8k(native_lib, gpu):
    """Size-independent properties at BASELINE config 5's 8K input (too big for golden files):
    determinism, row-slice independence (converting the top half alone gives the same rows - the
    invariant the reference's multithreading test pins), alpha is 0xff, and grey stays grey."""
    import torch
    w, h = 7680, 4320
    ii = V.video_info("NV12", w, h)
    oi = V.video_info("BGRA", w, h)
    src = torch.from_numpy(cases.frame_bytes(int(ii.size), "random", 8)).to(gpu)
    conv = V.VideoConverter(ii, oi)
    a = torch.zeros(int(oi.size), dtype=torch.uint8, device=gpu)
    b = torch.zeros_like(a)
    conv.frame(src, a)
    conv.frame(src, b)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert bool((a.view(-1, 4)[:, 3] == 255).all())
    # top half as its own frame: rows 0 .. h/2-2 are identical (the last row pairs with a row that
    # does not exist in the cropped frame)
    hh = h // 2
    ih = V.video_info("NV12", w, hh, colorimetry="bt709", chroma_site="mpeg2")
    top = torch.cat([src[: w * hh], src[w * h: w * h + w * hh // 2]])
    conv2 = V.VideoConverter(ih, V.video_info("BGRA", w, hh))
    c = torch.zeros(w * hh * 4, dtype=torch.uint8, device=gpu)
    conv2.frame(top, c)
    torch.cuda.synchronize()
    assert torch.equal(a[: w * 4 * (hh - 1)], c[: w * 4 * (hh - 1)])
    # neutral chroma + constant luma -> R == G == B everywhere
    grey = torch.full((int(ii.size),), 128, dtype=torch.uint8, device=gpu)
    grey[: w * h] = 90
    conv.frame(grey, a)
    torch.cuda.synchronize()
    px = a.view(-1, 4)
    assert bool((px[:, 0] == px[:, 1]).all()) and bool((px[:, 1] == px[:, 2]).all())
    assert int(px[0, 0]) == 84          # what the reference yields: its AYUV->ARGB kernel ignores the matrix offsets (-128/+128 instead of -16)
    conv.free()
    conv2.free()


def test_frame_list_equals_single_frames(native_lib, gpu):
    """gstamd_video_converter_frames (one launch for a list) == n x gstamd_video_converter_frame."""
    import torch
    w, h, n = 1280, 720, 19          # 19 > the 16-frame launch chunk
    ii, oi = V.video_info("NV12", w, h), V.video_info("BGRA", w, h)
    conv = V.VideoConverter(ii, oi)
    srcs = [torch.from_numpy(cases.frame_bytes(int(ii.size), "random", 300 + i)).to(gpu) for i in range(n)]
    one = [torch.zeros(int(oi.size), dtype=torch.uint8, device=gpu) for _ in range(n)]
    lst = [torch.zeros(int(oi.size), dtype=torch.uint8, device=gpu) for _ in range(n)]
    for s_, d_ in zip(srcs, one):
        conv.frame(s_, d_)
    conv.frames(srcs, lst)
    torch.cuda.synchronize()
    for a, b in zip(one, lst):
        assert torch.equal(a, b)
    # a plan the batch kernel does not cover falls back to per-frame launches with the same results
    oi2 = V.video_info("BGRA", 640, 360)
    conv2 = V.VideoConverter(ii, oi2, V.converter_config(**cases.LIN))
    a = [torch.zeros(int(oi2.size), dtype=torch.uint8, device=gpu) for _ in range(3)]
    b = [torch.zeros(int(oi2.size), dtype=torch.uint8, device=gpu) for _ in range(3)]
    for s_, d_ in zip(srcs[:3], a):
        conv2.frame(s_, d_)
    conv2.frames(srcs[:3], b)
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_frame_without_gpu_library_fails_loudly(native_lib):
    with pytest.raises(V.GstAmdError):
        c = V.VideoConverter(V.video_info("NV12", 64, 64), V.video_info("BGRA", 64, 64))
        c.frame(0, 0)


SCRATCH_PLANS = [   # plans that keep intermediate images between their kernels: one set per stream since round 3
    ("planar_pack", "BGRA", 1280, 720, "YUY2", 1280, 720, {}),
    ("two_pass_lanczos", "BGRA", 1280, 720, "RGBA", 900, 500, dict(resampler_method="lanczos")),
    ("plane_scaler", "I420", 1280, 720, "I420", 640, 360, dict(resampler_method="lanczos")),
    ("chain_scaler_planar_out", "I420", 1280, 720, "I420", 640, 480, dict(resampler_method="lanczos")),
    ("gamma_remap_scaled", "NV12", 1280, 720, "BGRA", 960, 540, dict(gamma_mode="remap")),
    ("gamma_remap_scaled_p010", "NV12", 1280, 720, "P010_10LE", 960, 540, dict(gamma_mode="remap", resampler_method="cubic")),
    ("deep_scaled", "P010_10LE", 1280, 720, "BGRA", 640, 360, dict(resampler_method="cubic")),
    ("floyd_steinberg_planar", "BGRA", 640, 1200, "NV12", 640, 1200, dict(dither_quantization=8, dither_method="floyd-steinberg")),
]


@pytest.mark.parametrize("case", SCRATCH_PLANS, ids=lambda c: c[0])
def test_hip_frames_in_flight_on_several_streams_keep_their_own_scratch(native_lib, gpu, case):
    """gstamd_video_converter_is_reentrant is 1 for every plan: 12 different frames sent round robin to 3 streams with nothing between
    them but the streams' own order - every stream's frames pass through that stream's intermediate images, so each output equals the
    one the same converter gives for that frame alone."""
    import torch
    name, ifmt, w, h, ofmt, ow, oh, cfg = case
    ii, oi = V.video_info(ifmt, w, h), V.video_info(ofmt, ow, oh)
    conv = V.VideoConverter(ii, oi, V.converter_config(**cfg))
    L = V.lib()
    L.gstamd_video_converter_is_reentrant.argtypes = [C.c_void_p]
    L.gstamd_stream_new.restype = C.c_void_p
    L.gstamd_stream_free.argtypes = [C.c_void_p]
    L.gstamd_stream_synchronize.argtypes = [C.c_void_p]
    assert L.gstamd_video_converter_is_reentrant(conv._h) == 1
    n = 12
    srcs = [torch.from_numpy(cases.frame_bytes(int(ii.size), "random", 7000 + i, w)).to(gpu) for i in range(n)]
    alone = []
    for i in range(n):
        d = torch.zeros(int(oi.size), dtype=torch.uint8, device=gpu)
        conv.frame(srcs[i], d)
        torch.cuda.synchronize()
        alone.append(d.cpu().numpy())
    streams = [C.c_void_p(L.gstamd_stream_new()) for _ in range(3)]
    assert all(st.value for st in streams)
    for rep in range(3):
        outs = [torch.zeros(int(oi.size), dtype=torch.uint8, device=gpu) for _ in range(n)]
        torch.cuda.synchronize()
        for i in range(n):
            conv.frame(srcs[i], outs[i], stream=streams[i % 3])
        for st in streams:
            assert L.gstamd_stream_synchronize(st) == 0
        for i in range(n):
            assert (outs[i].cpu().numpy() == alone[i]).all(), (name, rep, i)
    for st in streams:
        L.gstamd_stream_free(st)
    conv.free()


@pytest.mark.parametrize("case", cases.DEEP_NOFILL, ids=lambda c: "%s_%s_w%d_q%d" % (c[0], c[3], c[1], c[6].get("dither_quantization", 1)))
def test_hip_deep_plane_copies_without_border_fill(native_lib, gpu, ref, case):
    """the device side of test_deep_plane_copies_without_border_fill_on_host"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import fuzz_video
    ifmt, w, h, ofmt, ow, oh, cfg = case
    src = cases.frame_bytes(int(V.video_info(ifmt, w, h).size), "random", 4242 + w, w)
    dst = gpu_convert(gpu, ifmt, w, h, ofmt, ow, oh, cfg, None, None, src)
    ok, text = fuzz_video.matches_reference(ref, (ifmt, w, h, ofmt, ow, oh, cfg, None, None), src, dst, V.video_info(ofmt, ow, oh))
    assert ok, text
    assert not dst[:cfg["dest_y"] * V.video_info(ofmt, ow, oh).stride[0]].any()


@pytest.mark.parametrize("pair", [("GBR", "BGRA"), ("BGRA", "GBR"), ("GBR", "GBR"), ("GBR", "I420"), ("NV12", "GBR"), ("GBR", "ARGB64")])
def test_hip_frame_planes_takes_gbr_planes_in_frame_order(native_lib, gpu, ref, pair):
    """gstamd_video_converter_frame_planes gets the planes as GstVideoFrame.data[] holds them - G, B, R for GBR (video-info.c:1030-1041) -
    with pitches of the caller's choosing; inside a plan they are R, G, B.  (Round 4's element converted GBR with rotated channels on its
    single-buffer path: _frame / _frames were right, _frame_planes took the caller's order for the plan's.)"""
    import torch
    ifmt, ofmt = pair
    w, h = 70, 33
    ow, oh = (96, 40) if ifmt == ofmt else (70, 33)
    ii, oi = V.video_info(ifmt, w, h), V.video_info(ofmt, ow, oh)
    src = cases.frame_bytes(int(ii.size), "random", 515, w)
    exp = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh).frame(src)
    conv = V.VideoConverter(ii, oi)
    # every plane in its own allocation with a wider pitch than the default layout
    def scatter(info, fmt, data):
        planes, strides = [], []
        for i in range(info.n_planes):
            rows = (int(info.size) - int(info.offset[i])) // int(info.stride[i]) if i == info.n_planes - 1 else (int(info.offset[i + 1]) - int(info.offset[i])) // int(info.stride[i])
            pitch = int(info.stride[i]) + 64
            t = torch.zeros(rows * pitch, dtype=torch.uint8, device=gpu)
            if data is not None:
                a = torch.from_numpy(data[int(info.offset[i]):int(info.offset[i]) + rows * int(info.stride[i])].reshape(rows, int(info.stride[i]))).to(gpu)
                t.view(rows, pitch)[:, :int(info.stride[i])] = a
            planes.append(t)
            strides.append(pitch)
        return planes, strides
    sp, ss = scatter(ii, ifmt, src)
    dp, ds = scatter(oi, ofmt, None)
    conv.frame_planes(sp, ss, dp, ds)
    torch.cuda.synchronize()
    got = np.zeros(int(oi.size), np.uint8)
    for i in range(oi.n_planes):
        rows = dp[i].numel() // ds[i]
        got[int(oi.offset[i]):int(oi.offset[i]) + rows * int(oi.stride[i])] = dp[i].view(rows, ds[i])[:, :int(oi.stride[i])].cpu().numpy().reshape(-1)
    assert (got == exp).all(), int((got != exp).sum())
    # and the frame-base entry agrees
    d_dst = torch.zeros(int(oi.size), dtype=torch.uint8, device=gpu)
    conv.frame(torch.from_numpy(src).to(gpu), d_dst)
    torch.cuda.synchronize()
    assert (d_dst.cpu().numpy() == exp).all()
    conv.free()
