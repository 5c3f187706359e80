"""Random draws of the shape k_deep_scale_pack / k_deep_scale4 (video_deep_pack.h) serve - a 10 / 12 / 16-bit planar or semi-planar source that halves
(2-tap both ways) into an 8-bit planar / semi-planar or 4-byte destination, with input chroma sites, source crops, destination rectangles and borders -
against the reference (oracle/_ref): on the host emulator here, on the device under -m gpu.  Most draws take the one-kernel path (counted on the
host); the others (rectangles off the 4-byte grid, unaligned crops) take the multi-launch forms - the same bytes either way."""
import ctypes as C
import os
import random
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cases  # noqa: E402
from gstreamer_amd import video as V  # noqa: E402

SRC = ["P010_10LE", "I420_10LE", "I420_12LE", "P012_LE", "P016_LE", "I422_10LE", "I422_12LE"]
DST = ["NV12", "NV21", "I420", "YV12", "Y42B", "NV16", "NV61", "Y444", "NV24", "BGRA", "RGBA", "ARGB", "xBGR", "AYUV", "VUYA",
       "P010_10LE", "I420_10LE", "P016_LE", "I422_12LE", "I420_12BE", "Y444_10LE"]


def draw(rnd):
    ifmt, ofmt = rnd.choice(SRC), rnd.choice(DST)
    ow, oh = 4 * rnd.randint(2, 80), rnd.choice([rnd.randint(2, 40), rnd.randint(290, 330)])        # (above 576 source lines the defaults change)
    cw, ch = 2 * ow, 2 * oh
    cfg = dict(cases.LIN)
    w, h = cw, ch
    if rnd.random() < 0.4:          # a source crop
        sx, sy = rnd.choice([0, 8, 16, 2, 6]), 2 * rnd.randint(0, 5)
        w, h = cw + sx + 2 * rnd.randint(0, 4), ch + sy + 2 * rnd.randint(0, 4)
        cfg.update(src_x=sx, src_y=sy, src_width=cw, src_height=ch)
    fw, fh = ow, oh
    if rnd.random() < 0.4:          # a destination rectangle with borders
        dx, dy = rnd.choice([0, 4, 8, 12, 2]), 2 * rnd.randint(0, 4)
        fw, fh = ow + dx + 2 * rnd.randint(0, 6), oh + dy + 2 * rnd.randint(0, 4)
        cfg.update(dest_x=dx, dest_y=dy, dest_width=ow, dest_height=oh, border_argb=rnd.getrandbits(32))
    if rnd.random() < 0.2:
        cfg.update(alpha_mode="set", alpha_value=rnd.choice([0.25, 0.5, 1.0]))
    site = rnd.choice([None, None, "jpeg", "mpeg2", "cosited", "dv"])
    return ifmt, w, h, ofmt, fw, fh, cfg, site


def expected(ref, case, src):
    ifmt, w, h, ofmt, fw, fh, cfg, site = case
    return ref.VideoConverter(ifmt, w, h, ofmt, fw, fh, in_chroma_site=site, config=cases.ref_config_string(ref, cfg)).frame(src)


@pytest.mark.parametrize("seed", [11, 23])
def test_deep_pack_draws_on_host_match_reference(native_lib, emu_lib, ref, seed):
    emu_lib.emu_video_convert.argtypes = [C.POINTER(V.VideoInfo), C.POINTER(V.VideoInfo), C.POINTER(V.ConverterConfig), C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_int]
    emu_lib.emu_deep_pack_runs.restype = C.c_int
    rnd = random.Random(seed)
    fused, bad = 0, []
    for it in range(30):
        case = draw(rnd)
        ifmt, w, h, ofmt, fw, fh, cfg, site = case
        ii, oi = V.video_info(ifmt, w, h, chroma_site=site), V.video_info(ofmt, fw, fh)
        src = cases.frame_bytes(int(ii.size), "random", seed * 100 + it, w)
        dst = np.full(int(oi.size), 0x5a, np.uint8)
        want = expected(ref, case, src)
        before = emu_lib.emu_deep_pack_runs()
        c = V.converter_config(**cfg)
        r = emu_lib.emu_video_convert(C.byref(ii), C.byref(oi), C.byref(c), src.ctypes.data, dst.ctypes.data, 1, None, 0)
        fused += emu_lib.emu_deep_pack_runs() - before
        if r != 0:
            continue
        vb = lambda b: cases.visible_bytes(ofmt, fw, fh, list(oi.stride), list(oi.offset), b)          # noqa: E731
        if not (vb(dst) == vb(want)).all():
            bad.append(case)
    assert not bad, bad[:4]
    assert fused >= 15, fused


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [101, 202, 303, 404])
def test_deep_pack_draws_on_the_device_match_reference(native_lib, gpu, ref, seed):
    import torch
    rnd = random.Random(seed)
    bad, n = [], 0
    for it in range(60):
        case = draw(rnd)
        ifmt, w, h, ofmt, fw, fh, cfg, site = case
        ii, oi = V.video_info(ifmt, w, h, chroma_site=site), V.video_info(ofmt, fw, fh)
        src = cases.frame_bytes(int(ii.size), "random", seed * 100 + it, w)
        want = expected(ref, case, src)
        try:
            conv = V.VideoConverter(ii, oi, V.converter_config(**cfg))
        except Exception:
            continue
        d_src = torch.from_numpy(src).to(gpu)
        d_dst = torch.full((int(oi.size),), 0x5a, dtype=torch.uint8, device=gpu)
        conv.frame(d_src, d_dst)
        torch.cuda.synchronize()
        got = d_dst.cpu().numpy()
        conv.free()
        n += 1
        vb = lambda b: cases.visible_bytes(ofmt, fw, fh, list(oi.stride), list(oi.offset), b)          # noqa: E731
        if not (vb(got) == vb(want)).all():
            bad.append(case)
    assert not bad, bad[:4]
    assert n >= 50, n
