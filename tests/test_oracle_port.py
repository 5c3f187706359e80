"""Pins the numpy restatement (oracle/port.py) to the reference: against the reference-generated golden hashes
and, where loadable, against oracle/_ref directly.  CPU only."""
import json
import os

import numpy as np
import pytest

import cases
from oracle import port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GV = json.load(open(os.path.join(ROOT, "tests", "golden", "video_golden.json")))
GA = json.load(open(os.path.join(ROOT, "tests", "golden", "audio_golden.json")))
BY_NAME = {c[0]: c for c in cases.VIDEO_CASES}

PORT_VIDEO = ["nv12_bgra_2x2", "nv12_rgba_3x3", "nv12_bgra_322x241", "nv12_bgra_322x241_bt709", "nv12_bgra_640x360_mpeg2",
              "nv12_bgra_1280x720_jpeg", "nv12_bgra_1280x720_bt601", "nv12_bgra_324x242_w4mod8", "nv21_abgr_130x70",
              "nv12_argb_640x360", "nv12_bgra_half_cubic", "nv12_bgra_half_bilinear", "nv12_bgra_quarter_lanczos",
              "i420_rgba_quarter_lanczos", "nv12_bgra_up2_bilinear", "nv12_bgra_up2_cubic", "nv12_bgra_anamorphic_lanczos"]


def test_matrix_params():
    assert port.ayuv_to_argb_params("bt709") == (298, 459, 541, -55, -136)
    assert port.ayuv_to_argb_params("bt601") == (298, 409, 516, -100, -208)


@pytest.mark.parametrize("name", PORT_VIDEO)
def test_port_video_matches_reference_golden(name):
    _, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern = BY_NAME[name]
    s0 = (w + 3) // 4 * 4
    h2 = (h + 1) // 2 * 2
    size = s0 * h2 + (s0 * (h2 // 2) if ifmt.startswith("NV") else 2 * (((w + 1) // 2 * 2 // 2 + 3) // 4 * 4) * (h2 // 2))
    src = cases.frame_bytes(size, pattern, cases.case_seed(name), w)
    assert cases.sha(src) == GV[name]["in_sha256"]
    opt = {}
    method = cfg.get("resampler_method", "cubic")
    if "max_taps" in cfg:
        opt["max_taps"] = cfg["max_taps"]
    cosited = None if site is None else site in ("mpeg2", "cosited")
    out = port.convert_420_to_rgb(src, ifmt, w, h, ofmt, ow, oh, matrix=col, h_cosited=cosited, method=method, **opt)
    assert cases.sha(out) == GV[name]["sha256"], (list(out[:8]), GV[name]["head"][:8])


def test_port_blend_matches_reference(ref):
    for fmt, ab, func in (("BGRA", 3, "blend_bgra"), ("ARGB", 0, "blend_argb")):
        for alpha in (1.0, 0.5, 0.004):
            src = cases.frame_bytes(37 * 21 * 4, "random", 5)
            dst = cases.frame_bytes(64 * 48 * 4, "random", 6)
            exp = ref.compositor_blend(func, fmt, src, 37, 21, -9, 30, alpha, dst.copy(), 64, 48, 0, 48, 1)
            got = port.blend_a32(src, 37, 21, -9, 30, alpha, dst.copy(), 64, 48, ab)
            assert (exp == got).all()


def test_port_audio_matches_reference_golden():
    case = [c for c in cases.AUDIO_CASES if c[0] == "f32_48k_44k1_q4_mono"][0]
    name, fmt, ch, ir, orr, method, quality, bufs = case
    r = port.FloatResampler(ir, orr, ch)
    assert (r.n_taps, r.in_rate, r.out_rate) == (72, 160, 147)
    chunks = []
    for i, n in enumerate(list(bufs) + [None]):
        data = np.zeros((36, ch), np.float32) if n is None else cases.audio_buffer(fmt, ch, n, cases.case_seed(name) + i)
        no = r.get_out_frames(len(data))
        assert no == GA[name]["out_frames"][i]
        chunks.append(r.resample(data, no).reshape(-1))
    assert cases.sha(np.concatenate(chunks)) == GA[name]["sha256"]
