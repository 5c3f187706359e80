"""Random canvases through the HIP compositor against the reference's fill + blend loop (and the per-pad converter for scaled pads):
formats with per-pixel alpha (8 and 16 bits per component), 1..40 pads partly or wholly outside the canvas, every operator and
background, pad alphas down to 0."""
import os
import random

import numpy as np
import pytest

import cases
from gstreamer_amd import video as V

SEEDS = [int(x) for x in os.environ.get("GSTAMD_FUZZ_SEEDS", "11,22,33,44").split(",")]
FAM = {"BGRA": "bgra", "RGBA": "bgra", "ARGB": "argb", "ABGR": "argb", "AYUV": "argb"}
METHODS = ["nearest", "linear", "cubic", "lanczos"]


def expected(ref, fmt, background, pads, frames, dw, dh):
    wide = fmt.endswith("64")
    bpp = 8 if wide else 4
    yuv = fmt.startswith("AYUV")
    exp = np.zeros(dw * dh * bpp, np.uint8)
    sc = 256 if wide else 1
    black = (16 * sc, 128 * sc, 128 * sc) if yuv else (0, 0, 0)
    white = (235 * sc, 128 * sc, 128 * sc) if yuv else ((65535,) * 3 if wide else (255,) * 3)
    color_fn = "argb64" if wide else fmt.lower()
    if background == 0:
        ref.compositor_fill(0, fmt.lower() if (wide or yuv) else FAM[fmt], fmt, exp, dw, dh, 0, dh)
    elif background in (1, 2):
        ref.compositor_fill(1, color_fn, fmt, exp, dw, dh, 0, dh, *(black if background == 1 else white))
    func = ("overlay_" if background == 3 else "blend_") + ("argb64" if wide else FAM[fmt])
    for (w, h, ow, oh, method, x, y, alpha, mode), frame in zip(pads, frames):
        if ow:
            frame = ref.VideoConverter(fmt, w, h, fmt, ow, oh, config=cases.ref_config_string(ref, dict(resampler_method=method))).frame(frame)
            w, h = ow, oh
        ref.compositor_blend(func, fmt, frame, w, h, x, y, alpha, exp, dw, dh, 0, dh, mode)
    return exp


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_hip_random_canvases_match_reference(native_lib, gpu, ref, seed):
    import torch
    rnd = random.Random(seed)
    for it in range(12):
        fmt = rnd.choice(["BGRA", "RGBA", "ARGB", "ABGR", "AYUV", "ARGB64", "AYUV64"])
        wide = fmt.endswith("64")
        bpp = 8 if wide else 4
        dw, dh = rnd.randint(8, 200), rnd.randint(8, 120)
        background = rnd.randint(0, 3)
        n = rnd.choice([1, 2, 5, 17, 40])
        scaled_ok = not wide and rnd.random() < 0.5
        pads = []
        for i in range(n):
            w, h = rnd.randint(1, 90), rnd.randint(1, 60)
            ow = oh = 0
            if scaled_ok and rnd.random() < 0.4:
                ow, oh = rnd.randint(1, 120), rnd.randint(1, 80)
                if (ow, oh) == (w, h):
                    ow = oh = 0
            method = rnd.choice(METHODS)
            x, y = rnd.randint(-60, dw + 10), rnd.randint(-40, dh + 10)
            alpha = rnd.choice([1.0, 1.0, 0.75, 0.5, 0.3, 0.004, 0.0])
            pads.append((w, h, ow, oh, method, x, y, alpha, rnd.randint(0, 2)))
        frames = [cases.frame_bytes(p[0] * p[1] * bpp, "random", seed * 10000 + it * 100 + i) for i, p in enumerate(pads)]
        exp = expected(ref, fmt, background, pads, frames, dw, dh)
        d_frames = [torch.from_numpy(f).to(gpu) for f in frames]
        d = torch.empty(dw * dh * bpp, dtype=torch.uint8, device=gpu)
        convs = []
        if any(p[2] for p in pads):
            arr = (V.CompositorScaledPad * n)()
            for i, (w, h, ow, oh, method, x, y, alpha, mode) in enumerate(pads):
                arr[i].data, arr[i].width, arr[i].height, arr[i].stride = d_frames[i].data_ptr(), w, h, w * bpp
                arr[i].xpos, arr[i].ypos, arr[i].alpha, arr[i].blend_mode = x, y, alpha, mode
                if ow:
                    c = V.VideoConverter(V.video_info(fmt, w, h), V.video_info(fmt, ow, oh), V.converter_config(resampler_method=method))
                    assert V.lib().gstamd_compositor_pad_scaler_usable(c._h) == 1
                    convs.append(c)
                    arr[i].scaler = c._h
            r = V.lib().gstamd_compositor_aggregate_scaled(V.FORMATS[fmt], background, arr, n, d.data_ptr(), dw, dh, dw * bpp, None)
        else:
            arr = (V.CompositorPad * n)()
            for i, (w, h, ow, oh, method, x, y, alpha, mode) in enumerate(pads):
                arr[i].data, arr[i].width, arr[i].height, arr[i].stride = d_frames[i].data_ptr(), w, h, w * bpp
                arr[i].xpos, arr[i].ypos, arr[i].alpha, arr[i].blend_mode = x, y, alpha, mode
            r = V.lib().gstamd_compositor_aggregate(V.FORMATS[fmt], background, arr, n, d.data_ptr(), dw, dh, dw * bpp, None)
        assert r == 0, V.last_error()
        torch.cuda.synchronize()
        got = d.cpu().numpy()
        for c in convs:
            c.free()
        assert (got == exp).all(), (seed, it, fmt, background, n, int((got != exp).sum()), pads[:3])
