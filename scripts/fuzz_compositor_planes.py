"""DRAFT, NOT PART OF THE SUITE: random canvases for the outputs without per-pixel alpha (gstamd_compositor_aggregate_frame) against the
reference's fill + blend loop.  Its first runs on an MI355X (late round 2) produced two findings that are still open:
  * a wrong result after a 2-pixel-wide RGB pad in SOURCE mode (seed 44, draw 1: canvas 102 wide, pad (2, 28, 13, 4, alpha 0.0, source),
    592 bytes differ; the host emulator shows the same, so it is in compositor_planes.h, not in the launch);
  * a process abort on one of the seeds 11..88 with pads of 1..80 pixels (not yet narrowed down: reference blend or k_aggregate_plane).
Run:  GSTAMD_FUZZ_SEEDS=44 python -m pytest scripts/fuzz_compositor_planes.py -m gpu -q -p no:cacheprovider --rootdir tests -c /dev/null"""
import os
import random
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import cases  # noqa: E402
from gstreamer_amd import video as V  # noqa: E402

SEEDS = [int(x) for x in os.environ.get("GSTAMD_FUZZ_SEEDS", "11,22,33,44").split(",")]


PLANE_FMTS = ["I420", "YV12", "Y42B", "Y444", "NV12", "NV21", "RGB", "BGR"]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_hip_random_plane_canvases_match_reference(native_lib, gpu, ref, seed):
    """the same for the outputs without per-pixel alpha (gstamd_compositor_aggregate_frame: one launch per destination plane)"""
    import ctypes as C

    import torch
    rnd = random.Random(seed + 7)
    for it in range(10):
        fmt = rnd.choice(PLANE_FMTS)
        low = fmt.lower()
        dw, dh = rnd.randint(8, 160), rnd.randint(8, 100)
        background = rnd.randint(0, 3)
        n = rnd.choice([1, 3, 9, 20])
        # pads at least 4 pixels wide: a 2-pixel RGB pad in SOURCE mode came out wrong in the first run of this test (seed 44: 592 bytes
        # of a 102-wide canvas after the pad (2, 28, 13, 4, alpha 0.0, source) - open); narrower pads wait for that fix
        pads = [(rnd.randint(4, 80), rnd.randint(1, 50), rnd.randint(-40, dw + 6), rnd.randint(-30, dh + 6),
                 rnd.choice([1.0, 1.0, 0.7, 0.5, 0.3, 0.004, 0.0]), rnd.randint(0, 2)) for _ in range(n)]
        frames = [cases.frame_bytes(int(V.video_info(fmt, w, h).size), "random", seed * 10000 + it * 100 + k) for k, (w, h, *_r) in enumerate(pads)]
        exp = cases.frame_bytes(int(V.video_info(fmt, dw, dh).size), "random", 6999)
        strides, offsets = cases.default_layout(fmt, dw, dh)
        yuv = fmt not in ("RGB", "BGR")
        if background == 0:
            ref.compositor_fill(0, low, fmt, exp, dw, dh, 0, dh)
        elif background == 3:
            for i, (rb, rows) in enumerate(cases.visible_planes(fmt, dw, dh)):
                exp[offsets[i]:offsets[i] + strides[i] * rows].reshape(rows, strides[i])[:, :rb] = 0
        else:
            c = ((16, 128, 128) if yuv else (0, 0, 0)) if background == 1 else ((235, 128, 128) if yuv else (255, 255, 255))
            ref.compositor_fill(1, low, fmt, exp, dw, dh, 0, dh, *c)
        func = {"YV12": "blend_i420", "BGR": "blend_rgb"}.get(fmt, "blend_" + low)
        for src, (w, h, x, y, alpha, mode) in zip(frames, pads):
            ref.compositor_blend(func, fmt, src, w, h, x, y, alpha, exp, dw, dh, 0, dh, mode)
        srcs = [torch.from_numpy(s).to(gpu) for s in frames]
        arr = (V.CompositorFramePad * n)()
        for k, (w, h, x, y, alpha, mode) in enumerate(pads):
            st, of = cases.default_layout(fmt, w, h)
            for i in range(len(st)):
                arr[k].data[i] = srcs[k].data_ptr() + of[i]
                arr[k].stride[i] = st[i]
            arr[k].width, arr[k].height, arr[k].xpos, arr[k].ypos, arr[k].alpha, arr[k].blend_mode = w, h, x, y, alpha, mode
        d = torch.from_numpy(cases.frame_bytes(int(V.video_info(fmt, dw, dh).size), "random", 6999)).to(gpu)
        dp = (C.c_void_p * 3)(*[d.data_ptr() + o for o in offsets] + [None] * (3 - len(offsets)))
        ds = (C.c_int32 * 3)(*strides + [0] * (3 - len(strides)))
        V._check(V.lib().gstamd_compositor_aggregate_frame(V.FORMATS[fmt], background, None, None, arr, n, dp, ds, dw, dh, None))
        torch.cuda.synchronize()
        vis = lambda b: cases.visible_bytes(fmt, dw, dh, strides, offsets, b)
        got = d.cpu().numpy()
        assert (vis(got) == vis(exp)).all(), (seed, it, fmt, background, n, int((vis(got) != vis(exp)).sum()), pads[:3])
