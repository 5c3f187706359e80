"""Plane-scaler plans (convert_scale_planes): us per frame of k_plane_frame next to the pass-by-pass kernels, a few shapes.
    python scripts/bench_planes.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch                              # noqa: E402
from gstreamer_amd import video as V      # noqa: E402

CASES = [("NV12", 3840, 2160, "NV12", 1920, 1080, dict(resampler_method="linear")),
         ("I420", 3840, 2160, "I420", 1920, 1080, dict(resampler_method="linear")),
         ("NV12", 3840, 2160, "NV12", 1280, 720, dict(resampler_method="linear")),
         ("NV12", 3840, 2160, "NV12", 1920, 1080, dict(resampler_method="lanczos")),
         ("I420", 1920, 1080, "I420", 1280, 720, dict(resampler_method="cubic")),
         ("I420", 1920, 1080, "Y444", 1920, 1080, {}),
         ("NV12", 1920, 1080, "NV12", 3840, 2160, dict(resampler_method="linear"))]
dev = torch.device("cuda:0")
for case in CASES:
    ifmt, w, h, ofmt, ow, oh, cfg = case
    ii, oi = V.video_info(ifmt, w, h), V.video_info(ofmt, ow, oh)
    res = []
    for knob in (None, "GSTAMD_NO_PLANE_FRAME"):
        if knob:
            V.lib().gstamd_tuning_set(knob.encode(), 1)
        conv = V.VideoConverter(ii, oi, V.converter_config(**cfg))
        n_in = 24
        src = torch.randint(0, 255, (n_in, int(ii.size)), dtype=torch.uint8, device=dev)
        dst = torch.zeros((n_in, int(oi.size)), dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        for i in range(20):
            conv.frame(src[i % n_in], dst[i % n_in], st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 400
        for i in range(n):
            conv.frame(src[i % n_in], dst[i % n_in], st)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / n * 1e6)
        desc = conv.describe()
        conv.free()
        if knob:
            V.lib().gstamd_tuning_set(knob.encode(), -1)
    print("%s %dx%d -> %s %dx%d %s | %s | frame kernel %.1f us, passes %.1f us" % (ifmt, w, h, ofmt, ow, oh, cfg, desc, res[0], res[1]))
