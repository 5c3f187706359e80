#!/usr/bin/env python3
"""k_convert_strip: line pairs per lane (K) against frame size, one frame per launch.  GSTAMD_FAST_VARIANT=0,0,K,0 python scripts/strip_k_probe.py"""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gstreamer_amd import video as V
dev = torch.device("cuda:0")
for (w, h) in ((1280, 720), (1920, 1080), (2560, 1440), (3840, 2160)):
    ii, oi = V.video_info("NV12", w, h), V.video_info("BGRA", w, h)
    conv = V.VideoConverter(ii, oi)
    n_in = 24
    src = torch.randint(0, 255, (n_in, int(ii.size)), dtype=torch.uint8, device=dev)
    dst = torch.zeros((n_in, int(oi.size)), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for i in range(50):
        conv.frame(src[i % n_in], dst[i % n_in], st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 600
    e0.record()
    for i in range(n):
        conv.frame(src[i % n_in], dst[i % n_in], st)
    e1.record()
    torch.cuda.synchronize()
    print("%s %dx%d: %.2f us per frame" % (os.environ.get("GSTAMD_FAST_VARIANT", "shipped"), w, h, e0.elapsed_time(e1) * 1e3 / n), flush=True)
    conv.free()
