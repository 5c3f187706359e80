#!/bin/bash
# round 5: k_v210_fast_vec: parity (goldens with v210, fuzz) and timing at 4K in lists of 8 / single
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05t
timeout 600 python -m pytest tests/test_video_gpu.py -m gpu -q -p no:cacheprovider -k "v210" > gpurun_out/r05t/pytest_v210.log 2>&1
tail -2 gpurun_out/r05t/pytest_v210.log
python - <<'PY' | tee gpurun_out/r05t/v210_fast_vec_timing.log
import sys, time, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
from gstreamer_amd import video as V
dev = torch.device("cuda:0")
for a, b in (("I420", "v210"), ("v210", "I420"), ("UYVY", "v210"), ("v210", "UYVY"), ("Y42B", "v210"), ("v210", "YUY2")):
    ii, oi = V.video_info(a, 3840, 2160), V.video_info(b, 3840, 2160)
    conv = V.VideoConverter(ii, oi)
    src = torch.randint(0, 255, (16, int(ii.size)), dtype=torch.uint8, device=dev)
    dst = torch.zeros((16, int(oi.size)), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for i in range(5): conv.frame(src[i], dst[i], st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(200): conv.frame(src[i % 16], dst[i % 16], st)
    torch.cuda.synchronize(); us1 = (time.perf_counter() - t0) / 200 * 1e6
    srcs, dsts = [src[i] for i in range(8)], [dst[i] for i in range(8)]
    for i in range(3): conv.frames(srcs, dsts, st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(40): conv.frames(srcs, dsts, st)
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 320 * 1e6
    alg = conv.algorithmic_bytes()
    print("%s -> %s 4K: single %.1f us (frac %.3f), lists of 8 %.1f us per frame (frac %.3f)  %s" % (a, b, us1, alg / (us1 * 1e-6) / 8e12, us, alg / (us * 1e-6) / 8e12, conv.describe()))
PY
