#!/bin/bash
# round 5: v210 fastpaths on the device + the whole GPU suite + 30 fuzz seeds
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05p
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r05p/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05p/pytest_gpu.log; tail -5 gpurun_out/r05p/pytest_gpu.log
GSTAMD_FUZZ_SEEDS=9401-9430 timeout 600 python -m pytest tests/test_video_fuzz.py -m gpu -q -p no:cacheprovider > gpurun_out/r05p/fuzz_gpu_30_seeds.log 2>&1
tail -3 gpurun_out/r05p/fuzz_gpu_30_seeds.log
python - <<'PY'
import sys, time, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
from gstreamer_amd import video as V
dev = torch.device("cuda:0")
for a, b in (("I420", "v210"), ("v210", "I420"), ("UYVY", "v210"), ("v210", "UYVY")):
    ii, oi = V.video_info(a, 3840, 2160), V.video_info(b, 3840, 2160)
    conv = V.VideoConverter(ii, oi)
    src = torch.randint(0, 255, (16, int(ii.size)), dtype=torch.uint8, device=dev)
    dst = torch.zeros((16, int(oi.size)), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    srcs, dsts = [src[i] for i in range(8)], [dst[i] for i in range(8)]
    for i in range(3): conv.frames(srcs, dsts, st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(40): conv.frames(srcs, dsts, st)
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 320 * 1e6
    alg = conv.algorithmic_bytes()
    print("%s -> %s 4K lists of 8: %.1f us per frame, frac %.3f (%s)" % (a, b, us, alg / (us * 1e-6) / 8e12, conv.describe()))
PY
