#!/bin/bash
# round 5: RGB16 family on the device, element test, 30 fuzz seeds
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05o
timeout 900 python -m pytest tests/test_video_gpu.py tests/test_plugin_gpu.py -m gpu -q -p no:cacheprovider -k "r5g or round5 or refused or rgb" > gpurun_out/r05o/pytest_rgb16.log 2>&1
tail -3 gpurun_out/r05o/pytest_rgb16.log
GSTAMD_FUZZ_SEEDS=9301-9330 timeout 600 python -m pytest tests/test_video_fuzz.py -m gpu -q -p no:cacheprovider > gpurun_out/r05o/fuzz_gpu_30_seeds_rgb16.log 2>&1
tail -3 gpurun_out/r05o/fuzz_gpu_30_seeds_rgb16.log
