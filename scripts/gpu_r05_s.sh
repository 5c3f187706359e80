#!/bin/bash
# round 5: the 16-bit alpha-plane formats on the device + 30 fuzz seeds
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05s
timeout 900 python -m pytest tests/test_video_gpu.py tests/test_plugin_gpu.py -m gpu -q -p no:cacheprovider -k "r5a or round5 or a420 or r5p" > gpurun_out/r05s/pytest_alpha16.log 2>&1
tail -3 gpurun_out/r05s/pytest_alpha16.log
GSTAMD_FUZZ_SEEDS=9801-9830 timeout 900 python -m pytest tests/test_video_fuzz.py -m gpu -q -p no:cacheprovider > gpurun_out/r05s/fuzz_gpu_30_seeds.log 2>&1
tail -3 gpurun_out/r05s/fuzz_gpu_30_seeds.log
