#!/bin/bash
# round 5: the big-endian formats on the device: whole GPU suite + 40 fuzz seeds
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05u
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r05u/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05u/pytest_gpu.log; tail -4 gpurun_out/r05u/pytest_gpu.log
GSTAMD_FUZZ_SEEDS=9901-9940 timeout 900 python -m pytest tests/test_video_fuzz.py -m gpu -q -p no:cacheprovider > gpurun_out/r05u/fuzz_gpu_40_seeds.log 2>&1
tail -3 gpurun_out/r05u/fuzz_gpu_40_seeds.log
