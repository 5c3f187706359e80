#!/bin/bash
# round 5: A420 on the device, element test on both runtimes, 30 fuzz seeds
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05q
timeout 900 python -m pytest tests/test_video_gpu.py tests/test_plugin_gpu.py -m gpu -q -p no:cacheprovider -k "a420 or round5 or refused or planes" > gpurun_out/r05q/pytest_a420.log 2>&1
tail -3 gpurun_out/r05q/pytest_a420.log
GSTAMD_FUZZ_SEEDS=9501-9530 timeout 600 python -m pytest tests/test_video_fuzz.py -m gpu -q -p no:cacheprovider > gpurun_out/r05q/fuzz_gpu_30_seeds_a420.log 2>&1
tail -3 gpurun_out/r05q/fuzz_gpu_30_seeds_a420.log
