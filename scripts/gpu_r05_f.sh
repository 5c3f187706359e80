#!/bin/bash
# round 5: the whole GPU suite on the tree with 16-bit error diffusion, register windows, the column walk; then device fuzz seeds with the third generator
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r05_pytest_gpu_f.log 2>&1
tail -5 gpurun_out/r05_pytest_gpu_f.log
GSTAMD_FUZZ_SEEDS=5002-5021 timeout 1500 python -m pytest tests/test_video_fuzz.py -m gpu -q -p no:cacheprovider > gpurun_out/r05_fuzz_gpu_more_20_seeds.log 2>&1
tail -3 gpurun_out/r05_fuzz_gpu_more_20_seeds.log
