#!/bin/bash
# round 5: device fuzz, 1100 fresh seeds x 150 draws: 600 with the third generator (dither methods on 16-bit lines, gamma remap, ...) + rectangles,
# 500 with rectangles only
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05h
GSTAMD_FUZZ_SEEDS=7000-7599 timeout 1500 python -m pytest tests/test_video_fuzz.py -m gpu -q -p no:cacheprovider > gpurun_out/r05h/fuzz_gpu_600_seeds_more_rects.log 2>&1
tail -3 gpurun_out/r05h/fuzz_gpu_600_seeds_more_rects.log
GSTAMD_FUZZ_SEEDS=62000-62499 timeout 1500 python -m pytest tests/test_video_fuzz.py -m gpu -q -p no:cacheprovider > gpurun_out/r05h/fuzz_gpu_500_seeds_rects.log 2>&1
tail -3 gpurun_out/r05h/fuzz_gpu_500_seeds_rects.log
