#!/bin/bash
# round 5: word-load staging of 4-byte packed and packed 4:2:2 sources in the wave-tile scalers (k_hscale_wave, k_scale2x2_wave): survey + parity
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05m
timeout 300 python scripts/survey_item6.py 4 5 6 7 11 12 13 14 15 2>&1 | grep -- "->" | tee gpurun_out/r05m/survey_after_p4_p422_staging.log
timeout 900 python -m pytest tests/test_video_gpu.py -m gpu -q -p no:cacheprovider -k "scale or yuy2 or uyvy or bgra or lanczos or linear or cubic" > gpurun_out/r05m/pytest_scaled.log 2>&1
tail -3 gpurun_out/r05m/pytest_scaled.log
GSTAMD_FUZZ_SEEDS=9101-9120 timeout 600 python -m pytest tests/test_video_fuzz.py -m gpu -q -p no:cacheprovider > gpurun_out/r05m/fuzz_gpu_20_seeds.log 2>&1
tail -3 gpurun_out/r05m/fuzz_gpu_20_seeds.log
