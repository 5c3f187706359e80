#!/bin/bash
# round 5: the new formats (RGB10A2_LE / BGR10A2_LE, 64-bit orders and endiannesses, GRAY16) and the byte-read form of k_bilinear420_rows on the device
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05l
timeout 900 python -m pytest tests/test_video_gpu.py tests/test_plugin_gpu.py -m gpu -q -p no:cacheprovider -k "r5f or round5 or bilinear or scale or refused" > gpurun_out/r05l/pytest_new.log 2>&1
tail -4 gpurun_out/r05l/pytest_new.log
timeout 200 python scripts/survey_item6.py 0 1 3 2>&1 | grep -- "->" | tee gpurun_out/r05l/survey_after_bilr_fix.log
GSTAMD_FUZZ_SEEDS=9001-9030 timeout 600 python -m pytest tests/test_video_fuzz.py -m gpu -q -p no:cacheprovider > gpurun_out/r05l/fuzz_gpu_30_seeds_new_formats.log 2>&1
tail -3 gpurun_out/r05l/fuzz_gpu_30_seeds_new_formats.log
