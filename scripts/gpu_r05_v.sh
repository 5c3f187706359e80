#!/bin/bash
# round 5: frame lists through per-frame pack images (raw4_pack plans: BGRA -> NV12 / I420 bilinear-scaled): survey + list tests + fuzz
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05v
GSTAMD_LIST_DEBUG=1 timeout 300 python scripts/survey_item6.py 4 7 2>&1 | grep -- "->\|frame list" | tail -6 | cut -c1-260 | tee gpurun_out/r05v/survey_scratch_lists.log
timeout 900 python -m pytest tests/test_video_gpu.py -m gpu -q -p no:cacheprovider -k "list or frames or batch" > gpurun_out/r05v/pytest_lists.log 2>&1
tail -2 gpurun_out/r05v/pytest_lists.log
GSTAMD_FUZZ_SEEDS=101,404,505,707,909,61030,5001,6005,9951-9970 timeout 900 python -m pytest tests/test_video_fuzz.py -m gpu -q -p no:cacheprovider > gpurun_out/r05v/fuzz_gpu_28_seeds.log 2>&1
tail -3 gpurun_out/r05v/fuzz_gpu_28_seeds.log
