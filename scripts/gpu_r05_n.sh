#!/bin/bash
# round 5: frame lists of multi-launch plans fanned out over the converter's internal streams: survey (lists of 8), streams 1/2/3/4, whole GPU suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05n
for k in 1 2 3 4; do echo "== GSTAMD_FAN_STREAMS=$k" | tee -a gpurun_out/r05n/survey_fan.log
  GSTAMD_FAN_STREAMS=$k timeout 300 python scripts/survey_item6.py 2 3 4 5 6 8 9 11 12 14 2>&1 | grep -- "->" | cut -c1-150 | tee -a gpurun_out/r05n/survey_fan.log; done
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r05n/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05n/pytest_gpu.log; tail -4 gpurun_out/r05n/pytest_gpu.log
GSTAMD_FUZZ_SEEDS=9201-9240 timeout 600 python -m pytest tests/test_video_fuzz.py -m gpu -q -p no:cacheprovider > gpurun_out/r05n/fuzz_gpu_40_seeds.log 2>&1
tail -3 gpurun_out/r05n/fuzz_gpu_40_seeds.log
