#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_video_gpu.py -m gpu -q -k "enc16 or deepout or hd_ or frame_lists or p010 or 10le" > gpurun_out/r04_enc16_tests.log 2>&1
tail -4 gpurun_out/r04_enc16_tests.log
python scripts/bench_survey.py P010_10LE 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_survey_p010.log; cut -c1-170 gpurun_out/r04_survey_p010.log
