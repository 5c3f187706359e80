#!/bin/bash
# k_bilinear4_up on the device: parity of the bilinear / enlargement cases, then the survey rows it serves with its strip heights
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_video_gpu.py -m gpu -q -k "up or bilinear or golden or quad4" > gpurun_out/r04_up_tests.log 2>&1
tail -3 gpurun_out/r04_up_tests.log
{
python scripts/bench_survey.py BGRA NV12 2>&1 | grep "1920x1080 -> BGRA       3840x2160\|3840x2160 -> BGRA       1920x1080 bilinear" | cut -c1-160
for r in 2 4 16 32; do echo "ROWS=$r"; GSTAMD_BIL4_UP_ROWS=$r python scripts/bench_survey.py BGRA NV12 2>&1 | grep "1920x1080 -> BGRA       3840x2160" | cut -c1-120; done
echo "NO_UP"; GSTAMD_NO_BILINEAR4_UP=1 python scripts/bench_survey.py BGRA NV12 2>&1 | grep "1920x1080 -> BGRA       3840x2160" | cut -c1-120
} > gpurun_out/r04_up_survey.log 2>&1
cat gpurun_out/r04_up_survey.log
