#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash scripts/gpu_profiles.sh f8scale f5encode16 > gpurun_out/r04_profiles2.log 2>&1
python scripts/bench_survey.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_bench_survey_end.log
python scripts/bench_survey.py P010_10LE 2>&1 | grep -v amdgpu.ids | tail -8 >> gpurun_out/r04_bench_survey_end.log
cut -c1-150 gpurun_out/r04_bench_survey_end.log
for c in f8scale f5encode16; do grep -h '^{' gpurun_out/prof/bench_$c.json | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('$c', j['value'], j['ms_per_step'], j['roofline']['frac'])
"; done
