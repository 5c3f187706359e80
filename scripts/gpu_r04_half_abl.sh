#!/bin/bash
# k_bilinear420_half / k_bilinear4_up with the knobs of the TUNING build (GSTAMD_TUNING_LIB=1: numeric knobs exist only there):
# store forms (1 = halves traded through LDS, 2 = plain direct, 3 = streaming direct), ablations (5 = no passes / matrix, 6 = loads, chroma
# filter and stores only), rows per wave, waves per workgroup
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GSTAMD_TUNING_LIB=1
run() { python bench.py --config $1 --batch $2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 batch $2 STORE/ABL=$GSTAMD_BIL_HALF_STORE ROWS=$GSTAMD_BIL_HALF_ROWS WG=$GSTAMD_BIL_WG:', j['value'], j['roofline']['frac'], j['roofline'].get('avg_launch_us'))
"; }
{
run c5 16; run c5 1
for m in 2 3 5 6; do GSTAMD_BIL_HALF_STORE=$m run c5 16; done
for r in 4 16 32; do GSTAMD_BIL_HALF_ROWS=$r run c5 16; done
for r in 2 3 8; do GSTAMD_BIL_HALF_ROWS=$r run c5 1; done
for w in 2 4; do GSTAMD_BIL_WG=$w run c5 16; done
echo "survey: 4K -> 1080p single frames, rows per wave of the half kernel"
for r in 0 2 3 4 8; do echo "HALF_ROWS=$r"; GSTAMD_BIL_HALF_ROWS=$r python scripts/bench_survey.py NV12 2>&1 | grep "3840x2160 -> BGRA       1920x1080 bilinear" | cut -c1-120; done
echo "survey: enlargements, rows per wave of k_bilinear4_up"
for r in 2 4 8 16 32; do echo "UP_ROWS=$r"; GSTAMD_BIL4_UP_ROWS=$r python scripts/bench_survey.py BGRA 2>&1 | grep "1920x1080 -> BGRA       3840x2160" | cut -c1-120; done
} > gpurun_out/r04_half_abl.log 2>&1
cat gpurun_out/r04_half_abl.log
