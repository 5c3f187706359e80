#!/bin/bash
# k_bilinear420_half: where the time goes (ablation builds of the tuning library; 5 = no passes / matrix, 6 = loads, chroma filter and stores only)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
run() { python bench.py --config $1 --batch $2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 batch $2 STORE/ABL=$GSTAMD_BIL_HALF_STORE:', j['value'], j['roofline']['frac'], j['roofline'].get('avg_launch_us'))
"; }
{
run c5 16
for m in 5 6; do GSTAMD_BIL_HALF_STORE=$m run c5 16; done
run f8scale 8; run f8scale 1
} > gpurun_out/r04_half_abl.log 2>&1
cat gpurun_out/r04_half_abl.log
