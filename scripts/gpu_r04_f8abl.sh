#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
run() { python bench.py --config $1 --batch $2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 batch $2 ROWS=$GSTAMD_PLANE_QUAD_ROWS ONLY=$GSTAMD_PLANE_QUAD_ONLY NT=$GSTAMD_PLANE_QUAD_NT:', j['value'], j['roofline']['frac'], j['roofline'].get('avg_launch_us'))
"; }
export GSTAMD_TUNING_LIB=1
{
for nt in 0 4 8 12; do GSTAMD_PLANE_QUAD_ONLY=0 GSTAMD_PLANE_QUAD_NT=$nt GSTAMD_PLANE_QUAD_ROWS=1 run f8scale 8; done
for nt in 0 4 8 12; do GSTAMD_PLANE_QUAD_ONLY=0 GSTAMD_PLANE_QUAD_NT=$nt GSTAMD_PLANE_QUAD_ROWS=4 run f8scale 8; done
GSTAMD_PLANE_QUAD_ONLY=0 GSTAMD_PLANE_QUAD_NT=12 GSTAMD_PLANE_QUAD_ROWS=1 run f8scale 1
} > gpurun_out/r04_f8scale_ablation.log 2>&1
cat gpurun_out/r04_f8scale_ablation.log
