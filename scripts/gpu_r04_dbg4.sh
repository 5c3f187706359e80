#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
run() { python bench.py --config $1 --batch $2 --no-cpu-baseline 2>gpurun_out/err.txt | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 batch $2 WAVES=$GSTAMD_COL_WAVES OPL=$GSTAMD_COL_OPL:', j['value'], j['roofline']['frac'], j['roofline'].get('avg_launch_us'))
"; grep k_scale_col gpurun_out/err.txt | head -1; }
export GSTAMD_COL_DEBUG=1
{
run c3 8
for w in 2 3 4 5 6; do GSTAMD_COL_WAVES=$w run c3 8; done
GSTAMD_COL_OPL=1 run c3 8
for w in 3 4 6 8; do GSTAMD_COL_OPL=1 GSTAMD_COL_WAVES=$w run c3 8; done
GSTAMD_COL_WAVES=4 run c3 1
GSTAMD_COL_WAVES=4 run c3 4
} > gpurun_out/r04_dbg4.log 2>&1
cat gpurun_out/r04_dbg4.log
