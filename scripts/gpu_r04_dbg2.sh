#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
run() { python bench.py --config $1 --batch $2 --steps $3 --warmup $4 --preheat-ms $5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 batch $2 steps $3 warmup $4 preheat $5:', j['value'], j['roofline']['frac'], j['roofline'].get('avg_launch_us'))
"; }
{
run c3 8 40 8 60
run c3 8 150 20 60
run c3 8 600 20 60
run c3 8 40 2 0
run c3 8 2000 20 60
run f2p010in 1 50 20 60
run f2p010in 1 400 20 60
run f2p010in 8 50 20 60
run f2p010in 8 400 20 60
run c2 32 400 20 60
} > gpurun_out/r04_dbg2.log 2>&1
cat gpurun_out/r04_dbg2.log
