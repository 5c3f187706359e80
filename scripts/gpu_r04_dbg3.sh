#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
run() { python bench.py --config $1 --batch $2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 batch $2 KERNARG=$HIP_FORCE_DEV_KERNARG:', j['value'], j['roofline']['frac'], j['roofline'].get('avg_launch_us'))
"; }
{
for k in 0 1; do export HIP_FORCE_DEV_KERNARG=$k
run c3 8; run c3 1; run f2p010in 1; run f2p010in 8; run f8pack 1; run c2 1; run c2 32
done
} > gpurun_out/r04_dbg3.log 2>&1
cat gpurun_out/r04_dbg3.log
