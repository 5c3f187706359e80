#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
run() { python bench.py --config $1 --batch $2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 batch $2 NARROW=$GSTAMD_ENCODE16_NARROW:', j['value'], j['roofline']['frac'], j['roofline'].get('avg_launch_us'))
"; }
{
run f5encode16 8; run f5encode16 1
GSTAMD_ENCODE16_NARROW=1 run f5encode16 8; GSTAMD_ENCODE16_NARROW=1 run f5encode16 1
} > gpurun_out/r04_enc16_variants.log 2>&1
cat gpurun_out/r04_enc16_variants.log
