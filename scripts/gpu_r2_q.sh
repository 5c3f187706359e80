#!/bin/bash
# C4: what bounds k_aggregate - ablations of the tuning library (1: no blend arithmetic, 2: no pad loads, 3: no pads at all, 4: neither)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=gpurun_out; L=$R/q_c4_ablate.log; : > $L
for v in 0 1 2 3 4; do
  echo "== ablate=$v" >> $L
  GSTAMD_TUNING_LIB=1 GSTAMD_AGG_STRIP_ROWS=0 GSTAMD_AGG_ABLATE=$v timeout 200 python bench.py --config c4 --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('us/launch', j['roofline']['avg_launch_us'], 'frac', j['roofline']['frac'])
" >> $L
done
cat $L
