#!/bin/bash
# C4: k_aggregate_strip with 4 / 8 pixels per lane, rows per wave
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=gpurun_out; L=$R/s_c4_strip_px.log; : > $L
for v in "0 4" "2 4" "1 8" "2 8" "3 8" "4 8"; do
  set -- $v
  echo "== strip rows=$1 px=$2" >> $L
  GSTAMD_TUNING_LIB=1 GSTAMD_AGG_STRIP_ROWS=$1 GSTAMD_AGG_STRIP_PX=$2 timeout 200 python bench.py --config c4 --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('us/launch', j['roofline']['avg_launch_us'], 'frac', j['roofline']['frac'])
" >> $L
done
GSTAMD_TUNING_LIB=1 GSTAMD_AGG_STRIP_ROWS=2 GSTAMD_AGG_STRIP_PX=8 timeout 300 python -m pytest tests/test_compositor.py -m gpu -x -q 2>&1 | tail -2 >> $L
cat $L
