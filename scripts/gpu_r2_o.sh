#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=gpurun_out; L=$R/o_c5_wg.log; : > $L
for v in "1 6" "2 6" "4 6" "4 3" "4 -1" "2 -1"; do
  set -- $v
  echo "== wg=$1 rows=$2" >> $L
  GSTAMD_TUNING_LIB=1 GSTAMD_BIL_ROWS=$2 GSTAMD_BIL_WG=$1 GSTAMD_BIL_SLOTS=3840 timeout 200 python bench.py --config c5 --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('us/launch', j['roofline']['avg_launch_us'], 'frac', j['roofline']['frac'])
" >> $L
done
cat $L
