#!/bin/bash
# C5: frames per launch of k_bilinear420_rows (the converter's frame-list entry point)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=gpurun_out; L=$R/r_c5_batch.log; : > $L
for v in "1 -1" "2 -1" "4 -1" "8 -1" "8 4" "8 8" "8 3"; do
  set -- $v
  echo "== batch=$1 rows=$2" >> $L
  GSTAMD_TUNING_LIB=1 GSTAMD_BIL_ROWS=$2 timeout 200 python bench.py --config c5 --batch $1 --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('us/launch', j['roofline']['avg_launch_us'], 'frac', j['roofline']['frac'], 'fps', j['value'])
" >> $L
done
cat $L
