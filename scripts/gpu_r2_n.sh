#!/bin/bash
# C5: balanced strips of k_bilinear420_rows - how many wave slots to aim at
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=gpurun_out; L=$R/n_c5_slots.log; : > $L
for v in ${SLOTS_SWEEP:-3712 3840 3900 3968 4000 4032 4064}; do
  echo "== slots=$v" >> $L
  GSTAMD_TUNING_LIB=1 GSTAMD_BIL_ROWS=-1 GSTAMD_BIL_SLOTS=$v GSTAMD_BIL_VERBOSE=1 timeout 200 python bench.py --config c5 --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | python -c "
import sys, json
seen = False
for l in sys.stdin:
    if l.startswith('k_bilinear420_rows') and not seen:
        print(l.strip()); seen = True
    if l.startswith('{'):
        j = json.loads(l); print('us/launch', j['roofline']['avg_launch_us'], 'frac', j['roofline']['frac'])
" >> $L
done
cat $L
