#!/bin/bash
# C4 compositor: k_aggregate_strip rows-per-wave sweep (tuning library) + parity + product line
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=gpurun_out; L=$R/p_c4_strip.log; : > $L
timeout 300 python -m pytest tests/test_compositor.py -m gpu -x -q > $R/p_pytest.log 2>&1; tail -3 $R/p_pytest.log
for v in ${STRIP_SWEEP:-0 1 2 3 4 6 8}; do
  echo "== strip rows=$v" >> $L
  GSTAMD_TUNING_LIB=1 GSTAMD_AGG_STRIP_ROWS=$v timeout 200 python bench.py --config c4 --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('us/launch', j['roofline']['avg_launch_us'], 'frac', j['roofline']['frac'])
" >> $L
done
echo "== product" >> $L
timeout 200 python bench.py --config c4 --steps 150 --warmup 20 --no-cpu-baseline 2>&1 | cut -c1-250 >> $L
cat $L
