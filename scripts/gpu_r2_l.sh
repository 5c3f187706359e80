#!/bin/bash
# C4 compositor: rows per wave / load depth sweep on the tuning library, then the product library + parity tests
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=gpurun_out; L=$R/l_c4_variants.log; : > $L
timeout 300 python -m pytest tests/test_compositor.py -m gpu -x -q > $R/l_pytest.log 2>&1; tail -3 $R/l_pytest.log
for v in "0 4" "1 4" "2 4" "4 4" "8 4" "16 4" "4 6" "8 6" "4 8" "8 8" "16 8"; do
  set -- $v
  echo "== rows=$1 depth=$2" >> $L
  GSTAMD_TUNING_LIB=1 GSTAMD_AGG_ROWS=$1 GSTAMD_AGG_DEPTH=$2 timeout 200 python bench.py --config c4 --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('us/launch', j['roofline']['avg_launch_us'], 'frac', j['roofline']['frac'])
" >> $L
done
echo "== product" >> $L
timeout 200 python bench.py --config c4 --steps 150 --warmup 20 --no-cpu-baseline >> $L 2>&1
cat $L
