#!/bin/bash
# C5 (8K NV12 -> 4K BGRA bilinear): k_bilinear420_rows, rows-per-wave sweep on the tuning library + parity
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=gpurun_out; L=$R/m_c5_variants.log; : > $L
timeout 600 python -m pytest tests/test_video_gpu.py -m gpu -x -q -k "bil or c5 or 2to1 or half" > $R/m_pytest.log 2>&1; tail -3 $R/m_pytest.log
for v in ${ROWS_SWEEP:-0 1 2 4 8 16}; do
  echo "== rows=$v" >> $L
  GSTAMD_TUNING_LIB=1 GSTAMD_BIL_ROWS=$v timeout 200 python bench.py --config c5 --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('us/launch', j['roofline']['avg_launch_us'], 'frac', j['roofline']['frac'], j['roofline']['kernel'])
" >> $L
done
echo "== product" >> $L
timeout 200 python bench.py --config c5 --steps 150 --warmup 20 --no-cpu-baseline >> $L 2>&1
cat $L
