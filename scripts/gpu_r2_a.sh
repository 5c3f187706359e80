#!/bin/bash
# round 2, call A: fused C3 kernel - parity first, then timing of the geometry variants; then the per-config bench lines
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python -m pytest tests/test_video_gpu.py -m gpu -q -x -k "c3 or h420 or hscale420 or lanczos or frame_list_32 or c5_full" > $R/a_pytest_c3.log 2>&1; echo "exit $?" >> $R/a_pytest_c3.log
tail -3 $R/a_pytest_c3.log
for v in "8" "4" "2" "16"; do
  echo "== fused waves=$v" >> $R/a_c3_variants.log
  GSTAMD_FUSED_WAVES=$v timeout 300 python bench.py --config c3 --steps 60 --no-cpu-baseline >> $R/a_c3_variants.log 2>&1
done
for r in 16 24 48; do
  echo "== fused waves=8 rows=$r" >> $R/a_c3_variants.log
  GSTAMD_FUSED_WAVES=8 GSTAMD_FUSED_ROWS=$r timeout 300 python bench.py --config c3 --steps 60 --no-cpu-baseline >> $R/a_c3_variants.log 2>&1
done
echo "== two-pass" >> $R/a_c3_variants.log
GSTAMD_NO_FUSED420=1 timeout 300 python bench.py --config c3 --steps 60 --no-cpu-baseline >> $R/a_c3_variants.log 2>&1
grep -o '== .*\|"value": [0-9.]*\|"avg_launch_us": [0-9.]*\|"frac": [0-9.]*' $R/a_c3_variants.log | tr '\n' ' ' | sed 's/==/\n==/g'; echo
for c in c2 c5 c4 c4audio; do
  timeout 600 python bench.py --config $c > $R/a_bench_$c.log 2>&1; grep -o '"value": [0-9.]*\|"avg_launch_us": [0-9.]*\|"frac": [0-9.]*' $R/a_bench_$c.log | head -3 | tr '\n' ' '; echo " <- $c"
done
