#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python -m pytest tests/test_video_gpu.py -m gpu -q -x -k "c3 or h420 or hscale420 or lanczos or quarter" > $R/g_pytest_c3.log 2>&1; echo "exit $?" >> $R/g_pytest_c3.log
tail -4 $R/g_pytest_c3.log
rm -f $R/g_c3_variants.log
for v in 4 2 1; do
  echo "== mfma waves=$v" >> $R/g_c3_variants.log
  GSTAMD_MFMA_WAVES=$v timeout 300 python bench.py --config c3 --steps 60 --no-cpu-baseline >> $R/g_c3_variants.log 2>&1
done
for r in 16 32 64 135; do
  echo "== mfma waves=4 rows=$r" >> $R/g_c3_variants.log
  GSTAMD_MFMA_WAVES=4 GSTAMD_MFMA_ROWS=$r timeout 300 python bench.py --config c3 --steps 60 --no-cpu-baseline >> $R/g_c3_variants.log 2>&1
done
echo "== fused (no mfma)" >> $R/g_c3_variants.log
GSTAMD_NO_MFMA420=1 timeout 300 python bench.py --config c3 --steps 60 --no-cpu-baseline >> $R/g_c3_variants.log 2>&1
grep -o '== .*\|"value": [0-9.]*\|"avg_launch_us": [0-9.]*\|"frac": [0-9.]*\|Error.*\|error.*' $R/g_c3_variants.log | tr '\n' ' ' | sed 's/==/\n==/g'; echo
