#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python -m pytest tests/test_video_gpu.py -m gpu -q -x -k "c3 or h420 or hscale420 or lanczos or quarter or mfma" > $R/h_pytest_c3.log 2>&1; echo "exit $?" >> $R/h_pytest_c3.log
tail -3 $R/h_pytest_c3.log
rm -f $R/h_c3_variants.log
for f in 12 16 8; do for r in 0 28 44; do
  echo "== fused first=$f rows=$r" >> $R/h_c3_variants.log
  GSTAMD_FUSED_FIRST=$f GSTAMD_FUSED_ROWS=$r timeout 300 python bench.py --config c3 --steps 60 --no-cpu-baseline >> $R/h_c3_variants.log 2>&1
done; done
grep -o '== .*\|"value": [0-9.]*\|"avg_launch_us": [0-9.]*\|"frac": [0-9.]*\|Error.*\|error.*' $R/h_c3_variants.log | tr '\n' ' ' | sed 's/==/\n==/g'; echo
