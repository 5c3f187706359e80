#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=gpurun_out
for v in 16 8 4; do
  echo "===== waves=$v" >> $R/b_trace.log
  GSTAMD_FUSED_WAVES=$v timeout 300 python scripts/trace_fused.py >> $R/b_trace.log 2>&1
done
cat $R/b_trace.log
