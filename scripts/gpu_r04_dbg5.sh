#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for c in f2p010in f8pack; do
  ( cd /tmp; rm -rf /tmp/q_$c; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q_$c -o t -- python $GRAFT_REPO_ROOT/bench.py --config $c --batch 1 --no-cpu-baseline > /tmp/q_$c.log 2>&1 )
  grep '^{' /tmp/q_$c.log | cut -c1-200
  f=$(find /tmp/q_$c -name "*kernel_stats.csv" | head -1); head -3 $f | cut -c1-160
done > gpurun_out/r04_dbg5.log 2>&1
python bench.py --config c3 --no-cpu-baseline 2>&1 | grep '^{' | cut -c1-220 >> gpurun_out/r04_dbg5.log
cat gpurun_out/r04_dbg5.log
