#!/bin/bash
# the SURVEY 8(f)2 rows measured like the BASELINE configs: bench line + rocprofv3 kernel stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/f2
for c in f2gamma f2p010out f2p010in; do
  python $R/bench.py --config $c --no-cpu-baseline > $R/gpurun_out/f2/bench_$c.json 2> $R/gpurun_out/f2/bench_$c.err
  tail -c 600 $R/gpurun_out/f2/bench_$c.json; echo
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/f2/prof_$c -o $c -- python $R/bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  f=$(find $R/gpurun_out/f2/prof_$c -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -8 "$f" | cut -c1-200 > $R/gpurun_out/f2/kernel_stats_$c.csv
  find $R/gpurun_out/f2/prof_$c -name "*.db" -delete; find $R/gpurun_out/f2/prof_$c -name "*trace.csv" -delete
done
