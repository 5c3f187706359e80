#!/bin/bash
# round 3, call B: GPU suite with the direct compositor kernel and the defined-result plans; C4 bench + kernel trace
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3b; R=$GRAFT_REPO_ROOT/gpurun_out/r3b
timeout 120 python -m pytest tests/test_compositor.py tests/test_compositor_fuzz.py -m gpu -q -x > $R/pytest_comp.log 2>&1; rc=$?; echo "pytest exit $rc" >> $R/pytest_comp.log
tail -n 3 $R/pytest_comp.log
[ $rc -ne 0 ] && exit 1
timeout 600 python -m pytest tests -m gpu -q > $R/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $R/pytest_gpu.log
timeout 120 python bench.py --config c4 --no-cpu-baseline > $R/bench_c4.json 2> $R/bench_c4.err
( cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4 -o t -- python $GRAFT_REPO_ROOT/bench.py --config c4 --no-cpu-baseline > /tmp/p_c4.log 2>&1
  f=$(find /tmp/p_c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/c4_kernel_stats.csv )
tail -n 4 $R/pytest_gpu.log; cut -c1-700 $R/bench_c4.json; head -5 $R/c4_kernel_stats.csv | cut -c1-200
