#!/bin/bash
# round 3, call D: whole GPU suite (plugin harness tests incl.), swizzle bench
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3d; R=$GRAFT_REPO_ROOT/gpurun_out/r3d
timeout 800 python -m pytest tests -m gpu -q > $R/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $R/pytest_gpu.log
timeout 120 python scripts/bench_swizzle.py > $R/swizzle.log 2>&1
tail -n 12 $R/pytest_gpu.log; cat $R/swizzle.log
