#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python -m pytest tests/test_compositor.py -m gpu -q -x > $R/i_pytest_comp.log 2>&1; echo "exit $?" >> $R/i_pytest_comp.log; tail -3 $R/i_pytest_comp.log
timeout 300 python bench.py --config c4 --no-cpu-baseline > $R/i_bench_c4.log 2>&1; grep -o '"value": [0-9.]*\|"avg_launch_us": [0-9.]*\|"frac": [0-9.]*' $R/i_bench_c4.log | head -3 | tr '\n' ' '; echo
