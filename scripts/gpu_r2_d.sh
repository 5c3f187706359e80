#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=gpurun_out; rm -f $R/element_bench.jsonl
timeout 900 python -m pytest tests/test_plugin_gpu.py -m gpu -q -s > $R/d_pytest_plugin.log 2>&1; echo "exit $?" >> $R/d_pytest_plugin.log
tail -25 $R/d_pytest_plugin.log
cat $R/element_bench.jsonl
