#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_video_gpu.py -m gpu -q -x -k "p010 or i42010" > $R/k_pytest.log 2>&1; echo "exit $?" >> $R/k_pytest.log; tail -3 $R/k_pytest.log
timeout 600 python -m pytest tests/test_plugin_gpu.py -m gpu -q -x -k "p010" > $R/k_pytest_plugin.log 2>&1; echo "exit $?" >> $R/k_pytest_plugin.log; tail -5 $R/k_pytest_plugin.log
