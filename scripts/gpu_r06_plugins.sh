#!/bin/bash
# the plugin tests on the device (both runtimes)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_plugin_gpu.py -m gpu -q -p no:cacheprovider -n 6 2>&1 | tail -25 | tee $O/pytest_plugin_gpu.log
