#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05x
timeout 900 python -m pytest tests/test_plugin_gpu.py -m gpu -q -p no:cacheprovider -k "compositor" > gpurun_out/r05x/pytest_compositor.log 2>&1
tail -15 gpurun_out/r05x/pytest_compositor.log
