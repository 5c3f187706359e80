#!/bin/bash
# Round 6: interlaced frames on the device - the interlaced tests, the plugin tests that negotiate interlaced caps, then the whole GPU suite (the
# vertical chroma blend of every front kernel went from role-based 3:1 weights to vpair_get's weights over 8).  bash scripts/gpu_r06_interlaced.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_video_interlaced.py -m gpu -q -p no:cacheprovider -n 6 2>&1 | tail -15 | tee $O/pytest_interlaced_gpu.log
timeout 600 python -m pytest tests/test_plugin_gpu.py -m gpu -q -p no:cacheprovider -k "interlace" 2>&1 | tail -15 | tee $O/pytest_interlaced_plugin.log
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider -n 6 > $O/pytest_gpu_2.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_2.log; tail -n 6 $O/pytest_gpu_2.log
