#!/bin/bash
# round 5: Y41B / AV12 on the device, device fuzz with the 118-format table
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r05y41b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_video_gpu.py -m gpu -x -q -k "y41b or av12" > $O/pytest_y41b.log 2>&1; echo "y41b rc=$?" >> $O/pytest_y41b.log
timeout 900 python -m pytest tests/test_plugin_gpu.py -m gpu -x -q -k "round5" > $O/pytest_plugin.log 2>&1; echo "plugin rc=$?" >> $O/pytest_plugin.log
GSTAMD_FUZZ_SEEDS=43001-43100 timeout 1200 python -m pytest tests/test_video_fuzz.py -m gpu -q -p no:cacheprovider > $O/fuzz_gpu_100_seeds.log 2>&1; echo "fuzz rc=$?" >> $O/fuzz_gpu_100_seeds.log
for f in $O/*.log; do echo == $f; tail -n 4 $f; done
