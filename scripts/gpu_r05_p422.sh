#!/bin/bash
# round 5: k_bilinear422_rows (2-tap scaling straight from packed 4:2:2 frames)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r05p422; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_video_gpu.py -m gpu -x -q -k "yuy2 or uyvy or yvyu or vyuy or bgra or argb or ayuv or rgba" > $O/pytest_p422.log 2>&1; echo "p422 rc=$?" >> $O/pytest_p422.log
GSTAMD_FUZZ_SEEDS=${FUZZ_SEEDS:-44001-44060} timeout 1200 python -m pytest tests/test_video_fuzz.py -m gpu -q -p no:cacheprovider > $O/fuzz_gpu_60_seeds.log 2>&1; echo "fuzz rc=$?" >> $O/fuzz_gpu_60_seeds.log
python scripts/probe_p422.py 2>&1 | grep -v amdgpu.ids > $O/${PROBE_LOG:-probe_bilinear422_rows.log}
for f in $O/pytest_p422.log $O/fuzz_gpu_60_seeds.log; do tail -n 3 $f; done; cut -c1-200 $O/${PROBE_LOG:-probe_bilinear422_rows.log}
