#!/bin/bash
# C5 timing for the fused bilinear kernel at the tile widths in $TILES (auto = launcher's choice, 0 = kernel off); PARITY=1 runs the video tests first
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
if [ -n "$PARITY" ]; then timeout 600 python -m pytest tests/test_video_gpu.py -m gpu -x -q > gpurun_out/pytest_video.log 2>&1; tail -3 gpurun_out/pytest_video.log; fi
for t in $TILES; do
  if [ "$t" = auto ]; then e="X=1"; elif [ "$t" = 0 ]; then e="GSTAMD_NO_BILINEAR420=1"; else e="GSTAMD_BIL_TILE=$t"; fi
  echo "tile=$t $(env $e python scripts/bench_one.py ${CONFIG:-c5} 200 2>&1 | grep '^{' | grep -o '"us_per_frame": [0-9.]*\|"achieved": [0-9.]*' | tr '\n' ' ')"
done
