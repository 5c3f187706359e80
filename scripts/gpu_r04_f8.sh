#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_video_gpu.py -m gpu -q -k "quad_ or planes or gray8 or frame_lists" > gpurun_out/r04_quad_tests.log 2>&1
tail -3 gpurun_out/r04_quad_tests.log
run() { python bench.py --config $1 --batch $2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 batch $2 ROWS=$GSTAMD_PLANE_QUAD_ROWS SEQ=$GSTAMD_PLANE_QUAD_SEQ NODSTEP=$GSTAMD_PLANE_QUAD_NO_DSTEP ONLY=$GSTAMD_PLANE_QUAD_ONLY:', j['value'], j['roofline']['frac'], j['roofline'].get('avg_launch_us'))
"; }
{
for s in 1 0; do for r in 1 2 4; do GSTAMD_PLANE_QUAD_SEQ=$s GSTAMD_PLANE_QUAD_ROWS=$r run f8scale 8; done; done
for r in 1 2 4; do GSTAMD_PLANE_QUAD_SEQ=1 GSTAMD_PLANE_QUAD_ONLY=0 GSTAMD_PLANE_QUAD_ROWS=$r run f8scale 8; done
GSTAMD_PLANE_QUAD_SEQ=1 GSTAMD_PLANE_QUAD_NO_DSTEP=1 GSTAMD_PLANE_QUAD_ROWS=2 run f8scale 8
for s in 0 1; do GSTAMD_PLANE_QUAD_SEQ=$s run f8scale 1; done
} > gpurun_out/r04_f8scale_variants10.log 2>&1
cat gpurun_out/r04_f8scale_variants10.log
