#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_video_gpu.py -m gpu -q -k "quad or planes or gray8 or frame_lists or bilinear" > gpurun_out/r04_quad_tests.log 2>&1
tail -3 gpurun_out/r04_quad_tests.log
run() { python bench.py --config $1 --batch $2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 batch $2 ROWS=$GSTAMD_PLANE_QUAD_ROWS ONLY=$GSTAMD_PLANE_QUAD_ONLY:', j['value'], j['roofline']['frac'], j['roofline'].get('avg_launch_us'))
"; }
{
run f8scale 8; run f8scale 1
for r in 1 2 4; do GSTAMD_PLANE_QUAD_ROWS=$r run f8scale 8; done
for r in 1 2 4; do GSTAMD_PLANE_QUAD_ONLY=0 GSTAMD_PLANE_QUAD_ROWS=$r run f8scale 8; done
python scripts/bench_survey.py BGRA I420 2>&1 | grep "scale_planes\|as convert_scale" | cut -c1-150
} > gpurun_out/r04_f8scale_variants11.log 2>&1
cat gpurun_out/r04_f8scale_variants11.log
