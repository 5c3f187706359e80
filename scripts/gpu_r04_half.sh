#!/bin/bash
# k_bilinear420_half on the device: parity of the bilinear cases + the new compositor canvases, then C5 with the knobs of the new kernel
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_video_gpu.py -m gpu -q -k "half or bil420 or 2to1 or bilinear or frame_lists" > gpurun_out/r04_half_tests.log 2>&1
tail -3 gpurun_out/r04_half_tests.log
timeout 600 python -m pytest tests/test_compositor.py -m gpu -q > gpurun_out/r04_comp_tests.log 2>&1
tail -3 gpurun_out/r04_comp_tests.log
timeout 600 python -m pytest tests/test_plugin_gpu.py -m gpu -q -k "canvases" > gpurun_out/r04_canvas_tests.log 2>&1
tail -3 gpurun_out/r04_canvas_tests.log
run() { python bench.py --config $1 --batch $2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 batch $2 NOHALF=$GSTAMD_NO_BILINEAR_HALF ROWS=$GSTAMD_BIL_HALF_ROWS STORE=$GSTAMD_BIL_HALF_STORE WG=$GSTAMD_BIL_WG:', j['value'], j['roofline']['frac'], j['roofline'].get('avg_launch_us'))
"; }
{
run c5 16; run c5 1
for m in 2 3 4; do GSTAMD_BIL_HALF_STORE=$m run c5 16; GSTAMD_BIL_HALF_STORE=$m run c5 1; done
GSTAMD_NO_BILINEAR_HALF=1 run c5 16; GSTAMD_NO_BILINEAR_HALF=1 run c5 1
for r in 4 16; do GSTAMD_BIL_HALF_ROWS=$r run c5 16; done
GSTAMD_BIL_HALF_ROWS=8 run c5 1
for w in 2 4; do GSTAMD_BIL_WG=$w run c5 16; done
} > gpurun_out/r04_half_variants.log 2>&1
cat gpurun_out/r04_half_variants.log
