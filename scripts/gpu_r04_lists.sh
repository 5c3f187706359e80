#!/bin/bash
# round 4: the frame-list tests (library + element), the new plane / deep-plane kernels against the reference, then bench line + kernel stats
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_video_gpu.py tests/test_compositor.py tests/test_plugin_gpu.py -m gpu -q \
  -k "frame_list or buffer_lists or high_index or scale_col_frame or quad_ or planes" > gpurun_out/r04_list_tests.log 2>&1
tail -8 gpurun_out/r04_list_tests.log
GSTAMD_PROF_NO_PMC=1 bash scripts/gpu_profiles.sh "$@"
for c in "$@"; do grep -h '^{' gpurun_out/prof/bench_$c.json | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print(j['config']['workload'][:40], j['value'], j['ms_per_step'], j['roofline']['frac'])
"; done
