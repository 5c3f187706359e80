#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
GSTAMD_LIST_DEBUG=1 timeout 900 python -m pytest tests/test_video_gpu.py tests/test_compositor.py tests/test_plugin_gpu.py -m gpu -q -s \
  -k "frame_lists or buffer_lists or high_index" > gpurun_out/r04_list_tests.log 2>&1
tail -5 gpurun_out/r04_list_tests.log
for b in 1 8; do for c in f2p010in f2p010out c3 f8pack; do
  python bench.py --config $c --batch $b --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$c batch $b', j['value'], j['roofline']['frac'], j['roofline'].get('avg_launch_us'))
"; done; done > gpurun_out/r04_dbg_bench.log 2>&1
cat gpurun_out/r04_dbg_bench.log
timeout 60 python scripts/clock_ramp.py > gpurun_out/r04_clock.log 2>&1; tail -5 gpurun_out/r04_clock.log
