#!/bin/bash
# round 5: k_aggregate_walk occupancy variants (amdgpu_waves_per_eu 6 / 7 / 8) x rows per chunk
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
run () { GSTAMD_WALK_ROWS=$2 GSTAMD_LIB_PATH=$1 python bench.py --config c4a --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1'[-12:], $2, j['value'], j['roofline']['avg_launch_us'], j['roofline']['frac'])"; }
L=gstreamer_amd/lib
for r in 23 20; do run $L/libgstamddsp.so $r; done
for r in 23 20 19; do run $L/libgstamddsp_w7.so $r; done
for r in 20 17 15 12; do run $L/libgstamddsp_w8.so $r; done
