#!/bin/bash
# round 5: the column walk of the compositor's scaled pads (k_aggregate_walk): parity tests, the C4-A line, chunk-row sweep
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_compositor.py tests/test_compositor_fuzz.py tests/test_video_gpu.py -m gpu -q -x -p no:cacheprovider -k "scaled or walk or gbr or compositor" > gpurun_out/r05_pytest_walk.log 2>&1
tail -15 gpurun_out/r05_pytest_walk.log
for rows in 0 20 27 45 68; do
  GSTAMD_WALK_ROWS=$rows timeout 300 python bench.py --config c4a --no-cpu-baseline > gpurun_out/r05_bench_c4a_rows$rows.json 2> gpurun_out/r05_bench_c4a_rows$rows.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/r05_bench_c4a_rows$rows.json").read().strip().splitlines()[-1])
print("rows", $rows, j["value"], j["roofline"]["avg_launch_us"], j["roofline"]["frac"])
PY
done
