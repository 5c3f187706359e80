#!/bin/bash
# round 5: rows per wave of k_aggregate_direct_cull (1, 2, 4) on C4's layout with opaque pads
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r05cull2; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_compositor.py tests/test_compositor_fuzz.py -m gpu -x -q > $O/pytest_compositor.log 2>&1; echo "compositor rc=$?" >> $O/pytest_compositor.log
for r in 1 2 4; do
  GSTAMD_CULL_ROWS_EXPERIMENT=$r timeout 900 python -m pytest tests/test_compositor.py -m gpu -x -q -k opaque > $O/pytest_rows$r.log 2>&1; echo "rows $r rc=$?" >> $O/pytest_rows$r.log
  for h in map all; do
    GSTAMD_CULL_ROWS_EXPERIMENT=$r timeout 300 python bench.py --config c4opaque --opaque-hint $h > $O/bench_rows${r}_$h.json 2> $O/bench_rows${r}_$h.err
  done
done
timeout 300 python bench.py --config c4 > $O/bench_c4.json 2> $O/bench_c4.err
for f in $O/pytest_*.log; do tail -n 2 $f; done
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get("roofline",{})
    print(sys.argv[1].split("/")[-1], d["value"], d["unit"], "launch_us", round(d["ms_per_step"]*1000/d["config"].get("frames_per_step",1),2), "frac", r.get("frac"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
