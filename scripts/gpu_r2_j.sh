#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_video_gpu.py -m gpu -q -x -k "bilinear or c5 or golden" > $R/j_pytest.log 2>&1; echo "exit $?" >> $R/j_pytest.log; tail -3 $R/j_pytest.log
timeout 300 python bench.py --config c5 --no-cpu-baseline > $R/j_bench_c5.log 2>&1; grep -o '"value": [0-9.]*\|"avg_launch_us": [0-9.]*\|"frac": [0-9.]*' $R/j_bench_c5.log | head -3 | tr '\n' ' '; echo
cd /tmp; rm -rf /tmp/pj
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pj -o t -- python $GRAFT_REPO_ROOT/bench.py --config c5 --no-cpu-baseline --steps 30 > /tmp/pj.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/pj/**/*counter_collection.csv", recursive=True)
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "k_bilinear420" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("k_bilinear420 FETCH_SIZE raw KB avg", sum(v) / len(v), "x2 MB", 2 * sum(v) / len(v) * 1024 / 1e6, "(source frame 49.77 MB)")
PY
