#!/bin/bash
# Full GPU round: parity tests, smoke, bench (batched + per-frame), kernel-trace stats and PMC traffic
# of the SAME bench command.  Summaries land in gpurun_out/ (copied to profiles/ afterwards).
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=gpurun_out
timeout 900 python -m pytest tests -m gpu -q > $R/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $R/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; echo "smoke exit $?" >> $R/smoke.log
timeout 600 python bench.py > $R/bench.log 2>&1; echo "bench exit $?" >> $R/bench.log
timeout 600 python bench.py --batch 1 --no-cpu-baseline > $R/bench_batch1.log 2>&1
timeout 600 python bench.py --batch 8 --no-cpu-baseline > $R/bench_batch8.log 2>&1
BENCH="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline"
rm -rf $R/prof; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$R/prof/trace" -o c2 -- $BENCH > "$GRAFT_REPO_ROOT/$R/prof_trace.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$R/prof/pmc_fetch" -o c2 -- $BENCH > "$GRAFT_REPO_ROOT/$R/prof_fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$R/prof/pmc_write" -o c2 -- $BENCH > "$GRAFT_REPO_ROOT/$R/prof_write.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import csv, glob, json, os
R = "gpurun_out"
out = {}
def stats():
    f = glob.glob(R + "/prof/trace/**/*kernel_stats.csv", recursive=True)
    if not f: return
    rows = list(csv.DictReader(open(f[0])))
    for r in rows:
        if "k_convert" in r["Name"]:
            out["kernel"] = r["Name"]; out["calls"] = int(r["Calls"]); out["avg_ns"] = float(r["AverageNs"]); out["min_ns"] = float(r["MinNs"]); out["max_ns"] = float(r["MaxNs"])
def pmc(kind, name):
    f = glob.glob(R + "/prof/%s/**/*counter_collection.csv" % kind, recursive=True)
    if not f: return
    vals = []
    for r in csv.DictReader(open(f[0])):
        if "k_convert" in r.get("Kernel_Name", "") and r.get("Counter_Name") == name:
            vals.append(float(r["Counter_Value"]))
    if vals:
        out[name + "_avg_per_launch"] = sum(vals) / len(vals); out[name + "_n"] = len(vals)
stats(); pmc("pmc_fetch", "FETCH_SIZE"); pmc("pmc_write", "WRITE_SIZE")
json.dump(out, open(R + "/prof_summary.json", "w"), indent=1)
print(json.dumps(out))
PY
tail -2 $R/pytest_gpu.log; tail -1 $R/smoke.log; grep '^{' $R/bench.log | head -1 | cut -c1-400; grep -o '"value": [0-9.]*\|"achieved": [0-9.]*\|"avg_launch_us": [0-9.]*' $R/bench_batch1.log | tr '\n' ' '
