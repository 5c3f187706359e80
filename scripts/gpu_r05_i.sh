#!/bin/bash
# round 5 (re-entry): the whole GPU suite on the current tree, the default bench line, smoke, then per-config evidence
# (bench line + rocprofv3 kernel stats + PMC traffic) for the configs this round's kernels serve
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05i
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r05i/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05i/pytest_gpu.log; tail -4 gpurun_out/r05i/pytest_gpu.log
timeout 400 python bench.py 2>gpurun_out/r05i/bench_default.err > gpurun_out/r05i/bench_default.json; cut -c1-400 gpurun_out/r05i/bench_default.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/gpu_profiles.sh c2 c3 c4a c4 c5 c4audiomany > gpurun_out/r05i/profiles.log 2>&1
cp gpurun_out/prof/* gpurun_out/r05i/ 2>/dev/null
for c in c2 c3 c4a c4 c5 c4audiomany; do python - $c <<'PY'
import json,sys
c=sys.argv[1]
try:
    j=json.loads(open("gpurun_out/r05i/bench_%s.json"%c).read().strip().splitlines()[-1])
    r=j["roofline"]; print(c, j["value"], j["unit"], "launch_us", r.get("avg_launch_us"), "frac", r["frac"])
except Exception as e: print(c, "ERR", e)
PY
done
