#!/bin/bash
# round 5, closing run: the whole GPU suite, 800 fresh device fuzz seeds (120 000 draws, every format of the table), the default bench line, smoke,
# per-config evidence (bench line + rocprofv3 kernel stats + PMC traffic)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05f
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r05f/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05f/pytest_gpu.log; tail -4 gpurun_out/r05f/pytest_gpu.log
GSTAMD_FUZZ_SEEDS=50001-50800 timeout 1500 python -m pytest tests/test_video_fuzz.py -m gpu -q -p no:cacheprovider > gpurun_out/r05f/fuzz_gpu_800_seeds.log 2>&1
tail -3 gpurun_out/r05f/fuzz_gpu_800_seeds.log
timeout 400 python bench.py 2>gpurun_out/r05f/bench_default.err > gpurun_out/r05f/bench_default.json; cut -c1-300 gpurun_out/r05f/bench_default.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/gpu_profiles.sh c2 c3 c4a c4 c4opaque c5 c1 c4audio c4audiomany > gpurun_out/r05f/profiles.log 2>&1
cp gpurun_out/prof/* gpurun_out/r05f/ 2>/dev/null
for c in c2 c3 c4a c4 c4opaque c5 c1 c4audio c4audiomany; do python - $c <<'PY'
import json,sys
c=sys.argv[1]
try:
    j=json.loads(open("gpurun_out/r05f/bench_%s.json"%c).read().strip().splitlines()[-1])
    r=j["roofline"]; print(c, j["value"], j["unit"], "launch_us", r.get("avg_launch_us"), "frac", r["frac"])
except Exception as e: print(c, "ERR", e)
PY
done
timeout 300 python scripts/survey_item6.py > gpurun_out/r05f/survey_item6_end.log 2>&1; grep -- "->" gpurun_out/r05f/survey_item6_end.log | cut -c1-150
