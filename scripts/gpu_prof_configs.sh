#!/bin/bash
# per-kernel durations of the secondary configs (kernel-trace stats)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; rm -rf /tmp/pc
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o cfg -- python "$GRAFT_REPO_ROOT/scripts/bench_configs.py" --iters 100 > /tmp/pc.log 2>&1)
f=$(find /tmp/pc -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/configs_kernel_stats.csv
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/configs_kernel_stats.csv")))
for r in rows:
    if "gstamd" in r["Name"] or "k_" in r["Name"]:
        print("%-110s calls=%5s avg_us=%9.2f" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
