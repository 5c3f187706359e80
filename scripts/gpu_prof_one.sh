#!/bin/bash
# per-kernel durations of bench_one.py configs (kernel-trace stats):  gpu_prof_one.sh cfg [cfg ...]
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for c in "$@"; do
  rm -rf /tmp/po
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/po -o one -- python "$GRAFT_REPO_ROOT/scripts/bench_one.py" $c 50 > /tmp/po.log 2>&1)
  f=$(find /tmp/po -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/one_${c}_kernel_stats.csv
  echo "== $c"
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_" in r["Name"]:
        print("%-100s calls=%5s avg_us=%9.2f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
