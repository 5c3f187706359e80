#!/bin/bash
# kernel-trace stats of bench.py under each GSTAMD_FAST_VARIANT in $VARIANTS (checks which kernel really ran)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for v in $VARIANTS; do
  rm -rf /tmp/tv; (cd /tmp && GSTAMD_FAST_VARIANT="$v" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tv -o t -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > /tmp/tv.log 2>&1)
  f=$(find /tmp/tv -name "*kernel_stats.csv" | head -1)
  echo "== variant $v"; python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_convert" in r["Name"]:
        print("%-90s calls=%s avg_us=%.2f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
