#!/bin/bash
# round 5: where the item-6 pairs stand (single / lists of 8) and the per-kernel split of each (rocprofv3 kernel stats, one case per run)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05j; R=$GRAFT_REPO_ROOT/gpurun_out/r05j
timeout 600 python scripts/survey_item6.py > $R/survey_item6.log 2>&1; cat $R/survey_item6.log
for i in 0 1 4 5 8 9 11 12; do
  ( cd /tmp; rm -rf /tmp/s6_$i; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/s6_$i -o t -- python $GRAFT_REPO_ROOT/scripts/survey_item6.py $i > /tmp/s6_$i.log 2>&1 )
  f=$(find /tmp/s6_$i -name "*kernel_stats.csv" | head -1)
  echo "== case $i: $(grep -- '->' /tmp/s6_$i.log | cut -c1-60)" >> $R/kernel_split.log
  [ -n "$f" ] && python - "$f" >> $R/kernel_split.log <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "at::native" in r["Name"] or "rocclr" in r["Name"]: continue
    print("  %-90s calls %5s avg %8.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
cat $R/kernel_split.log
