#!/bin/bash
# Per-config evidence for profiles/: the bench line (roofline + cpu_baseline), rocprofv3 kernel-trace stats and the PMC HBM traffic
# (FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes) of the SAME bench command; plus the FETCH_SIZE /
# WRITE_SIZE calibration on known byte counts (scripts/fetchcal.hip).   usage: gpu_profiles.sh [configs...]   (GSTAMD_PROF_NO_PMC=1: bench line + kernel stats only)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/prof; R=$GRAFT_REPO_ROOT/gpurun_out/prof
CONFIGS="${@:-c2 c3 c4 c5 c4audio c1}"
for c in $CONFIGS; do
  timeout 600 python bench.py --config $c > $R/bench_$c.json 2> $R/bench_$c.err
  BENCH="python $GRAFT_REPO_ROOT/bench.py --config $c --no-cpu-baseline --no-secondary"   # (the secondary legs launch the same kernel on single frames: they would mix into its average)
  ( cd /tmp; rm -rf /tmp/p_$c
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$c/trace -o t -- $BENCH > /tmp/p_$c.log 2>&1
    [ -n "$GSTAMD_PROF_NO_PMC" ] ||
    timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_$c/fetch -o t -- $BENCH >> /tmp/p_$c.log 2>&1
    [ -n "$GSTAMD_PROF_NO_PMC" ] ||
    timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_$c/write -o t -- $BENCH >> /tmp/p_$c.log 2>&1 )
  f=$(find /tmp/p_$c/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/${c}_kernel_stats.csv
  python3 - $c /tmp/p_$c $R <<'PY'
import csv, glob, json, sys, collections
c, d, out = sys.argv[1:4]
res = {"config": c}
def counters(kind, name):
    f = glob.glob("%s/%s/**/*counter_collection.csv" % (d, kind), recursive=True)
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            if r.get("Counter_Name") == name:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
fe, wr = counters("fetch", "FETCH_SIZE"), counters("write", "WRITE_SIZE")
for k in sorted(set(fe) | set(wr)):
    if "elementwise" in k or "fill" in k.lower() and "border" not in k:
        continue
    res.setdefault("kernels", {})[k[:90]] = {"FETCH_SIZE_KB_avg": fe.get(k, (None, 0))[0], "WRITE_SIZE_KB_avg": wr.get(k, (None, 0))[0],
                                              "launches": max(fe.get(k, (0, 0))[1], wr.get(k, (0, 0))[1])}
json.dump(res, open("%s/pmc_%s.json" % (out, c), "w"), indent=1)
print(json.dumps(res)[:1500])
PY
done
[ -n "$GSTAMD_PROF_NO_PMC" ] && exit 0
( cd /tmp; rm -rf /tmp/p_cal
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_cal/fetch -o t -- $GRAFT_REPO_ROOT/scripts/fetchcal > /tmp/p_cal.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_cal/write -o t -- $GRAFT_REPO_ROOT/scripts/fetchcal >> /tmp/p_cal.log 2>&1 )
python3 - /tmp/p_cal $R <<'PY'
import csv, glob, json, sys, collections
d, out = sys.argv[1:3]
known = 512 << 20
res = {"known_bytes_per_launch": known}
for kind, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = glob.glob("%s/%s/**/*counter_collection.csv" % (d, kind), recursive=True)
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            if r.get("Counter_Name") == name:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        kb = sum(v) / len(v)
        res.setdefault(name, {})[k[:60]] = {"counter_KB": kb, "bytes_reported": kb * 1024, "reported_over_known": kb * 1024 / known}
json.dump(res, open("%s/fetchcal.json" % out, "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
