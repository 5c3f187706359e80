#!/bin/bash
# round 5: many audio streams per launch - bench line + kernel time from rocprofv3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05g
python bench.py --config c4audiomany --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r05g/bench_c4audiomany.json
cat gpurun_out/r05g/bench_c4audiomany.json
rocprofv3 --kernel-trace --stats -d gpurun_out/r05g/prof -o many -- python bench.py --config c4audiomany --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r05g/prof/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]:
        print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
