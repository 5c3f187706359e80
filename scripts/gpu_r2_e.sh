#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT/gpurun_out
export GST_PLUGIN_PATH=$GRAFT_REPO_ROOT/plugins:/opt/conda/lib/gstreamer-1.0 GST_PLUGIN_SYSTEM_PATH=/nonexistent GST_REGISTRY=/tmp/reg.bin GST_REGISTRY_FORK=no
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/gstreamer_amd/lib:$LD_LIBRARY_PATH LD_PRELOAD=/usr/lib/x86_64-linux-gnu/libstdc++.so.6
B=$GRAFT_REPO_ROOT/plugins/tests/bench_element
cd /tmp; rm -rf $R/prof_el
for l in 32 4; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_el/l$l -o el -- $B NV12 3840 2160 BGRA 3840 2160 640 3 bilinear $l > $R/e_prof_l$l.log 2>&1
  tail -2 $R/e_prof_l$l.log | cut -c1-300
  f=$(find $R/prof_el/l$l -name "*kernel_stats.csv" | head -1); head -5 $f
  t=$(find $R/prof_el/l$l -name "*kernel_trace.csv" | head -1)
  python3 - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows)//2:]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
gaps = [(int(rows[i+1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3 for i in range(len(rows)-1)]
print("kernels", len(rows), "dur us (first 12):", [round(x,1) for x in d[:12]])
print("gaps us (first 12):", [round(x,1) for x in gaps[:12]])
print("grid z / names:", set((r["Kernel_Name"][:40], r.get("Grid_Size_Z", r.get("Grid_Size","?"))) for r in rows[:6]))
PY
done
