#!/bin/bash
# SQ counters of k_deep_scale_pack on case 0 of scripts/deep_pack_probe.py (P010 4K -> NV12 1080p bilinear): two PMC passes
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $R
CMD="python $GRAFT_REPO_ROOT/scripts/deep_pack_probe.py 0"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
P3="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH"
cd /tmp; rm -rf /tmp/sqd
timeout 300 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d /tmp/sqd/p1 -o t -- $CMD > /tmp/sqd.log 2>&1
timeout 300 rocprofv3 --pmc $P3 --kernel-trace --output-format csv -d /tmp/sqd/p3 -o t -- $CMD >> /tmp/sqd.log 2>&1
python3 - /tmp/sqd $R/sq_deep_pack.json <<'PY'
import csv, glob, json, sys, collections
d, out = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in acc.items():
    if "deep_scale_pack" not in k:
        continue
    res[k] = {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}
    res[k]["launches"] = max(len(v) for v in cs.values())
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
tail -3 /tmp/sqd.log
