#!/bin/bash
# SQ counters of one bench config: usage gpu_sq.sh <tag> <config> [VAR=value ...]   (two PMC passes of 8 SQ counters each)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT/gpurun_out
tag=$1; cfg=$2; shift 2
for kv in "$@"; do export "$kv"; done
BENCH="python $GRAFT_REPO_ROOT/bench.py --config $cfg --no-cpu-baseline --steps 20 --warmup 5"
[ -f $R/sq_counters_available.txt ] || (cd /tmp; rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $R/sq_counters_available.txt)
cd /tmp; rm -rf /tmp/sq_$tag
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD"
P3="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES"
timeout 300 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d /tmp/sq_$tag/p1 -o t -- $BENCH > /tmp/sq_$tag.log 2>&1
timeout 300 rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d /tmp/sq_$tag/p2 -o t -- $BENCH >> /tmp/sq_$tag.log 2>&1
timeout 300 rocprofv3 --pmc $P3 --kernel-trace --output-format csv -d /tmp/sq_$tag/p3 -o t -- $BENCH >> /tmp/sq_$tag.log 2>&1
python3 - /tmp/sq_$tag $R/sq_$tag.json <<'PY'
import csv, glob, json, sys, collections
d, out = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in acc.items():
    if "elementwise" in k or "roll" in k or "copyBuffer" in k or "fill" in k.lower():
        continue
    res[k] = {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}
    res[k]["launches"] = max(len(v) for v in cs.values())
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1)[:4000])
PY
tail -3 /tmp/sq_$tag.log
