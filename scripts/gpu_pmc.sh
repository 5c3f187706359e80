#!/bin/bash
# SQ counter passes for one benchmark command:  scripts/gpu_pmc.sh <kernel-substring> <cmd...>
export TMPDIR=/tmp
K="$1"; shift
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
           "SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE"; do
  rm -rf /tmp/pmc
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc -o p -- "$@" > /tmp/pmc.log 2>&1)
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "no counters for: $set"; tail -3 /tmp/pmc.log; continue; }
  python - "$f" "$K" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r.get("Kernel_Name", ""):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-28s avg/launch %16.1f  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
done
