#!/bin/bash
# more counter passes for one benchmark command (issue / LDS / texture path):  scripts/gpu_pmc2.sh <kernel-substring> <cmd...>
export TMPDIR=/tmp
K="$1"; shift
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for set in "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_LDS_UNALIGNED_STALL" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_BUSY_CU_CYCLES SQ_CYCLES" \
           "TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TD_TD_BUSY_sum"; do
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
    print("%-34s avg/launch %16.1f  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
done
