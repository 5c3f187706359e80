#!/bin/bash
# Round 6 evidence: bench line + rocprofv3 kernel stats + PMC traffic of the headline and the configs VERDICT r05 names (scripts/gpu_profiles.sh), then the SQ
# counters of C3's k_scale_col (what binds it after the register-window change?) and of C4-A's k_aggregate_walk.
cd "$GRAFT_REPO_ROOT"
timeout 2400 bash scripts/gpu_profiles.sh c2 c3 c5 c4a c4 c1 c4audio 2>&1 | tail -5
mkdir -p gpurun_out/r06; cp gpurun_out/prof/* gpurun_out/r06/ 2>/dev/null
timeout 900 bash scripts/gpu_sq.sh c3 c3 2>&1 | tail -3
timeout 900 bash scripts/gpu_sq.sh c4a c4a 2>&1 | tail -3
cp gpurun_out/sq_c3.json gpurun_out/sq_c4a.json gpurun_out/r06/ 2>/dev/null
ls gpurun_out/r06 | wc -l
