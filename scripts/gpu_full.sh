#!/bin/bash
# gpu_round.sh (parity, smoke, bench, C2 kernel trace + PMC) followed by the secondary-config kernel stats
bash "$GRAFT_REPO_ROOT/scripts/gpu_round.sh"
bash "$GRAFT_REPO_ROOT/scripts/gpu_prof_configs.sh"
cd "$GRAFT_REPO_ROOT"; timeout 300 python scripts/bench_configs.py --iters 200 > gpurun_out/bench_configs.log 2>&1; tail -30 gpurun_out/bench_configs.log
