#!/bin/bash
# the round's closing run: the whole GPU suite, then bench line + kernel stats + PMC traffic of the configs whose kernels changed late
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04_pytest_gpu.log 2>&1
tail -3 gpurun_out/r04_pytest_gpu.log
bash scripts/gpu_profiles.sh c5 c3 f8scale > gpurun_out/r04_profiles.log 2>&1
for c in c5 c3 f8scale; do cut -c1-400 gpurun_out/prof/bench_$c.json; done
timeout 300 python bench.py 2>/dev/null | cut -c1-300
