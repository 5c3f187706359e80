#!/bin/bash
# parity tests + bench + kernel-trace profile of the same bench command (CSV stats -> gpurun_out/prof)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 100 --warmup 10 ${BENCH_ARGS} > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
rm -rf gpurun_out/prof; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o c2 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 30 --warmup 3 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/bench_prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"; find gpurun_out/prof -type f | head; tail -3 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/bench.log; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f"
