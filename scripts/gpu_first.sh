#!/bin/bash
# first GPU call: parity tests, smoke, bench, kernel-trace profile
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
{ ls -la /opt/conda/lib/libglib-2.0.so* ; ldd oracle/_ref/libgstref.so | grep -i "not found"; rocm-smi --showproductname | head -8; nproc; } > gpurun_out/env.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 600 python bench.py --steps 50 --warmup 5 > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o c2 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/bench_prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"; find gpurun_out/prof -name "*stats*" | head; tail -3 gpurun_out/pytest_gpu.log; cat gpurun_out/bench.log | tail -2
