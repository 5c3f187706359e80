#!/bin/bash
# round 5, first session: the whole GPU suite on the tree with the rectangle / GBR / pending-reads fixes, then this box's C3 / C4A / C2 lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r05_pytest_gpu_a.log 2>&1
tail -5 gpurun_out/r05_pytest_gpu_a.log
for c in c2 c3 c4a; do timeout 300 python bench.py --config $c > gpurun_out/r05_bench_${c}_a.json 2> gpurun_out/r05_bench_${c}_a.err; tail -1 gpurun_out/r05_bench_${c}_a.json | cut -c1-400; done
