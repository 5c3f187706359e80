#!/bin/bash
# round 3, call C: C4 variants (cache policy of the pad requests) on the tuning build, SQ counters of the product kernel
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3c; R=$GRAFT_REPO_ROOT/gpurun_out/r3c
timeout 120 python -m pytest tests/test_compositor.py tests/test_compositor_fuzz.py -m gpu -q -x > $R/pytest_comp.log 2>&1; rc=$?; tail -n 2 $R/pytest_comp.log
[ $rc -ne 0 ] && exit 1
for rep in 1 2; do
for nt in 0 1; do
  GSTAMD_TUNING_LIB=1 GSTAMD_AGG_NT=$nt timeout 120 python bench.py --config c4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('nt=$nt', d['ms_per_step']/d['config']['frames_per_step']*1000, 'us', d['roofline'])" | tee -a $R/c4_variants.log
done; done
bash scripts/gpu_sq.sh c4direct c4 > $R/sq.log 2>&1; cp gpurun_out/sq_c4direct.json $R/ 2>/dev/null; cat $R/sq_c4direct.json | head -40
