#!/bin/bash
# round 3, call E: C3 ablations, compile-time variants of the fused scaler (python -m gstreamer_amd.build -DGSTAMD_FUSED_ABL=n --suffix=abln)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3e; R=$GRAFT_REPO_ROOT/gpurun_out/r3e
rm -f $R/c3_ablation.log
run () { python bench.py --config c3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['roofline']['avg_launch_us'], 'us per launch;', round(d['ms_per_step'] / d['config']['frames_per_step'] * 1000, 2), 'us per frame')" | tee -a $R/c3_ablation.log; }
run product
for a in 1 2 3 4 5; do GSTAMD_LIB_PATH=$GRAFT_REPO_ROOT/gstreamer_amd/lib/libgstamddsp_abl$a.so run abl$a; done
run product_again
