#!/bin/bash
# video parity + bench.py sweep over the wide kernel's pairs-per-wave ($KS) and extra variants ($VARIANTS)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_video_gpu.py -m gpu -x -q > gpurun_out/pytest_video.log 2>&1; tail -3 gpurun_out/pytest_video.log
: > gpurun_out/c2_variants.log
run() {
  GSTAMD_WIDE_K="$1" GSTAMD_FAST_VARIANT="$2" python bench.py --steps 50 --warmup 5 --no-cpu-baseline $3 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('K=%-3s variant=%-10s %-10s us/launch=%8.3f us/frame=%6.3f GB/s=%8.1f frac=%.3f fps=%9.1f' % ('$1', '$2', '$3', d['roofline']['avg_launch_us'], d['roofline']['avg_launch_us'] / d['config']['frames_per_launch'], d['roofline']['achieved'], d['roofline']['frac'], d['value']))
    elif 'rror' in l: print(l.strip())
" >> gpurun_out/c2_variants.log
}
for k in $KS; do run $k "" ""; done
for k in $KS_ABL; do run $k "1024,1" ""; done
for k in $KS_B1; do run $k "" "--batch 1"; done
run 2 4096 ""
cat gpurun_out/c2_variants.log
for b in $BATCHES; do run 1 "" "--batch $b"; run 1 4096 "--batch $b"; done
cat gpurun_out/c2_variants.log | tail -8
