#!/bin/bash
# C2 iteration loop: video parity tests, then bench.py for the shipped kernel and the variants in $VARIANTS
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_video_gpu.py -m gpu -x -q > gpurun_out/pytest_video.log 2>&1; tail -3 gpurun_out/pytest_video.log
: > gpurun_out/c2_variants.log
for v in "" $VARIANTS; do
  GSTAMD_FAST_VARIANT="$v" python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('variant=%-12s us/launch=%8.3f us/frame=%6.3f GB/s=%8.1f frac=%.3f fps=%9.1f' % ('$v', d['roofline']['avg_launch_us'], d['roofline']['avg_launch_us'] / d['config']['frames_per_launch'], d['roofline']['achieved'], d['roofline']['frac'], d['value']))
    elif 'rror' in l: print(l.strip())
" >> gpurun_out/c2_variants.log
done
cat gpurun_out/c2_variants.log
