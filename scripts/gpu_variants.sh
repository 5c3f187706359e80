#!/bin/bash
# sweep kernel variants of the C2 fast path; prints avg_launch_us per variant
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/variants.log
for v in "" $VARIANTS; do
  GSTAMD_FAST_VARIANT="$v" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --batch ${BATCH:-1} ${EXTRA} 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('variant=%-12s batch=${BATCH:-1} us/launch=%8.3f GB/s=%8.1f fps=%9.1f' % ('$v', d['roofline']['avg_launch_us'], d['roofline']['achieved'], d['value']))
    elif 'Error' in l or 'error' in l: print(l.strip())
" >> gpurun_out/variants.log
done
cat gpurun_out/variants.log
