#!/bin/bash
# bench.py sweep: "K variant extra-args" triples, one per line in $RUNS (semicolon separated)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
if [ -n "$PARITY" ]; then timeout 600 python -m pytest tests/test_video_gpu.py -m gpu -x -q > gpurun_out/pytest_video.log 2>&1; tail -3 gpurun_out/pytest_video.log; fi
: > gpurun_out/c2_variants.log
run() {
  GSTAMD_WIDE_K="$1" GSTAMD_FAST_VARIANT="$2" python bench.py --steps ${STEPS:-400} --warmup 20 --no-cpu-baseline $3 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('K=%-3s variant=%-10s %-12s us/launch=%8.3f us/frame=%6.3f GB/s=%8.1f frac=%.3f fps=%9.1f' % ('$1', '$2', '$3', d['roofline']['avg_launch_us'], d['roofline']['avg_launch_us'] / d['config']['frames_per_launch'], d['roofline']['achieved'], d['roofline']['frac'], d['value']))
    elif 'rror' in l: print(l.strip())
" >> gpurun_out/c2_variants.log
}
IFS=';' read -ra R <<< "$RUNS"
for r in "${R[@]}"; do set -- $r; k=$1; v=$2; shift; shift; [ "$v" = "-" ] && v=""; run "$k" "$v" "$*"; done
cat gpurun_out/c2_variants.log
