#!/bin/bash
# kernel durations (rocprofv3 kernel-trace stats) per variant, to separate GPU time from host launch rate
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; : > gpurun_out/variants_prof.log
for v in "" $VARIANTS; do
  rm -rf /tmp/vp; (cd /tmp && GSTAMD_FAST_VARIANT="$v" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vp -o v -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > /tmp/vp.log 2>&1)
  f=$(find /tmp/vp -name "*kernel_stats.csv" | head -1)
  echo "variant=$v $(grep -m1 k_convert "$f" | sed 's/.*)",//' | awk -F, '{print "calls="$1" avg_ns="$3" min="$5" max="$6}') $(grep -o '"avg_launch_us": [0-9.]*' /tmp/vp.log)" >> gpurun_out/variants_prof.log
done
cat gpurun_out/variants_prof.log
