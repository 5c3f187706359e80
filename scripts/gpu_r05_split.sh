#!/bin/bash
# per-kernel split of the slowest item-6 pairs: YUY2 4K -> NV12 1080p bilinear (case 12), P010 4K -> NV12 1080p bilinear (case 8), UYVY -> I420 linear (11)
set -u
cd "$(dirname "$0")/.."
R=$PWD; O=gpurun_out/r05split; mkdir -p $O
export TMPDIR=/tmp
for c in ${CASES:-12 8 11}; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$c -o s -- python $R/scripts/survey_item6.py $c > $R/$O/run_$c.log 2>&1)
  f=$(find $O/prof_$c -name "*kernel_stats.csv" | head -1)
  echo "== case $c"; tail -n 1 $O/run_$c.log | cut -c1-230
  [ -n "$f" ] && head -8 "$f" | cut -d, -f1-5 | cut -c1-200
  [ -n "$f" ] && cp "$f" $O/case${c}_kernel_stats.csv
  rm -rf $O/prof_$c
done
