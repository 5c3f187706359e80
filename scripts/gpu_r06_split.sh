#!/bin/bash
# per-kernel split (rocprofv3 --kernel-trace --stats) of scripts/survey_item6.py cases: bash scripts/gpu_r06_split.sh 2 0 4
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O
for c in "$@"; do
  ( cd /tmp; rm -rf /tmp/sp_$c; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_$c -o t -- python $GRAFT_REPO_ROOT/scripts/survey_item6.py $c 2>&1 | grep -v "amdgpu.ids\|rocprofv3" | tail -1 )
  f=$(find /tmp/sp_$c -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp $f $O/split_case$c.csv; head -6 $f | cut -d, -f1-4 | cut -c1-170; fi
done
