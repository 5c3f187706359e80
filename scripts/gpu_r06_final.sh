#!/bin/bash
# end-of-round validation on the device: the whole GPU suite, fresh video fuzz seeds (10-bit sources weigh in through the format pool), kernel stats of the
# new kernels
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -n 4 > $O/pytest_gpu_final2.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_final2.log; tail -3 $O/pytest_gpu_final2.log
( cd /tmp; rm -rf /tmp/dpk; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dpk -o t -- python $GRAFT_REPO_ROOT/scripts/deep_pack_probe.py 0 10 > /tmp/dpk.log 2>&1; f=$(find /tmp/dpk -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/deep_pack_kernel_stats.csv && head -4 $f | cut -d, -f1-4 | cut -c1-200 )
bash scripts/gpu_r06_deep_pack_sq.sh > /dev/null 2>&1; cat $O/sq_deep_pack.json | head -40
