#!/bin/bash
# round 3, call A: the C4 access-pattern probe, the GPU suite (new plane-canvas fuzz), baseline bench lines of this box
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3a; R=gpurun_out/r3a
rocm-smi --showuse --showmemuse > $R/smi.log 2>&1
timeout 60 python -c "import torch; x=torch.ones(1<<20,device='cuda'); print('torch ok', float(x.sum()))" > $R/torch_ok.log 2>&1
timeout 300 scripts/c4_probe 200 > $R/c4_probe.jsonl 2> $R/c4_probe.err; echo "probe exit $?" >> $R/c4_probe.err
timeout 900 python -m pytest tests -m gpu -q -x > $R/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $R/pytest_gpu.log
for c in c4 c3 c5 c2; do
  timeout 300 python bench.py --config $c --no-cpu-baseline > $R/bench_$c.json 2> $R/bench_$c.err
done
cat $R/torch_ok.log; tail -n 3 $R/pytest_gpu.log; cat $R/c4_probe.jsonl | cut -c1-200; cat $R/bench_c*.json | cut -c1-300
