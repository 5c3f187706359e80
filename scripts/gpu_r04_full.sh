#!/bin/bash
# round 4 evidence: the whole GPU suite, then bench line + kernel stats + PMC traffic of every config
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r04_pytest_gpu.log 2>&1
tail -5 gpurun_out/r04_pytest_gpu.log
bash scripts/gpu_profiles.sh c2 c3 c5 c1 c4 c4a f8scale f8pack f8swizzle f2gamma f2p010in f2p010out f5encode16 c4audio > gpurun_out/r04_profiles.log 2>&1
tail -3 gpurun_out/r04_profiles.log | cut -c1-300
for c in c2 c3 c5 c1 c4 c4a f8scale f8pack f8swizzle f2gamma f2p010in f2p010out f5encode16 c4audio; do grep -h '^{' gpurun_out/prof/bench_$c.json | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('$c', j['value'], j['unit'], j['ms_per_step'], j['roofline']['frac'])
"; done
