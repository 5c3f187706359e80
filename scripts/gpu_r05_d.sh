#!/bin/bash
# round 5: the default bench line with its secondary object (C2 one frame per launch, the element per buffer / lists / batches), with and
# without device-resident kernel arguments
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python bench.py > gpurun_out/r05_bench_c2_d.json 2> gpurun_out/r05_bench_c2_d.err; tail -1 gpurun_out/r05_bench_c2_d.json | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['frac']); print(json.dumps(j.get('secondary'), indent=1)); print(j.get('cpu_baseline'))"
HIP_FORCE_DEV_KERNARG=0 python bench.py --no-cpu-baseline > gpurun_out/r05_bench_c2_d_hostkernarg.json 2>/dev/null; tail -1 gpurun_out/r05_bench_c2_d_hostkernarg.json | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('host kernarg:', j['value'], j['ms_per_step'], j['roofline']['frac']); print(json.dumps(j.get('secondary'), indent=1))"
