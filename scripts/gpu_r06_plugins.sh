#!/bin/bash
# Round 6: the plugin tests on the device (both runtimes), then a longer run of the interlaced device fuzz (GSTAMD_ILACE_SEEDS x 90 draws)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out/r06; mkdir -p $O; rm -f $O/ilace_tally.jsonl
timeout 1500 python -m pytest tests/test_plugin_gpu.py -m gpu -q -p no:cacheprovider -n 6 2>&1 | tail -25 | tee $O/pytest_plugin_gpu.log
GSTAMD_FUZZ_TALLY=$O/ilace_tally.jsonl GSTAMD_ILACE_SEEDS=${1:-100-399} timeout 2400 python -m pytest tests/test_video_interlaced.py -m gpu -q -p no:cacheprovider -n 6 -k random 2>&1 | tail -12 | tee $O/fuzz_interlaced_gpu.log
python - $O/ilace_tally.jsonl <<'PY' | tee $O/fuzz_interlaced_gpu_tally.txt
import json,sys
t={}
n=0
for l in open(sys.argv[1]):
    d=json.loads(l); n+=1
    for k,v in d.items():
        if k!="seed": t[k]=t.get(k,0)+v
print(n,"seeds:",t)
PY
