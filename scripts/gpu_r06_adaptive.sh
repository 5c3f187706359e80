#!/bin/bash
# Round 6: the store policy picked per launch size (video_kernels.hip launch_convert_pair) and the element's adaptive batch-buffers=0,
# measured by bench.py's headline + `secondary`; then the plugin tests that cover deferred launches.  bash scripts/gpu_r06_adaptive.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out/r06; mkdir -p $O
for i in 1 2; do
  timeout 400 python bench.py --no-cpu-baseline 2>$O/bench_adaptive_$i.err | tail -1 > $O/bench_adaptive_$i.json
  python - $O/bench_adaptive_$i.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read()); s=j.get("secondary",{})
print("list32 us/frame %.3f frac %.4f | one frame per launch %s us" % (j["ms_per_step"]*1e3/32, j["roofline"]["frac"], s.get("c_abi_one_frame_per_launch",{}).get("us_per_frame")))
for r in s.get("element",[]):
    print("   ", r.get("case","")[:70], r.get("us_per_frame"), r.get("frac"), r.get("error",""))
PY
done
timeout 900 python -m pytest tests/test_plugin_gpu.py -m gpu -q -p no:cacheprovider -x -k "deferred or batch or list" 2>&1 | tail -8
