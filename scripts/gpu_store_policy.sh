#!/bin/bash
# Round 6: how k_convert_strip's pixels leave the CU (video_fast.h store16_policy), per launch size: the headline (32-frame lists), one frame per launch
# at the C ABI and the element rows of bench.py's `secondary`, for every policy.  bash scripts/gpu_store_policy.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out/r06; mkdir -p $O
for p in 0 1 2 3 4; do
  GSTAMD_STORE_POLICY=$p timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/store_policy_$p.json
  python - $p $O/store_policy_$p.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[2]).read()); s=j.get("secondary",{})
print("policy",sys.argv[1],"list32 us/frame %.3f frac %.4f | one frame per launch %s us | element:" % (j["ms_per_step"]*1e3/32, j["roofline"]["frac"], s.get("c_abi_one_frame_per_launch",{}).get("us_per_frame")),
      [ (r.get("case","")[:22], r.get("us_per_frame")) for r in s.get("element",[]) if isinstance(r,dict)])
PY
done
