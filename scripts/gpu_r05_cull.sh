#!/bin/bash
# round 5: opaque culling in the compositor + AV12 on the device
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r05cull; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_compositor.py tests/test_compositor_fuzz.py -m gpu -x -q > $O/pytest_compositor.log 2>&1; echo "compositor rc=$?" >> $O/pytest_compositor.log
timeout 900 python -m pytest tests/test_video_gpu.py -m gpu -x -q -k "av12 or AV12" > $O/pytest_av12.log 2>&1; echo "av12 rc=$?" >> $O/pytest_av12.log
timeout 900 python -m pytest tests/test_plugin_gpu.py -m gpu -x -q -k "compositor or round5" > $O/pytest_plugin.log 2>&1; echo "plugin rc=$?" >> $O/pytest_plugin.log
for cfg in "c4" "c4opaque --opaque-hint none" "c4opaque --opaque-hint map" "c4opaque --opaque-hint all"; do
  n=$(echo $cfg | tr -d ' -')
  timeout 300 python bench.py --config $cfg > $O/bench_$n.json 2> $O/bench_$n.err
done
R=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_c4opaque -o c4opaque -- python $R/bench.py --config c4opaque --steps 30 --warmup 5 > $R/$O/prof.log 2>&1)
find $O/prof_c4opaque -name "*kernel_stats.csv" -exec cp {} $O/c4opaque_kernel_stats.csv \; ; rm -rf $O/prof_c4opaque
for f in $O/pytest_*.log; do tail -n 3 $f; done; for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get("roofline",{})
    print(sys.argv[1].split("/")[-1], d["value"], d["unit"], "launch_us", round(d["ms_per_step"]*1000/d["config"].get("frames_per_step",1),2), "frac", r.get("frac"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
