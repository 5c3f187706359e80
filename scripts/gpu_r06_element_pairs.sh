#!/bin/bash
# the videoconvertscale ELEMENT (GstHarness, HBM buffers: plugins/tests/bench_element) on the pairs of DESIGN 12.11 - 12.15: per buffer and in buffer lists of 8
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06; mkdir -p $O; : > $O/element_pairs.jsonl
export GST_PLUGIN_PATH=$GRAFT_REPO_ROOT/plugins:/opt/conda/lib/gstreamer-1.0 GST_PLUGIN_SYSTEM_PATH=/nonexistent GST_REGISTRY=/tmp/gstamd_pairs_registry.bin GST_REGISTRY_FORK=no GSTAMD_ELEMENT_STATS=0
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/gstreamer_amd/lib:$LD_LIBRARY_PATH
[ -f /usr/lib/x86_64-linux-gnu/libstdc++.so.6 ] && export LD_PRELOAD=/usr/lib/x86_64-linux-gnu/libstdc++.so.6
for p in "P010_10LE 3840 2160 NV12 1920 1080" "P010_10LE 3840 2160 BGRA 1920 1080" "P010_10LE 3840 2160 P010_10LE 1920 1080" "NV12 3840 2160 I420 1920 1080" "P010_10LE 3840 2160 I420_10LE 3840 2160" "NV12 3840 2160 I420_10LE 3840 2160"; do
  for ln in 1 8; do
    timeout 120 plugins/tests/bench_element $p 320 1 bilinear $ln 1 2>/dev/null | grep "^{" | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); d['pair'] = '$p'; d['buffers_per_list'] = $ln; print(json.dumps(d))" >> $O/element_pairs.jsonl
  done
done
python3 -c "
import json
for l in open('$O/element_pairs.jsonl'):
    d = json.loads(l); print(d['pair'], '| lists of', d['buffers_per_list'], '|', d['us_per_frame'], 'us per frame', round(d['algorithmic_gb_per_s'] / 8000, 3))
"
