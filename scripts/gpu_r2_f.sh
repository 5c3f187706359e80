#!/bin/bash
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
export GST_PLUGIN_SYSTEM_PATH=/nonexistent GST_REGISTRY_FORK=no
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/gstreamer_amd/lib:$LD_LIBRARY_PATH LD_PRELOAD=/usr/lib/x86_64-linux-gnu/libstdc++.so.6
for order in "/opt/conda/lib/gstreamer-1.0:$GRAFT_REPO_ROOT/plugins" "$GRAFT_REPO_ROOT/plugins:/opt/conda/lib/gstreamer-1.0"; do
  rm -f /tmp/regx.bin
  echo "=== $order"
  GST_REGISTRY=/tmp/regx.bin GST_PLUGIN_PATH=$order /opt/conda/bin/gst-inspect-1.0 videoconvert 2>&1 | head -24
  GST_REGISTRY=/tmp/regx.bin GST_PLUGIN_PATH=$order /opt/conda/bin/gst-inspect-1.0 2>&1 | grep -i "videoconvert\|audioresample\|videoscale"
done
