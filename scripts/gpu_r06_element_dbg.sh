cd "$GRAFT_REPO_ROOT"
export GST_PLUGIN_PATH=$GRAFT_REPO_ROOT/plugins:/opt/conda/lib/gstreamer-1.0 GST_PLUGIN_SYSTEM_PATH=/nonexistent GST_REGISTRY=/tmp/gstamd_pairs_registry.bin GST_REGISTRY_FORK=no GSTAMD_ELEMENT_STATS=0
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/gstreamer_amd/lib:$LD_LIBRARY_PATH TMPDIR=/tmp
[ -f /usr/lib/x86_64-linux-gnu/libstdc++.so.6 ] && export LD_PRELOAD=/usr/lib/x86_64-linux-gnu/libstdc++.so.6
for p in "NV12 3840 2160 I420 1920 1080" "P010_10LE 3840 2160 NV12 1920 1080"; do
  GST_DEBUG=GST_PERFORMANCE:5,amd*:4 GST_DEBUG_NO_COLOR=1 timeout 100 plugins/tests/bench_element $p 4 1 bilinear 1 1 2>&1 | grep "HIP plan" | head -3 | cut -c1-400
done
