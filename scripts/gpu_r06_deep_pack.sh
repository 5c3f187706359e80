#!/bin/bash
# k_deep_scale_pack: parity + timing against the multi-launch composite it replaces, then the P010 / 10-bit goldens and the device fuzz
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06; mkdir -p $O
timeout 600 python scripts/deep_pack_probe.py > $O/deep_pack_probe.log 2>&1; echo "rc=$?" >> $O/deep_pack_probe.log
GSTAMD_NO_DEEP_SCALE_PACK=1 timeout 600 python scripts/deep_pack_probe.py > $O/deep_pack_probe_before.log 2>&1
grep -v amdgpu.ids $O/deep_pack_probe.log; echo ---- before; grep -v amdgpu.ids $O/deep_pack_probe_before.log
