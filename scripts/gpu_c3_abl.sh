#!/bin/bash
# C3 horizontal-pass ablations: GSTAMD_ABLATE bits 0x100 (no filter phase) / 0x200 (no staging)
cd "$GRAFT_REPO_ROOT"
for a in 0 256 512; do
  echo "== ablate $a"
  GSTAMD_ABLATE=$a bash scripts/gpu_prof_one.sh c3 2>&1 | grep avg_us
done
