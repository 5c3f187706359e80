#!/bin/bash
# vertical pass variants on C3: GSTAMD_VSCALE_ROWS 1 / 2 / 4
cd "$GRAFT_REPO_ROOT"
for r in ${1:-1 2 4}; do
  echo "== vscale rows $r"
  GSTAMD_VSCALE_ROWS=$r bash scripts/gpu_prof_one.sh c3 2>&1 | grep avg_us
done
