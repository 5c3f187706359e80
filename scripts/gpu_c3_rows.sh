#!/bin/bash
# C3 horizontal pass (k_hscale420_dot4): rows per wave and ablations.  gpu_c3_rows.sh "rows list" "ablate list"
cd "$GRAFT_REPO_ROOT"
for r in ${1:-4}; do
  for a in ${2:-0}; do
    echo "== rows $r ablate $a"
    GSTAMD_H420_ROWS=$r GSTAMD_ABLATE=$a bash scripts/gpu_prof_one.sh c3 2>&1 | grep avg_us
  done
done
