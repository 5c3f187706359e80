#!/bin/bash
# C3 horizontal pass, k_hscale420_reg: lines per wave x request order.  gpu_c3_reg.sh "rows list" "late list"
cd "$GRAFT_REPO_ROOT"
for r in ${1:--1}; do
  for l in ${2:-1}; do
    echo "== lines/wave $r late $l"
    GSTAMD_H420_ROWS=$r GSTAMD_H420_LATE=$l bash scripts/gpu_prof_one.sh c3 2>&1 | grep avg_us
  done
done
