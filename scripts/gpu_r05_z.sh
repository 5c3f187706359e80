#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05z
timeout 300 python scripts/probe_col_semi.py 2>&1 | grep "us per frame" | tee gpurun_out/r05z/col_planar_vs_semi.log
