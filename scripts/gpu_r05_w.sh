#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05w
timeout 300 python scripts/survey_item6.py 4 5 7 13 2>&1 | grep -- "->" | cut -c1-230 | tee gpurun_out/r05w/survey_scratch_lists.log
