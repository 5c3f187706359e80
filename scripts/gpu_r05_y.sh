#!/bin/bash
# round 5: SQ counters of k_hscale_wave<SrcFront> (BGRA 4K -> NV12 1080p linear: the horizontal pass is 25 of the plan's 40 us)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05y
bash scripts/gpu_pmc.sh k_hscale_wave python $GRAFT_REPO_ROOT/scripts/survey_item6.py 5 > gpurun_out/r05y/hscale_wave_sq_counters.log 2>&1
cat gpurun_out/r05y/hscale_wave_sq_counters.log
