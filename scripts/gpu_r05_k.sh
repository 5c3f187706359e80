#!/bin/bash
# round 5: k_bilinear420_rows at 1.5:1 is 4-5x slower per output than at 2:1 - are the odd LDS byte offsets of its 16-bit pair reads the cost?
# variants: product / offsets forced even (wrong pixels, timing only) / two byte reads per pair
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05k
for v in "" _even _u8; do
  echo "== libgstamddsp$v" | tee -a gpurun_out/r05k/bilr_lds_align.log
  GSTAMD_LIB_PATH=gstreamer_amd/lib/libgstamddsp$v.so timeout 200 python scripts/survey_item6.py 0 1 3 2>&1 | grep -- "->" | tee -a gpurun_out/r05k/bilr_lds_align.log
done
