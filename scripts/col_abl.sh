#!/bin/bash
# k_scale_col ablations (profiling builds libgstamddsp_colabl<n>.so, results wrong on purpose): time per 8K frame with one part of the kernel left out
cd "$GRAFT_REPO_ROOT"
export GSTAMD_COL_OPL=${GSTAMD_COL_OPL:-1} GSTAMD_COL_WAVES=${GSTAMD_COL_WAVES:-2}
for v in "" colabl1 colabl2 colabl3 colabl4 colabl5; do
  if [ -n "$v" ]; then export GSTAMD_LIB_PATH=$PWD/gstreamer_amd/lib/libgstamddsp_$v.so; else unset GSTAMD_LIB_PATH; fi
  for b in 1 4; do
    python bench.py --config c3 --no-cpu-baseline --steps 40 --warmup 8 --batch $b 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-10s batch %d  us/frame %.2f' % ('$v' or 'product', $b, 1e6 / j['value']))"
  done
done
