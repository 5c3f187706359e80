#!/bin/bash
# Round 6, C4-A: what the column walk's extra traffic (1.47 x) is made of - time and FETCH_SIZE / WRITE_SIZE of k_aggregate_walk under the knobs that
# change who shares what in which L2 (GSTAMD_WALK_XCD: contiguous strips per XCD; GSTAMD_WALK_ROWS: rows per walk = the vertical halo's share)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06; mkdir -p $O
out=$O/walk_variants.txt; : > $out
run_one () {
  name="$1"; shift
  line=$(env "$@" timeout 300 python bench.py --config c4a --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1)
  us=$(python -c "import json,sys;d=json.loads(sys.argv[1]);print(round(d['ms_per_step']*1000,2), d['roofline']['frac'])" "$line")
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc
    (cd /tmp && env "$@" timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc -o p -- python $GRAFT_REPO_ROOT/bench.py --config c4a --no-cpu-baseline --steps 10 --warmup 2 > /tmp/pmc.log 2>&1)
    f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
    v=$(python - "$f" <<'PY'
import csv, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "k_aggregate_walk" in r.get("Kernel_Name", "")]
print(round(sum(v) / max(len(v), 1) / 1024, 2))
PY
)
    eval "$c=$v"
  done
  echo "$name | us/frame frac: $us | FETCH_SIZE MB raw $FETCH_SIZE (x2 = read) | WRITE_SIZE MB $WRITE_SIZE" | tee -a $out
}
run_one "default" A=1
run_one "xcd" GSTAMD_WALK_XCD=1
run_one "rows34" GSTAMD_WALK_ROWS=34
run_one "rows68" GSTAMD_WALK_ROWS=68
run_one "xcd+rows34" GSTAMD_WALK_XCD=1 GSTAMD_WALK_ROWS=34
run_one "xcd+rows24" GSTAMD_WALK_XCD=1 GSTAMD_WALK_ROWS=24
run_one "xcd+rows12" GSTAMD_WALK_XCD=1 GSTAMD_WALK_ROWS=12
timeout 600 python -m pytest tests/test_compositor.py tests/test_compositor_fuzz.py -m gpu -q -p no:cacheprovider -n 6 2>&1 | tail -3 | tee -a $out
