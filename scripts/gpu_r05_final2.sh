#!/bin/bash
# round 5, after the last kernel changes (SrcLean, k_bilinear422_rows, dot4 from packed 4:2:2): the whole GPU suite again, 2000 fresh device fuzz seeds
# (300 000 draws over the 118-format table), the default bench line, smoke, the item-6 survey
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05g
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r05g/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05g/pytest_gpu.log; tail -n 3 gpurun_out/r05g/pytest_gpu.log
GSTAMD_FUZZ_SEEDS=52001-54000 timeout 2400 python -m pytest tests/test_video_fuzz.py -m gpu -q -p no:cacheprovider -x > gpurun_out/r05g/fuzz_gpu_2000_seeds.log 2>&1
tail -n 3 gpurun_out/r05g/fuzz_gpu_2000_seeds.log
timeout 400 python bench.py 2>gpurun_out/r05g/bench_default.err > gpurun_out/r05g/bench_default.json; cut -c1-260 gpurun_out/r05g/bench_default.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python scripts/survey_item6.py > gpurun_out/r05g/survey_item6_end.log 2>&1; grep -- "->" gpurun_out/r05g/survey_item6_end.log | cut -c1-150
