#!/bin/bash
# the round's closing run: the whole GPU suite, a wider slice of the device fuzz, the default bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04_pytest_gpu.log 2>&1
tail -3 gpurun_out/r04_pytest_gpu.log
SEEDS=$(python -c "print(','.join(str(s) for s in list(range(20,140))+list(range(720,840))))")
GSTAMD_FUZZ_SEEDS=$SEEDS timeout 900 python -m pytest tests/test_video_fuzz.py -m gpu -q > gpurun_out/r04_fuzz_gpu_240_seeds.log 2>&1
tail -2 gpurun_out/r04_fuzz_gpu_240_seeds.log
timeout 300 python bench.py 2>/dev/null > gpurun_out/r04_bench_default.json; cut -c1-300 gpurun_out/r04_bench_default.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
