#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python -m pytest tests/test_audio_gpu.py -m gpu -q -x > $R/c_pytest_audio.log 2>&1; echo "exit $?" >> $R/c_pytest_audio.log; tail -3 $R/c_pytest_audio.log
for b in 1024 480000; do
  timeout 300 python bench.py --config c4audio --audio-block $b --no-cpu-baseline > $R/c_audio_$b.log 2>&1
  GSTAMD_NO_FIR_LDS=1 timeout 300 python bench.py --config c4audio --audio-block $b --no-cpu-baseline > $R/c_audio_old_$b.log 2>&1
  for f in $R/c_audio_$b.log $R/c_audio_old_$b.log; do grep -o '"value": [0-9.]*\|"avg_launch_us": [0-9.]*\|"gflops": [0-9.]*' $f | head -3 | tr '\n' ' '; echo " <- $f"; done
done
