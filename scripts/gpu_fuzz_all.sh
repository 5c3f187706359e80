#!/bin/bash
# Round 6, parity at scale on the device: the whole GPU suite, then fresh seeds of every fuzz
#   video converter    GSTAMD_FUZZ_SEEDS   1200 seeds x 150 draws = 180 000 (announced draws CHECKED: staged / masked / counted, scripts/fuzz_tally.py)
#   compositor         GSTAMD_COMP_SEEDS   1701 seeds x 12 scenes = 20 412 over aggregate / _opaque / _scaled / _frame, 1 .. 100 pads
#   audio resampler    GSTAMD_AUDIO_SEEDS   300 seeds, groups of 1 .. 70 streams (>= 5000 streams) through resample_many with update events
# logs and tallies under gpurun_out/r06/ (copied to profiles/r06/).  bash scripts/gpu_fuzz_all.sh [quick | final]
# (final: fresh seeds on the round's last tree - 75 000 converter draws over the 129-format table, 6 000 compositor scenes, 100 resampler seeds)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out/r06; mkdir -p $O; rm -f $O/*tally*.jsonl
W=${FUZZ_WORKERS:-6}
if [ "$1" = quick ]; then V=70001-70040; CS=200000-200039; AS=300000-300019; elif [ "$1" = final ]; then V=81001-81500; CS=210000-210499; AS=310000-310099; else V=70001-70600,5101-5700; CS=200000-201700; AS=300000-300299; fi
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider -n $W > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -n 4 $O/pytest_gpu.log
GSTAMD_FUZZ_TALLY=$O/video_tally.jsonl GSTAMD_FUZZ_SEEDS=$V timeout 3000 python -m pytest tests/test_video_fuzz.py -m gpu -q -p no:cacheprovider -n $W -k test_hip_random > $O/fuzz_video_gpu.log 2>&1
tail -n 3 $O/fuzz_video_gpu.log; python scripts/fuzz_tally.py $O/video_tally.jsonl | tee $O/fuzz_video_gpu_tally.txt
GSTAMD_FUZZ_TALLY=$O/comp_tally.jsonl GSTAMD_COMP_SEEDS=$CS timeout 3000 python -m pytest tests/test_compositor_fuzz.py -m gpu -q -p no:cacheprovider -n $W -k every_entry > $O/fuzz_compositor_gpu.log 2>&1
tail -n 3 $O/fuzz_compositor_gpu.log; python scripts/fuzz_tally.py $O/comp_tally.jsonl | tee $O/fuzz_compositor_gpu_tally.txt
GSTAMD_FUZZ_TALLY=$O/audio_tally.jsonl GSTAMD_AUDIO_SEEDS=$AS timeout 3000 python -m pytest tests/test_audio_fuzz.py -m gpu -q -p no:cacheprovider -n $W > $O/fuzz_audio_gpu.log 2>&1
tail -n 3 $O/fuzz_audio_gpu.log; python scripts/fuzz_tally.py $O/audio_tally.jsonl | tee $O/fuzz_audio_gpu_tally.txt
