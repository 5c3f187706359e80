#!/bin/bash
# Round 6: the test source on the device: library tests, the element's tests, the config-1-shaped end-to-end record with the frames born in HBM
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_video_testsrc.py -q -p no:cacheprovider -n 6 2>&1 | tail -6 | tee $O/pytest_testsrc_gpu.log
timeout 900 python -m pytest tests/test_plugin_gpu.py -m gpu -q -p no:cacheprovider -n 6 -k "amdhipvideotestsrc" 2>&1 | tail -12 | tee -a $O/pytest_testsrc_gpu.log
timeout 900 python scripts/config1_e2e.py 600 > $O/config1.json 2> $O/config1.err; tail -c 1500 $O/config1.json; tail -3 $O/config1.err
