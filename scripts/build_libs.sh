#!/bin/bash
# build the product and the tuning library from wherever the shell happens to be
cd "$(dirname "$0")/.." && python -c "
from gstreamer_amd import build
build.build(verbose=False); build.build(tuning=True, verbose=False)"
