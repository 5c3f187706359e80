#!/bin/bash
g++ -O2 -std=c++17 -fPIC -shared -ffp-contract=off -o tests/emu/libgstamdemu.so tests/emu/*.cpp gstreamer_amd/csrc/planner.cpp gstreamer_amd/csrc/audio_taps.cpp
