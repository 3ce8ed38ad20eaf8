#!/bin/bash
# builds tests/emu/_build/libemu.so: the HIP kernel sources compiled for the host SIMT interpreter
set -e
cd "$(dirname "$0")"
mkdir -p _build
g++ -O1 -g -std=c++17 -fPIC -shared -x c++ -I. -I../../ema_amd/csrc -I../../include \
    -Wno-unknown-pragmas -ldl -o _build/libemu.so harness.cpp ../../ema_amd/csrc/host_index.cpp
