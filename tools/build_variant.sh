#!/bin/bash
# Build a variant of libdrt_hip.so into build/<name>.so with extra compiler flags (A/B experiments, debug counters).
# Usage: tools/build_variant.sh <name> [-DFLAG ...]
set -eu
N=$1; shift
mkdir -p build
# (the device headers as string literals for hiprtc: the same step the Makefile and build_native() run first)
python3 differentiable-renderer_amd/csrc/embed_sources.py
hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -shared -Iinclude "$@" \
      -o build/$N.so differentiable-renderer_amd/csrc/drt_hip.hip -lrccl -lhiprtc
echo "build/$N.so"
