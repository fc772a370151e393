#!/bin/bash
# Rebuild score_i8.o with the given -D flags on the GPU box and run tools/diag/sweep_i8.sh under a label.
# usage: tools/diag/sweep_i8_build.sh <label> "<-D flags>"
set -e
( cd phamers_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result $2 -c score_i8.hip -o score_i8.o && make -s ) 2>&1 | grep -E "error" || true
tools/diag/sweep_i8.sh "$1"
