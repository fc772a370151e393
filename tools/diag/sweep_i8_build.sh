#!/bin/bash
# Rebuild score_i8.o as a DIAGNOSTIC build with the given -D flags on the GPU box and run tools/diag/sweep_i8.sh under a label.
# (A diagnostic library reports another ABI version; the bench loads it only because this script says so.)
# usage: tools/diag/sweep_i8_build.sh <label> "<-D flags>"
set -e
( cd phamers_amd/csrc && touch score_i8.hip phk_api.hip && make -s EXTRA_CXXFLAGS="-DPHK_DIAGNOSTIC_BUILD $2" ) 2>&1 | grep -E "error" || true
PHK_ALLOW_DIAGNOSTIC_BUILD=1 tools/diag/sweep_i8.sh "$1"
