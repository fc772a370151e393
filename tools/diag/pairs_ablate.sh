#!/bin/bash
# ON THE GPU BOX: the two-windows-per-add count kernel with one of its parts removed (DIAGNOSTIC builds: counts wrong, timing
# only; the library then reports another ABI version and is loaded only because this script says so).
# usage: tools/diag/pairs_ablate.sh <out file> ; afterwards the product build is restored
set -e
out=${1:-gpurun_out/pairs_ablate.txt}
cd phamers_amd/csrc
for abl in 1 2 3; do
    touch count.hip phk_api.hip
    make -s -j8 EXTRA_CXXFLAGS="-DPHK_DIAGNOSTIC_BUILD -DPAIRS_ABL=$abl" 2>&1 | grep -E "error" || true
    for l in p P; do
        echo "ABL=$abl (1 no LDS adds, 2 rows stored for 1 batch in 64, 3 all loads from two cached words) lanes=$l" >> ../../$out
        ( cd ../.. && PHK_ALLOW_DIAGNOSTIC_BUILD=1 python tools/bench_count.py --lanes $l --iters 20 --check 0 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:round(v,4) for k,v in d['per_kernel_ms'].items() if v>0.01})" >> $out )
    done
done
touch count.hip phk_api.hip
make -s -j8 2>&1 | grep -E "error" || true
