#!/bin/bash
# ON THE GPU BOX: the k = 4 sweep (phk_knn_f16h_kernel) with parts of its list maintenance removed (DIAGNOSTIC builds: scores
# wrong, timing only) -- what the sized candidates of DESIGN.md 7 could gain AT MOST, measured on one box before building them:
#   F16H_ABL=1  no bias multiply-add per value (the bias as a 17th k-step of the MFMA would remove it, at +1/16 MFMAs)
#   F16H_ABL=2  the ids settled every third block instead of every block
#   F16H_ABL=3  both
# usage: tools/diag/f16h_ablate.sh <out file> ; afterwards the product build is restored
out=${1:-gpurun_out/f16h_ablate.txt}
run() {
    python bench.py --min-seconds 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('$1', 'step %.4f ms' % d['ms_per_step'], {k: round(v['ms_per_step'], 4) for k, v in d['kernels'].items() if v['ms_per_step'] > 0.05})" >> $out
}
run "product build (before)"
cd phamers_amd/csrc
for abl in 1 2 3; do
    touch score_f16.hip phk_api.hip
    make -s -j8 EXTRA_CXXFLAGS="-DPHK_DIAGNOSTIC_BUILD -DF16H_ABL=$abl" 2>&1 | grep -E " error" || true
    ( cd ../.. && export PHK_ALLOW_DIAGNOSTIC_BUILD=1 && run "F16H_ABL=$abl" && run "F16H_ABL=$abl (again)" )
done
touch score_f16.hip phk_api.hip
make -s -j8 2>&1 | grep -E " error" || true
cd ../..
run "product build (after)"
