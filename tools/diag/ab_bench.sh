#!/bin/bash
# ON THE GPU BOX: A/B(/C) of whole-library builds on ONE box, interleaved and repeated (a box moves +-2 % from run to run).
# usage: tools/diag/ab_bench.sh <out file> <rounds> <bench args ...> -- <label>=<library path | ""> ...
#   ("" = the in-tree library).  Timing only for any library but the in-tree one.
out=$1; rounds=$2; shift 2
bargs=()
while [ "$1" != "--" ]; do bargs+=("$1"); shift; done
shift
for r in $(seq 1 $rounds); do
  for spec in "$@"; do
    label=${spec%%=*}; lib=${spec#*=}
    if [ -n "$lib" ]; then export PHK_ALLOW_DIAGNOSTIC_BUILD=1 PHAMERS_AB_LIB=$lib; else unset PHK_ALLOW_DIAGNOSTIC_BUILD PHAMERS_AB_LIB; fi
    python bench.py "${bargs[@]}" --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
ks=d['kernels']; tot=sum(v['ms_per_step'] for v in ks.values()); nl=sum(v['launches_per_step'] for v in ks.values())
print('%-10s round $r  step %.4f ms  kernels %.4f  rest %.4f  launches %.0f ' % ('$label', d['ms_per_step'], tot, d['ms_per_step']-tot, nl), {k.replace('phk_','').replace('_kernel',''): round(v['ms_per_step'], 4) for k, v in ks.items() if v['ms_per_step'] > 0.02})" >> $out
  done
done
unset PHK_ALLOW_DIAGNOSTIC_BUILD PHAMERS_AB_LIB
