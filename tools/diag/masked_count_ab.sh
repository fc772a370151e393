#!/bin/bash
# ON THE GPU BOX: k = 4 count kernels on masked (0.1 % and 1 % invalid bases) and ragged batches: the two-windows-per-add kernel
# with mask / sorted walk (phk_count_pairs2_kernel, library's choice) against the one-window-per-add slot kernel (count_lanes = d).
# usage: tools/diag/masked_count_ab.sh <out file> [<label>=<alternative library> ...]
out=$1; shift
run() {   # label, lanes, ppm
  python tools/bench_count.py --lanes "$2" --invalid-ppm $3 --iters 20 --check 64 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('%-28s ppm %-6s %.4f ms  exact=%s ' % ('$1', '$3', d['ms'], d['bit_exact_vs_oracle']), {k.replace('phk_','').replace('_kernel',''): round(v,4) for k,v in d['per_kernel_ms'].items() if v > 0.01})" >> $out
}
for ppm in 0 1000 10000 100000; do
  unset PHK_ALLOW_DIAGNOSTIC_BUILD PHAMERS_AB_LIB
  run "in-tree, library choice" "" $ppm
  run "in-tree, slot kernel (d)" "d" $ppm
  for spec in "$@"; do
    label=${spec%%=*}; lib=${spec#*=}
    export PHK_ALLOW_DIAGNOSTIC_BUILD=1 PHAMERS_AB_LIB=$lib
    run "$label, library choice" "" $ppm
  done
done
unset PHK_ALLOW_DIAGNOSTIC_BUILD PHAMERS_AB_LIB
for ppm in 0 1000; do
for lanes in "" d; do
PHK_BENCH_RAGGED_PPM=$ppm PHK_COUNT_LANES=$lanes python bench.py --workload ragged --min-seconds 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('ragged k=4 ppm $ppm lanes [$lanes]  step %.3f ms' % d['ms_per_step'], {k.replace('phk_','').replace('_kernel',''): round(v['ms_per_step'],4) for k,v in d['kernels'].items() if v['ms_per_step'] > 0.02 and 'count' in k}, d['parity']['counts_bit_exact'])" >> $out
done
done
exit 0
python bench.py --workload ragged --min-seconds 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('ragged k=4 (in-tree)  step %.3f ms' % d['ms_per_step'], {k.replace('phk_','').replace('_kernel',''): round(v['ms_per_step'],4) for k,v in d['kernels'].items() if v['ms_per_step'] > 0.01}, d['parity']['counts_bit_exact'])" >> $out
PHK_COUNT_LANES=d python bench.py --workload ragged --min-seconds 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('ragged k=4 (slot kernel d)  step %.3f ms' % d['ms_per_step'], {k.replace('phk_','').replace('_kernel',''): round(v['ms_per_step'],4) for k,v in d['kernels'].items() if v['ms_per_step'] > 0.01}, d['parity']['counts_bit_exact'])" >> $out
