#!/bin/bash
# ON THE GPU BOX: the two-part int8 sweep with each way of meeting the lists (PHK_I8_INSERT=0/1/2) at BASELINE configs 2 and 4.
# usage: tools/diag/i8_insert_modes.sh   (appends to gpurun_out/i8_insert_modes.txt)
out=gpurun_out/i8_insert_modes.txt
for cfg in 2 4; do
  for mode in 0 1 2; do
    PHK_I8_INSERT=$mode python bench.py --config $cfg --min-seconds 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); k=d['kernels']
print('config $cfg insert $mode step %.2f ms sweep %.2f decide %.2f err %.1e fb %d' % (d['ms_per_step'], k['phk_knn_i8_general_kernel']['ms_per_step'], k['phk_decide_gen_kernel']['ms_per_step'], d['parity']['max_rel_score_err'], d['parity']['fallback_queries']))" | tee -a $out
  done
done
