#!/bin/bash
# ON THE GPU BOX: multi-batch k = 4 calls with the batches' tails on the second stream (PHK_TAIL_ASIDE=1, default) and on one
# stream (=0): step time and decision statistics per 1M contigs must agree.  usage: tools/diag/large_n_probe.sh <out file>
out=${1:-gpurun_out/large_n_probe.txt}
for n in 12500000 20000000; do
  for ta in 1 0; do
    PHK_TAIL_ASIDE=$ta python bench.py --config 3 --contigs $n --min-seconds 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('contigs $n tail_aside $ta', 'step %.3f ms' % d['ms_per_step'], 'steps', d['steps'], d['parity']['decision_stats'], {k: round(v['ms_per_step'], 3) for k, v in d['kernels'].items() if v['ms_per_step'] > 0.5})" >> $out
  done
done
