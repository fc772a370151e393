#!/bin/bash
# ON THE GPU BOX: where does a single k = 4 call stop certifying?  (round 5: 100M contigs in one call sent 83M rows to the brute force)
out=${1:-gpurun_out/large_n_probe2.txt}
for n in 16000000 17000000 30000000; do
    python bench.py --config 3 --contigs $n --min-seconds 1 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
s=d['parity']['decision_stats']
print('contigs $n step %.1f ms' % d['ms_per_step'], 'second_chance', s['second_chance'], 'brute', s['brute_forced'], 'cen_uncert', s['centroid_leader_uncertified'], 'per M:', round(s['second_chance']/($n/1e6)))" >> $out
done
