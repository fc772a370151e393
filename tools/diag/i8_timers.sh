#!/bin/bash
# ON THE GPU BOX: phase timers of the int8 sweep (diagnostic build -DI8_TIMERS=<1|2>) at one BASELINE config; the device
# printf lines of two workgroups go to gpurun_out/i8_timers_<config>_<level>.txt.  usage: tools/diag/i8_timers.sh <config> <1|2>
set -e
CFG=${1:-2}; LVL=${2:-1}
( cd phamers_amd/csrc && touch score_i8.hip phk_api.hip && make -s EXTRA_CXXFLAGS="-DPHK_DIAGNOSTIC_BUILD -DI8_TIMERS=$LVL" ) 2>&1 | grep -E "error" || true
PHK_ALLOW_DIAGNOSTIC_BUILD=1 timeout -k 10 300 python bench.py --config $CFG --steps 1 --warmup 0 --min-seconds 0 --no-cpu-baseline --parity-contigs 8 \
    > gpurun_out/i8_timers_raw.txt 2> gpurun_out/i8_timers_err.txt || true
grep "i8 timers" gpurun_out/i8_timers_raw.txt | head -16 > gpurun_out/i8_timers_${CFG}_${LVL}.txt
python - gpurun_out/i8_timers_${CFG}_${LVL}.txt <<'PY'
import re, sys
rows = [list(map(int, re.findall(r"(?:total|wait|barrier|body|loop) (\d+)", l))) for l in open(sys.argv[1])]
if rows:
    n = len(rows); s = [sum(r[i] for r in rows) / n for i in range(5)]
    print("mean of %d waves: total %.0f wait %.1f%% barrier %.1f%% body %.1f%% epilogue+loop %.1f%%" % (n, s[0], *(100 * x / s[0] for x in s[1:])))
PY
