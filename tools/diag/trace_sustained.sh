#!/bin/bash
# ON THE GPU BOX: rocprofv3 kernel-trace stats of the DEFAULT bench command over a sustained timed region (the three-step traces
# of tools/profile_gpu.sh start cold and read ~10 % slower per kernel).  usage: tools/diag/trace_sustained.sh <tag> [min seconds]
TAG=${1:-r04}
SEC=${2:-3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ROOT/bench.py --no-cpu-baseline --min-seconds $SEC > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err" || echo "trace run failed"
f=$(ls -t "$OUT"/trace/*/*_kernel_stats.csv | head -1)
cp "$f" "$OUT/kernel_stats.csv"
head -8 "$OUT/kernel_stats.csv" | cut -c1-60,200-330
python3 - "$OUT/bench_under_trace.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("bench under trace:", round(d["ms_per_step"], 3), "ms/step", d["steps"], "steps", {k: round(v["ms_per_step"] / v["launches_per_step"], 4) for k, v in d["kernels"].items() if v["ms_per_step"] > 0.1})
PY
