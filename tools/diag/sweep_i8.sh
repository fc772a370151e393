#!/bin/bash
# The int8 general-D kernel on BASELINE configs 4 and 2, one short bench run per variant (diagnostic; appends to
# gpurun_out/sweep_i8.txt).  usage: tools/diag/sweep_i8.sh [label] [ENV=VALUE ...]
set -e
out=gpurun_out/sweep_i8.txt
mkdir -p gpurun_out
label=${1:-default}; shift || true
for cfg in 4 2; do
    env "$@" timeout -k 10 240 python bench.py --config $cfg --steps 3 --warmup 1 --min-seconds 0 --no-cpu-baseline --parity-contigs 8 \
        > gpurun_out/sweep_tmp.json 2> gpurun_out/sweep_tmp.err
    python - "$label" "$cfg" >> $out <<'PY'
import json, sys
d = json.load(open("gpurun_out/sweep_tmp.json"))
k = d["kernels"]
print(sys.argv[1], "config", sys.argv[2], "step %.2f ms" % d["ms_per_step"],
      " ".join("%s=%.2f" % (n.replace("phk_", "").replace("_kernel", ""), v["ms_per_step"]) for n, v in k.items() if v["ms_per_step"] > 0.3),
      "err %.1e" % d["parity"]["max_rel_score_err"], "fb", d["parity"]["fallback_queries"])
PY
    tail -1 $out
done
