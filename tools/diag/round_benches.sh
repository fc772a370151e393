#!/bin/bash
# ON THE GPU BOX: the round's bench lines (one JSON file each under gpurun_out/<tag>/), then the rocprofv3 summaries.
# usage: tools/diag/round_benches.sh <tag>
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --workload ragged --min-seconds 3 --no-cpu-baseline > $OUT/bench_ragged.json 2>> $OUT/err.log
python bench.py --config 2 --min-seconds 3 --no-cpu-baseline > $OUT/bench_config2.json 2>> $OUT/err.log
python bench.py --config 2 --workload ragged --min-seconds 3 --no-cpu-baseline > $OUT/bench_config2_ragged.json 2>> $OUT/err.log
python bench.py --config 3 --contigs 12500000 --min-seconds 3 --no-cpu-baseline > $OUT/bench_config3_share.json 2>> $OUT/err.log
python bench.py --config 4 --min-seconds 3 --no-cpu-baseline > $OUT/bench_config4.json 2>> $OUT/err.log
# (RCCL prints banner lines on stdout in front of the bench line: only the JSON line is kept)
PHK_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 python bench.py --min-seconds 2 --no-cpu-baseline 2>> $OUT/err.log | grep '^{' > $OUT/bench_one_rank_rccl.json
# BASELINE configs[3] whole on ONE GPU: 100M contigs in one call (125 GB of packed bases + 102 GB of counts resident)
timeout -k 10 400 python bench.py --config 3 --steps 3 --warmup 1 --min-seconds 0 --no-cpu-baseline > $OUT/bench_config3_100M_one_gpu.json 2>> $OUT/err.log || echo "configs[3] on one GPU: failed (see err.log)"
python - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    print(os.path.basename(f), round(d["value"], 1), d["unit"], "%.3f ms" % d["ms_per_step"], "steps", d["steps"],
          {k: round(v["ms_per_step"], 3) for k, v in d["kernels"].items() if v["ms_per_step"] > 0.05},
          "roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3), d.get("gather"))
PY
