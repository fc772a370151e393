#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats of bench.py plus separate PMC
# passes (never combined with trace domains other than kernel-trace).  Results land under
# gpurun_out/prof_<tag>/; the summaries worth judging are copied into profiles/ afterwards.
set -u
TAG=${1:-r01}
N=${2:-1000000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="$ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --contigs $N"
echo "== kernel trace + stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $BENCH > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err" || echo "trace run failed"
for pass in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  echo "== pmc $pass"
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --contigs 200000 > /dev/null 2> "$OUT/pmc_$name.err" || echo "pmc pass failed: $pass"
done
find "$OUT" -name "*.csv" | head -50
