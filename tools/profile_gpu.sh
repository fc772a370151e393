#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats of bench.py for one BASELINE configuration plus
# separate PMC passes for HBM traffic (FETCH_SIZE / WRITE_SIZE never share a pass; no other trace domains).
# Results land under gpurun_out/prof_<tag>/; summaries are copied into profiles/ afterwards.
#   tools/profile_gpu.sh <tag> <config> [contigs]
set -u
TAG=${1:-r03}
CFG=${2:-1}
N=${3:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="--config $CFG --no-cpu-baseline --min-seconds 0"
[ -n "$N" ] && ARGS="$ARGS --contigs $N"
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace + stats"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ROOT/bench.py --steps 3 --warmup 1 $ARGS > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err" || echo "trace run failed"
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-30)
  echo "== pmc $pass"
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- python3 $ROOT/bench.py --steps 1 --warmup 0 $ARGS > /dev/null 2> "$OUT/pmc_$name.err" || echo "pmc pass failed: $pass"
done
PROF_ROOT="$OUT" python3 - <<'PY'
import csv,glob,collections,os,json
root=os.environ['PROF_ROOT']
res={}
for d in sorted(glob.glob(root+'/pmc_*')):
    if not os.path.isdir(d): continue
    for f in glob.glob(d+'/*/*_counter_collection.csv'):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); nd=collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0].replace('void ','')
            if 'phk' in k:
                agg[k][r['Counter_Name']]+=float(r['Counter_Value']); nd[k].add(r['Dispatch_Id'])
        for k in agg:
            # per bench step: the run is one step (+ the parity spot check outside the profiled kernels' names)
            for c,v in agg[k].items(): res.setdefault(k,{})[c]=v
            res[k]['dispatches']=len(nd[k])
    for f in glob.glob(d+'/*/*_kernel_trace.csv'):
        tot=collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0].replace('void ','')
            if 'phk' in k: tot[k]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
        for k,v in tot.items(): res.setdefault(k,{})['ms_under_pmc']=v
json.dump(res,open(root+'/pmc_summary.json','w'),indent=1,sort_keys=True)
print(json.dumps({k:{c:v for c,v in d.items() if c in ('FETCH_SIZE','WRITE_SIZE','ms_under_pmc','dispatches')} for k,d in res.items()},indent=1,sort_keys=True))
PY
cat "$OUT"/trace/*/*_kernel_stats.csv | cut -c1-200
