#!/bin/bash
# ON THE GPU BOX: PMC passes over the count microbenchmark for the lane-pair kernel.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_lanes
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/p$i" -- python3 $ROOT/tools/bench_count.py --contigs 200000 --iters 2 --check 0 > "$OUT/p$i.log" 2>&1 || echo "pass failed $i"
done
python3 - <<'PY'
import csv,glob,collections,os,json
root=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/prof_lanes'
res={}
for d in sorted(glob.glob(root+'/p*')):
    if not os.path.isdir(d): continue
    for f in glob.glob(d+'/*/*_counter_collection.csv'):
        agg=collections.defaultdict(float); nd=collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if 'phk_count_slots' in r['Kernel_Name']:
                agg[r['Counter_Name']]+=float(r['Counter_Value']); nd[r['Counter_Name']]+=1
        disp=max(nd.values()) if nd else 1
        for k,v in agg.items(): res[k]=v/max(nd[k],1)
    for f in glob.glob(d+'/*/*_kernel_trace.csv'):
        ds=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in csv.DictReader(open(f)) if 'phk_count_slots' in r['Kernel_Name']]
        if ds: res['ms_under_pmc']=sum(ds)/len(ds); res['dispatches']=len(ds)
json.dump(res,open(root+'/summary.json','w'),indent=1,sort_keys=True)
print(json.dumps(res,indent=1,sort_keys=True))
PY
