#!/bin/bash
# ON THE GPU BOX: PMC passes over bench.py (200k contigs, 1 step) focused on the scoring kernels.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_score
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_FLAT SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/p$i" -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --contigs ${PROF_CONTIGS:-393216} > "$OUT/p$i.log" 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv,glob,collections,os,json
root=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/prof_score'
res={}
for d in sorted(glob.glob(root+'/p*')):
    if not os.path.isdir(d): continue
    for f in glob.glob(d+'/*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0].replace('void ','')
            if 'phk' in k: res.setdefault(k,{}).setdefault(r['Counter_Name'],0.0); res[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for f in glob.glob(d+'/*/*_kernel_trace.csv'):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0].replace('void ','')
            if 'phk' in k: res.setdefault(k,{})['ms_under_pmc']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6; res[k]['vgpr']=r.get('VGPR_Count'); res[k]['lds']=r.get('LDS_Block_Size')
json.dump(res,open(root+'/summary.json','w'),indent=1,sort_keys=True)
for k in res:
    if 'knn' in k or 'rerank' in k or 'decide' in k: print(k, json.dumps(res[k],indent=0,sort_keys=True))
PY
