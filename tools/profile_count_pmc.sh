#!/bin/bash
# ON THE GPU BOX: issue / LDS counters of the k = 4 count kernels on the BASELINE batch (1M x 5 kb), separate --pmc passes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_count_r02
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- python3 $ROOT/tools/bench_count.py > /dev/null 2> "$OUT/pmc_$name.err" || echo "pmc pass failed: $pass"
done
python3 - <<'PY'
import csv,glob,collections,os,json
root=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/prof_count_r02'
res={}
for d in sorted(glob.glob(root+'/pmc_*')):
    if not os.path.isdir(d): continue
    for f in glob.glob(d+'/*/*_counter_collection.csv'):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); nd=collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0].replace('void ','')
            if 'phk_count' in k:
                agg[k][r['Counter_Name']]+=float(r['Counter_Value']); nd[k].add(r['Dispatch_Id'])
        for k in agg:
            for c,v in agg[k].items(): res.setdefault(k,{})[c]=v/max(len(nd[k]),1)
json.dump(res,open(root+'/pmc_count.json','w'),indent=1,sort_keys=True)
print(json.dumps(res,indent=1,sort_keys=True))
PY
