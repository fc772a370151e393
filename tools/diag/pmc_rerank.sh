cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM"; do
  n=$(echo $pass | cut -c1-12 | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/pmc_rr_$n -- python3 $R/bench.py --config 2 --contigs 2000000 --steps 1 --warmup 0 --no-cpu-baseline --min-seconds 0 --parity-contigs 8 > /dev/null 2> $R/gpurun_out/pmc_rr_$n.err || echo fail $pass
done
python3 - <<'PY'
import csv,glob,collections,os
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out'
for f in glob.glob(root+'/pmc_rr_*/*/*_counter_collection.csv'):
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'phk_rerank_kernel' in r['Kernel_Name']: agg[r['Counter_Name']]+=float(r['Counter_Value'])
    print({k:round(v/1e9,3) for k,v in agg.items()})
PY
