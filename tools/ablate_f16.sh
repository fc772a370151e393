#!/bin/bash
# ON THE GPU BOX: rebuild the library with -DPHK_ABL=<n> for each n given and time the kernels.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/abl
for v in "$@"; do
  touch $ROOT/phamers_amd/csrc/score_f16.hip $ROOT/phamers_amd/csrc/score_mfma.hip
  make -C $ROOT/phamers_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DPHK_ABL=$v" > $ROOT/gpurun_out/abl/build_$v.log 2>&1 || { echo "build $v failed"; tail -5 $ROOT/gpurun_out/abl/build_$v.log; exit 1; }
  timeout -k 10 200 python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --contigs ${ABL_CONTIGS:-1000000} > $ROOT/gpurun_out/abl/bench_$v.json 2> $ROOT/gpurun_out/abl/bench_$v.err || { echo "bench $v failed"; tail -5 $ROOT/gpurun_out/abl/bench_$v.err; exit 1; }
  python3 - "$ROOT/gpurun_out/abl/bench_$v.json" $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print('ABL',sys.argv[2],'ms/step',d['ms_per_step'],'value',d['value'],{k:round(v['ms_per_step'],3) for k,v in d['kernels'].items()}, d.get('parity'))
PY
done
