#!/bin/bash
set -x
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "head or matrix_core" > gpurun_out/o_tests.log 2>&1; tail -3 gpurun_out/o_tests.log
for lib in tools/_libs/libkge_hip_noprefetch.so pykg2vec_amd/libkge_hip.so tools/_libs/libkge_hip_noprefetch.so pykg2vec_amd/libkge_hip.so; do
  echo "== $lib"
  KGE_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python tools/head_perf.py 2>&1 | grep "B=4096\|B=16384\|B=1000"
  KGE_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,e in d['extra'].items(): print(k, 'eval ms', round(e['eval_ms_per_pass'],3), 'TF', round(e['eval_TFLOPs'],1))
"
done 2>&1 | tee gpurun_out/o_ab.log
