#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_hip_staged.py -x -q -m gpu --timeout 60 > gpurun_out/s_tests.log 2>&1; tail -3 gpurun_out/s_tests.log
for lib in tools/_libs/libkge_no_nt.so pykg2vec_amd/libkge_hip.so tools/_libs/libkge_no_nt.so pykg2vec_amd/libkge_hip.so; do
  echo "== $lib"
  KGE_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
e=d['extra']['C3']; print('C3 step_us', round(e['step_us'],1), 'M/s', round(e['scored_triples_per_s']/1e6,1))
"
done 2>&1 | tee gpurun_out/s_ab.log
