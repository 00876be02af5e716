#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_hip_staged.py -x -q -m gpu --timeout 60 > gpurun_out/u_tests.log 2>&1; tail -15 gpurun_out/u_tests.log
for st in 0 1 0 1; do
  echo "== KGE_STAGED=$st"
  KGE_STAGED=$st timeout 300 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,e in d['extra'].items(): print(k, e['mode'][:24], 'step_us', round(e['step_us'],1))
"
done 2>&1 | tee gpurun_out/u_ab.log
