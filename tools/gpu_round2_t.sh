#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_hip_staged.py tests/test_hip_parity.py -x -q -m gpu --timeout 120 -k "staged or optim or weights or step" > gpurun_out/t_tests.log 2>&1; tail -3 gpurun_out/t_tests.log
for nt in 0 1 0 1; do
  echo "== KGE_OPT_NT=$nt KGE_STAGED=0"
  KGE_STAGED=0 KGE_OPT_NT=$nt timeout 300 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C1', round(d['value']/1e9,3))
for k,e in d['extra'].items(): print(k, 'step_us', round(e['step_us'],1))
"
done 2>&1 | tee gpurun_out/t_ab.log
