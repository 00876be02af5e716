#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -x -q -m gpu --timeout 120 -k "rescal or Rescal or graph or fuzz" > gpurun_out/z_tests.log 2>&1; tail -5 gpurun_out/z_tests.log
timeout 300 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,e in d['extra'].items(): print(k, e['mode'][:24], 'step_us', round(e['step_us'],1))
"
