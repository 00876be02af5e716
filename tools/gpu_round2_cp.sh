#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_pull.py tests/test_hip_edges.py -x -q -m gpu --timeout 400 > gpurun_out/cp_tests.log 2>&1; tail -5 gpurun_out/cp_tests.log
for B in 128 1024; do
timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 200 --batch $B --eval-triples 256 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$B', d['config']['step_path'][:40], round(d['ms_per_step']*1e3,2), 'us/step', round(d['value']/1e6,1), 'M/s; ref-default record:', d.get('train_reference_default_batch'))"
done
