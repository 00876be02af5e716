#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_pull.py -x -q -m gpu --timeout 300 > gpurun_out/tm_tests.log 2>&1; tail -8 gpurun_out/tm_tests.log
for pl in 0 1; do KGE_PULL=$pl ONLY="TransM FB15k d=100" N_EVAL=0 timeout 200 python tools/config_perf.py 2>&1 | grep -v amdgpu.ids; done
timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C1', d['value']/1e9, d['roofline']['avg_launch_ms'])"
