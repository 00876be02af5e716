#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
for B in 128 1024 4096 8192; do for pl in 1; do
KGE_PULL=$pl timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 200 --batch $B --eval-triples 256 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$B KGE_PULL=$pl', round(d['ms_per_step']*1e3,2), 'us/step', round(d['value']/1e6,1), 'M/s kernel', round(d['roofline']['avg_launch_ms']*1e3,2))"
done; done 2>&1 | tee gpurun_out/sm_ab.log
