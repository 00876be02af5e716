#!/bin/bash
# Same-box A/B of the staged (atomic-free) step against the atomic-scatter step (KGE_STAGED=0/1) through bench.py's C2/C3/C4
# records, plus a rocprofv3 kernel table of the staged run (profiles/r02_experiments.md).
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_hip_staged.py -x -q -m gpu --timeout 60 > gpurun_out/u_tests.log 2>&1; tail -5 gpurun_out/u_tests.log
for cfg in "KGE_STAGED=0" "KGE_STAGED=1"; do
  echo "== $cfg"
  env $cfg timeout 300 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,e in d['extra'].items(): print(k, e['mode'][:24], 'step_us', round(e['step_us'],1))
"
done 2>&1 | tee gpurun_out/u_ab.log
KGE_STAGED=1 timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/z_prof_v -o bench -- python bench.py --no-cpu-baseline --steps 20 > gpurun_out/z_prof_v.log 2>&1
python tools/rocpd_summary.py gpurun_out/z_prof_v/bench_results.db gpurun_out/v_kernel_table.md > /dev/null; grep "pointwise\|opt_staged\|stage_rel\|rotate" gpurun_out/v_kernel_table.md | cut -c1-200
rm -rf gpurun_out/z_prof_v
