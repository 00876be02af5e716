#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_hip_staged.py -x -q -m gpu --timeout 60 > gpurun_out/r_tests.log 2>&1; tail -25 gpurun_out/r_tests.log
KGE_STAGED=1 timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/z_prof_s -o bench -- python bench.py --no-cpu-baseline --steps 20 > gpurun_out/z_prof_s.log 2>&1
python tools/rocpd_summary.py gpurun_out/z_prof_s/bench_results.db gpurun_out/r_kernel_table.md > /dev/null; grep "rotate\|staged\|k_opt\|fillBuffer" gpurun_out/r_kernel_table.md | cut -c1-260
rm -rf gpurun_out/z_prof_s
