#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
KGE_STAGED=1 timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/z_prof_v -o bench -- python bench.py --no-cpu-baseline --steps 20 > gpurun_out/z_prof_v.log 2>&1
python tools/rocpd_summary.py gpurun_out/z_prof_v/bench_results.db gpurun_out/v_kernel_table.md > /dev/null; grep "pointwise\|staged\|stage_rel\|k_opt\|fillBuffer\|rotate" gpurun_out/v_kernel_table.md | cut -c1-250
rm -rf gpurun_out/z_prof_v
