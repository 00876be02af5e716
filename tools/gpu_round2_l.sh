#!/bin/bash
set -x
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for flag in 0 1; do
  KGE_EVAL_GEMM=$flag ONLY="C3 " timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/l_prof$flag -o t -- python tools/config_perf.py > gpurun_out/l_c3_gemm$flag.log 2>&1
  python tools/rocpd_summary.py gpurun_out/l_prof$flag/t_results.db gpurun_out/l_c3_kernels_gemm$flag.md > /dev/null
  rm -rf gpurun_out/l_prof$flag
done
grep -h eval gpurun_out/l_c3_gemm*.log
