#!/bin/bash
# GPU call I: kernel-level breakdown of the C2 evaluation pass under the VALU and the matrix-core sweep
set -x
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for flag in 0 1; do
  KGE_EVAL_GEMM=$flag ONLY="C2 " timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/i_prof$flag -o t -- python tools/config_perf.py > gpurun_out/i_c2_gemm$flag.log 2>&1
  python tools/rocpd_summary.py gpurun_out/i_prof$flag/t_results.db gpurun_out/i_c2_kernels_gemm$flag.md > /dev/null
  rm -rf gpurun_out/i_prof$flag
done
