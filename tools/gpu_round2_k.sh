#!/bin/bash
set -x
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 120 tools/_libs/mfma_bench > gpurun_out/k_mfma_bench.txt 2>&1
for flag in 0 1; do
  KGE_EVAL_GEMM=$flag ONLY="C2 " timeout 200 python tools/config_perf.py > gpurun_out/k_c2_gemm$flag.log 2>&1
done
grep -h eval gpurun_out/k_c2_gemm*.log
cat gpurun_out/k_mfma_bench.txt
