#!/bin/bash
# GPU call H: matrix-core dot-product sweep -- parity, then same-box A/B on the dot-product configs
set -x
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_hip_parity.py -x -q -k "matrix_core" 2>&1 | tail -25 ) > gpurun_out/h_pytest.log
tail -5 gpurun_out/h_pytest.log
grep -q passed gpurun_out/h_pytest.log || exit 1
grep -q failed gpurun_out/h_pytest.log && exit 1
for flag in 0 1; do
  KGE_EVAL_GEMM=$flag ONLY="C2 " timeout 200 python tools/config_perf.py > gpurun_out/h_c2_gemm$flag.log 2>&1
  KGE_EVAL_GEMM=$flag ONLY="C4 " timeout 200 python tools/config_perf.py > gpurun_out/h_c4_gemm$flag.log 2>&1
  KGE_EVAL_GEMM=$flag ONLY="DistMult" timeout 200 python tools/config_perf.py > gpurun_out/h_dm_gemm$flag.log 2>&1
  KGE_EVAL_GEMM=$flag ONLY="ANALOGY" timeout 200 python tools/config_perf.py > gpurun_out/h_an_gemm$flag.log 2>&1
  KGE_EVAL_GEMM=$flag ONLY="C3 " timeout 200 python tools/config_perf.py > gpurun_out/h_c3_gemm$flag.log 2>&1
done
grep -h "eval" gpurun_out/h_*_gemm*.log
