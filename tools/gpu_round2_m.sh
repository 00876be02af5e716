#!/bin/bash
# GPU call M: full GPU suite + C3 A/B after the prepare change
set -x
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/m_pytest.log
tail -3 gpurun_out/m_pytest.log
for flag in 0 1; do
  KGE_EVAL_GEMM=$flag ONLY="C3 " timeout 200 python tools/config_perf.py > gpurun_out/m_c3_gemm$flag.log 2>&1
done
ONLY="TransD" timeout 200 python tools/config_perf.py > gpurun_out/m_transd.log 2>&1
grep -h eval gpurun_out/m_c3_gemm*.log gpurun_out/m_transd.log
