#!/bin/bash
# GPU call G: LDS-staged long-table sweep -- parity, then same-box A/B on the C2 / C3 / C4 eval legs
set -x
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_hip_parity.py -x -q -k "lds_staged or eval_sweep" 2>&1 | tail -15 ) > gpurun_out/g_pytest.log
( timeout 600 python -m pytest tests/test_fullsize_golden.py tests/test_hip_fullsize_configs.py -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/g_pytest_full.log
for flag in 0 1; do
  KGE_EVAL_LDS=$flag ONLY="C2 " timeout 200 python tools/config_perf.py > gpurun_out/g_c2_lds$flag.log 2>&1
  KGE_EVAL_LDS=$flag ONLY="C3 " timeout 200 python tools/config_perf.py > gpurun_out/g_c3_lds$flag.log 2>&1
  KGE_EVAL_LDS=$flag ONLY="C4 " timeout 200 python tools/config_perf.py > gpurun_out/g_c4_lds$flag.log 2>&1
done
tail -3 gpurun_out/g_pytest.log
tail -3 gpurun_out/g_pytest_full.log
grep -h "eval" gpurun_out/g_c*_lds*.log
