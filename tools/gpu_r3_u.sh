#!/bin/bash
# Round 3, call U: TransH / TransD gradients without atomics (kge_pullx.hip): parity, then same-box A/B at FB15k B=32768
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 300 python -m pytest tests/test_hip_transx.py -x -q -m gpu --timeout 120 > $O/u3_tests.log 2>&1; tail -15 $O/u3_tests.log | cut -c1-300
run() { ONLY="$1" N_EVAL=64 timeout 200 python tools/config_perf.py 2>&1 | tail -1 | cut -c1-110; }
for own in 0 1; do echo "== KGE_TRANSX_OWN=$own"; KGE_TRANSX_OWN=$own run "TransH FB15k d=100 B=32768"; KGE_TRANSX_OWN=$own run "TransD FB15k d=100 B=32768"; done | tee $O/u3_ab.log
