#!/bin/bash
# Round 3, call X: RESCAL pair step for even hidden sizes (float2 rows: the reference's k = 50): parity + preset timing A/B
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 400 python -m pytest tests/test_hip_parity.py tests/test_fullsize_golden.py tests/test_hip_edges.py tests/test_hip_transx.py -x -q -m gpu --timeout 200 -k "rescal or graph_replayed or transx" > $O/x3_tests.log 2>&1; tail -6 $O/x3_tests.log | cut -c1-300
run() { ONLY="$1" N_EVAL=64 timeout 120 python tools/config_perf.py 2>&1 | tail -1 | cut -c1-120; }
for u in 1 0; do
  if [ $u = 1 ]; then export KGE_RESCAL_UNFUSED=1; else unset KGE_RESCAL_UNFUSED; fi
  echo "== KGE_RESCAL_UNFUSED=$u"; run "RESCAL FB15k k=50"; run "C4 "
done | tee $O/x3_ab.log
