#!/bin/bash
# Round 3, call AA: model-generic staged owner-computes step (ANALOGY / CP / SimplE / QuatE): parity, then A/B against the atomic step
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_hip_own.py -x -q -m gpu --timeout 300 > $O/aa3_tests.log 2>&1; tail -15 $O/aa3_tests.log | cut -c1-400
run() { ONLY="$1" N_EVAL=64 timeout 200 python tools/config_perf.py 2>&1 | tail -1 | cut -c1-110; }
for v in 0 1; do echo "== KGE_PW_PULL=$v"; for c in "ANALOGY FB15k d=200 B=4096" "CP FB15k d=50 B=128" "CP FB15k d=100 B=32768" "SimplE FB15k d=100 B=128" "SimplE FB15k d=100 B=32768" "QuatE FB15k d=200 B=100" "QuatE FB15k d=100 B=32768"; do KGE_PW_PULL=$v run "$c"; done; done | tee $O/aa3_ab.log
