#!/bin/bash
# Round 3, call J: the pairwise RESCAL step as one launch per (relation, 16 pairs) tile: parity, then C4 timing A/B + kernel table
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu --timeout 100 -k "rescal_pair_step or touched_gradient" > $O/j3_pair.log 2>&1; tail -15 $O/j3_pair.log | cut -c1-300
timeout 400 python -m pytest tests/test_hip_parity.py tests/test_fullsize_golden.py tests/test_hip_edges.py -x -q -m gpu --timeout 200 -k "rescal or transr or graph_replayed or ntn" > $O/j3_tests.log 2>&1; tail -4 $O/j3_tests.log | cut -c1-300
run() { ONLY="$1" N_EVAL=64 timeout 120 python tools/config_perf.py 2>&1 | tail -1; }
for u in 1 0; do
  if [ $u = 1 ]; then export KGE_RESCAL_UNFUSED=1; else unset KGE_RESCAL_UNFUSED; fi
  echo "== KGE_RESCAL_UNFUSED=$u"; run "C4 "; run "RESCAL FB15k"
done | tee $O/j3_ab.log
ONLY="C4 " N_EVAL=64 timeout 300 rocprofv3 --kernel-trace --stats -d $O/j3_p0 -o c4 -- python tools/config_perf.py > $O/j3_p0.log 2>&1
python tools/rocpd_summary.py $O/j3_p0/c4_results.db $O/j3_c4_kernels.md > /dev/null; head -12 $O/j3_c4_kernels.md | cut -c1-200
rm -rf $O/j3_p0
