#!/bin/bash
# Round 3, call L: large-batch RESCAL (split G + block-aggregated grouping): parity, B=32768 tables, C4 regression check
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu --timeout 100 -k "rescal_pair_step or touched_gradient" > $O/l3_pair.log 2>&1; tail -8 $O/l3_pair.log | cut -c1-300
timeout 400 python -m pytest tests/test_hip_parity.py tests/test_fullsize_golden.py tests/test_hip_edges.py -x -q -m gpu --timeout 200 -k "rescal or transr or graph_replayed" > $O/l3_tests.log 2>&1; tail -3 $O/l3_tests.log | cut -c1-300
for tag in "RESCAL YAGO" "RESCAL FB15k k=200"; do
  f=$(echo "$tag" | tr ' =/' '___')
  ONLY="mfma-batch $tag" timeout 300 rocprofv3 --kernel-trace --stats -d $O/l3_p -o t -- python tools/config_perf.py > $O/l3_$f.log 2>&1
  grep "mfma-batch" $O/l3_$f.log
  python tools/rocpd_summary.py $O/l3_p/t_results.db $O/l3_$f.md > /dev/null; head -9 $O/l3_$f.md | cut -c1-220
  rm -rf $O/l3_p
done
ONLY="C4 " N_EVAL=64 timeout 120 python tools/config_perf.py 2>&1 | tail -1
