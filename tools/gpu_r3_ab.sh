#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_hip_own.py -x -q -m gpu --timeout 300 > $O/aa3_tests.log 2>&1; tail -6 $O/aa3_tests.log | cut -c1-400
for c in "ANALOGY FB15k d=200 B=4096" "QuatE FB15k d=100 B=32768"; do
  f=$(echo "$c" | tr ' =/' '___')
  KGE_PW_PULL=1 ONLY="$c" N_EVAL=64 timeout 300 rocprofv3 --kernel-trace --stats -d $O/ab3_p -o t -- python tools/config_perf.py > $O/ab3_$f.log 2>&1
  python tools/rocpd_summary.py $O/ab3_p/t_results.db $O/ab3_$f.md > /dev/null; echo "== $c"; head -6 $O/ab3_$f.md | cut -c1-200
  rm -rf $O/ab3_p
done
