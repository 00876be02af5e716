#!/bin/bash
# Round 3, call F: k_own_step with the per-triple working set, resident-wave targets 4 / 5 / 6 / 8 (same box)
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 300 python -m pytest tests/test_hip_own.py -x -q -m gpu --timeout 100 > $O/f3_tests.log 2>&1; tail -3 $O/f3_tests.log | cut -c1-300
run() { ONLY="$1" N_EVAL=64 timeout 120 python tools/config_perf.py 2>&1 | tail -1; }
for w in 4 3; do
  if [ $w = 4 ]; then unset KGE_HIP_LIB; else export KGE_HIP_LIB=$PWD/tools/_libs/libkge_own_w$w.so; fi
  echo "== waves=$w"; run "C2 "; KGE_PW_PULL=1 run "DistMult"
done 2>&1 | tee $O/f3_ab.log
unset KGE_HIP_LIB
ONLY="C2 " N_EVAL=64 timeout 300 rocprofv3 --kernel-trace --stats -d $O/f3_p0 -o c2 -- python tools/config_perf.py > $O/f3_p0.log 2>&1
python tools/rocpd_summary.py $O/f3_p0/c2_results.db $O/f3_kernels.md > /dev/null; grep "k_own" $O/f3_kernels.md | head -4 | cut -c1-200
rm -rf $O/f3_p0
