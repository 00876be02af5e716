#!/bin/bash
# Round 3, call C: the two-phase owner-computes step of the pointwise models: parity tests, then C2 timing A/B and kernel table
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_hip_own.py -x -q -m gpu --timeout 120 > $O/c3_own.log 2>&1; tail -15 $O/c3_own.log
for v in 0 1; do
  echo "== KGE_PW_PULL=$v"; ONLY="C2 " N_EVAL=64 KGE_PW_PULL=$v timeout 200 python tools/config_perf.py 2>&1 | tail -1
  ONLY="DistMult" N_EVAL=64 KGE_PW_PULL=$v timeout 200 python tools/config_perf.py 2>&1 | tail -1
done | tee $O/c3_ab.log
ONLY="C2 " N_EVAL=64 KGE_PW_PULL=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/c3_p0 -o c2 -- python tools/config_perf.py > $O/c3_p0.log 2>&1
python tools/rocpd_summary.py $O/c3_p0/c2_results.db $O/c3_c2_own_kernels.md > /dev/null; grep "k_own\|k_pull_sample" $O/c3_c2_own_kernels.md | cut -c1-220
rm -rf $O/c3_p0
