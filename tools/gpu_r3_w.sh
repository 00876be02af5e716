#!/bin/bash
# Round 3, call W: staged form of the pointwise owner-computes step (k_own_eval + k_own_step<STAGED> + k_own_apply): parity, A/B
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_hip_own.py -x -q -m gpu --timeout 200 > $O/w3_tests.log 2>&1; tail -12 $O/w3_tests.log | cut -c1-300
run() { ONLY="$1" N_EVAL=64 timeout 200 python tools/config_perf.py 2>&1 | tail -1 | cut -c1-110; }
for st in 0 1; do echo "== KGE_OWN_STAGED=$st"; KGE_OWN_STAGED=$st run "C2 "; KGE_OWN_STAGED=$st run "DistMult FB15k d=100 B=32768"; done | tee $O/w3_ab.log
ONLY="C2 " N_EVAL=64 timeout 300 rocprofv3 --kernel-trace --stats -d $O/w3_p -o t -- python tools/config_perf.py > $O/w3_p.log 2>&1
python tools/rocpd_summary.py $O/w3_p/t_results.db $O/w3_c2_kernels.md > /dev/null; head -6 $O/w3_c2_kernels.md | cut -c1-190
rm -rf $O/w3_p
