#!/bin/bash
# Round 3, call I: RESCAL step: single-launch relation grouping, renormalisation folded into the optimiser; parity then C4 timing A/B
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 400 python -m pytest tests/test_hip_parity.py tests/test_fullsize_golden.py tests/test_hip_edges.py -x -q -m gpu --timeout 200 -k "rescal or transr or graph_replayed or ntn" > $O/i3_tests.log 2>&1; tail -4 $O/i3_tests.log | cut -c1-300
run() { ONLY="$1" N_EVAL=64 timeout 120 python tools/config_perf.py 2>&1 | tail -1; }
for f in 0 1; do echo "== KGE_RESCAL_FUSED=$f"; KGE_RESCAL_FUSED=$f run "C4 "; KGE_RESCAL_FUSED=$f run "RESCAL FB15k"; done | tee $O/i3_ab.log
run "TransR FB15k" | tee -a $O/i3_ab.log
ONLY="C4 " N_EVAL=64 timeout 300 rocprofv3 --kernel-trace --stats -d $O/i3_p0 -o c4 -- python tools/config_perf.py > $O/i3_p0.log 2>&1
python tools/rocpd_summary.py $O/i3_p0/c4_results.db $O/i3_c4_kernels.md > /dev/null; head -16 $O/i3_c4_kernels.md | cut -c1-200
rm -rf $O/i3_p0
