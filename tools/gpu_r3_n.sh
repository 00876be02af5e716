#!/bin/bash
# Round 3, call N: RotatE bundle with the rows split over two waves: parity, then C3 A/B + kernel table
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 500 python -m pytest tests -x -q -m gpu --timeout 200 -k "rotate or staged" > $O/n3_tests.log 2>&1; tail -6 $O/n3_tests.log | cut -c1-300
run() { ONLY="$1" N_EVAL=64 timeout 120 python tools/config_perf.py 2>&1 | tail -1; }
for u in 0 2 4; do echo "== KGE_ROTATE_SPLIT=$u"; KGE_ROTATE_SPLIT=$u run "C3 "; done | tee $O/n3_ab.log
ONLY="C3 " N_EVAL=64 timeout 300 rocprofv3 --kernel-trace --stats -d $O/n3_p -o t -- python tools/config_perf.py > $O/n3_p.log 2>&1
python tools/rocpd_summary.py $O/n3_p/t_results.db $O/n3_c3_kernels.md > /dev/null; head -7 $O/n3_c3_kernels.md | cut -c1-220
rm -rf $O/n3_p
