#!/bin/bash
# Round 3, call AG: cooperative chain target/filter kernel, workgroup-per-triple query vectors, finer K split of the re-layout:
# rank tests, eval timings, kernel durations of the C3 / C2 eval passes
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_fullsize_golden.py tests/test_hip_fullsize_configs.py tests/test_hip_edges.py -x -q -m gpu --timeout 200 -k "matrix_core or eval or rank or fullsize or filter" > $O/ag3_tests.log 2>&1; tail -4 $O/ag3_tests.log | cut -c1-300
for sh in c2 c3 c4; do SHAPE=$sh REPS=5 timeout 200 python tools/eval_only.py 2>&1 | tail -1; done | tee $O/ag3_eval.log
for sh in c3 c2; do
SHAPE=$sh REPS=3 timeout 300 rocprofv3 --kernel-trace -d $O/ag_kt_$sh -o ev -- python tools/eval_only.py > $O/ag_kt_$sh.log 2>&1
python tools/rocpd_summary.py $(find $O/ag_kt_$sh -name "*.db") $O/ag3_kernels_$sh.md > /dev/null 2>&1
rm -rf $O/ag_kt_$sh
grep -E "k_eval" $O/ag3_kernels_$sh.md | cut -c1-70,110-180
done
