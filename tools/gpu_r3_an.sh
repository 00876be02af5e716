#!/bin/bash
# Round 3, call AN: RESCAL large-batch step with the relation-matrix gradient as a GEMM over gathered rows (k_rescal_g): tests,
# step time with the form off / on, kernel table
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize_configs.py tests/test_fullsize_golden.py -x -q -m gpu --timeout 300 -k "rescal" > $O/an3_tests.log 2>&1; tail -6 $O/an3_tests.log | cut -c1-300
for v in 0 1; do echo "== KGE_RESCAL_G=$v"; KGE_RESCAL_G=$v ONLY="mfma-batch RESCAL" timeout 300 python tools/config_perf.py 2>&1 | grep RESCAL; done | tee $O/an3_perf.log
ONLY="mfma-batch RESCAL YAGO" timeout 300 rocprofv3 --kernel-trace -d $O/an_kt -o r -- python tools/config_perf.py > $O/an_kt.log 2>&1
python tools/rocpd_summary.py $(find $O/an_kt -name "*.db") $O/an3_rescal_kernels.md > /dev/null 2>&1
rm -rf $O/an_kt
awk -F'|' '{print substr($2,1,70), "|", $5, "|", $7}' $O/an3_rescal_kernels.md | head -8
