#!/bin/bash
# Round 3, call AM: RESCAL large-batch pair step with V / U as batch-as-M GEMMs (k_rescal_rows): tests, step time off / on, kernels
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu --timeout 300 -k "rescal_pair_step" > $O/am3_tests.log 2>&1; tail -3 $O/am3_tests.log | cut -c1-300
for v in 0 1; do echo "== KGE_RESCAL_ROWS=$v"; KGE_RESCAL_ROWS=$v ONLY="mfma-batch RESCAL" timeout 300 python tools/config_perf.py 2>&1 | grep RESCAL; done | tee $O/am3_perf.log
KGE_RESCAL_ROWS=1 ONLY="mfma-batch RESCAL YAGO" timeout 300 rocprofv3 --kernel-trace -d $O/am_kt -o r -- python tools/config_perf.py > $O/am_kt.log 2>&1
python tools/rocpd_summary.py $(find $O/am_kt -name "*.db") $O/am3_rescal_kernels.md > /dev/null 2>&1
rm -rf $O/am_kt
awk -F'|' '{print substr($2,1,70), "|", $5, "|", $7}' $O/am3_rescal_kernels.md | head -9
