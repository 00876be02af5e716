#!/bin/bash
# Round 3, call AI: large-batch NTN kernels (batch-as-M GEMMs): oracle tests with the forms forced on, then the B = 32 768 step
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu --timeout 300 -k "ntn" > $O/ai3_tests.log 2>&1; tail -15 $O/ai3_tests.log | cut -c1-300
for v in 1; do echo "KGE_NTN_BIG=$v $(KGE_NTN_BIG=$v ONLY="mfma-batch NTN" timeout 300 python tools/config_perf.py 2>&1 | tail -1)"; done | tee $O/ai3_perf.log
KGE_NTN_BIG=1 ONLY="mfma-batch NTN" timeout 300 rocprofv3 --kernel-trace -d $O/ai_kt -o ntn -- python tools/config_perf.py > $O/ai_kt.log 2>&1
python tools/rocpd_summary.py $(find $O/ai_kt -name "*.db") $O/ai3_ntn_kernels.md > /dev/null 2>&1
rm -rf $O/ai_kt
awk -F'|' '{print substr($2,1,70), "|", $5, "|", $7}' $O/ai3_ntn_kernels.md | head -24
