#!/bin/bash
# Round 3, call AK: NTN large-batch forms incl. the linear maps as GEMMs: oracle tests, step time, kernel table
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu --timeout 300 -k "ntn" > $O/ak3_tests.log 2>&1; tail -5 $O/ak3_tests.log | cut -c1-300
for v in 1; do echo "KGE_NTN_BIG=$v $(KGE_NTN_BIG=$v ONLY="mfma-batch NTN" timeout 300 python tools/config_perf.py 2>&1 | tail -1)"; done | tee $O/ak3_perf.log
KGE_NTN_BIG=1 ONLY="mfma-batch NTN" timeout 300 rocprofv3 --kernel-trace -d $O/ak_kt -o ntn -- python tools/config_perf.py > $O/ak_kt.log 2>&1
python tools/rocpd_summary.py $(find $O/ak_kt -name "*.db") $O/ak3_ntn_kernels.md > /dev/null 2>&1
rm -rf $O/ak_kt
awk -F'|' '{print substr($2,1,70), "|", $5, "|", $7}' $O/ak3_ntn_kernels.md | head -18
