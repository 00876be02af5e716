#!/bin/bash
# TransR large-batch step: parity tests of the rows path, same-box A/B against the tile kernels, kernel table of the B = 32 768 step.
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-t1}
O=gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_edges.py -x -q -k "transr" --timeout 300 > $O/${TAG}_tests.log 2>&1; tail -25 $O/${TAG}_tests.log
for r in 1 2; do for arm in "KGE_TRANSR_ROWS=0" "KGE_TRANSR_ROWS=2 KGE_TRANSR_G=0" "KGE_TRANSR_ROWS=1 KGE_TRANSR_G=0" "KGE_TRANSR_ROWS=1"; do echo "== $arm"; env $arm ONLY="TransR" timeout 300 python tools/config_perf.py 2>&1 | grep -v amdgpu.ids | tail -4; done; done | tee $O/${TAG}_ab.txt
ONLY="mfma-batch TransR" timeout 300 rocprofv3 --kernel-trace --stats -d $O/_p0 -o b -- python tools/config_perf.py > $O/${TAG}_prof.log 2>&1
python tools/rocpd_summary.py $(find $O/_p0 -name '*.db' | head -1) $O/${TAG}_kernels.md > /dev/null; head -8 $O/${TAG}_kernels.md | cut -c1-200
rm -rf $O/_p0
