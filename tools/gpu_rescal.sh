#!/bin/bash
# RESCAL large-batch step: parity tests, same-box A/B against the previous build, kernel table of the B = 32 768 steps.
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-rs1}
O=gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "rescal" --timeout 300 > $O/${TAG}_tests.log 2>&1; tail -5 $O/${TAG}_tests.log
for r in 1 2; do for arm in "KGE_HIP_LIB=tools/_libs/old_dense.so" "KGE_RESCAL_G2=0" "KGE_X=1"; do echo "== $arm"; env $arm ONLY="RESCAL" timeout 300 python tools/config_perf.py 2>&1 | grep -v amdgpu.ids | tail -5; done; done | tee $O/${TAG}_ab.txt
ONLY="mfma-batch RESCAL" timeout 300 rocprofv3 --kernel-trace --stats -d $O/_p0 -o b -- python tools/config_perf.py > $O/${TAG}_prof.log 2>&1
python tools/rocpd_summary.py $(find $O/_p0 -name '*.db' | head -1) $O/${TAG}_kernels.md > /dev/null; head -12 $O/${TAG}_kernels.md | cut -c1-200
rm -rf $O/_p0
