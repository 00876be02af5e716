#!/bin/bash
# Large-batch steps of the relation-matrix models (one gpurun lease): parity tests of the rows path, same-box A/B of its switches,
# kernel table of the B = 32 768 step.  Usage (through gpurun):  bash tools/gpu_relmat.sh <transr|rescal> [tag]
#   transr: arms KGE_TRANSR_ROWS=0 (tile kernels) / KGE_TRANSR_ROWS=1 KGE_TRANSR_G=0 (dword gathers) / default
#   rescal: arms KGE_HIP_LIB=tools/_libs/base.so if present (a previous build) / KGE_RESCAL_G2=0 / default
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
MODEL=${1:?transr|rescal}; TAG=${2:-$MODEL}
O=gpurun_out
if [ "$MODEL" = transr ]; then
  KEY="transr"; ONLY_ALL="TransR"; ONLY_BIG="mfma-batch TransR"
  ARMS=("KGE_TRANSR_ROWS=0" "KGE_TRANSR_ROWS=1 KGE_TRANSR_G=0" "KGE_X=1")
else
  KEY="rescal"; ONLY_ALL="RESCAL"; ONLY_BIG="mfma-batch RESCAL"
  ARMS=("KGE_RESCAL_G2=0" "KGE_X=1"); [ -f tools/_libs/base.so ] && ARMS=("KGE_HIP_LIB=tools/_libs/base.so" "${ARMS[@]}")
fi
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_edges.py -x -q -k "$KEY" --timeout 300 > $O/${TAG}_tests.log 2>&1; tail -3 $O/${TAG}_tests.log
for r in 1 2; do for arm in "${ARMS[@]}"; do echo "== $arm"; env $arm ONLY="$ONLY_ALL" timeout 300 python tools/config_perf.py 2>&1 | grep -v amdgpu.ids | tail -5; done; done | tee $O/${TAG}_ab.txt
ONLY="$ONLY_BIG" timeout 300 rocprofv3 --kernel-trace --stats -d $O/_p0 -o b -- python tools/config_perf.py > $O/${TAG}_prof.log 2>&1
python tools/rocpd_summary.py $(find $O/_p0 -name '*.db' | head -1) $O/${TAG}_kernels.md > /dev/null; head -12 $O/${TAG}_kernels.md | cut -c1-200
rm -rf $O/_p0
