#!/bin/bash
# C1 headline step: parity tests of the owner-computes step, same-box A/B against tools/_libs/base.so, kernel table
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-c1}
O=gpurun_out
timeout 900 python -m pytest tests/test_hip_pull.py -x -q --timeout 300 > $O/${TAG}_tests.log 2>&1; tail -3 $O/${TAG}_tests.log
for r in 1 2 3; do for arm in "KGE_HIP_LIB=tools/_libs/base.so" "KGE_PULL_FOLD=0" "KGE_X=1"; do echo "== $arm"; env $arm timeout 300 python bench.py --no-cpu-baseline --no-live-pmc --no-extra-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])"; done; done | tee $O/${TAG}_ab.txt
