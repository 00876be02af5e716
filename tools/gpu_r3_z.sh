#!/bin/bash
# Round 3, call Z5: the pointwise owner-computes step under a DENSE optimiser (Adam) at small batches
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { ONLY="$1" timeout 200 python tools/config_perf.py 2>&1 | tail -1 | cut -c1-100; }
for v in 0 1; do echo "== KGE_PW_PULL=$v"; for c in "DistMult FB15k d=100 B=128 adam" "ComplEx WN18RR d=200 B=128 adam" "ComplEx WN18RR d=200 B=1024 adam"; do KGE_PW_PULL=$v run "$c"; done; done | tee gpurun_out/z3_ab5.log
