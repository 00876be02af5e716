#!/bin/bash
# Round 3, call Z: where does the two-launch TransE step start to pay?  (B = 8192 / 16384; TransH / DistMult at 8192 for their rules)
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { ONLY="$1" timeout 200 python tools/config_perf.py 2>&1 | tail -1 | cut -c1-100; }
for dir in 0 1; do echo "== KGE_PULL_DIR=$dir"; KGE_PULL_DIR=$dir run "C1 TransE FB15k d=100 B=8192"; KGE_PULL_DIR=$dir run "C1 TransE FB15k d=100 B=16384"; done | tee gpurun_out/z3_ab.log
