#!/bin/bash
# Round 3, call AL: NTN, where the large-batch forms start to pay: B = 128 ... 8 192 with the forms off / on
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
for v in 0 1; do echo "== KGE_NTN_BIG=$v"; KGE_NTN_BIG=$v ONLY="NTN FB15k d=k=100 B=" timeout 400 python tools/config_perf.py 2>&1 | grep "NTN"; done | tee $O/al3_ntn_threshold.log
