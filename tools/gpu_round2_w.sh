#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
for st in 0 1; do
  echo "== KGE_STAGED=$st"
  ONLY="DistMult" N_EVAL=0 KGE_STAGED=$st timeout 200 python tools/config_perf.py 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee gpurun_out/w_ab.log
