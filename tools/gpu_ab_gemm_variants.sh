#!/bin/bash
# Same-box A/B of k_eval_gemm builds (profiles/r02_experiments.md).  The variant libraries are built by compiling kge_eval.hip
# with -D switches (e.g. -DKGE_NO_LDS_PREFETCH) and linking it with the other objects of pykg2vec_amd/csrc/build into
# tools/_libs/libkge_<name>.so; KGE_HIP_LIB selects the library inside ONE gpurun call (boxes differ by up to +-8 %).
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in pykg2vec_amd/libkge_hip.so tools/_libs/libkge_ks32.so tools/_libs/libkge_pf2.so tools/_libs/libkge_wps3.so tools/_libs/libkge_ks32pf2.so; do
  echo "== $lib"
  KGE_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,e in d['extra'].items(): print(k, 'eval ms', round(e['eval_ms_per_pass'],3), 'TF', round(e['eval_TFLOPs'],1))
"
done; done 2>&1 | tee gpurun_out/q_ab.log
