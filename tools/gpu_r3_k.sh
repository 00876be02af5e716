#!/bin/bash
# Round 3, call K: relation-matrix models at B=32768 (kernel table per model), for profiles/r03_mfma_models_vs_batch.md
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
for tag in "RESCAL YAGO" "RESCAL FB15k k=200" "TransR FB15k 100" "NTN FB15k d=k=100 B=32768"; do
  f=$(echo "$tag" | tr ' =/' '___')
  ONLY="mfma-batch $tag" timeout 300 rocprofv3 --kernel-trace --stats -d $O/k3_p -o t -- python tools/config_perf.py > $O/k3_$f.log 2>&1
  grep "mfma-batch" $O/k3_$f.log
  python tools/rocpd_summary.py $O/k3_p/t_results.db $O/k3_$f.md > /dev/null; head -9 $O/k3_$f.md | cut -c1-220
  rm -rf $O/k3_p
done
