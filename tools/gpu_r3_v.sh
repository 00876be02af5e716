#!/bin/bash
# Round 3, call V: kernel tables of the TransH / TransD atomic-free step at FB15k B=32768
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
for mdl in TransH TransD; do
  ONLY="$mdl FB15k d=100 B=32768" N_EVAL=64 timeout 300 rocprofv3 --kernel-trace --stats -d $O/v3_p -o t -- python tools/config_perf.py > $O/v3_$mdl.log 2>&1
  python tools/rocpd_summary.py $O/v3_p/t_results.db $O/v3_${mdl}_kernels.md > /dev/null; echo "== $mdl"; head -8 $O/v3_${mdl}_kernels.md | cut -c1-190
  rm -rf $O/v3_p
done
