#!/bin/bash
# Round 3, call Q3: where does phase 2 of the two-phase step spend its time?  (a build whose owners skip their visits: timing only)
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
export KGE_PULL_DIR=1 KGE_PULL_SEGMENT=32
for lib in default tools/_libs/libkge_dir_novisit.so; do
  if [ $lib = default ]; then unset KGE_HIP_LIB; else export KGE_HIP_LIB=$PWD/$lib; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/q3_p -o t -- python bench.py --no-cpu-baseline --no-extra-configs > $O/q3_prof.log 2>&1
  python tools/rocpd_summary.py $O/q3_p/t_results.db $O/q3_k.md > /dev/null; echo "== $lib"; grep "k_pull" $O/q3_k.md | cut -c1-170 | head -3
  rm -rf $O/q3_p
done | tee $O/q3_novisit.log
