#!/bin/bash
# timing experiments: kernel durations of the TransR B = 32 768 step with parts switched off (KGE_TRANSR_DBG bit mask)
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
for d in ${DBGS:-0 1 2 4 8 64 128 16 32}; do
  KGE_TRANSR_DBG=$d ONLY="mfma-batch TransR" timeout 300 rocprofv3 --kernel-trace --stats -d $O/_p$d -o b -- python tools/config_perf.py > $O/dbg_prof.log 2>&1
  python tools/rocpd_summary.py $(find $O/_p$d -name '*.db' | head -1) $O/dbg_$d.md > /dev/null
  echo "== dbg $d"; grep -E "k_transr" $O/dbg_$d.md | cut -d'|' -f2,7 
  rm -rf $O/_p$d
done
