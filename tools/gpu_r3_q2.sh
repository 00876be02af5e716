#!/bin/bash
# Round 3, call Q2: two-phase step, incidences per work item (segment 8 = default / 16 / 32)
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
export KGE_PULL_DIR=1
for rep in 1 2; do
for seg in 8 16 32; do
  KGE_PULL_SEGMENT=$seg timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > $O/q3_tmp.json 2> $O/q3_tmp.err
  python - <<PY
import json
d=json.load(open('gpurun_out/q3_tmp.json'))
print("segment $seg", "ms_per_step %.4f" % d["ms_per_step"], "value %.3f G" % (d["value"]/1e9), "setup", round(d["setup_ms"],2))
PY
done; done | tee $O/q3_seg.log
tail -2 $O/q3_tmp.err
