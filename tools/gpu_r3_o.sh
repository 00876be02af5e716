#!/bin/bash
# Round 3, call O: pipeline depth of the visit loop in k_pull_step (2 = shipped, 3, 4): same-box A/B through KGE_HIP_LIB
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
for rep in 1 2; do
for lib in default tools/_libs/libkge_pull_d3.so tools/_libs/libkge_pull_d4.so; do
  if [ $lib = default ]; then unset KGE_HIP_LIB; else export KGE_HIP_LIB=$PWD/$lib; fi
  timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > $O/o3_tmp.json 2> $O/o3_tmp.err
  python - <<PY
import json
d=json.load(open('gpurun_out/o3_tmp.json'))
print("$lib", "ms_per_step %.4f" % d["ms_per_step"], "region %.4f" % d["roofline"]["avg_launch_ms"], "burst %.4f" % d["roofline"]["burst_launch_ms"])
PY
done; done | tee $O/o3_ab.log
