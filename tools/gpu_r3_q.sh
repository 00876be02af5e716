#!/bin/bash
# Round 3, call Q: two-phase step, visits in flight per batch (2 / 4 = default / 6 / 8) with the optimiser-state prefetch
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
export KGE_PULL_DIR=1
timeout 600 python -m pytest tests/test_hip_pull.py -x -q -m gpu --timeout 200 -k "two_phase" > $O/q3_tests.log 2>&1; tail -3 $O/q3_tests.log | cut -c1-300
for rep in 1 2; do
for lib in default tools/_libs/libkge_dir_b2.so tools/_libs/libkge_dir_b6.so tools/_libs/libkge_dir_b8.so; do
  if [ $lib = default ]; then unset KGE_HIP_LIB; else export KGE_HIP_LIB=$PWD/$lib; fi
  timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > $O/q3_tmp.json 2> $O/q3_tmp.err
  python - <<PY
import json
d=json.load(open('gpurun_out/q3_tmp.json'))
print("$lib", "ms_per_step %.4f" % d["ms_per_step"], "value %.3f G" % (d["value"]/1e9))
PY
done; done | tee $O/q3_ab.log
