#!/bin/bash
# Round 3, call S: two-phase step, direction codes as a byte per element (default) against 2 bits per element (previous build)
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_hip_pull.py -x -q -m gpu --timeout 200 -k "two_phase or bit_reproducible or reference_weights" > $O/s3_tests.log 2>&1; tail -3 $O/s3_tests.log | cut -c1-300
for rep in 1 2; do
for lib in default tools/_libs/libkge_dir_2bit.so; do
  if [ $lib = default ]; then unset KGE_HIP_LIB; else export KGE_HIP_LIB=$PWD/$lib; fi
  timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > $O/s3_tmp.json 2> $O/s3_tmp.err
  python - <<PY
import json
d=json.load(open('gpurun_out/s3_tmp.json'))
print("$lib", "ms_per_step %.4f" % d["ms_per_step"], "value %.3f G" % (d["value"]/1e9), "burst %.4f" % d["roofline"]["burst_launch_ms"])
PY
done; done | tee $O/s3_ab.log
