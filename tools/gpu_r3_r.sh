#!/bin/bash
# Round 3, call R: two-phase step as the default for L1 at B >= 16384: pull / dist / fullsize tests + bench line
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_hip_pull.py tests/test_hip_dist.py tests/test_hip_fullsize.py tests/test_fullsize_golden.py -x -q -m gpu --timeout 300 > $O/r3_tests.log 2>&1; tail -6 $O/r3_tests.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > $O/r3_bench.json 2> $O/r3_bench.err
python - <<PY
import json
d=json.load(open('gpurun_out/r3_bench.json'))
r=d["roofline"]
print("default: ms_per_step %.4f value %.3f G frac %.3f burst %.4f setup %.2f ms | %s" % (d["ms_per_step"], d["value"]/1e9, r["frac"], r["burst_launch_ms"], d["setup_ms"], d["config"]["step_path"][:100]))
PY
tail -2 $O/r3_bench.err
