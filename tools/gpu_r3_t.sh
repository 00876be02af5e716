#!/bin/bash
# Round 3, call T: the next epoch's first sampler rides in the epoch's last step (pull and own runs): tests + bench at 200 / 20 steps
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests/test_hip_pull.py tests/test_hip_own.py tests/test_hip_dist.py tests/test_hip_fullsize.py tests/test_hip_edges.py -x -q -m gpu --timeout 300 > $O/t3_tests.log 2>&1; tail -5 $O/t3_tests.log | cut -c1-300
for st in "--steps 200 --warmup 20" "--steps 20 --warmup 5"; do
  timeout 300 python bench.py $st --no-cpu-baseline > $O/t3_tmp.json 2> $O/t3_tmp.err
  python - <<PY
import json
d=json.load(open('gpurun_out/t3_tmp.json'))
r=d["roofline"]
print("$st: ms_per_step %.4f value %.3f G frac %.3f traffic %s | C2 %.1f us" % (d["ms_per_step"], d["value"]/1e9, r["frac"], r["traffic"], d["extra"]["C2"]["step_us"]))
PY
done | tee $O/t3_bench.log
