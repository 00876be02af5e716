#!/bin/bash
set -x
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "head or optim or weights or step" > gpurun_out/n_tests.log 2>&1; tail -3 gpurun_out/n_tests.log
KGE_HEAD_TILE=0 timeout 300 python tools/head_perf.py > gpurun_out/n_head_tile64.log 2>&1
timeout 300 python tools/head_perf.py > gpurun_out/n_head_tile128.log 2>&1
cat gpurun_out/n_head_tile64.log gpurun_out/n_head_tile128.log
timeout 300 python bench.py --no-cpu-baseline --steps 50 > gpurun_out/n_bench.json 2> gpurun_out/n_bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/n_bench.json").read().strip().splitlines()[-1])
print(d["value"]/1e9)
for k,e in d["extra"].items(): print(k, e["step_us"], e["scored_triples_per_s"]/1e6, e["eval_ms_per_pass"])
PY
