#!/bin/bash
# Round 3, call E: sampler-written visit descriptors in both owner-computes kernels: parity, then timings
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 400 python -m pytest tests/test_hip_own.py tests/test_hip_pull.py tests/test_hip_dist.py -x -q -m gpu --timeout 100 > $O/e3_tests.log 2>&1; tail -6 $O/e3_tests.log | cut -c1-300
run() { ONLY="$1" N_EVAL=64 timeout 120 python tools/config_perf.py 2>&1 | tail -1; }
run "C2 " | tee $O/e3_perf.log
run "DistMult" | tee -a $O/e3_perf.log
KGE_PW_PULL=1 run "DistMult" | tee -a $O/e3_perf.log
timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-extra-configs > $O/e3_bench200.json 2> $O/e3_bench200.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/e3_bench200.json').read().strip().splitlines()[-1])
print('value %.1f M  ms/step %.4f kernel %.4f setup_ms %.3f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('setup_ms', -1)))
print('small', d.get('train_reference_default_batch'))
PY
