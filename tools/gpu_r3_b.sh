#!/bin/bash
# Round 3, call B: device-built incidence index (equality with the numpy rule, full suite on it, bench line with setup_ms)
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 300 python -m pytest tests/test_hip_pull.py -x -q -m gpu --timeout 300 -k "device_built" > $O/b3_index.log 2>&1; tail -5 $O/b3_index.log
timeout 900 python -m pytest tests -q -m gpu --timeout 300 > $O/b3_tests.log 2>&1; tail -6 $O/b3_tests.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > $O/b3_bench20.json 2> $O/b3_bench20.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/b3_bench20.json').read().strip().splitlines()[-1])
print('value %.1f M  ms/step %.4f kernel %.4f setup %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('setup')))
print('small', d.get('train_reference_default_batch'))
PY
