#!/bin/bash
# Round 3, call H: matrix-core rank sweep on v_mfma_f32_16x16x4_f32 (default) vs 32x32x2 (A/B library): exactness tests, eval timings
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 400 python -m pytest tests/test_hip_parity.py tests/test_fullsize_golden.py tests/test_hip_fullsize_configs.py -x -q -m gpu --timeout 200 -k "matrix_core or eval or rank or fullsize" > $O/h3_tests.log 2>&1; tail -4 $O/h3_tests.log | cut -c1-300
ev() { timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,e in d['extra'].items(): print(k, 'step_us %.1f' % e['step_us'], 'eval_ms %.3f' % e['eval_ms_per_pass'], 'eval M/s %.3f' % (e['eval_test_triples_per_s']/1e6), 'TF %.1f' % e['eval_TFLOPs'], e['mode'][:30])
"; }
for rep in 1 2; do
echo "== 16x16x4 (default)"; unset KGE_HIP_LIB; ev
echo "== 32x32x2"; KGE_HIP_LIB=$PWD/tools/_libs/libkge_gemm32.so ev
done 2>&1 | tee $O/h3_ab.log
