#!/bin/bash
# Round 3, call AD: matrix-core rank sweep, operands of a k-step as ONE 16-byte LDS read per lane (default) vs four 4-byte reads
# (A/B library built with -DKGE_GEMM_B32READS): exactness tests, eval timings
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
ev() { timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,e in d['extra'].items(): print(k, 'step_us %.1f' % e['step_us'], 'eval_ms %.3f' % e['eval_ms_per_pass'], 'eval M/s %.3f' % (e['eval_test_triples_per_s']/1e6), 'TF %.1f' % e['eval_TFLOPs'], e['mode'][:30])
"; }
for rep in 1 2 3; do
echo "== 16-byte operand reads (default)"; unset KGE_HIP_LIB; ev
echo "== 4-byte operand reads"; KGE_HIP_LIB=$PWD/tools/_libs/libkge_gemm_b32reads.so ev
echo "== default + s_setprio 1 around the MFMA cluster"; KGE_HIP_LIB=$PWD/tools/_libs/libkge_gemm_prio.so ev
done 2>&1 | tee $O/ad3_ab.log
