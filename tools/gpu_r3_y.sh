#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_hip_transx.py -x -q -m gpu --timeout 120 > gpurun_out/y3_tests.log 2>&1; tail -5 gpurun_out/y3_tests.log | cut -c1-300
ITERS=96 SEED=5 timeout 900 python tools/fuzz_own.py > gpurun_out/fuzz_own.log 2>&1; echo "fuzz_own rc=$?"; tail -3 gpurun_out/fuzz_own.log
