#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
ITERS=${FUZZ_OWN:-96} SEED=5 timeout 900 python tools/fuzz_own.py > gpurun_out/fuzz_own.log 2>&1; echo "fuzz_own rc=$?"; tail -4 gpurun_out/fuzz_own.log
ITERS=64 SEED=31 timeout 900 python tools/fuzz_own.py > gpurun_out/fuzz_own2.log 2>&1; echo "fuzz_own (seed 31) rc=$?"; tail -3 gpurun_out/fuzz_own2.log
