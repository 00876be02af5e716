#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
ITERS=128 SEED=17 timeout 1200 python tools/fuzz_own.py > gpurun_out/fuzz_own3.log 2>&1; echo "fuzz_own (seed 17) rc=$?"; tail -6 gpurun_out/fuzz_own3.log | cut -c1-400
