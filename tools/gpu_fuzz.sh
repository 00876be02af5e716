#!/bin/bash
# Randomised sweeps on one MI355X: fused train step + rank sweep vs the numpy oracle over random shapes / models
# (tools/fuzz_parity.py) and hipGraph replay vs eager (tools/fuzz_graph.py).  Non-zero exit on any violation.
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
ITERS=32 SEED=78 timeout 600 python tools/fuzz_parity.py > gpurun_out/fuzz_parity.log 2>&1; echo "fuzz_parity rc=$?"; tail -4 gpurun_out/fuzz_parity.log
ITERS=28 SEED=10 timeout 600 python tools/fuzz_graph.py > gpurun_out/fuzz_graph.log 2>&1; echo "fuzz_graph rc=$?"; tail -4 gpurun_out/fuzz_graph.log
ITERS=64 SEED=3 timeout 600 python tools/fuzz_pull.py > gpurun_out/fuzz_pull.log 2>&1; echo "fuzz_pull rc=$?"; tail -6 gpurun_out/fuzz_pull.log
