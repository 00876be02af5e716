#!/bin/bash
# Randomised sweeps on one MI355X: fused train step + rank sweep vs the numpy oracle over random shapes / models
# (tools/fuzz_parity.py), hipGraph replay vs eager (tools/fuzz_graph.py), owner-computes vs atomic path (tools/fuzz_pull.py),
# staged vs atomic path (tools/fuzz_staged.py).  Non-zero exit code printed on any violation.
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
ITERS=${FUZZ_PARITY:-160} SEED=77 timeout 600 python tools/fuzz_parity.py > gpurun_out/fuzz_parity.log 2>&1; echo "fuzz_parity rc=$?"; tail -2 gpurun_out/fuzz_parity.log
ITERS=${FUZZ_GRAPH:-84} SEED=9 timeout 600 python tools/fuzz_graph.py > gpurun_out/fuzz_graph.log 2>&1; echo "fuzz_graph rc=$?"; tail -2 gpurun_out/fuzz_graph.log
ITERS=${FUZZ_PULL:-64} SEED=3 timeout 600 python tools/fuzz_pull.py > gpurun_out/fuzz_pull.log 2>&1; echo "fuzz_pull rc=$?"; tail -4 gpurun_out/fuzz_pull.log
ITERS=${FUZZ_STAGED:-60} SEED=21 timeout 600 python tools/fuzz_staged.py > gpurun_out/fuzz_staged.log 2>&1; echo "fuzz_staged rc=$?"; tail -6 gpurun_out/fuzz_staged.log
ITERS=${FUZZ_OWN:-96} SEED=5 timeout 900 python tools/fuzz_own.py > gpurun_out/fuzz_own.log 2>&1; echo "fuzz_own rc=$?"; tail -6 gpurun_out/fuzz_own.log
ITERS=${FUZZ_EVAL_GEMM:-80} SEED=31 timeout 900 python tools/fuzz_eval_gemm.py > gpurun_out/fuzz_eval_gemm.log 2>&1; echo "fuzz_eval_gemm rc=$?"; tail -6 gpurun_out/fuzz_eval_gemm.log
ITERS=${FUZZ_BIG:-60} SEED=41 timeout 600 python tools/fuzz_big_paths.py > gpurun_out/fuzz_big_paths.log 2>&1; echo "fuzz_big_paths rc=$?"; tail -4 gpurun_out/fuzz_big_paths.log
