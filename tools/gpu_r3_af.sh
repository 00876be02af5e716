#!/bin/bash
# Round 3, call AF: k_eval_gemm with static wave priorities that differ between co-resident workgroups (two guesses of the
# workgroup -> CU placement), against the default (all waves at priority 0)
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
for rep in 1 2 3; do
for lib in default prio2 prio3; do
  if [ $lib = default ]; then unset KGE_HIP_LIB; else export KGE_HIP_LIB=$PWD/tools/_libs/libkge_gemm_$lib.so; fi
  for sh in c2 c3 c4; do echo "$lib $(SHAPE=$sh REPS=5 timeout 200 python tools/eval_only.py 2>&1 | tail -1)"; done
done; done 2>&1 | tee $O/af3_ab.log
