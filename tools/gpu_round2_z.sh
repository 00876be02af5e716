#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -x -q -m gpu --timeout 120 -k "rescal or Rescal or transr or TransR or graph or fuzz or ntn or NTN" > gpurun_out/z_tests.log 2>&1; tail -5 gpurun_out/z_tests.log
ONLY="NTN" N_EVAL=0 timeout 200 python tools/config_perf.py 2>&1 | grep -v amdgpu.ids
