#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 400 python tools/mfma_models_perf.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/x_mfma_models.md
