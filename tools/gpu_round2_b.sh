#!/bin/bash
# GPU call B: parity suite + bench at the driver's and the default step counts
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/b_pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/b_bench20.log 2> gpurun_out/b_bench20.err
timeout 600 python bench.py --no-cpu-baseline --no-extra-configs > gpurun_out/b_bench200.log 2> gpurun_out/b_bench200.err
tail -3 gpurun_out/b_pytest.log
head -c 300 gpurun_out/b_bench20.log
