#!/bin/bash
# GPU call E: owner-computes step, 16-lane vs 32-lane owner groups (same box), parity first
set -x
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_hip_pull.py -x -q 2>&1 | tail -15 ) > gpurun_out/e_pytest_pull.log
( KGE_PULL_G=32 timeout 300 python -m pytest tests/test_hip_pull.py -x -q 2>&1 | tail -5 ) > gpurun_out/e_pytest_pull_g32.log
timeout 150 python bench.py --no-cpu-baseline --no-extra-configs --steps 20 --warmup 5 > gpurun_out/e_bench20_g16.log 2> gpurun_out/e_bench20_g16.err || { tail -3 gpurun_out/e_bench20_g16.err; exit 1; }
KGE_PULL_G=32 timeout 150 python bench.py --no-cpu-baseline --no-extra-configs > gpurun_out/e_bench200_g32.log 2> gpurun_out/e_bench200_g32.err
timeout 150 python bench.py --no-cpu-baseline --no-extra-configs > gpurun_out/e_bench200_g16.log 2> gpurun_out/e_bench200_g16.err
timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/e_prof -o bench -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs > gpurun_out/e_bench_prof.log 2>&1
python tools/rocpd_summary.py gpurun_out/e_prof/bench_results.db gpurun_out/e_kernel_stats.md > /dev/null
rm -rf gpurun_out/e_prof
tail -3 gpurun_out/e_pytest_pull.log gpurun_out/e_pytest_pull_g32.log
