#!/bin/bash
# GPU call C: owner-computes step -- parity tests, same-box A/B against the push path, kernel trace
set -x
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_hip_pull.py -x -q 2>&1 | tail -25 ) > gpurun_out/c_pytest_pull.log
KGE_PULL=1 timeout 150 python bench.py --no-cpu-baseline --no-extra-configs --steps 20 --warmup 5 > gpurun_out/c_bench20_pull.log 2> gpurun_out/c_bench20_pull.err || { tail -3 gpurun_out/c_bench20_pull.err; exit 1; }
KGE_PULL=0 timeout 150 python bench.py --no-cpu-baseline --no-extra-configs --steps 20 --warmup 5 > gpurun_out/c_bench20_push.log 2> gpurun_out/c_bench20_push.err
KGE_PULL=0 timeout 150 python bench.py --no-cpu-baseline --no-extra-configs > gpurun_out/c_bench200_push.log 2> gpurun_out/c_bench200_push.err
KGE_PULL=1 timeout 150 python bench.py --no-cpu-baseline --no-extra-configs > gpurun_out/c_bench200_pull.log 2> gpurun_out/c_bench200_pull.err
timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/c_prof -o bench -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs > gpurun_out/c_bench_prof.log 2>&1
DB=$(ls gpurun_out/c_prof/*.db 2>/dev/null | head -1)
python tools/rocpd_summary.py $DB gpurun_out/c_kernel_stats.md > /dev/null
rm -rf gpurun_out/c_prof
tail -5 gpurun_out/c_pytest_pull.log
tail -3 gpurun_out/c_bench20_pull.err
