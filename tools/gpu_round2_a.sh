#!/bin/bash
# GPU call A (round 2): parity suite, VALU micro-benchmark, bench at the driver's and the default step counts, kernel trace
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/a_pytest.log
timeout 120 tools/_libs/valu_bench > gpurun_out/a_valu_bench.txt 2>&1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/a_bench20.log 2> gpurun_out/a_bench20.err
timeout 600 python bench.py --no-cpu-baseline --no-extra-configs > gpurun_out/a_bench200.log 2> gpurun_out/a_bench200.err
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/a_prof -o bench -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/a_bench_prof.log 2>&1
DB=$(ls gpurun_out/a_prof/*/*.db gpurun_out/a_prof/*.db 2>/dev/null | head -1)
python tools/rocpd_summary.py $DB gpurun_out/a_kernel_stats.md > /dev/null
rm -rf gpurun_out/a_prof
tail -3 gpurun_out/a_pytest.log
head -c 600 gpurun_out/a_bench20.log
