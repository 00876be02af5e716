#!/bin/bash
# GPU call D: owner-computes step -- parity, bench A/B, kernel trace + PMC passes (FETCH_SIZE / WRITE_SIZE / SQ), each its own run
set -x
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_hip_pull.py -x -q 2>&1 | tail -15 ) > gpurun_out/d_pytest_pull.log
KGE_PULL=1 timeout 150 python bench.py --no-cpu-baseline --no-extra-configs --steps 20 --warmup 5 > gpurun_out/d_bench20_pull.log 2> gpurun_out/d_bench20_pull.err || { tail -3 gpurun_out/d_bench20_pull.err; exit 1; }
KGE_PULL=1 timeout 150 python bench.py --no-cpu-baseline --no-extra-configs > gpurun_out/d_bench200_pull.log 2> gpurun_out/d_bench200_pull.err
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --eval-triples 2048"
timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/d_prof -o bench -- $B > gpurun_out/d_prof.log 2>&1
python tools/rocpd_summary.py gpurun_out/d_prof/bench_results.db gpurun_out/d_kernel_stats.md > /dev/null
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/d_pmc_f -o bench -- $B > gpurun_out/d_pmc_f.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/d_pmc_w -o bench -- $B > gpurun_out/d_pmc_w.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --kernel-trace -d gpurun_out/d_pmc_s -o bench -- $B > gpurun_out/d_pmc_s.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --kernel-trace -d gpurun_out/d_pmc_g -o bench -- $B > gpurun_out/d_pmc_g.log 2>&1
python tools/rocpd_pmc.py gpurun_out/d_pmc.json "rocprofv3 --pmc (one counter set per run, each with --kernel-trace only) -- $B ; MI355X, round 2" gpurun_out/d_pmc_f/bench_results.db gpurun_out/d_pmc_w/bench_results.db gpurun_out/d_pmc_s/bench_results.db gpurun_out/d_pmc_g/bench_results.db
rm -rf gpurun_out/d_prof gpurun_out/d_pmc_f gpurun_out/d_pmc_w gpurun_out/d_pmc_s gpurun_out/d_pmc_g
tail -3 gpurun_out/d_pytest_pull.log
