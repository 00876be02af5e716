#!/bin/bash
# Round-2 evidence run: bench lines (driver's step count and default), same-box push-vs-pull A/B, kernel trace of the whole
# bench command, PMC passes (one counter set per run, each with --kernel-trace only)
set -x
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_line_20steps.json 2> gpurun_out/z_b20.err
timeout 600 python bench.py > gpurun_out/r02_bench_line.json 2> gpurun_out/z_b200.err
KGE_PULL=0 timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > gpurun_out/r02_bench_line_push_path.json 2> gpurun_out/z_push.err
KGE_PULL=0 timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 20 --warmup 5 > gpurun_out/r02_bench_line_push_path_20steps.json 2>> gpurun_out/z_push.err
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline"
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/z_prof -o bench -- $B > gpurun_out/z_prof.log 2>&1
python tools/rocpd_summary.py gpurun_out/z_prof/bench_results.db gpurun_out/r02_kernel_stats_table.md > /dev/null
P="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --eval-triples 2048"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/z_pmc_f -o bench -- $P > gpurun_out/z_pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/z_pmc_w -o bench -- $P > gpurun_out/z_pmc_w.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --kernel-trace -d gpurun_out/z_pmc_s -o bench -- $P > gpurun_out/z_pmc_s.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d gpurun_out/z_pmc_g -o bench -- $P > gpurun_out/z_pmc_g.log 2>&1
python tools/rocpd_pmc.py gpurun_out/r02_pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* / GRBM_GUI_ACTIVE (one counter set per run, each with --kernel-trace only) -- $P ; MI355X, round 2 final build; summarised by tools/rocpd_pmc.py" gpurun_out/z_pmc_f/bench_results.db gpurun_out/z_pmc_w/bench_results.db gpurun_out/z_pmc_s/bench_results.db gpurun_out/z_pmc_g/bench_results.db
rm -rf gpurun_out/z_prof gpurun_out/z_pmc_f gpurun_out/z_pmc_w gpurun_out/z_pmc_s gpurun_out/z_pmc_g
head -c 400 gpurun_out/r02_bench_line.json
