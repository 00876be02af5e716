#!/bin/bash
set -x
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 120 tools/_libs/mfma_bench > gpurun_out/r02_mfma_bench.txt 2>&1; cat gpurun_out/r02_mfma_bench.txt
P="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --eval-triples 2048"
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace -d gpurun_out/z_pmc_m -o bench -- $P > gpurun_out/z_pmc_m.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --kernel-trace -d gpurun_out/z_pmc_m2 -o bench -- $P > gpurun_out/z_pmc_m2.log 2>&1
tail -3 gpurun_out/z_pmc_m2.log
python tools/rocpd_pmc.py gpurun_out/r02_pmc_gemm.json "rocprofv3 --pmc (two passes) --kernel-trace -- $P ; MI355X round 2; matrix-core kernels of the C2/C3/C4 extra records" gpurun_out/z_pmc_m/bench_results.db gpurun_out/z_pmc_m2/bench_results.db
python tools/rocpd_summary.py gpurun_out/z_pmc_m/bench_results.db gpurun_out/p_kernel_table.md > /dev/null
rm -rf gpurun_out/z_pmc_m gpurun_out/z_pmc_m2
