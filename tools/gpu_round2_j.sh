#!/bin/bash
# GPU call J: PMC of the matrix-core sweep (C2 evaluation): is the MFMA pipe busy?
set -x
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|SQ_BUSY|LDS_BANK|SQ_WAIT|SQ_ACTIVE_INST" | head -60 > gpurun_out/j_counters.txt
B='python tools/eval_only.py'
KGE_EVAL_GEMM=1 ONLY="C2 " N_EVAL=3134 timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/j_pmc1 -o t -- python tools/config_perf.py > gpurun_out/j_pmc1.log 2>&1
KGE_EVAL_GEMM=1 ONLY="C2 " timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE --kernel-trace -d gpurun_out/j_pmc2 -o t -- python tools/config_perf.py > gpurun_out/j_pmc2.log 2>&1
python tools/rocpd_pmc.py gpurun_out/j_pmc.json "C2 eval, matrix-core sweep, rocprofv3 --pmc, MI355X round 2" gpurun_out/j_pmc1/t_results.db gpurun_out/j_pmc2/t_results.db
rm -rf gpurun_out/j_pmc1 gpurun_out/j_pmc2
