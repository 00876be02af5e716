#!/bin/bash
# Round 3, call D: own-step parity (bounded), A/B of pipelined / single-buffer visits and of the segment length, counters
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 240 python -m pytest tests/test_hip_own.py -x -q -m gpu --timeout 60 -k 'reproducible or oracle' > $O/d3_own.log 2>&1; tail -8 $O/d3_own.log | cut -c1-300
run() { ONLY="$1" N_EVAL=64 timeout 120 python tools/config_perf.py 2>&1 | tail -1; }
ab() { echo "== lib=${1:-default(pipelined)} seg=$2"; if [ -n "$1" ]; then export KGE_HIP_LIB=$PWD/$1; else unset KGE_HIP_LIB; fi; KGE_PULL_SEGMENT=$2 run "C2 "; KGE_PULL_SEGMENT=$2 run "DistMult"; unset KGE_HIP_LIB; }
for seg in 8 4; do ab "" $seg; ab tools/_libs/libkge_own_single.so $seg; done 2>&1 | tee $O/d3_ab.log
export ONLY="C2 " N_EVAL=64
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --kernel-trace -d $O/d3_p1 -o c2 -- python tools/config_perf.py > $O/d3_p1.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -d $O/d3_p2 -o c2 -- python tools/config_perf.py > $O/d3_p2.log 2>&1
python tools/rocpd_pmc.py $O/d3_c2_own_pmc.json "ONLY='C2 ' tools/config_perf.py, own path" $O/d3_p1/c2_results.db $O/d3_p2/c2_results.db
rm -rf $O/d3_p1 $O/d3_p2
