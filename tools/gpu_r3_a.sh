#!/bin/bash
# Round 3, call A: full GPU suite + smoke on the hygiene build, row-traffic microbenchmark, and a counter-level look at the C2
# (ComplEx WN18RR) step in both forms (atomic scatter / staged rows) to decide the shape of the atomic-free pointwise step.
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu --timeout 300 > $O/a3_tests.log 2>&1; tail -4 $O/a3_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/a3_smoke.log 2>&1; tail -2 $O/a3_smoke.log
timeout 120 tools/_libs/gather_bench > $O/a3_gather_bench.txt 2>&1; tail -3 $O/a3_gather_bench.txt
rocprofv3 -L > $O/a3_counters.txt 2>&1
for st in 0 1; do
  echo "== KGE_STAGED=$st"; ONLY="C2 " N_EVAL=64 KGE_STAGED=$st timeout 200 python tools/config_perf.py 2>&1 | tail -1
done | tee $O/a3_c2_ab.log
C="python tools/config_perf.py"
export ONLY="C2 " N_EVAL=64 KGE_STAGED=1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/a3_p0 -o c2 -- $C > $O/a3_p0.log 2>&1
python tools/rocpd_summary.py $O/a3_p0/c2_results.db $O/a3_c2_staged_kernels.md > /dev/null
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --kernel-trace -d $O/a3_p1 -o c2 -- $C > $O/a3_p1.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --kernel-trace -d $O/a3_p2 -o c2 -- $C > $O/a3_p2.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/a3_p3 -o c2 -- $C > $O/a3_p3.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/a3_p4 -o c2 -- $C > $O/a3_p4.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/a3_p5 -o c2 -- $C > $O/a3_p5.log 2>&1
dbs=""; for p in 1 2 3 4 5; do [ -f $O/a3_p$p/c2_results.db ] && dbs="$dbs $O/a3_p$p/c2_results.db"; done
python tools/rocpd_pmc.py $O/a3_c2_staged_pmc.json "KGE_STAGED=1 ONLY='C2 ' tools/config_perf.py, one counter set per pass" $dbs
export KGE_STAGED=0
timeout 300 rocprofv3 --kernel-trace --stats -d $O/a3_q0 -o c2 -- $C > $O/a3_q0.log 2>&1
python tools/rocpd_summary.py $O/a3_q0/c2_results.db $O/a3_c2_atomic_kernels.md > /dev/null
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --kernel-trace -d $O/a3_q1 -o c2 -- $C > $O/a3_q1.log 2>&1
python tools/rocpd_pmc.py $O/a3_c2_atomic_pmc.json "KGE_STAGED=0 ONLY='C2 ' tools/config_perf.py" $O/a3_q1/c2_results.db
unset ONLY N_EVAL KGE_STAGED
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/a3_bench20.json 2> $O/a3_bench20.err
head -c 600 $O/a3_bench20.json
rm -rf $O/a3_p? $O/a3_q?
