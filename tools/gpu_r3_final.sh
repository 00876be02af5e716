#!/bin/bash
# Round 3, final evidence run: bench lines (default 200 steps, the driver's 20 steps), rocprofv3 kernel table of the bench
# command, PMC traffic passes (FETCH_SIZE / WRITE_SIZE in separate passes, --kernel-trace only), SQ counters of the headline kernel.
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python bench.py > $O/r03_bench_line.json 2> $O/r03_bench.err; head -c 400 $O/r03_bench_line.json; echo
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r03_bench_line_20steps.json 2> $O/r03_bench20.err; head -c 300 $O/r03_bench_line_20steps.json; echo
B="python bench.py --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/fp0 -o b -- $B > $O/r03_prof0.log 2>&1
python tools/rocpd_summary.py $O/fp0/b_results.db $O/r03_kernel_stats.md > /dev/null; head -14 $O/r03_kernel_stats.md | cut -c1-200
B2="python bench.py --no-cpu-baseline --no-extra-configs"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fp1 -o b -- $B2 > $O/r03_prof1.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/fp2 -o b -- $B2 > $O/r03_prof2.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --kernel-trace -d $O/fp3 -o b -- $B2 > $O/r03_prof3.log 2>&1
dbs=""; for p in 1 2 3; do [ -f $O/fp$p/b_results.db ] && dbs="$dbs $O/fp$p/b_results.db"; done
python tools/rocpd_pmc.py $O/r03_pmc_traffic.json "python bench.py --no-cpu-baseline --no-extra-configs; one counter set per pass (FETCH_SIZE | WRITE_SIZE | SQ_*), --kernel-trace only" $dbs
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_pmc_traffic.json'))
for k,v in d["kernels"].items():
    if "k_pull_step" in k or "k_eval_sweep" in k:
        print(k[:110], {c:(round(x.get("avg_KB",x.get("avg",0)),1)) for c,x in v.items() if c in ("FETCH_SIZE","WRITE_SIZE")})
PY
rm -rf $O/fp0 $O/fp1 $O/fp2 $O/fp3
