#!/bin/bash
# One gpurun lease = the round's evidence set.  Usage (through gpurun):  bash tools/gpu_evidence.sh <tag> [parts]
#   tag    prefix of everything written under gpurun_out/ (e.g. r04a); copy what is to be judged into profiles/
#   parts  any of: tests smoke bench bench20 stats pmc fuzz   (default: all but fuzz, in that order)
#     tests    pytest -m gpu                                   -> <tag>_tests.log
#     smoke    __graft_entry__.smoke()
#     bench    python bench.py (default flags: 200 steps, live counter passes, extras, CPU baseline) -> <tag>_bench_line.json
#     bench20  the driver's shape: --steps 20 --warmup 5        -> <tag>_bench_line_20steps.json
#     stats    rocprofv3 --kernel-trace --stats of the bench command, split by launch geometry -> <tag>_kernel_stats.md
#     fuzz     the randomised sweeps tools/fuzz_*.py (parity, graph, pull, staged, own, eval gemm, big paths, round-4 entry points, 1-N head)
#                                                                -> <tag>_fuzz.txt
#     pmc      FETCH_SIZE / WRITE_SIZE / SQ_* passes (separate passes, --kernel-trace only) of the bench command,
#              per (kernel, grid)                               -> <tag>_pmc_traffic.json
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:?tag}; shift
PARTS=${*:-tests smoke bench bench20 stats pmc}
O=gpurun_out
for part in $PARTS; do
  case $part in
    tests)   timeout 1500 python -m pytest tests -x -q -m gpu --timeout 300 > $O/${TAG}_tests.log 2>&1; tail -4 $O/${TAG}_tests.log ;;
    smoke)   timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
    bench)   timeout 900 python bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench.err; cp $O/bench_detail.json $O/${TAG}_bench_detail.json 2>/dev/null
             tail -n 1 $O/${TAG}_bench_line.json | head -c 600; echo; tail -3 $O/${TAG}_bench.err ;;
    bench20) timeout 900 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_line_20steps.json 2> $O/${TAG}_bench20.err; cp $O/bench_detail.json $O/${TAG}_bench_detail_20steps.json 2>/dev/null
             tail -n 1 $O/${TAG}_bench_line_20steps.json | head -c 300; echo ;;
    stats)   timeout 600 rocprofv3 --kernel-trace --stats -d $O/_p0 -o b -- python bench.py --no-cpu-baseline --no-live-pmc > $O/${TAG}_prof0.log 2>&1
             python tools/rocpd_summary.py $(find $O/_p0 -name '*.db' | head -1) $O/${TAG}_kernel_stats.md > /dev/null; head -16 $O/${TAG}_kernel_stats.md | cut -c1-180
             rm -rf $O/_p0 ;;
    pmc)     B2="python bench.py --no-cpu-baseline --no-live-pmc"
             timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/_p1 -o b -- $B2 > $O/${TAG}_prof1.log 2>&1
             timeout 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/_p2 -o b -- $B2 > $O/${TAG}_prof2.log 2>&1
             timeout 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --kernel-trace -d $O/_p3 -o b -- $B2 > $O/${TAG}_prof3.log 2>&1
             python tools/rocpd_pmc.py $O/${TAG}_pmc_traffic.json "$B2; one counter set per pass (FETCH_SIZE | WRITE_SIZE | SQ_*), --kernel-trace only" $(find $O/_p1 $O/_p2 $O/_p3 -name '*.db')
             rm -rf $O/_p1 $O/_p2 $O/_p3 ;;
    fuzz)    for spec in "fuzz_parity 120 77" "fuzz_graph 60 9" "fuzz_pull 48 3" "fuzz_staged 40 21" "fuzz_own 64 5" "fuzz_eval_gemm 60 31" "fuzz_big_paths 40 41" "fuzz_r4 60 1" "fuzz_head 60 3"; do
               set -- $spec
               ITERS=$2 SEED=$3 timeout 900 python tools/$1.py > $O/_fuzz_$1.log 2>&1; echo "== $1 ITERS=$2 SEED=$3 rc=$?" | tee -a $O/${TAG}_fuzz.txt; tail -4 $O/_fuzz_$1.log | tee -a $O/${TAG}_fuzz.txt
             done ;;
    *) echo "unknown part $part" ;;
  esac
done
