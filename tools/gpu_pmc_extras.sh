#!/bin/bash
# PMC traffic (FETCH_SIZE / WRITE_SIZE, one counter per pass, each with --kernel-trace only) of the C2 / C3 / C4 records of bench.py
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
P="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --eval-triples 256"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/z_pmc_xf -o bench -- $P > gpurun_out/z_pmc_xf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/z_pmc_xw -o bench -- $P > gpurun_out/z_pmc_xw.log 2>&1
python tools/rocpd_pmc.py gpurun_out/r02_pmc_traffic_c2_c3_c4.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one counter per run, --kernel-trace only) -- $P ; MI355X, round 2 final build; kernels of the C2 ComplEx / C3 RotatE / C4 RESCAL records (train + eval) next to the headline ones.  FETCH_SIZE is raw: double it for 16-byte-per-lane streaming reads (MI355X_MICROARCH.md, HBM section)" gpurun_out/z_pmc_xf/bench_results.db gpurun_out/z_pmc_xw/bench_results.db
rm -rf gpurun_out/z_pmc_xf gpurun_out/z_pmc_xw
