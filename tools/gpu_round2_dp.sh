#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
KGE_BENCH_SHARE_GPU=1 KGE_BENCH_CHECK_REPLICAS=1 KGE_PULL=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 6 --warmup 2 --batch 4096 --eval-triples 256 > gpurun_out/dp_out.log 2> gpurun_out/dp_err.log
grep -v "^W\|^$" gpurun_out/dp_err.log | grep -B2 -A12 "Traceback" | head -60
tail -3 gpurun_out/dp_out.log | cut -c1-300
