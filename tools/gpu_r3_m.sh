#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_hip_edges.py -x -q -m gpu --timeout 400 -k "bench_two_ranks" > $O/m3_bench2.log 2>&1; tail -12 $O/m3_bench2.log | cut -c1-400
