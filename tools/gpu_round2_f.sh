#!/bin/bash
# GPU call F: segment-length sweep of the owner-computes step (host-side parameter), same box
set -x
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for seg in 4 6 8 12 16; do
  KGE_PULL_SEGMENT=$seg timeout 150 python bench.py --no-cpu-baseline --no-extra-configs > gpurun_out/f_bench200_seg$seg.log 2> gpurun_out/f_bench200_seg$seg.err
done
