#!/bin/bash
# Same-box A/B inside ONE gpurun lease (box-to-box spread is +-8 %, so accept / reject decisions need both arms on one box).
# Usage (through gpurun):  bash tools/gpu_ab.sh <tag> <rounds> "<arm A: env assignments>" "<arm B: env assignments>" -- <command ...>
#   an arm is a (possibly empty) list of VAR=value words, e.g. "KGE_RESCAL_SLAB=0" or "KGE_HIP_LIB=tools/_libs/variant.so";
#   the command's stdout of every run is appended to gpurun_out/<tag>_ab.log under a header naming the arm.
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:?tag}; ROUNDS=${2:?rounds}; ARM_A=$3; ARM_B=$4; shift 4
[ "$1" = "--" ] && shift
for r in $(seq 1 "$ROUNDS"); do
  for arm in "$ARM_A" "$ARM_B"; do
    echo "== round $r arm [$arm]" | tee -a gpurun_out/${TAG}_ab.log
    env $arm timeout 900 "$@" 2>> gpurun_out/${TAG}_ab.err | tee -a gpurun_out/${TAG}_ab.log | tail -c 1500
    echo
  done
done
