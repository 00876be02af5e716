#!/bin/bash
# Dev tool: build a variant of kge_head.hip (extra -D flags) into tools/_libs/<name>.so for a same-box A/B through KGE_HIP_LIB.
# Usage: bash tools/head_variant.sh <name> [-DHEAD_PD=2 -DHEAD_OCC=3 ...]
set -e
cd "$(dirname "$0")/../pykg2vec_amd/csrc"
name=$1; shift
mkdir -p ../../tools/_libs /tmp/kge_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function "$@" -c kge_head.hip -o /tmp/kge_variants/head_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v kge_head.o) /tmp/kge_variants/head_$name.o -o ../../tools/_libs/$name.so
echo built tools/_libs/$name.so
