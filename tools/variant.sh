#!/bin/bash
# Dev tool: build a variant of ONE translation unit (extra -D flags) into tools/_libs/<name>.so for a same-box A/B through KGE_HIP_LIB.
# Usage: bash tools/variant.sh <unit, e.g. kge_head> <name> [-DSOMETHING=2 ...]     (the other units come from csrc/build/)
set -e
cd "$(dirname "$0")/../pykg2vec_amd/csrc"
unit=$1; name=$2; shift 2
mkdir -p ../../tools/_libs /tmp/kge_variants
extra=$(sed -n "s/^FLAGS_${unit} := //p" Makefile)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function $extra "$@" -c $unit.hip -o /tmp/kge_variants/${unit}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v "/$unit.o") /tmp/kge_variants/${unit}_$name.o -o ../../tools/_libs/$name.so
echo built tools/_libs/$name.so
