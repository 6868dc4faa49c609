#!/bin/bash
# Variant of the library that differs in ONE source: tools/variant_one.sh <name> <source basename> [-DFLAG=1 ...] -> ab_libs/lib<name>.so
# (the other objects are the product build's: run make -C pctrans_amd/csrc first)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift; shift
mkdir -p ab_libs/_obj_$name
FL="-DPCT_EXPERIMENT_BUILD -O3 -std=c++20 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -fvisibility=hidden -Wall -Wno-unused-result"
/opt/rocm/bin/hipcc $FL "$@" -c pctrans_amd/csrc/$src.hip -o ab_libs/_obj_$name/$src.o
objs=""
for o in pctrans_amd/csrc/_obj/*.o; do
  if [ "$(basename $o .o)" = "$src" ]; then objs="$objs ab_libs/_obj_$name/$src.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_libs/lib$name.so $objs
echo "built ab_libs/lib$name.so ($src: $*)"
