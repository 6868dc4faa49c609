#!/bin/bash
# Build a variant of the library that differs only in the compile flags of ONE source file:
#   tools/variant_file.sh <name> <file-stem> [-DFLAG=1 ...]   ->  ab_libs/lib<name>.so   (travels to the GPU box with gpurun)
set -e
cd "$(dirname "$0")/.."
name=$1; stem=$2; shift; shift
mkdir -p ab_libs/_obj_$name
make -C pctrans_amd/csrc -j8 >/dev/null
FL="-DPCT_EXPERIMENT_BUILD -O3 -std=c++20 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -fvisibility=hidden -Wall -Wno-unused-result"
objs=""
for f in pctrans_amd/csrc/_obj/*.o; do
  b=$(basename $f .o)
  if [ "$b" = "$stem" ]; then
    /opt/rocm/bin/hipcc $FL "$@" -c pctrans_amd/csrc/$b.hip -o ab_libs/_obj_$name/$b.o
    objs="$objs ab_libs/_obj_$name/$b.o"
  else
    objs="$objs $f"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_libs/lib$name.so $objs
echo "built ab_libs/lib$name.so ($*)"
