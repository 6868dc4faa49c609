#!/bin/bash
# Variant of msda_forward_col.hip on top of a whole-library variant built by tools/variant_all.sh (its other objects are reused):
#   tools/variant_col_from.sh <base> <name> [-DFLAG ...]  ->  ab_libs/lib<name>.so   (flags of the base must be repeated)
set -e
cd "$(dirname "$0")/.."
base=$1; name=$2; shift; shift
mkdir -p ab_libs/_obj_$name
FL="-DPCT_EXPERIMENT_BUILD -O3 -std=c++20 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -fvisibility=hidden -Wall -Wno-unused-result"
/opt/rocm/bin/hipcc $FL "$@" -c pctrans_amd/csrc/msda_forward_col.hip -o ab_libs/_obj_$name/msda_forward_col.o
objs=""
for f in ab_libs/_obj_$base/*.o; do
  b=$(basename $f .o)
  if [ "$b" = msda_forward_col ]; then objs="$objs ab_libs/_obj_$name/$b.o"; else objs="$objs $f"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_libs/lib$name.so $objs
echo "built ab_libs/lib$name.so from $base ($*)"
