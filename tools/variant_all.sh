#!/bin/bash
# Build a variant of the WHOLE library with extra compile flags on every source (flags that change something the translation
# units share, e.g. -DPCT_QUEUE_STRIDE=64); COLFLAGS adds flags for msda_forward_col.hip only (knock-outs):
#   [COLFLAGS="-DPCT_COL_KO_..."] tools/variant_all.sh <name> [-DFLAG=1 ...]   ->  ab_libs/lib<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p ab_libs/_obj_$name
FL="-DPCT_EXPERIMENT_BUILD -O3 -std=c++20 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -fvisibility=hidden -Wall -Wno-unused-result"
objs=""
n=0
for src in pctrans_amd/csrc/*.hip; do
  b=$(basename $src .hip)
  extra=""
  if [ "$b" = msda_forward_col ]; then extra="$COLFLAGS"; fi
  if [ "$b" = masked_attention ] || [ "$b" = cross_attention ]; then extra="-fno-slp-vectorize"; fi
  /opt/rocm/bin/hipcc $FL "$@" $extra -c $src -o ab_libs/_obj_$name/$b.o &
  objs="$objs ab_libs/_obj_$name/$b.o"
  n=$((n+1))
  if [ $((n % 8)) = 0 ]; then wait; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_libs/lib$name.so $objs
echo "built ab_libs/lib$name.so ($* | col: $COLFLAGS)"
