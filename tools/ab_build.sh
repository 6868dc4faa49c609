#!/bin/bash
# Same-box A/B of the library: A = HEAD (git stash), B = working tree.  Builds both into ab_libs/ (travels with gpurun);
# then on the GPU box:  for v in A B A B; do cp ab_libs/lib$v.so pctrans_amd/lib/libpctrans_hip.so; <bench>; done
set -e
cd "$(dirname "$0")/.."
if git diff --quiet HEAD -- pctrans_amd include; then
  echo "ab_build.sh: the working tree has no changes against HEAD under pctrans_amd/ or include/: A and B would be the same library" >&2
  exit 1
fi
mkdir -p ab_libs
make -C pctrans_amd/csrc >/dev/null
cp pctrans_amd/lib/libpctrans_hip.so ab_libs/libB.so
git stash -q
make -C pctrans_amd/csrc >/dev/null || { git stash pop -q; exit 1; }
cp pctrans_amd/lib/libpctrans_hip.so ab_libs/libA.so
git stash pop -q
make -C pctrans_amd/csrc >/dev/null
cmp ab_libs/libB.so pctrans_amd/lib/libpctrans_hip.so && echo "A = HEAD, B = working tree: built"
