#!/bin/bash
# On the GPU box: run a command against library variants in turn:  tools/ab_libs_cmd.sh "<command>" <rounds> name ...   ("prod" = the product library)
cd "$GRAFT_REPO_ROOT"
cp pctrans_amd/lib/libpctrans_hip.so /tmp/prod_c.so
cmd=$1; rounds=$2; shift; shift
for r in $(seq 1 $rounds); do
  for name in "$@"; do
    if [ "$name" = prod ]; then cp /tmp/prod_c.so pctrans_amd/lib/libpctrans_hip.so; else cp ab_libs/lib$name.so pctrans_amd/lib/libpctrans_hip.so || exit 1; fi
    echo "=== round $r: $name"
    timeout -k 10 300 bash -c "$cmd" 2>&1 | grep -v amdgpu.ids
  done
done
cp /tmp/prod_c.so pctrans_amd/lib/libpctrans_hip.so
