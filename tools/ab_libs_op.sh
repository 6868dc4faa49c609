#!/bin/bash
# On the GPU box: alternate library variants (ab_libs/lib<name>.so; "prod" = the product library) on the MSDeformAttn op bench.
#   tools/ab_libs_op.sh "<bench_msda_op args>" <rounds> name[:ENV=VAL,...] ...
cd "$GRAFT_REPO_ROOT"
cp pctrans_amd/lib/libpctrans_hip.so /tmp/prod.so
opargs=$1; rounds=$2; shift; shift
for r in $(seq 1 $rounds); do
  for spec in "$@"; do
    name=${spec%%:*}; envs=""
    if [ "$spec" != "$name" ]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
    if [ "$name" = prod ]; then cp /tmp/prod.so pctrans_amd/lib/libpctrans_hip.so; else cp ab_libs/lib$name.so pctrans_amd/lib/libpctrans_hip.so || exit 1; fi
    echo "=== round $r: $spec"
    env $envs timeout -k 10 300 python3 tools/bench_msda_op.py $opargs 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
cp /tmp/prod.so pctrans_amd/lib/libpctrans_hip.so
