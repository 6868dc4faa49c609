#!/bin/bash
# On the GPU box: alternate backward-kernel library variants (ab_libs/lib<name>.so) on the same box.
#   tools/ab_bwd.sh "<bench_msda_bwd args>" name[:ENV=VAL,...] ...
cd "$GRAFT_REPO_ROOT"
opargs=$1; shift
mkdir -p gpurun_out
for spec in "$@"; do
  name=${spec%%:*}; envs=""
  if [ "$spec" != "$name" ]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
  cp ab_libs/lib$name.so pctrans_amd/lib/libpctrans_hip.so || exit 1
  echo "=== $spec"
  env $envs timeout -k 10 200 python3 tools/bench_msda_bwd.py $opargs || exit 1
done
