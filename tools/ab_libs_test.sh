#!/bin/bash
# On the GPU box: run a pytest selection against a library variant (ab_libs/lib<name>.so), then restore the product library.
#   tools/ab_libs_test.sh <name> <pytest args...>
cd "$GRAFT_REPO_ROOT"
name=$1; shift
cp pctrans_amd/lib/libpctrans_hip.so /tmp/prod_t.so
cp ab_libs/lib$name.so pctrans_amd/lib/libpctrans_hip.so || exit 1
PCT_ALLOW_EXPERIMENT_BUILD=1 timeout -k 10 600 python3 -m pytest "$@" 2>&1 | tail -15
rc=$?
cp /tmp/prod_t.so pctrans_amd/lib/libpctrans_hip.so
exit $rc
