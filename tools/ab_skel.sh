#!/bin/bash
# On the GPU box: the skeleton knock-out (no records, no staging, no gather, no stores: WRONG RESULTS, timing only) and the
# product library at 1, 2 and 3 workgroups per CU, plus the skeleton's in-kernel stamps.  tools/variant.sh skel ... builds it.
cd "$GRAFT_REPO_ROOT"
cp pctrans_amd/lib/libpctrans_hip.so /tmp/prod.so
for wg in 1 2 3; do
  echo "=== product, $wg workgroups per CU"
  PCT_COL_GRID_WG=$wg timeout -k 10 200 python3 tools/bench_msda_op.py --shapes P2 --dists M,I --batches 128 --iters 30 || exit 1
done
cp ab_libs/libskel.so pctrans_amd/lib/libpctrans_hip.so
for wg in 1 2 3; do
  echo "=== skeleton, $wg workgroups per CU"
  PCT_COL_GRID_WG=$wg timeout -k 10 200 python3 tools/bench_msda_op.py --shapes P2 --dists M,I --batches 128 --iters 30 || exit 1
done
echo "=== skeleton stamps"
timeout -k 10 200 python3 tools/stamp_msda.py M 128 || exit 1
cp /tmp/prod.so pctrans_amd/lib/libpctrans_hip.so
