cd $GRAFT_REPO_ROOT
for kb in 0 60 70 78; do echo "== PCT_COL_POOL_KB=$kb"; PCT_COL_POOL_KB=$kb timeout -k 10 200 python3 tools/bench_msda_op.py --shapes P2 --batches 128 --dists I,M --dtypes f32 2>/dev/null; done
