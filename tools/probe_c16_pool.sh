cd $GRAFT_REPO_ROOT
for kb in 0 32 40 52; do echo "== PCT_COL_POOL_KB=$kb"; PCT_COL_POOL_KB=$kb timeout -k 10 200 python3 tools/bench_msda_op.py --shapes P3 --batches 1,4 --dists I,M --dtypes f16 --iters 50 2>/dev/null; done
