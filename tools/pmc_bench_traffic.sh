#!/bin/bash
# roofline.traffic of bench.py: memory-side requests of the MSDeformAttn forward kernel, by request size, in separate
# --pmc passes (run on the GPU box from the repo root):   tools/pmc_bench_traffic.sh [M|I]
# writes gpurun_out/r04_msda_traffic_batch<B>_dist<D>.json for bench.py's default batch; copy it to profiles/.
cd /tmp && export TMPDIR=/tmp
export PCT_BENCH_SETTLE_BLOCKS=2    # the profiler does not survive a full-length settle with counters on (bench.py)
R=$GRAFT_REPO_ROOT
D=${1:-M}
rm -rf $R/gpurun_out/pmc_traffic_$D
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_traffic_$D/g$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --loc-dist $D > $R/gpurun_out/pmc_traffic_${D}_g$i.log 2>&1 || echo "pass $i failed"
done
B=$(python3 -c "import sys; sys.argv=['bench.py']; sys.path.insert(0, \"$R\"); import bench; print(bench.parse().batch)")
python3 $R/tools/summarize_pmc.py --kernel msda_forward_col --batch $B --levels 4 --dist $D --out $R/gpurun_out/r04_msda_traffic_batch${B}_dist$D.json $R/gpurun_out/pmc_traffic_$D/g1 $R/gpurun_out/pmc_traffic_$D/g2 $R/gpurun_out/pmc_traffic_$D/g3 $R/gpurun_out/pmc_traffic_$D/g4
find $R/gpurun_out/pmc_traffic_$D -name "*.csv" -size +1M -delete
