#!/bin/bash
# roofline.traffic of bench.py: memory-side requests of the MSDeformAttn forward kernel, by request size, in separate
# --pmc passes (run on the GPU box from the repo root; writes gpurun_out/r01_msda_traffic_batch<B>.json for bench.py's default batch; copy it to profiles/).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_traffic
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_traffic/g$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_traffic_g$i.log 2>&1 || echo "pass $i failed"
done
B=$(python3 -c "import sys; sys.path.insert(0, \"$R\"); import bench; print(bench.parse().batch)")
python3 $R/tools/summarize_pmc.py --kernel msda_forward_win --batch $B --levels 4 --out $R/gpurun_out/r01_msda_traffic_batch$B.json $R/gpurun_out/pmc_traffic/g1 $R/gpurun_out/pmc_traffic/g2 $R/gpurun_out/pmc_traffic/g3 $R/gpurun_out/pmc_traffic/g4
find $R/gpurun_out/pmc_traffic -name "*.csv" -size +1M -delete
