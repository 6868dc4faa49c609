cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/bench_msda_bwd.py --cases P1:8,P2:2,P2:8,P2:32,P2:128 --dists I,M,U 2>/dev/null > $R/gpurun_out/r03_bwd_op_P2_col.txt
PCT_MSDA_BWD_KERNEL=win python3 $R/tools/bench_msda_bwd.py --cases P1:1,P1:2,P1:4,P2:1,P2:2 --dists I,M 2>/dev/null > $R/gpurun_out/r03_bwd_small_win.txt
PCT_MSDA_BWD_KERNEL=col python3 $R/tools/bench_msda_bwd.py --cases P1:1,P1:2,P1:4,P2:1,P2:2 --dists I,M 2>/dev/null > $R/gpurun_out/r03_bwd_small_col.txt
rm -rf $R/gpurun_out/stats_bwd
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_bwd -- python3 $R/tools/bench_msda_bwd.py --cases P2:2,P2:8,P2:32 --dists M > $R/gpurun_out/r03_bwd_op_M_rocprof.txt 2>&1 || echo bwd stats failed
find $R/gpurun_out/stats_bwd -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r03_bwd_col_kernel_stats_P2_distM.csv \;
find $R/gpurun_out/stats_bwd -name "*.csv" -size +1M -delete
cat $R/gpurun_out/r03_bwd_op_P2_col.txt; echo; paste -d'\n' $R/gpurun_out/r03_bwd_small_win.txt $R/gpurun_out/r03_bwd_small_col.txt; head -4 $R/gpurun_out/r03_bwd_col_kernel_stats_P2_distM.csv | cut -c1-200
