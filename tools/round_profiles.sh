# End-of-round records (run on the GPU box from the repo root: tools/round_profiles.sh); copy what matters to profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${T:-r04}
python3 $R/bench.py > $R/gpurun_out/${T}_bench_distM.json 2> $R/gpurun_out/${T}_bench_distM.err || echo bench failed
python3 $R/bench.py --loc-dist I --no-cpu-baseline > $R/gpurun_out/${T}_bench_distI.json 2> $R/gpurun_out/${T}_bench_distI.err || echo benchI failed
rm -rf $R/gpurun_out/stats_b
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_b -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/stats_b.log 2>&1 || echo stats failed
find $R/gpurun_out/stats_b -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${T}_bench_kernel_stats_batch128_distM.csv \;
find $R/gpurun_out/stats_b -name "*.csv" -size +1M -delete
python3 $R/tools/bench_msda_op.py --shapes P2 --batches 8,32,128 --dists I,M,U --dtypes f32 > $R/gpurun_out/${T}_op_col_P2.txt 2>/dev/null
python3 $R/tools/bench_msda_op.py --shapes P1,P4 --batches 8,128 --dists I,M --dtypes f32 >> $R/gpurun_out/${T}_op_col_P2.txt 2>/dev/null
python3 $R/tools/bench_msda_op.py --shapes P3 --batches 1,4 --dists I,M,U --dtypes f16,bf16 --iters 50 > $R/gpurun_out/${T}_op_col16_P3.txt 2>/dev/null
python3 $R/tools/bench_msda_op.py --shapes P2 --batches 8,128 --dists I,M --dtypes f16,bf16 >> $R/gpurun_out/${T}_op_col16_P3.txt 2>/dev/null
# MSDeformAttn backward (pyramid-column kernel): op table, kernel stats (rocprofv3) of the op bench
bash $R/tools/record_bwd.sh > $R/gpurun_out/${T}_bwd_record.txt 2>&1
