#!/bin/bash
# Issue-side counters of the step's hand-written kernels inside bench.py (GPU box, repo root): which unit each one keeps busy.
#   tools/pmc_bench_issue.sh   ->  gpurun_out/r04_bench_issue.txt  (copy to profiles/)
cd /tmp && export TMPDIR=/tmp
export PCT_BENCH_SETTLE_BLOCKS=2    # the profiler does not survive a full-length settle with counters on (bench.py)
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_issue
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_issue/g$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_issue_g$i.log 2>&1 || echo "pass $i failed"
done
python3 - > $R/gpurun_out/r04_bench_issue.txt <<'PY'
import csv, glob, os, collections
R = os.environ['GRAFT_REPO_ROOT']
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + '/gpurun_out/pmc_issue/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'pct::' in k:
            vals[k.split('(')[0][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
print("per launch (mean over the launches of a bench run); SQ_* quad-cycle counters x 4 = cycles; 1 024 SIMDs")
for k, d in sorted(vals.items(), key=lambda kv: -sum(kv[1].get('GRBM_GUI_ACTIVE', [0]))):
    c = {n: sum(v) / len(v) for n, v in d.items()}
    if 'GRBM_GUI_ACTIVE' not in c:
        continue
    cyc = c['GRBM_GUI_ACTIVE'] / 8
    n = len(d['GRBM_GUI_ACTIVE'])
    line = "%-70s launches %4d  cycles %10.0f" % (k, n, cyc)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
        line += "  mfma_busy %5.1f%%" % (100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc)
    if 'SQ_ACTIVE_INST_VALU' in c:
        line += "  valu_issue %5.1f%%" % (100 * 4 * c['SQ_ACTIVE_INST_VALU'] / 1024 / cyc)
    if 'SQ_ACTIVE_INST_LDS' in c:
        line += "  lds_issue %5.1f%%" % (100 * 4 * c['SQ_ACTIVE_INST_LDS'] / 1024 / cyc)
    if 'SQ_INSTS_VALU' in c:
        line += "  valu/wave-cycle insts %.3g" % c['SQ_INSTS_VALU']
    if 'SQ_WAVE_CYCLES' in c and 'SQ_WAIT_ANY' in c:
        line += "  wait_any/wave_cycles %4.1f%%" % (100 * c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'])
    print(line)
PY
find $R/gpurun_out/pmc_issue -name "*.csv" -size +1M -delete
cat $R/gpurun_out/r04_bench_issue.txt
