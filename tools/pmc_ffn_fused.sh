#!/bin/bash
# SQ counters of the fused FFN kernel at the bench's row count (development tool; GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_ffnf
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 160 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ffnf/g$i -- python3 $R/tools/bench_ffn.py > $R/gpurun_out/pmc_ffnf_g$i.log 2>&1 || echo "group $i failed"
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
vals=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R+'/gpurun_out/pmc_ffnf/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'ffn_fused' in r['Kernel_Name'] or 'linear_' in r['Kernel_Name']:
            vals[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in vals.items():
    print(k)
    c={n: sum(v)/len(v) for n,v in d.items()}
    for n in sorted(c): print("   %-28s %.4g" % (n, c[n]))
    if 'GRBM_GUI_ACTIVE' in c and 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
        cyc=c['GRBM_GUI_ACTIVE']/8
        print("   kernel cycles (GUI_ACTIVE/8) %.4g ; MFMA busy per SIMD %.4g = %.1f %%" % (cyc, c['SQ_VALU_MFMA_BUSY_CYCLES']/1024, 100*c['SQ_VALU_MFMA_BUSY_CYCLES']/1024/cyc))
PY
