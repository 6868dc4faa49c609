cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_lin/g$i -- python3 $R/tools/prof_linear_one.py > $R/gpurun_out/pmc_lin_g$i.log 2>&1 || echo "group $i failed"
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
vals=collections.defaultdict(list)
for f in glob.glob(R+'/gpurun_out/pmc_lin/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'linear_k128' in r['Kernel_Name']:
            vals[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(vals.items()):
    print(k, len(v), sum(v)/len(v))
PY
