#!/bin/bash
# Memory-side read requests of the MSDeformAttn forward kernel by request size (development tool; run on the GPU box).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_RDREQ_IO_32B_sum" "TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mem/g$i -- python3 $R/tools/bench_msda_fused.py 64 > $R/gpurun_out/pmc_mem_g$i.log 2>&1 || echo "group $i failed"
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
vals=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R+'/gpurun_out/pmc_mem/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'msda_forward' in k:
            vals['fused' if 'Lb1ELb0' in k or 'true, false' in k else 'plain'][r['Counter_Name']].append(float(r['Counter_Value']))
for var, d in vals.items():
    for k,v in sorted(d.items()):
        print(var, k, len(v), sum(v)/len(v))
PY
