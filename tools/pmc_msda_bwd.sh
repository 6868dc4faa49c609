#!/bin/bash
# SQ + memory-side counters of ONE MSDeformAttn backward configuration (development tool; run on the GPU box):
#   tools/pmc_msda_bwd.sh <tag> <case e.g. P2:32> <dist>
# One rocprofv3 --pmc pass per counter group (never mixed with trace domains), summary JSON under gpurun_out/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; cs=$2; dist=$3
rm -rf $R/gpurun_out/pmcb_$tag
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum"; do
  i=$((i+1))
  timeout -k 10 180 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmcb_$tag/g$i -- python3 $R/tools/bench_msda_bwd.py --cases $cs --dists $dist --iters 4 > $R/gpurun_out/pmcb_${tag}_g$i.log 2>&1 || echo "group $i failed"
done
python3 - "$tag" "$cs" "$dist" <<'PY'
import csv, glob, json, os, sys, collections
R = os.environ['GRAFT_REPO_ROOT']
tag = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + '/gpurun_out/pmcb_%s/**/*counter_collection.csv' % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'msda_backward' in k:
            vals[k.split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
out = {"workload": "tools/bench_msda_bwd.py --cases %s --dists %s" % (sys.argv[2], sys.argv[3]), "counters": {}}
for k, d in vals.items():
    out["counters"][k] = {c: sum(v[1:]) / max(1, len(v[1:])) for c, v in sorted(d.items())}   # first launch dropped
    c = out["counters"][k]
    if "TCC_EA0_RDREQ_128B_sum" in c:
        c["hbm_read_bytes"] = 128 * c["TCC_EA0_RDREQ_128B_sum"] + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 32 * c.get("TCC_EA0_RDREQ_32B_sum", 0)
    if "TCC_EA0_WRREQ_sum" in c:
        w64 = c.get("TCC_EA0_WRREQ_64B_sum", 0)
        c["hbm_write_bytes"] = 64 * w64 + 32 * (c["TCC_EA0_WRREQ_sum"] - w64)
    if "SQ_WAVES" in c and c["SQ_WAVES"]:
        c["valu_per_wave"] = c.get("SQ_INSTS_VALU", 0) / c["SQ_WAVES"]
json.dump(out, open(R + '/gpurun_out/pmcb_%s.json' % tag, 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
