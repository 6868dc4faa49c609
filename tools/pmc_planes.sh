#!/bin/bash
# SQ counters of the column kernel on the reference layout vs piece planes (development tool; run on the GPU box):
#   tools/pmc_planes.sh <tag> [dist] [batch]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
rm -rf $R/gpurun_out/pmc_$tag
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; do
  i=$((i+1))
  timeout -k 10 180 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag/g$i -- python3 $R/tools/prof_msda_planes.py "$@" > $R/gpurun_out/pmc_${tag}_g$i.log 2>&1 || echo "group $i failed"
done
python3 - "$tag" <<'PY'
import csv, glob, json, os, sys, collections
R = os.environ['GRAFT_REPO_ROOT']
tag = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + '/gpurun_out/pmc_%s/**/*counter_collection.csv' % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'msda_forward_col' in k:
            vals[k.split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, d in vals.items():
    out[k] = {c: sum(v[1:]) / max(1, len(v[1:])) for c, v in sorted(d.items())}
json.dump(out, open(R + '/gpurun_out/pmc_%s.json' % tag, 'w'), indent=1)
for k, c in out.items():
    print(k)
    print("   ", {n: ("%.4g" % v) for n, v in c.items()})
PY
