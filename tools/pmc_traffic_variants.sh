#!/bin/bash
# On the GPU box: memory-side read/write requests of the MSDeformAttn op for several library variants (ab_libs/lib<name>.so):
#   tools/pmc_traffic_variants.sh "<prof_msda_one args>" name1 name2 ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
args=$1; shift
for name in "$@"; do
  cp $R/ab_libs/lib$name.so $R/pctrans_amd/lib/libpctrans_hip.so || exit 1
  rm -rf $R/gpurun_out/pmcv_$name
  i=0
  for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
             "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" ${EXTRA_GROUPS:+"$EXTRA_GROUPS"}; do
    i=$((i+1))
    timeout -k 10 180 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmcv_$name/g$i -- python3 $R/tools/prof_msda_one.py $args > $R/gpurun_out/pmcv_${name}_g$i.log 2>&1 || echo "group $i failed"
  done
  python3 - "$name" <<'PY'
import csv, glob, os, sys, collections
R = os.environ['GRAFT_REPO_ROOT']; name = sys.argv[1]
vals = collections.defaultdict(list); dur = []
for f in glob.glob(R + '/gpurun_out/pmcv_%s/**/*counter_collection.csv' % name, recursive=True):
    for r in csv.DictReader(open(f)):
        if 'msda_forward' in r['Kernel_Name']:
            vals[r['Counter_Name']].append(float(r['Counter_Value']))
c = {k: sum(v[1:]) / max(1, len(v[1:])) for k, v in vals.items()}
rd = 128 * c.get("TCC_EA0_RDREQ_128B_sum", 0) + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 32 * c.get("TCC_EA0_RDREQ_32B_sum", 0)
w64 = c.get("TCC_EA0_WRREQ_64B_sum", 0); wr = 64 * w64 + 32 * (c.get("TCC_EA0_WRREQ_sum", 0) - w64)
extra = " ".join("%s=%.4g" % (k, v) for k, v in sorted(c.items()) if not k.startswith("TCC_EA0"))
print("%-14s read %.3f GB  write %.3f GB  total %.3f GB   %s" % (name, rd / 1e9, wr / 1e9, (rd + wr) / 1e9, extra))
PY
done
