#!/bin/bash
# On the GPU box: alternate library variants built by tools/variant.sh (ab_libs/lib<name>.so) on the same box.
#   tools/ab_run.sh "<op-bench args>" <bench.py args or -> name[:ENV=VAL,...] ...
# For every variant: the MSDeformAttn op sweep (tools/bench_msda_op.py) and, unless BENCH=0, bench.py's roofline record.
cd "$GRAFT_REPO_ROOT"
opargs=$1; shift
mkdir -p gpurun_out
for spec in "$@"; do
  name=${spec%%:*}; envs=""
  if [ "$spec" != "$name" ]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
  cp ab_libs/lib$name.so pctrans_amd/lib/libpctrans_hip.so || exit 1
  echo "=== $spec"
  env $envs timeout -k 10 300 python3 tools/bench_msda_op.py $opargs || exit 1
  if [ "${BENCH:-1}" != 0 ]; then
    env $envs timeout -k 10 400 python3 bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline ${BENCHARGS} > gpurun_out/ab_bench_$name.json || exit 1
    python3 - gpurun_out/ab_bench_$name.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("bench: %.1f samples/s  %.2f ms/step  msda %.4f ms  frac %.4f  kernel %s" % (d["value"], d["ms_per_step"], r["mean_launch_ms"], r["frac"], r.get("kernel", "")[:40]))
PY
  fi
done
