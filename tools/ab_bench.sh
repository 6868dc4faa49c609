#!/bin/bash
# On the GPU box: bench.py (short) for several library variants; prints samples/s, ms per step, linear1 and MSDeformAttn ms.
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  cp ab_libs/lib$v.so pctrans_amd/lib/libpctrans_hip.so || exit 1
  python3 bench.py --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline 2>/dev/null > gpurun_out/abb_$v.json
  python3 - $v <<'PY'
import json, sys
d = json.loads(open("gpurun_out/abb_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("%-10s %.1f samples/s  %.2f ms/step  linear1 %.3f ms  msda %.3f ms" % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline_mfma"]["mean_launch_ms"], d["roofline"]["mean_launch_ms"]))
PY
done
