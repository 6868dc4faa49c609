#!/bin/bash
# On the GPU box: bench.py N times in a row (fresh processes), one summary line per run: value, settle blocks, slowest timed step.
cd "$GRAFT_REPO_ROOT"
n=${1:-3}; tag=${2:-rep}
for i in $(seq 1 $n); do
  python3 bench.py > gpurun_out/${tag}_$i.json 2> gpurun_out/${tag}_$i.err || exit 1
  python3 - "$i" gpurun_out/${tag}_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read())
sm = d["step_ms"]
print("run %s value %.1f ms/step %.2f settle %s  timed-step max %.1f (index %d) median %.2f  frac %.3f" % (
    sys.argv[1], d["value"], d["ms_per_step"], d["settle_ms"], max(sm), sm.index(max(sm)), sorted(sm)[len(sm) // 2],
    d["roofline"]["frac"]), flush=True)
PY
done
