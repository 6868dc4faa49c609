#!/usr/bin/env python3
"""Print the per-step times of bench.py JSON lines (files given on the command line)."""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], "settle", d.get("settle_ms"), "steps", d.get("step_ms"))
    except Exception as e:  # noqa: BLE001
        print(f, "unreadable:", e)
