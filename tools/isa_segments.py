#!/usr/bin/env python3
"""Static instruction mix of one kernel in a `hipcc -S` listing, split at s_barrier (development tool).
    python tools/isa_segments.py listing.s <substring of the mangled kernel name>"""
import collections
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
i = s.index("\n" + [l for l in s.split("\n") if l.startswith("_Z") and name in l.split(":")[0] and ":" in l][0])
j = s.index(".Lfunc_end", i)
segs, cur = [], collections.Counter()
for ln in s[i:j].split("\n"):
    t = ln.strip()
    if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    if op == "s_barrier":
        segs.append(cur)
        cur = collections.Counter()
        continue
    if op.startswith("v_"):
        cur["valu"] += 1
        for key in ("fma_mix", "pk_fma", "readfirstlane", "readlane", "writelane", "_f64", "_dpp"):
            if key in op or key in t:
                cur[key] += 1
    elif op.startswith("ds_"):
        cur["lds"] += 1
    elif op.startswith(("global_", "buffer_")):
        cur["vmem"] += 1
    elif op.startswith("scratch_"):
        cur["scratch"] += 1
    elif op.startswith("s_"):
        cur["salu"] += 1
segs.append(cur)
tot = collections.Counter()
for k, c in enumerate(segs):
    print(k, dict(c))
    tot.update(c)
print("total", dict(tot))
