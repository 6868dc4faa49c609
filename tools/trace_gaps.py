#!/usr/bin/env python3
"""Idle time of the GPU inside the bench step from a rocprofv3 --kernel-trace CSV: busy fraction over the last steps and where the
gaps are (by the kernel that FOLLOWS the gap).  usage: trace_gaps.py <dir with *_kernel_trace.csv>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# the last 40 % of the trace: steady state
t_lo = ev[0][0] + int(0.6 * (ev[-1][1] - ev[0][0]))
ev = [e for e in ev if e[0] >= t_lo]
span = ev[-1][1] - ev[0][0]
busy = 0
gaps = collections.Counter()
cnt = collections.Counter()
last_end = ev[0][0]
for s, e, n in ev:
    if s > last_end:
        gaps[n[:70]] += s - last_end
        cnt[n[:70]] += 1
    busy += max(0, e - max(s, last_end))
    last_end = max(last_end, e)
print("span %.1f ms, busy %.1f ms (%.1f %%), idle %.1f ms, kernels %d" % (span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, len(ev)))
for n, g in gaps.most_common(18):
    print("  idle before %-70s %7.3f ms in %5d gaps (%.1f us each)" % (n, g / 1e6, cnt[n], g / 1e3 / cnt[n]))
