#!/usr/bin/env python3
"""Turn rocprofv3 --pmc counter_collection CSVs into the per-launch traffic record bench.py reports as
roofline.traffic (profiles/r01_msda_traffic_batch<N>.json).

    python tools/summarize_pmc.py --kernel msda_forward_win --batch 64 --levels 4 \
        --out profiles/r01_msda_traffic_batch64.json  DIR_WITH_FETCH_PASS  DIR_WITH_WRITE_PASS ...

Every directory is searched for *counter_collection.csv; per counter the mean over the kernel's dispatches is kept.
Counters are collected in separate passes (FETCH_SIZE alone; WRITE_SIZE with the TCC hit/miss sums) as
MI355X_MICROARCH.md prescribes."""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("dirs", nargs="+")
ap.add_argument("--kernel", default="msda_forward_win")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--levels", type=int, default=4)
ap.add_argument("--dist", default=None, help="location distribution of the bench run (M / I)")
ap.add_argument("--skip-first", type=int, default=0, help="dispatches to drop per counter (warm-up)")
ap.add_argument("--out", required=True)
a = ap.parse_args()

vals = defaultdict(list)
name = None
for d in a.dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if a.kernel in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
                name = r["Kernel_Name"]
mean = {k: sum(v[a.skip_first:]) / max(1, len(v[a.skip_first:])) for k, v in vals.items()}
rec = {
    "kernel": name,
    "workload": "bench.py: batch %d, %d levels, M=8 D=16 P=4, fp32" % (a.batch, a.levels),
    "method": "rocprofv3 --pmc FETCH_SIZE (own pass) and --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (own pass) around "
              "bench.py; per-launch means over %s dispatches" % {k: len(v) for k, v in vals.items()},
    "note": "FETCH_SIZE = TCC_EA0_RDREQ x 64 B; MI355X_MICROARCH.md (HBM): the counter reads 1/2 of the bytes of a WIDE "
            "coalesced 16-B/lane stream and is uncalibrated for other widths. This kernel mixes 8-B/4-B per-lane loads "
            "(offsets, logits, reference points) with 16-B LDS-DMA window loads, so both the raw value and the doubled "
            "upper bound are kept; WRITE_SIZE is exact for 16-B/lane stores.",
    "batch": a.batch,
    "levels": a.levels,
}
if a.dist:
    rec["location_dist"] = a.dist
    rec["workload"] += ", sampling offsets: distribution " + a.dist
for k, v in mean.items():
    rec[k + ("_KB" if k.endswith("_SIZE") else "")] = v
if "TCC_EA0_RDREQ_128B_sum" in mean:
    # memory-side requests by size: the calibration MI355X_MICROARCH.md asks for ("calibrate on a known byte count in your
    # own access pattern"): on this kernel every read request is a 128-B line, so FETCH_SIZE (requests x 64 B) is half
    rd = 128.0 * mean["TCC_EA0_RDREQ_128B_sum"] + 64.0 * mean.get("TCC_EA0_RDREQ_64B_sum", 0.0) + \
        32.0 * mean.get("TCC_EA0_RDREQ_32B_sum", 0.0)
    wr64 = mean.get("TCC_EA0_WRREQ_64B_sum", 0.0)
    wr = 64.0 * wr64 + 32.0 * (mean.get("TCC_EA0_WRREQ_sum", wr64) - wr64)
    rec["hbm_read_bytes"] = rd
    rec["hbm_write_bytes"] = wr
    rec["hbm_bytes"] = rd + wr
    rec["method"] = ("rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum "
                     "(own pass) and --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum ... (own pass) around bench.py; "
                     "bytes = sum over request sizes; per-launch means over %s dispatches" % {k: len(v) for k, v in vals.items()})
    rec["note"] = ("L2 memory-side (fabric) requests: Infinity-Cache hits are included, so this is the traffic the L2 sends "
                   "out, an upper bound on DRAM traffic.  FETCH_SIZE of the same launches reads half of hbm_read_bytes "
                   "(gfx950: 128-B requests tallied at 64 B).")
if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
    rec["hbm_bytes_raw"] = (mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024.0
    rec["hbm_bytes_fetch_doubled"] = (2 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024.0
with open(a.out, "w") as f:
    json.dump(rec, f, indent=1)
print(json.dumps(rec, indent=1))
