#!/bin/bash
# VGPR / spill summary of every kernel in one .hip file (development tool): tools/kernel_regs.sh pctrans_amd/csrc/x.hip [filter]
f=$1; flt=${2:-.}
/opt/rocm/bin/hipcc -O3 -std=c++20 --offload-arch=gfx950 -munsafe-fp-atomics --cuda-device-only -c "$f" -I"$(dirname "$f")" \
  -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
cur = None
for ln in sys.stdin:
    m = re.search(r"Function Name: (\S+)", ln)
    if m: cur = {"name": m.group(1)}; continue
    for key in ("VGPRs", "SGPRs Spill", "VGPRs Spill", "ScratchSize \[bytes/lane\]", "Occupancy \[waves/SIMD\]", "LDS Size \[bytes/block\]"):
        m = re.search(r"remark: +" + key + r": (\d+)", ln)
        if m and cur is not None: cur[key] = m.group(1)
    if "LDS Size" in ln and cur:
        print("%-110s vgpr %s sgpr-spill %s vgpr-spill %s scratch %s occ %s" % (cur["name"][:110], cur.get("VGPRs"), cur.get("SGPRs Spill"), cur.get("VGPRs Spill"), cur.get("ScratchSize \\[bytes/lane\\]"), cur.get("Occupancy \\[waves/SIMD\\]")))
        cur = None
' | grep -E "$flt"
