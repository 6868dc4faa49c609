#!/usr/bin/env python3
"""Build-time check of the column kernels' work-queue fetch (development tool; run after a compiler bump):

    python tools/check_queue_atomic.py pctrans_amd/csrc/msda_forward_col.hip [more .hip files]

The queue's returning `global_atomic_add` is issued through inline asm and its result is only waited for ~350 source lines
later (`s_waitcnt vmcnt(0)` in a second asm block, where the memory counter is at zero anyway).  The compiler does not know
that a memory write to that VGPR is pending: a copy, a phi move or a spill it inserted in between would read a stale
register.  This script compiles each file to assembly and, for every kernel, checks that NO instruction between the atomic
and the asm wait that names the same register reads or overwrites that register."""
import re
import subprocess
import sys
import tempfile

HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "-std=c++20", "--offload-arch=gfx950", "-munsafe-fp-atomics", "--cuda-device-only", "-S"]


def check(path):
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run([HIPCC] + FLAGS + [path, "-I", path.rsplit("/", 1)[0], "-o", f.name], check=True,
                       stderr=subprocess.DEVNULL)
        text = open(f.name).read()
    bad = kernels = sites = 0
    for m in re.finditer(r"^(_Z\w+):.*?^\.Lfunc_end", text, re.S | re.M):
        name, body = m.group(1), m.group(0).split("\n")
        if "msda_forward_col" not in name:
            continue
        kernels += 1
        i = 0
        while i < len(body):
            am = re.search(r"global_atomic_add (v\d+),", body[i])
            if not am or "s_nop 4" not in body[i - 1]:          # only the inline-asm fetch (the asm string starts with s_nop 4)
                i += 1
                continue
            reg = am.group(1)
            sites += 1
            j = i + 1
            found = False
            while j < len(body):
                ln = body[j].split(";")[0]
                if "s_waitcnt vmcnt(0)" in ln and "ASMSTART" in body[j - 1]:
                    # the asm block that parks the value: its operand comment names the register on the next lines
                    found = True
                    break
                # (`v_mov_b32 reg, 0` is the `f_new = 0` of the path on which no atomic was issued: textually behind the
                # atomic, never executed after it)
                if re.search(r"\b%s\b" % reg, ln) and not ln.strip().startswith(".") and \
                        not re.fullmatch(r"\s*v_mov_b32_e32 %s, 0\s*" % reg, ln):
                    print("%s: %s touched before the wait: %s" % (name[:60], reg, ln.strip()))
                    bad += 1
                j += 1
            if not found:
                print("%s: no asm wait after the atomic at line %d" % (name[:60], i))
                bad += 1
            i = j
    print("%s: %d kernels, %d atomic sites, %d problems" % (path, kernels, sites, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if sum(check(p) for p in sys.argv[1:]) else 0)
