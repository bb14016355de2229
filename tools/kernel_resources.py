#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table from `make -C wildfire_control_python_amd/csrc asm`
(wf_kernels.resources.txt).  Usage: python tools/kernel_resources.py [filter-substring ...]"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "..", "wildfire_control_python_amd", "csrc", "wf_kernels.resources.txt")


def table(path=PATH):
    txt = open(path).read()
    rows = []
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = b.split("\n")[0].strip()

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        rows.append((name, g("VGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
                     g(r"LDS Size \[bytes/block\]")))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    return [(n,) + r[1:] for n, r in zip(names, rows)]


if __name__ == "__main__":
    pats = sys.argv[1:]
    print(f"{'kernel':72s} {'VGPR':>5s} {'SGPR':>5s} {'scratch':>7s} {'occ':>4s} {'LDS':>6s}")
    for r in table(sys.argv[1] if pats and os.path.exists(pats[0]) else PATH):
        if not pats or any(p in r[0] for p in pats if not os.path.exists(p)):
            print(f"{r[0][:72]:72s} {r[1]:5d} {r[2]:5d} {r[3]:7d} {r[4]:4d} {r[5]:6d}")
