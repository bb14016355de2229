#!/usr/bin/env python3
"""Basic-block instruction census of one kernel in wf_kernels.s (after `make -C wildfire_control_python_amd/csrc asm`).
Usage: python tools/asm_blocks.py <mangled-name-substring> [--dump LABEL]"""
import os
import re
import sys

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "wildfire_control_python_amd", "csrc", "wf_kernels.s")


def kernel_body(sub):
    s = open(PATH).read()
    m = re.search(r"^(_Z[^\n:]*" + re.escape(sub) + r"[^\n:]*):", s, re.M)
    if not m:
        sys.exit("no such kernel")
    end = s.index(".Lfunc_end", m.end())
    return m.group(1), s[m.end():end].split("\n")


def blocks(lines):
    out, cur = [], ("entry", [])
    for l in lines:
        if re.match(r"^\.LBB\d+_\d+:", l):
            out.append(cur)
            cur = (l.split(":")[0], [])
        elif l.startswith("\t") and not l.strip().startswith((".", ";")):
            cur[1].append(l.strip())
    out.append(cur)
    return out


if __name__ == "__main__":
    name, lines = kernel_body(sys.argv[1])
    bl = blocks(lines)
    if "--dump" in sys.argv:
        lab = sys.argv[sys.argv.index("--dump") + 1]
        for n, b in bl:
            if n == lab:
                print("\n".join(b))
        sys.exit(0)
    print(name, sum(len(b) for _, b in bl), "instructions")
    for n, b in bl:
        v = sum(i.startswith("v_") for i in b)
        sc = sum(i.startswith("s_") for i in b)
        mem = sum(i.startswith(("global_", "ds_", "buffer_", "scratch_", "flat_")) for i in b)
        print(f"{n:12s} total {len(b):4d} valu {v:4d} salu {sc:4d} mem {mem:3d}  last: {b[-1] if b else ''}")
