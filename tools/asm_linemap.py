#!/usr/bin/env python3
"""Static scalar/vector instruction counts per source line of one kernel (compiles wf_kernels.hip with line
tables).  usage: asm_linemap.py [kernel-substring] [top]
e.g. asm_linemap.py wf_step_rowsILi14ELi14ELi2ELb0 30   (default);  asm_linemap.py wf_step_groupILi10ELi10ELb0"""
import re
import subprocess
import sys
from collections import Counter

pat = sys.argv[1] if len(sys.argv) > 1 else "wf_step_rowsILi14ELi14ELi2ELb0"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
csrc = "wildfire_control_python_amd/csrc"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Iinclude", f"-I{csrc}",
                       "--cuda-device-only", "-S", "-gline-tables-only", "-x", "hip", f"{csrc}/wf_kernels.hip",
                       "-o", "/tmp/wfk_g.s"], stderr=subprocess.DEVNULL)
lines = open("/tmp/wfk_g.s").read().split("\n")
files = {}                      # .file <n> "dir" "name"
for l in lines:
    m = re.match(r'\s+\.file\s+(\d+)\s+(?:"([^"]*)"\s+)?"([^"]*)"', l)
    if m:
        files[int(m.group(1))] = m.group(3)
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l)][0]
end = [i for i, l in enumerate(lines) if i > start and l.startswith(".Lfunc_end")][0]
print(lines[start])
cur = None
sal, val = Counter(), Counter()
spill = 0
for l in lines[start:end]:
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    if re.match(r"^\s+v_(writelane|readlane)", l):
        spill += 1
    if re.match(r"^\s+s_", l) and not re.match(r"^\s+s_(waitcnt|nop|endpgm)", l):
        sal[cur] += 1
    elif re.match(r"^\s+v_", l):
        val[cur] += 1
cache = {}


def text(key):
    if not key:
        return ""
    f, ln = key
    name = files.get(f, "?").split("/")[-1]
    if name not in cache:
        try:
            cache[name] = open(f"{csrc}/{name}").read().split("\n")
        except OSError:
            cache[name] = []
    src = cache[name]
    return f"{name}:{ln}: {src[ln - 1].strip()[:90] if 0 < ln <= len(src) else ''}"


print(f"static SALU {sum(sal.values())}  VALU {sum(val.values())}  writelane/readlane {spill}")
tot = Counter()
for k in set(sal) | set(val):
    tot[k] = sal[k] + val[k]
for k, c in tot.most_common(top):
    print(f"{c:4d} s={sal[k]:3d} v={val[k]:3d}  {text(k)}")
