#!/usr/bin/env python3
"""Static scalar/vector instruction counts per source line for one kernel
(compiles wf_kernels.hip with line tables).  usage: asm_linemap.py [W H] [top]"""
import re
import subprocess
import sys
from collections import Counter

W, H = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("14", "14")
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
csrc = "wildfire_control_python_amd/csrc"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Iinclude", f"-I{csrc}",
                       "--cuda-device-only", "-S", "-gline-tables-only", "-x", "hip", f"{csrc}/wf_kernels.hip",
                       "-o", "/tmp/wfk_g.s"], stderr=subprocess.DEVNULL)
lines = open("/tmp/wfk_g.s").read().split("\n")
fused = sys.argv[4] if len(sys.argv) > 4 else "0"
name = f"_ZN12_GLOBAL__N_112wf_step_waveILi{W}ELi{H}ELb{fused}EEEv6WfArgs:"
start = [i for i, l in enumerate(lines) if l.startswith(name)][0]
end = [i for i, l in enumerate(lines) if i > start and l.startswith(".Lfunc_end")][0]
cur = None
sal, val = Counter(), Counter()
spill = 0
for l in lines[start:end]:
    m = re.match(r"\s+\.loc\s+\d+\s+(\d+)\s+(\d+)", l)
    if m:
        cur = int(m.group(1))
        continue
    if re.match(r"^\s+v_(writelane|readlane)", l):
        spill += 1
    if re.match(r"^\s+s_", l) and not re.match(r"^\s+s_(waitcnt|nop|endpgm)", l):
        sal[cur] += 1
    elif re.match(r"^\s+v_", l):
        val[cur] += 1
src = open(f"{csrc}/wf_kernels.hip").read().split("\n")
print(f"static SALU {sum(sal.values())}  VALU {sum(val.values())}  writelane/readlane {spill}")
for ln, c in sal.most_common(top):
    print(f"{c:4d} v={val[ln]:3d} L{ln}: {src[ln - 1].strip()[:100] if ln else ''}")
