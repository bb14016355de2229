#!/usr/bin/env python3
"""Dirty-line write amplification of the in-place tiled step (BASELINE configs[3]: 1 024 x 256x256, ~1 % pre-ignited,
no-op agents).  wf_step_tiled stores a lane's 16-byte quad only when one of its four cells changed, but HBM is
written in whole dirty lines.  This counts, on the host, the quads (16 B) and 128-byte lines that really change per
tick in the same regime the rocprofv3 WRITE_SIZE pass of tools/profile_round.sh samples (ticks 60..120), so that
WRITE_SIZE / changed-quad bytes can be reported.   usage (GPU box): python tools/c4_write_amp.py > profiles/..."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wildfire_control_python_amd import BatchedForestFire

B, S = 1024, 256
env = BatchedForestFire(B, size=S, seed=1234, auto_reset=True, fuel_range=(10, 40), ignite_ppm=10000)
env.reset(seed=1234)
out = []
prev = None
for t in range(121):
    if t in (60, 80, 100, 120):
        planes, _ = env.download()
        w = planes.view(np.uint32).reshape(B, S * S)
        if prev is not None and t - prev[0] == 1:
            pass
        prev = (t, w.copy())
    env.step_device(None, 1, None, None, update_mask=1)
    env.sync()
    if t in (60, 80, 100, 120):
        planes, _ = env.download()
        w2 = planes.view(np.uint32).reshape(B, S * S)
        ch = prev[1] != w2
        quads = ch.reshape(B, -1, 4).any(axis=2)
        lines = ch.reshape(B, -1, 32).any(axis=2)
        out.append({"tick": t, "changed_cells": int(ch.sum()), "changed_quads": int(quads.sum()), "changed_quad_bytes": int(quads.sum()) * 16,
                    "changed_128B_lines": int(lines.sum()), "changed_line_bytes": int(lines.sum()) * 128,
                    "burning_fraction": float(((w2 & 0xff) == 1).mean())})
env.close()
m = lambda k: float(np.mean([o[k] for o in out]))
print(json.dumps({"workload": "1024 x 256x256, fuel U{10..40}, 10000 ppm pre-ignited, no-op agents; ticks 60, 80, 100, 120",
                  "state_bytes": B * S * S * 4, "samples": out,
                  "mean_changed_quad_bytes": m("changed_quad_bytes"), "mean_changed_line_bytes": m("changed_line_bytes")}, indent=1))
