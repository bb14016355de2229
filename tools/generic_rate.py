"""Throughput of the generic kernel family (any grid shape; wind with a direction): wf_big_pre -> wf_step_env, and the
thread-per-cell sequence it replaced (WF_CELLS_LEGACY=1), next to the tuned families at the same sizes.
usage (GPU box): python tools/generic_rate.py [only]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wildfire_control_python_amd as pkg

ONLY = len(sys.argv) > 1 and sys.argv[1] == "only"          # "only": the headline size with wind "random" only (profiling)
SWEEP = [(B, S, [0.54, (0, 0)], fam) for S, B in ((16, 65536), (20, 32768), (24, 32768), (32, 16384), (48, 8192), (64, 4096), (100, 2048), (128, 1024))
         for fam in (0, 2, 3)]
CASES = SWEEP if len(sys.argv) > 1 and sys.argv[1] == "sweep" else [(65536, 14, "random", 0), (65536, 14, "random", 1), (65536, 14, [0.54, (0, 0)], 0),
         (65536, 15, [0.54, (0, 0)], 0), (65536, 15, [0.54, (0, 0)], 1), (65536, 13, [0.54, (0, 0)], 0),
         (8192, 50, [0.54, (0, 0)], 0), (8192, 50, [0.54, (0, 0)], 1), (8192, 52, [0.54, (0, 0)], 0),
         (1024, 250, [0.54, (0, 0)], 0), (1024, 250, [0.54, (0, 0)], 1), (1024, 256, [0.54, (0, 0)], 0), (1024, 256, "random", 0)]
for B, S, wind, legacy in CASES[:1 if ONLY else None]:
    if True:
        os.environ.pop("WF_CELLS_LEGACY", None)
        os.environ.pop("WF_FAMILY", None)
        if legacy == 1:
            os.environ["WF_CELLS_LEGACY"] = "1"
        if legacy == 2:
            os.environ["WF_FAMILY"] = "cells"
        if legacy == 3:
            os.environ["WF_FAMILY"] = "tiled"
        env = pkg.BatchedForestFire(B, size=S, seed=1234, auto_reset=True, fuel_range=(10, 40), metadata={"wind": wind},
                                    ignite_ppm=10000 if S >= 250 else 0, containment=(S * S <= 60000) and None)
        env.reset(seed=5)
        act = torch.randint(0, 4, (64, B), dtype=torch.int8, device="cuda")
        torch.cuda.synchronize()
        for i in range(30):
            env.step_device(act[i % 64], 1)
        env.sync()
        n = 200
        env.timer_start()
        for i in range(n):
            env.step_device(act[i % 64], 1)
        ms = env.timer_stop()
        cu = B * S * S * n / (ms * 1e-3)
        print(f"{B:6d} x {S:3d}x{S:<3d} wind {wind!s:18s} {env.step_kernel:28s}: {ms / n * 1e3:8.2f} us per step, {cu:.3e} cell-updates/s, "
              f"{cu * 8 / 8e12:.3f} of the HBM roofline", flush=True)
        env.close()
