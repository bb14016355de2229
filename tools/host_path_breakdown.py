"""Where the time of a host-buffer step at 65 536 x 14x14 goes: the same single step with its three small arrays (actions in, reward / done out)
in device memory or in the pinned host buffers the library lends, a stream synchronisation after every step.
usage (GPU box): python tools/host_path_breakdown.py"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wildfire_control_python_amd as pkg
from wildfire_control_python_amd import _native as N

B, S = 65536, 14
env = pkg.BatchedForestFire(B, size=S, seed=1, auto_reset=True, fuel_range=(10, 40))
env.reset(seed=1)
dev = torch.device("cuda", 0)
d_act = torch.randint(0, 4, (B,), dtype=torch.int8, device=dev)
d_rew = torch.empty(B, dtype=torch.float64, device=dev)
d_done = torch.empty(B, dtype=torch.uint8, device=dev)
h_act, h_rew, h_done = env.host_staging(1)
h_act[0] = np.random.default_rng(0).integers(0, 4, size=B)
pa, pr, pd = h_act.ctypes.data, h_rew.ctypes.data, h_done.ctypes.data
torch.cuda.synchronize()


def run(name, a, r, d, n=300):
    for _ in range(20):
        env.step_device(a, 1, r, d)
    env.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        env.step_device(a, 1, r, d)
        env.sync()
    dt = (time.perf_counter() - t0) / n * 1e6
    print(f"{name:64s}: {dt:6.1f} us per step + sync", flush=True)


run("all three in device memory", d_act, d_rew, d_done)
run("actions read from pinned host memory by the kernel", pa, d_rew, d_done)
run("done written to pinned host memory by the kernel", d_act, d_rew, pd)
run("reward written to pinned host memory by the kernel", d_act, pr, d_done)
run("reward + done to pinned host memory", d_act, pr, pd)
run("all three in pinned host memory (what wf_step does)", pa, pr, pd)
run("no reward / done at all (NULL), actions on the device", d_act, None, None)
# back-to-back without the per-step synchronisation, for the asynchronous cost of the PCIe accesses alone
for name, a, r, d in (("async, device memory", d_act, d_rew, d_done), ("async, pinned host memory", pa, pr, pd)):
    env.sync()
    env.timer_start()
    for _ in range(1000):
        env.step_device(a, 1, r, d)
    print(f"{name:64s}: {env.timer_stop() / 1000 * 1e3:6.1f} us per step (no sync in between)", flush=True)
env.close()
