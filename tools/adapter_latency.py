"""Latency of the single-environment drop-in (ForestFire.step, what the reference's learners call) and of small batches
through the numpy API.  usage (GPU box): python tools/adapter_latency.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wildfire_control_python_amd as pkg

for size in (10, 14):
    sim = pkg.ForestFire(size=size)
    np.random.seed(1)
    sim.reset()
    n, t0 = 0, time.perf_counter()
    rng = np.random.default_rng(0)
    while n < 2000:
        _, _, done, _ = sim.step(int(rng.integers(0, 4)))
        n += 1
        if done:
            sim.reset()
    dt = (time.perf_counter() - t0) / n
    print(f"ForestFire(size={size}).step(): {dt * 1e6:7.1f} us per step (resets included)", flush=True)
    sim.close()
for B in (1, 64, 4096):
    env = pkg.BatchedForestFire(B, size=14, auto_reset=True)
    env.reset(seed=1)
    a = np.zeros(B, np.int8)
    for _ in range(20):
        env.step(a)
    t0 = time.perf_counter()
    for _ in range(500):
        env.step(a)
    dt = (time.perf_counter() - t0) / 500
    t0 = time.perf_counter()
    for _ in range(500):
        env.step_many(a[None, :])
    dt2 = (time.perf_counter() - t0) / 500
    print(f"BatchedForestFire({B}).step(): {dt * 1e6:7.1f} us (with observation), step_many: {dt2 * 1e6:7.1f} us (reward / done only)", flush=True)
    env.close()
