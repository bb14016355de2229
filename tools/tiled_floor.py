"""Config 4 (1024 x 256x256): step time with and without the CA update (update_mask = 0 reads every cell, reduces, writes nothing)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wildfire_control_python_amd import BatchedForestFire
B, S, K = 1024, 256, 60
for cont in (None, False):
    env = BatchedForestFire(B, size=S, seed=1, auto_reset=True, fuel_range=(10, 40), ignite_ppm=10000, containment=cont)
    env.reset(seed=1)
    rew = torch.empty((1, B), dtype=torch.float64, device="cuda"); done = torch.empty((1, B), dtype=torch.uint8, device="cuda")
    for upd in (1, 0, 1):
        for s in range(10): env.step_device(None, 1, rew, done, update_mask=upd)
        env.timer_start()
        for s in range(K): env.step_device(None, 1, rew, done, update_mask=upd)
        ms = env.timer_stop()
        print(f"containment={cont} update={upd}: {ms * 1e3 / K:.1f} us per step", flush=True)
    env.close()
