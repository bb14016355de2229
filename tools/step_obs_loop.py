"""One-tick device-API steps WITH the observation (World.get_state() of the resulting state): written by the step kernel itself (fused) against the
stand-alone observation kernel behind the step.  usage (GPU box): python3 tools/step_obs_loop.py --size 64 --batch 4096"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import wildfire_control_python_amd as pkg

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--size", type=int, default=64)
ap.add_argument("--calls", type=int, default=300)
a = ap.parse_args()
B, S = a.batch, a.size
env = pkg.BatchedForestFire(B, size=S, seed=1234, auto_reset=True, fuel_range=(10, 40))
env.reset(seed=1234)
d_act = torch.from_numpy(np.random.default_rng(1).integers(0, 4, size=(64, 1, B)).astype(np.int8)).cuda()
d_r = torch.empty((1, B), dtype=torch.float64, device="cuda")
d_d = torch.empty((1, B), dtype=torch.uint8, device="cuda")
d_obs = torch.empty((B, S, S, 3), dtype=torch.int8, device="cuda")
res = {}
for mode in ("step only", "fused observation", "step + get_state"):
    for timed in (False, True):
        if timed:
            env.sync()
            env.timer_start()
        for i in range(a.calls):
            env.step_device(d_act[i % 64], 1, d_r, d_d, update_mask=1, d_obs=d_obs if mode == "fused observation" else None)
            if mode == "step + get_state":
                env.get_state_device(d_obs)
        if timed:
            res[mode] = env.timer_stop() / a.calls * 1e3
print(f"{B} x {S}x{S}, us per step: " + ", ".join(f"{k} {v:.2f}" for k, v in res.items()), flush=True)
env.close()
