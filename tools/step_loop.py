"""A bare device-side step loop for profiling passes (rocprofv3 --pmc / --kernel-trace): `calls` calls of `n` steps each on
`batch` environments of size x size, actions resident on the device, nothing else running.
usage (GPU box): python3 tools/step_loop.py --batch 65536 --size 14 --n 1 --calls 400 [--warm 400]   (WF_* knobs from the environment)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import wildfire_control_python_amd as pkg

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--size", type=int, default=14)
ap.add_argument("--n", type=int, default=1)
ap.add_argument("--calls", type=int, default=400)
ap.add_argument("--warm", type=int, default=400, help="untimed steps before the timed calls (episodes in steady state)")
ap.add_argument("--noop", action="store_true")
ap.add_argument("--fuel", default="10,40", help="fuel range lo,hi")
ap.add_argument("--no-containment", action="store_true")
a = ap.parse_args()
lo, hi = (int(x) for x in a.fuel.split(","))
env = pkg.BatchedForestFire(a.batch, size=a.size, seed=1234, auto_reset=True, fuel_range=(lo, hi), containment=(False if a.no_containment else None))
env.reset(seed=1234)
T, B = a.n, a.batch
rows = 64
d_act = None if a.noop else torch.from_numpy(np.random.default_rng(1).integers(0, 4, size=(rows, T, B)).astype(np.int8)).cuda()
d_r = torch.empty((T, B), dtype=torch.float64, device="cuda")
d_d = torch.empty((T, B), dtype=torch.uint8, device="cuda")
mask = (1 << T) - 1
for i in range(max(1, a.warm // T)):
    env.step_device(None if d_act is None else d_act[i % rows], T, d_r, d_d, update_mask=mask)
env.sync()
env.timer_start()
for i in range(a.calls):
    env.step_device(None if d_act is None else d_act[i % rows], T, d_r, d_d, update_mask=mask)
us = env.timer_stop() / a.calls * 1e3
print(f"{B} x {a.size}x{a.size}, {T} step(s) per call: {us:.2f} us per call = {us / T:.2f} us per step "
      f"[{env._lib.wf_call_kernel_name(env._h, T).decode()} x{int(env._lib.wf_call_launches(env._h, T))}]", flush=True)
env.close()
