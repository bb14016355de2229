"""World.get_state() of a whole batch on the device (wf_get_state_device): us per call and fraction of the 8 TB/s
roofline on 7 B per cell (4 read + 3 written).  usage (GPU box): python tools/get_state_rate.py [json-out]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wildfire_control_python_amd as pkg

out = {}
for B, W, H in ((1024, 256, 256), (4096, 128, 128), (8192, 50, 50), (65536, 15, 15), (1024, 250, 250), (65536, 14, 14), (16384, 64, 64)):
    env = pkg.BatchedForestFire(B, width=W, height=H, seed=1, fuel_range=(10, 40), ignite_ppm=10000, containment=False)
    env.reset(seed=1)
    d_obs = torch.empty((B, W, H, 3), dtype=torch.int8, device="cuda")
    for _ in range(5):
        env.get_state_device(d_obs)
    env.sync()
    n = 50
    env.timer_start()
    for _ in range(n):
        env.get_state_device(d_obs)
    us = env.timer_stop() / n * 1e3
    gbs = 7 * B * W * H / (us * 1e-6) / 1e9
    out[f"{B}x{W}x{H}"] = {"us_per_call": us, "achieved_GBs": gbs, "frac_of_8TBs": gbs / 8000.0, "bytes_per_cell": 7}
    print(f"get_state {B} x {W}x{H}: {us:8.1f} us  {gbs:7.0f} GB/s  {gbs / 8000:.3f} of 8 TB/s", flush=True)
    env.close()
    del d_obs
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
