"""k single-step launch pairs per API call (WF_NO_MULTI: the rows kernel's own step loop off), issued directly and replayed
from a HIP graph (WF_GRAPH=1), against the in-kernel step loop - us per step at 65 536 x 14x14.
usage (GPU box): python tools/graph_steps.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wildfire_control_python_amd as pkg

B, S, K = 65536, 14, 16
dev = torch.device("cuda", 0)
acts = torch.from_numpy(np.random.default_rng(0).integers(0, 4, size=(K, B), dtype=np.int8)).to(dev)
rew = torch.empty((K, B), dtype=torch.float64, device=dev)
done = torch.empty((K, B), dtype=torch.uint8, device=dev)
for name, kv in (("in-kernel step loop (default)", {}), ("16 x 2 launches, issued", {"WF_NO_MULTI": "1"}),
                 ("16 x 2 launches, HIP graph", {"WF_NO_MULTI": "1", "WF_GRAPH": "1"}),
                 ("16 x 1 launch, issued", {"WF_NO_MULTI": "1", "WF_NO_SPLIT": "1"}),
                 ("16 x 1 launch, HIP graph", {"WF_NO_MULTI": "1", "WF_NO_SPLIT": "1", "WF_GRAPH": "1"})):
    for k in ("WF_NO_MULTI", "WF_GRAPH", "WF_NO_SPLIT"):
        os.environ.pop(k, None)
    os.environ.update(kv)
    env = pkg.BatchedForestFire(B, size=S, seed=1234, auto_reset=True, fuel_range=(10, 40))
    env.reset(seed=1234)
    mask = (1 << K) - 1
    for _ in range(5):
        env.step_device(acts, K, rew, done, update_mask=mask)
    env.sync()
    n = 60
    env.timer_start()
    for _ in range(n):
        env.step_device(acts, K, rew, done, update_mask=mask)
    ms = env.timer_stop()
    us = ms / (n * K) * 1e3
    print(f"{name:34s}: {us:6.2f} us per step, {8 * B * S * S / (us * 1e-6) / 8e12:.3f} of the roofline "
          f"({env._lib.wf_call_launches(env._h, K)} launches per call)", flush=True)
    env.close()
