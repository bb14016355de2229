"""Time the single-step path for different numbers of concurrent part-launches (WF_SPLIT) and group sizes."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wildfire_control_python_amd import BatchedForestFire

B, S, K = int(os.environ.get("B", 65536)), 14, 300
UPD = int(os.environ.get("UPD", 1))
CONT = None if int(os.environ.get("CONT", 1)) else False
NOACT = int(os.environ.get("NOACT", 0))
dev = torch.device("cuda", 0)
acts = torch.from_numpy(np.random.default_rng(0).integers(0, 4, size=(K, B)).astype(np.int8)).to(dev)
rew = torch.empty((1, B), dtype=torch.float64, device=dev)
done = torch.empty((1, B), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
combos = [(1, 2, 0), (1, 4, 0), (1, 6, 0), (1, 8, 0), (2, 4, 0), (2, 6, 0), (2, 8, 0), (2, 10, 0), (2, 12, 0)]
for parts, group, shared in combos:
    os.environ["WF_SPLIT"] = str(parts)
    if shared:
        os.environ["WF_SHARED"] = str(shared)
    else:
        os.environ.pop("WF_SHARED", None)
    if group:
        os.environ["WF_GROUP"] = str(group)
    else:
        os.environ.pop("WF_GROUP", None)
    env = BatchedForestFire(B, size=S, seed=1, auto_reset=True, fuel_range=(10, 40), containment=CONT)
    env.reset(seed=1)
    for s in range(20):
        env.step_device(None if NOACT else acts[s], 1, rew, done, update_mask=UPD)
    env.timer_start()
    for s in range(K):
        env.step_device(None if NOACT else acts[s], 1, rew, done, update_mask=UPD)
    ms = env.timer_stop()
    print(f"parts={parts} group={group or 'auto'} shared={shared} us_per_step={ms * 1e3 / K:.2f} cells/s={B * S * S * K / (ms * 1e-3):.4g}", flush=True)
    env.close()
