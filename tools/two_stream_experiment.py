import sys, time, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wildfire_control_python_amd import BatchedForestFire
import bench
S, K = 14, 300
def run(nctx):
    B = 65536 // nctx
    envs = [BatchedForestFire(B, size=S, seed=1, auto_reset=True, fuel_range=(10, 40), env_id0=i * B) for i in range(nctx)]
    acts = torch.from_numpy(bench.make_actions(K + 20, 0, 65536, 1)).cuda()
    rew = [torch.empty((1, B), dtype=torch.float64, device='cuda') for _ in range(nctx)]
    done = [torch.empty((1, B), dtype=torch.uint8, device='cuda') for _ in range(nctx)]
    for e in envs: e.reset(seed=1)
    torch.cuda.synchronize()
    for s in range(20):
        for i, e in enumerate(envs): e.step_device(acts[s, i * B:(i + 1) * B], 1, rew[i], done[i], update_mask=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(20, 20 + K):
        for i, e in enumerate(envs): e.step_device(acts[s, i * B:(i + 1) * B].contiguous(), 1, rew[i], done[i], update_mask=1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"contexts={nctx} us_per_step={dt*1e6:.2f} cells/s={65536*196/dt:.4g}")
    for e in envs: e.close()
for n in (1, 2, 4, 1, 2):
    run(n)
