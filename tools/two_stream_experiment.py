"""Would N independent contexts on N streams (each a share of the batch) beat one context?
usage: python tools/two_stream_experiment.py [size] [total_batch] [noop] [ignite_ppm]
(this experiment led to the two-launch split inside the library for the wave family)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from wildfire_control_python_amd import BatchedForestFire
import bench

S = int(sys.argv[1]) if len(sys.argv) > 1 else 14
TOTAL = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
NOOP = len(sys.argv) > 3 and sys.argv[3] == "noop"
PPM = int(sys.argv[4]) if len(sys.argv) > 4 else 0
K = 300 if S <= 16 else 60
os.environ["WF_SPLIT"] = "1"          # the library's own split off: contexts are the unit here


def run(nctx):
    B = TOTAL // nctx
    envs = [BatchedForestFire(B, size=S, seed=1, auto_reset=True, fuel_range=(10, 40), env_id0=i * B, ignite_ppm=PPM)
            for i in range(nctx)]
    a = bench.make_actions(K + 20, 0, TOTAL, 1)
    if NOOP:
        a[:] = 5
    acts = [torch.from_numpy(np.ascontiguousarray(a[:, i * B:(i + 1) * B])).cuda() for i in range(nctx)]
    rew = [torch.empty((1, B), dtype=torch.float64, device='cuda') for _ in range(nctx)]
    done = [torch.empty((1, B), dtype=torch.uint8, device='cuda') for _ in range(nctx)]
    for e in envs:
        e.reset(seed=1)
    torch.cuda.synchronize()
    for s in range(20):
        for i, e in enumerate(envs):
            e.step_device(acts[i][s], 1, rew[i], done[i], update_mask=1)
    for e in envs:
        e.sync()
    t0 = time.perf_counter()
    for s in range(20, 20 + K):
        for i, e in enumerate(envs):
            e.step_device(acts[i][s], 1, rew[i], done[i], update_mask=1)
    for e in envs:
        e.sync()
    dt = (time.perf_counter() - t0) / K
    print(f"size={S} contexts={nctx} us_per_step={dt * 1e6:.2f} cells/s={TOTAL * S * S / dt:.4g}", flush=True)
    for e in envs:
        e.close()


for n in (1, 2, 4, 1, 2):
    run(n)
