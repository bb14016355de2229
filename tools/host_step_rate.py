"""us per synchronous host-buffer step (wf_step: numpy in / out) at 65 536 x 14x14, caller-owned arrays and lent pinned buffers.
usage (GPU box): [WF_SYNC_SPLIT=0] python tools/host_step_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
import gc
from wildfire_control_python_amd import BatchedForestFire
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = BatchedForestFire(B, size=14, seed=1234, auto_reset=True, fuel_range=(10, 40))
env.reset(seed=1234)
rng = np.random.default_rng(0)
a = rng.integers(0, 4, size=(64, B)).astype(np.int8)
for i in range(200):
    env.step_many(a[i % 64])
gc.collect(); gc.freeze()
med = lambda v: sorted(v)[len(v) // 2]
t = []
for i in range(200):
    t0 = time.perf_counter(); env.step_many(a[i % 64]); t.append(time.perf_counter() - t0)
l_act, _, _ = env.host_staging(1)
tl = []
for i in range(200):
    t0 = time.perf_counter(); l_act[0] = a[i % 64]; env.step_many(l_act, borrow=True); tl.append(time.perf_counter() - t0)
print(f"{B} envs: caller arrays median {med(t) * 1e6:.1f} us (mean {sum(t) / len(t) * 1e6:.1f}), lent buffers median {med(tl) * 1e6:.1f} us (mean {sum(tl) / len(tl) * 1e6:.1f}) per step")
env.close()
