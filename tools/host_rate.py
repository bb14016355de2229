"""How fast can the host enqueue steps?  (Python -> ctypes -> wf_step_obs_device -> 2 kernel launches)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wildfire_control_python_amd import BatchedForestFire
import bench
B, S, K = 65536, 14, 2000
env = BatchedForestFire(B, size=S, seed=1, auto_reset=True, fuel_range=(10, 40))
env.reset(seed=1)
acts = torch.from_numpy(bench.make_actions(64, 0, B, 1)).cuda()
rew = torch.empty((1, B), dtype=torch.float64, device="cuda"); done = torch.empty((1, B), dtype=torch.uint8, device="cuda")
rows = [acts[i] for i in range(64)]
for s in range(50): env.step_device(rows[s % 64], 1, rew, done, update_mask=1)
env.sync()
t0 = time.perf_counter()
for s in range(K): env.step_device(rows[s % 64], 1, rew, done, update_mask=1)
t1 = time.perf_counter()
env.sync()
t2 = time.perf_counter()
print(f"enqueue loop {1e6*(t1-t0)/K:.2f} us/step; with final sync {1e6*(t2-t0)/K:.2f} us/step")
# raw pointers prepared once (no per-call tensor -> pointer conversion)
import ctypes
lib, h = env._lib, env._h
pa = [ctypes.c_void_p(r.data_ptr()) for r in rows]; pr = ctypes.c_void_p(rew.data_ptr()); pd = ctypes.c_void_p(done.data_ptr())
one = ctypes.c_uint64(1)
env.sync()
t0 = time.perf_counter()
for s in range(K): lib.wf_step_obs_device(h, pa[s % 64], 1, one, pr, pd, None)
t1 = time.perf_counter()
env.sync()
t2 = time.perf_counter()
print(f"raw ctypes: enqueue loop {1e6*(t1-t0)/K:.2f} us/step; with final sync {1e6*(t2-t0)/K:.2f} us/step")
