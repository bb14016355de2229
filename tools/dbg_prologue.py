"""Phase times inside wf_tiled_prologue for the environments that rebuild their reachability board (a WF_DBG_PRO build:
make -C wildfire_control_python_amd/csrc variant NAME=dbg EXTRA=-DWF_DBG_PRO; WILDFIRE_LIB=.../libwildfire_hip_dbg.so).
usage (GPU box): WILDFIRE_LIB=$PWD/wildfire_control_python_amd/libwildfire_hip_dbg.so python3 tools/dbg_prologue.py --size 64 --batch 4096"""
import argparse
import ctypes
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
ap.add_argument("--warm", type=int, default=300)
a = ap.parse_args()
env = pkg.BatchedForestFire(a.batch, size=a.size, seed=1234, auto_reset=True, fuel_range=(10, 40))
env.reset(seed=1234)
B = a.batch
d_act = torch.from_numpy(np.random.default_rng(1).integers(0, 4, size=(64, 1, B)).astype(np.int8)).cuda()
d_r = torch.empty((1, B), dtype=torch.float64, device="cuda")
d_d = torch.empty((1, B), dtype=torch.uint8, device="cuda")
for i in range(a.warm):
    env.step_device(d_act[i % 64], 1, d_r, d_d, update_mask=1)
env.sync()
out = (ctypes.c_ulonglong * 16)()
env._lib.wf_debug_read(out)
env.timer_start()
for i in range(a.calls):
    env.step_device(d_act[i % 64], 1, d_r, d_d, update_mask=1)
us = env.timer_stop() / a.calls * 1e3
env._lib.wf_debug_read(out)
n = max(1, out[4])
names = ["agent logic + barrier", "bitmap build", "relaxation", "convert + write out"]
print(f"{B} x {a.size}x{a.size}: {us:.2f} us per step; flooding workgroups {out[4]} of {out[8]} ({100.0 * out[4] / max(1, out[8]):.2f} %)")
for i in range(4):
    print(f"  {names[i]:24s} mean {out[i] / n * 0.01:.2f} us")
print(f"  whole flooding workgroup: mean {sum(out[:4]) / n * 0.01:.2f} us, max {out[5] * 0.01:.2f} us; relaxation max {out[6] * 0.01:.2f} us")
print(f"  workgroup without a flood: mean {out[9] / max(1, out[8] - out[4]) * 0.01:.2f} us")
env.close()
