"""n-step calls (step_many, 16 steps per call, actions every step) of larger batches: the rows kernel's step loop (two part
launches on two streams) against wf_step_ticks (one launch, one environment per wave).
usage (GPU box): python tools/nstep_kernels.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import wildfire_control_python_amd as pkg

T = 16
for S in (8, 10, 14, 16):
    for B in (8192, 16384, 65536):
        for acts in (True, False):
            row = []
            for name, kv in (("rows step loop", {}), ("wf_step_ticks", {"WF_NO_MULTI": "1", "WF_NO_SPLIT": "1"})):
                for k in ("WF_NO_MULTI", "WF_NO_SPLIT"):
                    os.environ.pop(k, None)
                os.environ.update(kv)
                env = pkg.BatchedForestFire(B, size=S, seed=1234, auto_reset=True, fuel_range=(10, 40))
                env.reset(seed=5)
                d_act = None
                if acts:
                    d_act = torch.from_numpy(np.random.default_rng(1).integers(0, 4, size=(T, B)).astype(np.int8)).cuda()
                d_r = torch.empty((T, B), dtype=torch.float64, device="cuda")
                d_d = torch.empty((T, B), dtype=torch.uint8, device="cuda")
                mask = (1 << T) - 1
                for _ in range(10):
                    env.step_device(d_act, T, d_r, d_d, update_mask=mask)
                env.sync()
                n = 100
                env.timer_start()
                for _ in range(n):
                    env.step_device(d_act, T, d_r, d_d, update_mask=mask)
                us = env.timer_stop() / n * 1e3
                kname = env._lib.wf_call_kernel_name(env._h, T).decode()
                row.append((name, us, kname))
                env.close()
            print(f"{S:2d}x{S:<2d} B={B:6d} {'actions' if acts else 'no-op  '}: " +
                  "   ".join(f"{nm}: {us:8.2f} us per call = {us / T:6.2f} per step [{kn[:34]}]" for nm, us, kn in row), flush=True)
