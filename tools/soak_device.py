"""Long differential run of the DEVICE API (asynchronous step_device on the context's own streams - the path bench.py times) against
the CPU oracle: every step rewards and done flags, every `check` steps the whole state and the fused observation.
usage (GPU box): python tools/soak_device.py [steps] [batch] [size] [seed] [check]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
from wildfire_control_python_amd import BatchedForestFire
from oracle import oracle as O

T = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
S = int(sys.argv[3]) if len(sys.argv) > 3 else 14
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 5
check = int(sys.argv[5]) if len(sys.argv) > 5 else 25
VIS = np.uint16(~np.uint16(128 | 256))
env = BatchedForestFire(B, size=S, seed=seed, auto_reset=True, containment=True, fuel_range=(3, 25), ignite_ppm=3000)
env.reset(seed=seed)
op = O.default_params(containment=1, fuel_lo=3, fuel_hi=25, ignite_ppm=3000)
planes, envs = O.new_state(B, S, S)
O.reset(planes, envs, seed=seed, params=op)
rng = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
d_r = torch.empty((1, B), dtype=torch.float64, device=dev)
d_d = torch.empty((1, B), dtype=torch.uint8, device=dev)
d_obs = torch.empty((B, S, S, 3), dtype=torch.int8, device=dev)
t0 = time.time()
paid = resets = 0
for t in range(T):
    a = rng.integers(0, 6, size=B).astype(np.int8)
    d_a = torch.from_numpy(a).to(dev)
    torch.cuda.synchronize()
    want_obs = t % check == check - 1
    env.step_device(d_a, 1, d_r, d_d, update_mask=1, d_obs=d_obs if want_obs else None)
    stopped = (envs["flags"] & O.F_RUNNING) == 0
    ro, do = O.step(planes, envs, a, op)
    if stopped.any():
        O.reset(planes, envs, seed=seed, mask=stopped, params=op)
        ro[stopped], do[stopped] = 0.0, 0
        resets += int(stopped.sum())
    paid += int((ro == 1000.0).sum())
    env.sync()
    assert np.array_equal(d_r.cpu().numpy()[0], ro), f"reward differs at step {t}"
    assert np.array_equal(d_d.cpu().numpy()[0], do), f"done differs at step {t}"
    if want_obs:
        gp, ge = env.download()
        assert np.array_equal(gp, planes), f"planes differ at step {t}"
        ge = ge.copy(); ge["flags"] &= VIS
        assert np.array_equal(ge, envs), f"records differ at step {t}"
        assert np.array_equal(d_obs.cpu().numpy(), O.get_state(planes)), f"observation differs at step {t}"
print(f"soak_device ok: {env.step_kernel} x{env.step_launches}; {B} envs x {T} steps of {S}x{S} bit-exact vs oracle "
      f"({resets} resets, {paid} containment payments, {time.time() - t0:.0f} s)")
env.close()
