"""Long randomized differential run of the HIP step against the CPU oracle (not part of pytest).
usage (GPU box): [WF_GROUP=8] [STEPS_PER_CALL=k] [OPTIONS=wind,rivers,toggle] python tools/soak.py [steps] [batch] [size] [seed]
Every step: rewards and done flags; every 25 steps: full planes, records and the observation."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wildfire_control_python_amd import BatchedForestFire
from oracle import oracle as O

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192 + 13
S = int(sys.argv[3]) if len(sys.argv) > 3 else 14
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 5
VIS = np.uint16(~np.uint16(128 | 256))

OPTS = set(filter(None, os.environ.get("OPTIONS", "").split(",")))      # the reference's off-by-default options
md, okw = {}, {}
if "wind" in OPTS:
    md["wind"] = "random"
    okw.update(wind_mode=2, wind_speeds=[0.0, 0.7, 0.85, 0.0])
if "rivers" in OPTS:
    md["make_rivers"] = True
    okw.update(make_rivers=1)
if "toggle" in OPTS:
    md.update(allow_dig_toggle=True, n_actions=5)
    okw.update(allow_dig_toggle=1)
env = BatchedForestFire(B, size=S, seed=seed, auto_reset=True, containment=True, fuel_range=(3, 25), ignite_ppm=3000, metadata=md or None)
env.reset(seed=seed)
op = O.default_params(containment=1, fuel_lo=3, fuel_hi=25, ignite_ppm=3000, **okw)
planes, envs = O.new_state(B, S, S)
counts = np.zeros(planes.shape, np.uint8) if "wind" in OPTS else None
O.reset(planes, envs, seed=seed, params=op, counts=counts)
rng = np.random.default_rng(seed)
t0 = time.time()
paid = resets = deaths = 0
KPC = int(os.environ.get("STEPS_PER_CALL", "1"))
pending = []                       # (actions, device reward, device done) of a multi-step call being replayed
for t in range(T):
    a = rng.integers(0, 6, size=B).astype(np.int8)
    # a third of the agents walk in circles (rings -> floods, containment), the rest wander
    k = B // 3
    x, y = envs["x"][:k].astype(int), envs["y"][:k].astype(int)
    m = S // 2
    a[:k] = np.where((x >= m) & (y > m), rng.choice([1, 3], k), np.where((x > m) & (y <= m), rng.choice([1, 2], k),
                     np.where((x <= m) & (y < m), rng.choice([0, 2], k), rng.choice([0, 3], k))))
    if KPC > 1:
        if not pending:            # draw the whole call's actions now (the circle policy then lags by < KPC steps)
            acts = [a] + [rng.integers(0, 6, size=B).astype(np.int8) for _ in range(KPC - 1)]
            rr, dd = env.step_many(np.stack(acts))
            pending = [(acts[i], rr[i], dd[i].astype(bool)) for i in range(KPC)]
        a, r, d = pending.pop(0)
        obs = None
    else:
        obs, r, d, _ = env.step(a)
    stopped = (envs["flags"] & O.F_RUNNING) == 0
    ro, do = O.step(planes, envs, a, op, counts=counts)
    if stopped.any():
        O.reset(planes, envs, seed=seed, mask=stopped, params=op, counts=counts)
        ro[stopped], do[stopped] = 0.0, 0
        resets += int(stopped.sum())
    assert np.array_equal(r, ro), f"reward t={t}: {np.flatnonzero(r != ro)[:8]}"
    assert np.array_equal(d, do.astype(bool)), f"done t={t}"
    paid += int((ro == 1000.0).sum())
    deaths += int((ro == -1000.0).sum())
    if (t % 25 == 0 or t == T - 1) and not pending:
        gp, ge = env.download()
        ge = ge.copy()
        ge["flags"] &= VIS
        assert np.array_equal(gp, planes), f"planes t={t}: envs {np.unique(np.nonzero(gp != planes)[0])[:8]}"
        assert np.array_equal(ge, envs), f"records t={t}"
        if counts is not None:
            assert np.array_equal(env.download_counts(), counts), f"wind counts t={t}"
        assert np.array_equal(env.get_state() if obs is None else obs, O.get_state(planes)), f"obs t={t}"
    if t % 250 == 0:
        print(f"t={t} ok  resets={resets} contained={paid} deaths={deaths}  {time.time() - t0:.0f}s", flush=True)
print(f"SOAK OK kernel={env.step_kernel} x{env.step_launches}: {T} steps x {B} envs {S}x{S}, resets={resets} contained={paid} deaths={deaths}")
