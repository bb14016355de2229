"""Where the single-environment drop-in's step goes: cProfile over ForestFire.step() (one environment, host round trip per step).
usage (GPU box): python tools/adapter_profile.py"""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
from wildfire_control_python_amd import ForestFire
sim = ForestFire(size=14)
np.random.seed(1)
sim.reset()
rng = np.random.default_rng(0)
acts = [int(x) for x in rng.integers(0, 4, size=6000)]
def run(n0, n):
    for i in range(n0, n0 + n):
        if sim.step(acts[i])[2]:
            sim.reset()
run(0, 500)
t0 = time.perf_counter(); run(500, 2500); dt = time.perf_counter() - t0
print(f"un-profiled: {dt / 2500 * 1e6:.1f} us per step (resets included)")
pr = cProfile.Profile(); pr.enable(); run(3000, 2500); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])
sim.close()
