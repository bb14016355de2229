"""End-to-end rate of the batched learners (environment + epsilon-greedy policy + replay step), 65536 envs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wildfire_control_python_amd import BatchedForestFire
from wildfire_control_python_amd.learner import BatchedLearner
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for algo in ("DQN", "DSARSA"):
    env = BatchedForestFire(B, size=14, seed=1, auto_reset=True)
    L = BatchedLearner(env, algo, store_per_step=1024)
    L.learn(20)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    L.learn(200, reset=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{algo}: {B * 200 / dt:.4g} env-steps/s ({dt / 200 * 1e3:.2f} ms per batched step), mean reward {sum(L.logs['mean_reward'][-50:]) / 50:.2f}", flush=True)
    env.close()
