"""wf_step_ticks (n fused steps, one environment per wave): time per call against ticks per call and batch size, i.e. the
fixed cost of a call, the time per tick, and how that scales with the waves a SIMD holds (1 024 SIMDs: batch / 1 024).
usage (GPU box): python tools/ticks_scaling.py [legacy]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "legacy":
    os.environ["WF_FUSED_LEGACY"] = "1"
os.environ["WF_NO_MULTI"] = "1"      # large batches too: the one-environment-per-wave kernel, not the rows step loop
os.environ["WF_NO_SPLIT"] = "1"
import wildfire_control_python_amd as pkg

S = 10
for B in (1024, 2048, 4096, 8192, 16384):
    env = pkg.BatchedForestFire(B, size=S, seed=1234, auto_reset=True, fuel_range=(10, 40))
    row = []
    for T in (1, 16, 32, 64):
        env.reset(seed=5)
        mask = (1 << T) - 1
        for _ in range(10):
            env.step_device(None, T, update_mask=mask)
        env.sync()
        n = 200
        env.timer_start()
        for _ in range(n):
            env.step_device(None, T, update_mask=mask)
        us = env.timer_stop() / n * 1e3
        row.append((T, us))
    kname = env._lib.wf_call_kernel_name(env._h, 16).decode()
    per_tick = (row[3][1] - row[1][1]) / 48
    print(f"B={B:6d} ({B / 1024:4.1f} waves per SIMD) [{kname[:40]}] " + "  ".join(f"T={t}: {u:7.2f} us" for t, u in row) +
          f"   per tick {per_tick * 1e3:6.0f} ns = {per_tick * 2400:5.0f} cycles at 2.4 GHz, fixed {row[1][1] - 16 * per_tick:5.2f} us", flush=True)
    env.close()
