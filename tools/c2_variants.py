"""BASELINE configs[1] (4 096 x 10x10, no-op actions, 16 ticks per call): time per call under kernel variants.
usage (GPU box): python tools/c2_variants.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wildfire_control_python_amd as pkg

B, S, T = 4096, 10, 16
for B in (4096, 2048, 8192):
    env = pkg.BatchedForestFire(B, size=S, seed=1234, auto_reset=True, fuel_range=(10, 40))
    for name, kv in (("default", {}), ("wf_step_ticks, one launch", {"WF_NO_MULTI": "1", "WF_NO_SPLIT": "1"}),
                     ("round-1/2 kernel", {"WF_NO_MULTI": "1", "WF_NO_SPLIT": "1", "WF_FUSED_LEGACY": "1"}),
                     ("resident loop G=2", {"WF_RESIDENT": "1", "WF_MULTI_SMALL": "1", "WF_GROUP": "2"}),
                     ("resident loop G=4", {"WF_RESIDENT": "1", "WF_MULTI_SMALL": "1", "WF_GROUP": "4"}),
                     ("resident loop G=8", {"WF_RESIDENT": "1", "WF_MULTI_SMALL": "1", "WF_GROUP": "8"}),
                     ("loop through L2 G=2", {"WF_MULTI_SMALL": "1", "WF_GROUP": "2"}),
                     ("loop through L2 G=4", {"WF_MULTI_SMALL": "1", "WF_GROUP": "4"}),
                     ("16 single steps", None)):
        for k in ("WF_MULTI_SMALL", "WF_GROUP", "WF_RESIDENT", "WF_NO_MULTI", "WF_NO_SPLIT", "WF_FUSED_LEGACY"):
            os.environ.pop(k, None)
        os.environ.update(kv or {})
        env.reload_env()
        env.reset(seed=5)
        mask = (1 << T) - 1
        def call():
            if kv is None:
                for _ in range(T):
                    env.step_device(None, 1, update_mask=1)
            else:
                env.step_device(None, T, update_mask=mask)
        for _ in range(20):
            call()
        env.sync()
        n = 300
        env.timer_start()
        t0 = time.perf_counter()
        for _ in range(n):
            call()
        ms = env.timer_stop()
        wall = time.perf_counter() - t0
        cu = B * S * S * T * n / (ms * 1e-3)
        kname = env._lib.wf_call_kernel_name(env._h, T).decode() if kv is not None else env.step_kernel
        if kv == {} and B >= 8192:      # (the introspection speaks of calls with actions; this one has none)
            kname = "wf_step_ticks, one launch (call without actions)"
        print(f"B={B:5d} {name:22s} [{kname[:60]}]: {ms / n * 1e3:7.2f} us per {T}-tick call (wall {wall / n * 1e6:7.2f}), {cu:.3e} cell-updates/s, "
              f"{cu * 8 / 8e12:.3f} of the HBM roofline", flush=True)
    env.close()
