#!/usr/bin/env python3
"""Per-wave timeline of the headline step kernel (measurement build: `make -C wildfire_control_python_amd/csrc variant
NAME=timing EXTRA=-DWF_WAVE_TIMING`, run with WILDFIRE_LIB=.../libwildfire_hip_timing.so).

Every wave of wf_step_rows records the 100 MHz wall clock at kernel entry, after phase A, after phase B and at its end.
For a few steady-state steps this prints, relative to the first wave's start of the step: when waves start (launch ramp),
how long they live, how long each phase takes, and when the last one ends (tail) - separately for waves that had to
reset an environment.  usage: python tools/wave_timeline.py [--batch 65536] [--nosplit]"""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--nosplit", action="store_true")
ap.add_argument("--steps", type=int, default=6)
args = ap.parse_args()
if args.nosplit:
    os.environ["WF_NO_SPLIT"] = "1"
import torch
import bench
from wildfire_control_python_amd import BatchedForestFire

B, S = args.batch, 14
env = BatchedForestFire(B, size=S, seed=1234, auto_reset=True, fuel_range=(10, 40))
env.reset(seed=1234)
lib = env._lib
lib.wf_debug_timing.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda", 0)
G = 8
nw = (B + G - 1) // G + 64
buf = torch.zeros((nw, 10), dtype=torch.int64, device=dev)
acts = torch.from_numpy(bench.make_actions(400, 0, B, 1234)).to(dev)
rew = torch.empty((1, B), dtype=torch.float64, device=dev)
done = torch.empty((1, B), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for s in range(300):
    env.step_device(acts[s], 1, rew, done, update_mask=1)
env.sync()
assert lib.wf_debug_timing(env._h, ctypes.c_void_p(buf.data_ptr())) == 0
print(f"kernel {env.step_kernel} x{env.step_launches}, {B} envs; times in us relative to the step's first wave start")
prev_end = None
for s in range(300, 300 + args.steps):
    buf.zero_()
    torch.cuda.synchronize()
    # queue three steps back to back, look at the middle one?  no: the buffer holds one step - run one, but keep the queue warm
    env.step_device(acts[s], 1, rew, done, update_mask=1)
    env.sync()
    r = buf.cpu().numpy()
    r = r[r[:, 0] != 0]
    t0, tA, tB, t1 = (r[:, k].astype(np.float64) / 100.0 for k in range(4))     # 100 MHz -> us
    nres = r[:, 4] & 0xff
    nflood = (r[:, 4] >> 8) & 0xff
    nsweep = (r[:, 4] >> 16) & 0xffff
    xcc = (r[:, 4] >> 32) & 0xf
    z = t0.min()
    life = t1 - t0
    q = lambda a: " ".join(f"{np.percentile(a, p):6.2f}" for p in (0, 10, 50, 90, 99, 100))
    print(f"step {s}: {len(r)} waves, {int((nres > 0).sum())} with a reset; kernel span {t1.max() - z:6.2f}")
    print(f"   start  (min p10 p50 p90 p99 max): {q(t0 - z)}")
    print(f"   end                              : {q(t1 - z)}")
    print(f"   life                             : {q(life)}")
    print(f"   phase A                          : {q(tA - t0)}")
    print(f"   phase B                          : {q(tB - tA)}")
    print(f"   phase C                          : {q(t1 - tB)}")
    if (nres > 0).any():
        print(f"   life, waves with a reset         : {q(life[nres > 0])}")
        print(f"   life, waves without              : {q(life[nres == 0])}")
    print(f"   waves by floods 0/1/2/3+: " + " ".join(str(int((nflood == k).sum())) for k in (0, 1, 2)) + f" {int((nflood >= 3).sum())}"
          f"; sweeps per flood mean {nsweep.sum() / max(1, nflood.sum()):.1f}; median life by floods: " +
          " ".join(f"{k}:{np.median(life[nflood == k]):.1f}" for k in (0, 1, 2, 3, 4) if (nflood == k).any()))
    print(f"   corr(life, floods) {np.corrcoef(life, nflood)[0, 1]:.2f}  corr(life, sweeps) {np.corrcoef(life, nsweep)[0, 1]:.2f}  corr(life, start) {np.corrcoef(life, t0)[0, 1]:.2f}  corr(phaseA, start) {np.corrcoef(tA - t0, t0)[0, 1]:.2f}")
    hw = r[:, 5]
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    ids, inv = np.unique(cuid, return_inverse=True)
    per_cu_n = np.bincount(inv)
    per_cu_life = np.bincount(inv, weights=life) / per_cu_n
    print(f"   distinct CUs {len(ids)}; waves per CU min/median/max {per_cu_n.min()}/{int(np.median(per_cu_n))}/{per_cu_n.max()}; "
          f"mean life per CU min/p50/max {per_cu_life.min():.1f}/{np.median(per_cu_life):.1f}/{per_cu_life.max():.1f}; "
          f"share of life variance between CUs {np.var(per_cu_life[inv]) / np.var(life):.2f}")
    sid = cuid * 4 + simd
    ids2, inv2 = np.unique(sid, return_inverse=True)
    n2 = np.bincount(inv2)
    l2 = np.bincount(inv2, weights=life) / n2
    print(f"   distinct SIMDs {len(ids2)}; waves per SIMD min/median/max {n2.min()}/{int(np.median(n2))}/{n2.max()}; share of life variance between SIMDs {np.var(l2[inv2]) / np.var(life):.2f}; "
          f"corr(mean life of SIMD, waves on SIMD) {np.corrcoef(l2, n2)[0, 1]:.2f}")
    print("   waves per SIMD histogram: " + " ".join(f"{k}:{int((n2 == k).sum())}" for k in range(1, n2.max() + 1)))
    # progress of the 8 waves of a few SIMDs: start, end of A, ends of the 4 iterations (stamp before the store), end
    its = r[:, 6:10].astype(np.float64) / 100.0
    for pick in (0, 137, 700):
        sel = np.nonzero(inv2 == pick)[0]
        sel = sel[np.argsort(t0[sel])]
        print(f"   SIMD #{pick} (waves in start order; start | A end | iteration ends | end):")
        for w in sel:
            print("      " + f"{t0[w] - z:5.1f} | {tA[w] - z:5.1f} | " + " ".join(f"{v - z:5.1f}" for v in its[w]) + f" | {t1[w] - z:5.1f}   floods {nflood[w]} resets {nres[w]} hw_wave {hw[w] & 15}")
    late = (t0 - z) > 5.0
    print(f"   waves starting later than 5 us: {int(late.sum())} ({100.0 * late.mean():.1f} %); per XCC: " +
          " ".join(f"{x}:{int((late & (xcc == x)).sum())}" for x in sorted(set(xcc.tolist()))))
    hist, edges = np.histogram(t0 - z, bins=[0, 0.5, 1, 2, 3, 4, 6, 8, 10, 12, 14, 16, 20, 30])
    print("   start histogram: " + " ".join(f"<{e:g}:{h}" for h, e in zip(hist, edges[1:])))
    print("   per XCC: waves / median start / median end: " +
          "  ".join(f"{x}:{int((xcc == x).sum())}/{np.median((t0 - z)[xcc == x]):.1f}/{np.median((t1 - z)[xcc == x]):.1f}" for x in sorted(set(xcc.tolist()))))
env.close()
