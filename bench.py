#!/usr/bin/env python3
"""bench.py - headline benchmark of the batched wildfire CA step on MI355X.

Workload (BASELINE.json configs[2], the one `metric` is quoted on): 65 536 batched
14x14 grids per GPU, random fuel U{10..40}, agent dig-actions (uniform N/S/E/W)
injected every step, containment test on, finished environments auto-reset on
the device so every environment does real work in every timed step.
A "step" = ONE wf_step_device call = one ForestFire.step() for every env = one CA
tick.  State and actions are resident in HBM before the timed region starts.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher: this process spawns N fresh rank processes (one per GPU) BEFORE it makes
any GPU call and rank 0 prints the one JSON line.  Under torch.distributed.run (RANK / WORLD_SIZE set)
it is one of the ranks; WORLD_SIZE must equal --gpus.  More ranks than visible GPUs is refused unless
--share-gpu is given (rehearsal on one card: gloo carries the barrier, the line says "shared_card").

Multi-GPU: environments are independent -> the global batch is sharded contiguously, one process per
GPU, NO collective on the data path (weak scaling: --batch envs per GPU).  torch.distributed (RCCL)
carries only the timing barrier and the max-over-ranks reduction.

The timed region has a minimum duration (--min-ms, default 50): when K steps take less, the region is
K * reps steps (same action rows again) and `steps` reports what was actually timed.

Extra legs inside the same JSON line (rank 0 of a 1-GPU run, skipped by --no-extras): the other
BASELINE configs and a true-HBM size, each with its own roofline: `configs: {c2, c4, c5_shard, hbm_1m}`.
"""
import argparse
import ctypes
import json
import math
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_CELL_UPDATE = 8      # 4 int8 planes read + 4 written


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=65536, help="environments per GPU")
    ap.add_argument("--size", type=int, default=14)
    ap.add_argument("--min-ms", type=float, default=50.0, help="minimum duration of the timed region")
    ap.add_argument("--fused", type=int, default=16, help="steps per launch of the extra fused measurement (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="headline loop only: no fused / observation / host-API / single-launch / config legs (profiling passes)")
    ap.add_argument("--legs", default="c2,c2_n64,c4,c5_shard,hbm_1m,c3_wind,mid_64,mid_128", help="extra config legs to run (comma separated)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--ignite-ppm", type=int, default=0, help="extra pre-ignited cells per million at reset (config 4)")
    ap.add_argument("--noop", action="store_true", help="no agent actions (configs 2 and 4)")
    ap.add_argument("--no-containment", action="store_true", help="experiment: containment test off")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal: allow more ranks than visible GPUs (ranks share cards, gloo barrier)")
    return ap.parse_args()


def config_label(S, B, noop, ppm):
    """Which BASELINE.json config (if any) these inputs are."""
    if (S, B) == (14, 65536) and not noop:
        return "BASELINE configs[2]"
    if (S, B) == (10, 4096) and noop:
        return "BASELINE configs[1]"
    if (S, B) == (256, 1024) and noop and ppm:
        return "BASELINE configs[3]"
    if (S, B) == (14, 125000) and not noop:
        return "BASELINE configs[4], one of the 8 shards"
    return "custom size"


def make_actions(steps, lo, hi, seed):
    """(steps, hi-lo) int8 uniform N/S/E/W; column i belongs to GLOBAL env lo+i, so the
    action stream of an environment does not depend on how the batch is sharded."""
    out = np.empty((steps, hi - lo), dtype=np.int8)
    blk = 8192
    for g0 in range((lo // blk) * blk, hi, blk):
        r = np.random.default_rng([seed, g0 // blk]).integers(0, 4, size=(steps, blk), dtype=np.int8)
        a, b = max(g0, lo), min(g0 + blk, hi)
        out[:, a - lo:b - lo] = r[:, a - g0:b - g0]
    return out


# ------------------------------------------------------------------ CPU baseline (oracle; before any GPU call)
def _cpu_worker(args):
    size, seed, seconds, tid, B = args
    from oracle import oracle as O
    p = O.default_params(fuel_lo=10, fuel_hi=40)
    planes, envs = O.new_state(B, size, size)
    O.reset(planes, envs, seed=seed, env_id0=tid * B, params=p)
    rng = np.random.default_rng([seed, tid])
    steps, t0 = 0, time.perf_counter()
    while True:
        a = rng.integers(0, 4, size=B).astype(np.int8)
        _, done = O.step(planes, envs, a, p)
        if done.any():
            O.reset(planes, envs, seed=seed, mask=done, env_id0=tid * B, params=p)
        steps += 1
        el = time.perf_counter() - t0
        if el >= seconds:
            return steps, el


def reference_python_numbers(size):
    """The reference's own Python path, timed in the build container by tools/time_reference.py
    (the reference cannot travel to the GPU box; only this JSON does)."""
    try:
        d = json.load(open(os.path.join(REPO, "profiles", "ref_cpu_baseline.json")))
        runs = [r for r in d["runs"] if r["size"] == size] or d["runs"]
        one = min(runs, key=lambda r: r["processes"])
        many = max(runs, key=lambda r: r["processes"])
        return {"cell_updates_per_s_1_process": one["step"]["cell_updates_per_s"],
                "cell_updates_per_s_all_cores": many["step"]["cell_updates_per_s"], "processes": many["processes"],
                "update_only_cell_updates_per_s_1_process": one["update"]["cell_updates_per_s"],
                "grid": one["size"], "cpu_model": d["cpu_model"],
                "measured_by": "tools/time_reference.py in the build container (misc.time_simulation_run shape, misc.py:79-94)"}
    except Exception:
        return None


def cpu_baseline(size, seed, seconds):
    """The oracle (plain-C restatement of the reference step) on a bounded sample of the same workload:
    2048 envs per worker process, same fuel law / action law / auto-reset, one worker per core of this
    process's CPU share (capped at 16).  Must run BEFORE this process touches the GPU (it forks)."""
    import multiprocessing as mp
    B = 2048
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(16, cores))
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(size, seed, seconds, t, B) for t in range(cores)])
    value = sum(B * size * size * st / el for st, el in res)
    ref = reference_python_numbers(size)
    ref_txt = (f"reference Python path: {ref['cell_updates_per_s_1_process']:.3g} cell-updates/s on 1 core, "
               f"{ref['cell_updates_per_s_all_cores']:.3g} on {ref['processes']} ({ref['cpu_model']}; profiles/ref_cpu_baseline.json)"
               if ref else "reference Python path: profiles/ref_cpu_baseline.json missing")
    out = {"value": value, "unit": "cell-updates/s", "cores": cores, "kind": "port",
           "sample": f"{cores} processes x {B} envs x {size}x{size}, {sum(st for st, _ in res)} steps in {seconds:.0f}s each, "
                     f"oracle/wf_oracle.c ({value / cores:.3g} cell-updates/s per core); {ref_txt}"}
    if ref:
        out["reference_python"] = ref
    return out


def kernel_source_sha256():
    """Hash of the kernel and ABI sources the loaded library was built from (the .so itself is not reproducible byte for
    byte; the sources are).  profiles/traffic.json carries the hash of the sources its counters were collected with."""
    import glob
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(REPO, "wildfire_control_python_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h")) + glob.glob(os.path.join(src, "*.cpp"))
                    + [os.path.join(REPO, "include", "wildfire.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def measured_traffic(leg, algorithmic_bytes):
    """(bytes, note): L2 <-> fabric bytes per launch of the leg's dominant kernel from the committed rocprofv3 PMC passes
    (profiles/traffic.json, written by tools/collect_profiles.py; a process cannot collect PMC counters on itself).
    FETCH_SIZE x 2 (gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE.  None when the passes were collected with other
    kernel sources than the ones in this tree, or for another workload size."""
    try:
        t = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))
        e = t["legs"][leg]
        if t.get("kernel_source_sha256") != kernel_source_sha256():
            return None, "profiles/traffic.json was collected with other kernel sources: stale, not reported"
        if e["algorithmic_bytes_per_launch"] != algorithmic_bytes:
            return None, "profiles/traffic.json holds another workload size for this leg"
        return e["traffic_bytes_per_launch"], (f"{e['pmc_pass']}; L2<->fabric bytes (FETCH_SIZE x2 + WRITE_SIZE), Infinity-Cache hits included"
                                               + ("" if e["state_bytes"] > 256 << 20 else "; the state fits the 256 MB Infinity Cache: not HBM traffic"))
    except Exception as ex:
        return None, f"profiles/traffic.json: {ex!r}"


# ------------------------------------------------------------------ --gpus N without a launcher
def spawn_ranks(args):
    """Parent of an N-rank run: never loads torch or the HIP runtime (the GPUs are counted from the KFD topology in
    sysfs), starts one fresh process per rank, relays rank 0's JSON line."""
    import socket
    from wildfire_control_python_amd.launcher import visible_gpu_count
    ndev = visible_gpu_count()
    if ndev is None:
        ndev = args.gpus                       # no topology to read: the ranks themselves check (and refuse) below
    if args.gpus > ndev and not args.share_gpu:
        sys.exit(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) visible; "
                 "one process per GPU is the contract (pass --share-gpu to rehearse on fewer cards)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    sys.exit(rc)


# ------------------------------------------------------------------ one measured loop
class Stepper:
    """Pre-built ctypes arguments for wf_step_obs_device: the per-step host cost is one foreign call."""

    def __init__(self, env, d_act, d_rew, d_done, n_steps=1, mask=1, d_obs=None):
        self.fn, self.h = env._lib.wf_step_obs_device, env._h
        self.rows = d_act.shape[0] // n_steps if d_act is not None else 1
        stride = d_act.shape[-1] * n_steps if d_act is not None else 0
        base = d_act.data_ptr() if d_act is not None else 0
        self.acts = [ctypes.c_void_p(base + i * stride) if d_act is not None else None for i in range(self.rows)]
        self.n, self.mask = n_steps, ctypes.c_uint64(mask)
        self.rew, self.done = ctypes.c_void_p(d_rew.data_ptr()), ctypes.c_void_p(d_done.data_ptr())
        self.obs = ctypes.c_void_p(d_obs.data_ptr()) if d_obs is not None else None

    def run(self, first, count):
        fn, h, acts, n, mask, rew, done, obs, rows = self.fn, self.h, self.acts, self.n, self.mask, self.rew, self.done, self.obs, self.rows
        for s in range(first, first + count):
            rc = fn(h, acts[s % rows], n, mask, rew, done, obs)
            if rc:
                raise RuntimeError(f"wf_step_obs_device failed: {rc}")


def run_leg(name, S, B, *, noop, ppm=0, env_id0=0, steps, warmup, ticks_per_call=1, seed=1234, dev=None, wind=None):
    """One extra config on this GPU: HIP-event time per call, cell-updates/s, its own roofline entry."""
    import torch
    from wildfire_control_python_amd import BatchedForestFire
    env = BatchedForestFire(B, size=S, device=dev.index, seed=seed, env_id0=env_id0, auto_reset=True,
                            fuel_range=(10, 40), ignite_ppm=ppm, metadata={"wind": wind} if wind else None)
    env.reset(seed=seed)
    n = ticks_per_call
    rows = 1 if noop else 64
    d_act = None
    if not noop:
        d_act = torch.from_numpy(make_actions(rows * n, env_id0, env_id0 + B, seed)).to(dev)
    d_rew = torch.empty((n, B), dtype=torch.float64, device=dev)
    d_done = torch.empty((n, B), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    st = Stepper(env, d_act, d_rew, d_done, n_steps=n, mask=(1 << n) - 1)
    st.run(0, warmup)
    env.sync()
    env.timer_start()
    st.run(warmup, steps)
    ms = env.timer_stop()
    cells = B * S * S
    us = ms / steps * 1e3
    achieved = BYTES_PER_CELL_UPDATE * cells * n / (us * 1e-6) / 1e9
    traffic, traffic_note = measured_traffic(name, BYTES_PER_CELL_UPDATE * cells * n) if name != "c2_n64" else (None, "no counter pass for this informational leg")
    get_state = None
    if name == "c4":
        # World.get_state() of the whole batch as its own call (what reset() returns; the step kernels write it themselves)
        d_obs = torch.empty((B, S, S, 3), dtype=torch.int8, device=dev)
        for _ in range(3):
            env.get_state_device(d_obs)
        env.sync()
        env.timer_start()
        for _ in range(20):
            env.get_state_device(d_obs)
        gus = env.timer_stop() / 20 * 1e3
        gbs = 7 * cells / (gus * 1e-6) / 1e9
        get_state = {"us_per_call": gus, "bytes_per_cell": 7, "achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                     "kernel": "wf_get_state_cols (LDS-tiled transpose)"}
        del d_obs
    out = {"workload": f"{B} x {S}x{S}, fuel U{{10..40}}, " + ("no-op actions" if noop else "uniform N/S/E/W actions") +
                       (f", {ppm} ppm pre-ignited" if ppm else "") + (f", wind {wind!r} (environment.py:187-193)" if wind else "") + f", containment {'on' if env.containment else 'off'}, auto-reset"
                       + (f", env ids from {env_id0}" if env_id0 else "") + f" ({config_label(S, B, noop, ppm)})",
           "calls": steps, "warmup": warmup, "ticks_per_call": n, "us_per_call": us,
           "cell_updates_per_s": cells * n * steps / (ms * 1e-3),
           "state_bytes": cells * 4,
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "kernel": env._lib.wf_call_kernel_name(env._h, n).decode(),
                        "launches_per_call": int(env._lib.wf_call_launches(env._h, n)),
                        "algorithmic_bytes_per_call": BYTES_PER_CELL_UPDATE * cells * n,
                        "traffic": traffic, "traffic_note": traffic_note,
                        "note": "8 B per cell-update x ticks per call; fused ticks keep the grid in registers, so HBM moves 8 B per cell per CALL"
                                if n > 1 else "8 B per cell-update"}}
    if get_state:
        out["get_state"] = get_state
    if name.startswith("mid_"):
        # what a learner that reads World.get_state() after every step pays on the tiled family: the step kernel writes
        # the observation itself (11 B per cell-update: + 3 B of observation)
        d_obs = torch.empty((B, S, S, 3), dtype=torch.int8, device=dev)
        so = Stepper(env, d_act, d_rew, d_done, n_steps=1, mask=1, d_obs=d_obs)
        so.run(0, 10)
        env.sync()
        env.timer_start()
        so.run(10, 40)
        ous = env.timer_stop() / 40 * 1e3
        out["step_plus_observation"] = {"fused_us_per_step": ous, "achieved": 11 * cells / (ous * 1e-6) / 1e9, "unit": "GB/s",
                                        "frac": 11 * cells / (ous * 1e-6) / 1e9 / HBM_PEAK_GBS, "bytes_per_cell": 11}
        del d_obs
    env.close()
    del d_act, d_rew, d_done
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)                                   # never returns
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.size, args.seed, args.cpu_seconds)      # forks: before anything touches the GPU
    import torch
    ndev = torch.cuda.device_count()
    if ndev < 1 or not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    shared_card = world > ndev                 # rehearsal: more ranks than GPUs (ranks share cards, gloo)
    if shared_card and not args.share_gpu:
        sys.exit(f"bench.py: {world} ranks but only {ndev} GPU(s) visible (pass --share-gpu to rehearse on fewer cards)")
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    rccl_ranks = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL (backend "nccl") needs one GPU per rank; it only carries the barrier + one max-reduce
        dist.init_process_group("gloo" if shared_card else "nccl", rank=rank, world_size=world)
        if not shared_card:
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)                             # one real RCCL collective: every rank is there
            rccl_ranks = int(probe.item())
            assert rccl_ranks == dist.get_world_size()
    n_gpus = world

    from wildfire_control_python_amd import make_shard
    S, B, K, Wm = args.size, args.batch, args.steps, args.warmup
    cells = B * S * S
    env = make_shard(B * world, S, rank=rank, world_size=world, device=local, seed=args.seed,
                     auto_reset=True, fuel_range=(10, 40), containment=(False if args.no_containment else None), ignite_ppm=args.ignite_ppm)
    lo = rank * B
    env.reset(seed=args.seed)
    rows = min(Wm + K, 1024)                   # action rows resident in HBM; longer runs walk over them again
    acts = make_actions(rows, lo, lo + B, args.seed)
    if args.noop:
        acts[:] = 5
    d_act = torch.from_numpy(acts).to(dev)
    d_rew = torch.empty((1, B), dtype=torch.float64, device=dev)
    d_done = torch.empty((1, B), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    stepper = Stepper(env, d_act, d_rew, d_done)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(*vals):
        t = torch.tensor(vals, dtype=torch.float64, device=("cpu" if shared_card or dist is None else dev))
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t.tolist()]

    def timed_region(first, count):
        barrier()
        t0 = time.perf_counter()
        env.timer_start()
        stepper.run(first, count)
        ev = env.timer_stop()
        barrier()
        return time.perf_counter() - t0, ev

    # Everything the setup allocated (torch, numpy, ctypes) moves to the permanent generation: a full collection of the
    # cyclic garbage collector over those objects takes ~40 ms, and one landing inside a host-side timing loop (29 calls of
    # 50 us) turned BENCH_r03's lent-buffer figure into 1 400 us per step (round 4: per-call times in gpurun_out/r4/).
    import gc
    gc.collect()
    gc.freeze()
    stepper.run(0, Wm)
    env.sync()
    wall, ev_ms = timed_region(Wm, K)
    wall_max, ev_max = reduce_max(wall, ev_ms)
    steps_timed = K
    for _ in range(4):
        if wall_max * 1e3 >= args.min_ms:
            break
        # too short a sample (the driver's --steps 20 is 0.5 ms): time K * reps steps as ONE region instead
        # (every rank derives the same reps from the max-reduced time)
        reps = min(int(math.ceil(1.25 * args.min_ms * steps_timed / (wall_max * 1e3) / K)) + 1, 1 << 16)
        steps_timed = K * reps
        wall, ev_ms = timed_region(Wm + K, steps_timed)
        wall_max, ev_max = reduce_max(wall, ev_ms)
    done_frac = float(d_done.float().mean().item())

    extras = rank == 0 and not args.no_extras
    # the same steps issued as ONE launch per step (the library's two-stream split of large batches
    # switched off): this is the figure rocprofv3's per-kernel average duration can be compared with
    single_ms = None
    if env.step_launches > 1 and not args.no_extras:
        os.environ["WF_NO_SPLIT"] = "1"
        env.reload_env()
        n_s = max(300, min(steps_timed, 2000))
        stepper.run(0, 20)
        env.timer_start()
        stepper.run(20, n_s)
        single_ms = env.timer_stop() / n_s
        del os.environ["WF_NO_SPLIT"]
        env.reload_env()

    # extra: n steps per API call (step_many).  Small batches run them in ONE launch with the grid in registers
    # between ticks; large batches of a rows-layout shape in one launch per stream whose waves step their
    # environments n times (a kernel boundary less per step), other large batches as n single-step launches.
    fused = None
    if args.fused > 1 and not args.no_extras:
        F = args.fused
        reps = 40
        d_act_f = torch.from_numpy(make_actions(F, lo, lo + B, args.seed + 1)).to(dev)
        d_rew_f = torch.empty((F, B), dtype=torch.float64, device=dev)
        d_done_f = torch.empty((F, B), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        sf = Stepper(env, d_act_f, d_rew_f, d_done_f, n_steps=F, mask=(1 << F) - 1)
        sf.run(0, 3)
        env.sync()
        env.timer_start()
        sf.run(0, reps)
        f_ms = env.timer_stop()
        fused = {"steps_per_call": F, "calls": reps,
                 "launches_per_call": int(env._lib.wf_call_launches(env._h, F)),
                 "cell_updates_per_s_per_gpu": cells * F * reps / (f_ms * 1e-3)}

    # extra: step + observation.  Fused (the step kernel writes World.get_state() itself) against
    # the same step followed by the stand-alone observation kernel.
    obs_extra = None
    if not args.no_extras:
        d_obs = torch.empty((B, S, S, 3), dtype=torch.int8, device=dev)
        torch.cuda.synchronize()
        n_o = 300
        so = Stepper(env, d_act, d_rew, d_done, d_obs=d_obs)
        so.run(0, 5)
        env.timer_start()
        so.run(5, n_o)
        fused_ms = env.timer_stop()
        env.timer_start()
        for s in range(n_o):
            stepper.run(s, 1)
            env.get_state_device(d_obs)
        sep_ms = env.timer_stop()
        obs_extra = {"fused_us_per_step": fused_ms / n_o * 1e3, "separate_us_per_step": sep_ms / n_o * 1e3,
                     "fused_cell_updates_per_s": cells * n_o / (fused_ms * 1e-3)}
        del d_obs

    # extra: the synchronous host-buffer entry point (numpy actions in, reward/done out): PCIe and
    # host synchronisation included.  Reported for completeness; it is never `value`.
    host_api = None
    if world == 1 and not args.no_extras:
        a_np = make_actions(30, lo, lo + B, args.seed + 2)
        env.step_many(a_np[0])
        t_own = []
        for s in range(1, 30):
            t0 = time.perf_counter()
            env.step_many(a_np[s])
            t_own.append(time.perf_counter() - t0)
        th = sum(t_own)
        # the same with the pinned buffers the library lends (wf_host_staging): no host copies of actions / reward / done
        l_act, _, _ = env.host_staging(1)
        l_act[0] = a_np[0]
        env.step_many(l_act, borrow=True)
        t_lent = []
        for s in range(1, 30):
            t0 = time.perf_counter()
            l_act[0] = a_np[s]
            env.step_many(l_act, borrow=True)
            t_lent.append(time.perf_counter() - t0)
        tl = sum(t_lent)
        if os.environ.get("WF_BENCH_DEBUG"):
            print("host_api per call us: own", [round(x * 1e6) for x in t_own], "lent", [round(x * 1e6) for x in t_lent], file=sys.stderr)
        med = lambda v: sorted(v)[len(v) // 2]
        host_api = {"cell_updates_per_s": cells * 29 / th, "us_per_step": th / 29 * 1e6,
                    "lent_buffers_us_per_step": tl / 29 * 1e6, "lent_buffers_cell_updates_per_s": cells * 29 / tl,
                    "median_us_per_step": med(t_own) * 1e6, "max_us_per_step": max(t_own) * 1e6,
                    "lent_buffers_median_us_per_step": med(t_lent) * 1e6, "lent_buffers_max_us_per_step": max(t_lent) * 1e6,
                    "note": "wf_step(): numpy in / out, PCIe and host synchronisation included (the kernels read actions from and write "
                            "reward / done to pinned host memory); lent_buffers = the same call on the pinned buffers wf_host_staging lends: "
                            "no host copies.  Never `value`: large batches belong on the device API"}

    # extra: the single-environment drop-in the reference's learners call (ForestFire.step: numpy in / out, the host view
    # refreshed after every step), against the reference's own Python step timed in the build container
    adapter = None
    if world == 1 and not args.no_extras:
        from wildfire_control_python_amd import ForestFire
        sim = ForestFire(size=S, device=local)
        np.random.seed(1)
        sim.reset()
        arng = np.random.default_rng(0)
        for _ in range(50):
            if sim.step(int(arng.integers(0, 4)))[2]:
                sim.reset()
        n_a, ta = 0, time.perf_counter()
        while n_a < 1500:
            _, _, dn, _ = sim.step(int(arng.integers(0, 4)))
            n_a += 1
            if dn:
                sim.reset()
        ta = (time.perf_counter() - ta) / n_a
        sim.close()
        ref = reference_python_numbers(S)
        adapter = {"us_per_step": ta * 1e6, "grid": [S, S], "resets_included": True,
                   "reference_python_us_per_step": (S * S / ref["cell_updates_per_s_1_process"] * 1e6) if ref else None,
                   "note": "ForestFire(size).step(action) -> [state, reward, done, {}] (forest_fire.py:30-49), one environment, host round trip per step"}

    kernel_name, step_kernel, step_launches, containment = env.kernel_name, env.step_kernel, env.step_launches, env.containment
    env.close()
    del d_act, stepper
    torch.cuda.empty_cache()

    # extra legs: the other BASELINE configs and a size that cannot sit in the 256 MB Infinity Cache
    legs = {}
    if extras and world == 1:
        want = [x for x in args.legs.split(",") if x]
        specs = {
            "c2": dict(S=10, B=4096, noop=True, steps=200, warmup=10, ticks_per_call=16),
            # (informational: the same configuration at the API's maximum of 64 ticks per call - the call's fixed cost spread over four times the ticks)
            "c2_n64": dict(S=10, B=4096, noop=True, steps=100, warmup=5, ticks_per_call=64),
            "c4": dict(S=256, B=1024, noop=True, ppm=10000, steps=200, warmup=20),
            "c5_shard": dict(S=14, B=125000, noop=False, env_id0=875000, steps=400, warmup=40),
            "hbm_1m": dict(S=14, B=1048576, noop=False, steps=120, warmup=20),
            # informational (no BASELINE config): the reference's off-by-default wind option at the headline size (the wave
            # family's group-layout kernel with per-direction heat counts), and two mid-size grids (tiled family, one launch per step and
            # part: a workgroup per environment - 64x64: two waves of four 8-row strips, 128x128: eight waves of two)
            "c3_wind": dict(S=14, B=65536, noop=False, steps=200, warmup=20, wind="random"),
            "mid_64": dict(S=64, B=4096, noop=False, steps=200, warmup=20),
            "mid_128": dict(S=128, B=1024, noop=False, steps=200, warmup=20),
        }
        for nm in want:
            if nm in specs:
                try:
                    legs[nm] = run_leg(nm, dev=dev, seed=args.seed, **specs[nm])
                except Exception as e:                      # a leg must never take the headline line down with it
                    legs[nm] = {"error": repr(e)}

    if rank == 0:
        value = cells * world * steps_timed / wall_max
        launch_ms = ev_max / steps_timed
        traffic, traffic_note = measured_traffic("c3" if (S, B) == (14, 65536) else "c5_shard" if (S, B) == (14, 125000) else "custom",
                                                 BYTES_PER_CELL_UPDATE * cells)
        achieved = BYTES_PER_CELL_UPDATE * cells / (launch_ms * 1e-3) / 1e9
        out = {
            "metric": "cell-updates/sec",
            "value": value,
            "unit": "cell-updates/s",
            "n_gpus": n_gpus,
            "steps": steps_timed,
            "steps_requested": K,
            "warmup": Wm,
            "ms_per_step": wall_max / steps_timed * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": f"{B} batched {S}x{S} grids per GPU, random fuel U{{10..40}}, "
                                   + ("no-op actions, " if args.noop else "uniform N/S/E/W dig-action injected each step, ")
                                   + ("containment test on, " if containment else "containment test off, ") +
                                   f"device auto-reset of finished envs ({config_label(S, B, args.noop, args.ignite_ppm)})",
                       "global_batch": B * world, "grid": [S, S], "ticks_per_launch": 1,
                       "parallelism": f"env-shard x{world}, no collective",
                       "kernel": kernel_name},
            "env_steps_per_s": value / (S * S),
            "rccl_ranks": rccl_ranks,
            "shared_card": bool(shared_card),
            # one step = `launches_per_step` concurrent launches of `kernel` on separate streams, each over an
            # equal share of the batch; achieved = algorithmic bytes of the whole step / HIP-event time per step
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": step_kernel, "launches_per_step": step_launches,
                         "launch_us": launch_ms * 1e3,
                         "algorithmic_bytes_per_launch": BYTES_PER_CELL_UPDATE * cells,
                         "note": ("launch_us/bytes are per STEP (all concurrent launches of it); "
                                  "single_launch = the same step as one launch, comparable with rocprofv3 kernel stats")},
            "done_frac_last_step": done_frac,
            # `value` is the step alone (reward and done written, no observation).  A learner that wants World.get_state()
            # after every step pays step_plus_observation.fused_us_per_step (11 B per cell-update: + 3 B of observation)
            "value_includes_observation": False,
        }
        if single_ms is not None:
            a1 = BYTES_PER_CELL_UPDATE * cells / (single_ms * 1e-3) / 1e9
            out["roofline"]["single_launch"] = {"launch_us": single_ms * 1e3, "achieved": a1, "frac": a1 / HBM_PEAK_GBS}
        if fused:
            out["steps_per_call"] = fused
        if obs_extra:
            out["step_plus_observation"] = obs_extra
        if host_api:
            out["host_buffer_api"] = host_api
        if adapter:
            out["single_env_adapter"] = adapter
        if legs:
            out["configs"] = legs
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
