#!/usr/bin/env python3
"""bench.py - headline benchmark of the batched wildfire CA step on MI355X.

Workload (BASELINE.json configs[2], the one `metric` is quoted on): 65 536 batched
14x14 grids per GPU, random fuel U{10..40}, agent dig-actions (uniform N/S/E/W)
injected every step, containment test on, finished environments auto-reset on
the device so every environment does real work in every timed step.
A "step" = ONE wf_step_device call = one ForestFire.step() for every env = one CA
tick (the library issues it as two concurrent half-batch launches on its own two
streams; `roofline.single_launch` is the same step as one launch, and the rate of
16 steps per call is reported separately as `steps_per_call`).  State and actions
are resident in HBM before the timed region starts.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: environments are independent -> the global batch is sharded
contiguously, one process per GPU, NO collective on the data path (weak scaling:
65 536 envs per GPU).  torch.distributed is used only for the timing barrier and
the max-over-ranks reduction.
"""
import argparse
import json
import os
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
    ap.add_argument("--fused", type=int, default=16, help="steps per launch of the extra fused measurement (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="headline loop only: no fused / observation / host-API / single-launch legs (profiling passes)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--ignite-ppm", type=int, default=0, help="extra pre-ignited cells per million at reset (config 4)")
    ap.add_argument("--noop", action="store_true", help="no agent actions (configs 2 and 4)")
    ap.add_argument("--no-containment", action="store_true", help="experiment: containment test off")
    return ap.parse_args()


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
    return {"value": value, "unit": "cell-updates/s", "cores": cores,
            "kind": "port",
            "sample": f"{cores} processes x {B} envs x {size}x{size}, {sum(st for st, _ in res)} steps in {seconds:.0f}s each, "
                      f"oracle/wf_oracle.c ({value / cores:.3g} cell-updates/s per core; reference Python path: 5.5e5 "
                      f"cell-updates/s on 1 core, BASELINE.md section 2)"}


def measured_traffic(batch, size):
    """HBM-side bytes per launch of the dominant kernel, from the committed rocprofv3 PMC
    passes (profiles/traffic.json; bench.py cannot collect PMC counters on itself)."""
    try:
        t = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))
        if t["algorithmic_bytes_per_launch"] == BYTES_PER_CELL_UPDATE * batch * size * size:
            return t["traffic_bytes_per_launch"]
    except Exception:
        pass
    return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.size, args.seed, args.cpu_seconds)      # forks: before anything touches the GPU
    import torch
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    ndev = torch.cuda.device_count()
    shared_card = world > ndev                 # rehearsal: more ranks than GPUs (ranks share cards, gloo)
    local = local % ndev
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL (backend "nccl") needs one GPU per rank; it only carries the barrier + one max-reduce
        dist.init_process_group("gloo" if shared_card else "nccl", rank=rank, world_size=world)
    n_gpus = world

    from wildfire_control_python_amd import make_shard
    S, B, K, Wm = args.size, args.batch, args.steps, args.warmup
    cells = B * S * S
    env = make_shard(B * world, S, rank=rank, world_size=world, device=local, seed=args.seed,
                     auto_reset=True, fuel_range=(10, 40), containment=(False if args.no_containment else None), ignite_ppm=args.ignite_ppm)
    lo = rank * B
    env.reset(seed=args.seed)
    total = Wm + K
    dev = torch.device("cuda", local)
    acts = make_actions(total, lo, lo + B, args.seed)
    if args.noop:
        acts[:] = 5
    d_act = torch.from_numpy(acts).to(dev)
    d_rew = torch.empty((1, B), dtype=torch.float64, device=dev)
    d_done = torch.empty((1, B), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for s in range(Wm):
        env.step_device(d_act[s], 1, d_rew, d_done, update_mask=1)
    env.sync()

    barrier()
    t0 = time.perf_counter()
    env.timer_start()
    for s in range(Wm, total):
        env.step_device(d_act[s], 1, d_rew, d_done, update_mask=1)
    ev_ms = env.timer_stop()
    barrier()
    wall = time.perf_counter() - t0
    done_frac = float(d_done.float().mean().item())

    # the same steps issued as ONE launch per step (the library's two-stream split of large batches
    # switched off): this is the figure rocprofv3's per-kernel average duration can be compared with
    single_ms = None
    if env.step_launches > 1 and not args.no_extras:
        os.environ["WF_NO_SPLIT"] = "1"
        n_s = min(K, 300)
        env.step_device(d_act[0], 1, d_rew, d_done, update_mask=1)
        env.timer_start()
        for s in range(n_s):
            env.step_device(d_act[Wm + s], 1, d_rew, d_done, update_mask=1)
        single_ms = env.timer_stop() / n_s
        del os.environ["WF_NO_SPLIT"]

    # extra: n steps per API call (step_many).  Small batches run them in ONE launch with the grid in registers
    # between ticks; large batches of a rows-layout shape in one launch per stream whose waves step their
    # environments n times (a kernel boundary less per step), other large batches as n single-step launches.
    fused = None
    if args.fused > 1 and not args.no_extras:
        F = args.fused
        reps = max(1, min(K, 320) // F)
        d_act_f = torch.from_numpy(make_actions(F, lo, lo + B, args.seed + 1)).to(dev)
        d_rew_f = torch.empty((F, B), dtype=torch.float64, device=dev)
        d_done_f = torch.empty((F, B), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        env.step_device(d_act_f, F, d_rew_f, d_done_f, update_mask=(1 << F) - 1)
        env.sync()
        env.timer_start()
        for _ in range(reps):
            env.step_device(d_act_f, F, d_rew_f, d_done_f, update_mask=(1 << F) - 1)
        f_ms = env.timer_stop()
        fused = {"steps_per_call": F, "calls": reps,
                 "launches_per_call": int(env._lib.wf_call_launches(env._h, F)),
                 "cell_updates_per_s_per_gpu": cells * F * reps / (f_ms * 1e-3)}

    # extra: step + observation.  Fused (the step kernel writes World.get_state() itself) against
    # the same step followed by the stand-alone observation kernel.
    obs_extra = None
    if env.step_kernel.startswith(("wf_step_group", "wf_step_rows")) and not args.no_extras:
        d_obs = torch.empty((B, S, S, 3), dtype=torch.int8, device=dev)
        torch.cuda.synchronize()
        n_o = min(K, 300)
        env.step_device(d_act[0], 1, d_rew, d_done, update_mask=1, d_obs=d_obs)
        env.timer_start()
        for s in range(n_o):
            env.step_device(d_act[Wm + s], 1, d_rew, d_done, update_mask=1, d_obs=d_obs)
        fused_ms = env.timer_stop()
        env.timer_start()
        for s in range(n_o):
            env.step_device(d_act[Wm + s], 1, d_rew, d_done, update_mask=1)
            env.get_state_device(d_obs)
        sep_ms = env.timer_stop()
        obs_extra = {"fused_us_per_step": fused_ms / n_o * 1e3, "separate_us_per_step": sep_ms / n_o * 1e3,
                     "fused_cell_updates_per_s": cells * n_o / (fused_ms * 1e-3)}

    # extra: the synchronous host-buffer entry point (numpy actions in, reward/done out): PCIe and
    # host synchronisation included.  Reported for completeness; it is never `value`.
    host_api = None
    if world == 1 and not args.no_extras:
        a_np = make_actions(30, lo, lo + B, args.seed + 2)
        env.step_many(a_np[0])
        th = time.perf_counter()
        for s in range(1, 30):
            env.step_many(a_np[s])
        th = time.perf_counter() - th
        host_api = {"cell_updates_per_s": cells * 29 / th, "us_per_step": th / 29 * 1e6,
                    "note": "wf_step(): H2D actions + launch + D2H reward/done + stream sync per step"}

    t = torch.tensor([wall, ev_ms], dtype=torch.float64, device=("cpu" if shared_card else dev))
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max, ev_max = float(t[0].item()), float(t[1].item())

    if rank == 0:
        value = cells * world * K / wall_max
        launch_ms = ev_max / K
        achieved = BYTES_PER_CELL_UPDATE * cells / (launch_ms * 1e-3) / 1e9
        out = {
            "metric": "cell-updates/sec",
            "value": value,
            "unit": "cell-updates/s",
            "n_gpus": n_gpus,
            "steps": K,
            "warmup": Wm,
            "ms_per_step": wall_max / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": f"{B} batched {S}x{S} grids per GPU, random fuel U{{10..40}}, "
                                   + ("no-op actions, " if args.noop else "uniform N/S/E/W dig-action injected each step, ")
                                   + ("containment test on, " if env.containment else "containment test off, ") +
                                   "device auto-reset of finished envs (BASELINE configs[2])",
                       "global_batch": B * world, "grid": [S, S], "ticks_per_launch": 1,
                       "parallelism": f"env-shard x{world}, no collective",
                       "kernel": env.kernel_name},
            "env_steps_per_s": value / (S * S),
            # one step = `launches_per_step` concurrent launches of `kernel` on separate streams, each over an
            # equal share of the batch; achieved = algorithmic bytes of the whole step / HIP-event time per step
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(B, S),
                         "kernel": env.step_kernel, "launches_per_step": env.step_launches,
                         "launch_us": launch_ms * 1e3,
                         "algorithmic_bytes_per_launch": BYTES_PER_CELL_UPDATE * cells,
                         "note": ("launch_us/bytes are per STEP (all concurrent launches of it); "
                                  "single_launch = the same step as one launch, comparable with rocprofv3 kernel stats")},
            "done_frac_last_step": done_frac,
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
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
