#!/usr/bin/env python3
"""Time the REFERENCE's own Python path in the build container.

The reference cannot travel to the GPU box, so its CPU numbers are measured here, by this committed
script, and only the resulting JSON (profiles/ref_cpu_baseline.json) travels; bench.py quotes it next
to the GPU figures.  Workloads:

  * `step`   : the shape of the reference's only timing harness, misc.time_simulation_run
               (misc.py:79-94): `sim.reset(); sim.step("D"); while sim.W.RUNNING: sim.step(" ")`,
               i.e. full no-op episodes including get_state + get_reward (A*).
  * `update` : ForestFire.update() alone (forest_fire.py:85-106) over the same episodes.

One process per grid size (the reference freezes WIDTH/HEIGHT at import, environment.py:22-23);
`--procs N` runs N independent copies at once (the code is single-threaded: an N-core figure is N
processes).  The reference is imported unchanged and read-only from /root/reference with the two
stand-ins of tests/golden/gen_golden.py (PyPI `colour`; a loader for the reference's own astar.cpp
compiled into oracle/_ref/astar.so).

    python tools/time_reference.py            # sizes 10 and 14, 1 process and all cores
"""
import argparse
import json
import os
import platform
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "profiles", "ref_cpu_baseline.json")


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def worker(size, seconds):
    """One process: time both workloads on the reference at `size`; prints one JSON line."""
    sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
    import gen_golden
    gen_golden._install_stand_ins()
    sys.path.insert(0, REF)
    import numpy as np
    import random
    from Simulation import constants
    constants.METADATA["width"] = size
    constants.METADATA["height"] = size
    from Simulation.forest_fire import ForestFire          # the reference, unchanged
    sim = ForestFire()
    assert sim.W.WIDTH == size and sim.W.HEIGHT == size
    np.random.seed(0)
    random.seed(0)
    cells = size * size

    # misc.time_simulation_run shape (misc.py:85-90)
    steps, episodes, t0 = 0, 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        sim.reset()
        sim.step("D")
        steps += 1
        while sim.W.RUNNING:
            sim.step(" ")
            steps += 1
        episodes += 1
    t_step = time.perf_counter() - t0

    # update() alone over the same kind of episodes (agent idle; the episode ends when the fire is out)
    ticks, t_upd = 0, 0.0
    while t_upd < seconds:
        sim.reset()
        t1 = time.perf_counter()
        while sim.W.RUNNING:
            sim.update()
            ticks += 1
        t_upd += time.perf_counter() - t1
    print(json.dumps({"size": size, "episodes": episodes,
                      "step": {"env_steps": steps, "seconds": t_step, "env_steps_per_s": steps / t_step,
                               "cell_updates_per_s": steps * cells / t_step},
                      "update": {"ticks": ticks, "seconds": t_upd, "ticks_per_s": ticks / t_upd,
                                 "cell_updates_per_s": ticks * cells / t_upd}}), flush=True)


def run(size, procs, seconds):
    cmd = [sys.executable, os.path.abspath(__file__), "--worker", "--size", str(size), "--seconds", str(seconds)]
    ps = [subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True) for _ in range(procs)]
    outs = []
    for p in ps:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise SystemExit(f"worker failed (size {size})")
        outs.append(json.loads(out.strip().splitlines()[-1]))
    agg = {"size": size, "processes": procs}
    for k in ("step", "update"):
        agg[k] = {"cell_updates_per_s": sum(o[k]["cell_updates_per_s"] for o in outs),
                  "per_process_cell_updates_per_s": sum(o[k]["cell_updates_per_s"] for o in outs) / procs}
    agg["step"]["env_steps_per_s"] = sum(o["step"]["env_steps_per_s"] for o in outs)
    agg["update"]["ticks_per_s"] = sum(o["update"]["ticks_per_s"] for o in outs)
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--size", type=int, default=0)
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--procs", type=int, default=0, help="0 = 1 process and one per core")
    args = ap.parse_args()
    if args.worker:
        worker(args.size, args.seconds)
        return
    if not os.path.isdir(REF):
        sys.exit("reference not present: its CPU path can only be timed in the build container")
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle"), "ref"])
    cores = len(os.sched_getaffinity(0))
    res = {"what": "reference Python path (Simulation/forest_fire.py + environment.py), imported unchanged from /root/reference",
           "harness": "misc.time_simulation_run shape (misc.py:79-94) = 'step'; ForestFire.update() alone = 'update'",
           "cpu_model": cpu_model(), "cores_available": cores, "python": platform.python_version(),
           "seconds_per_workload": args.seconds, "runs": []}
    for size in ([args.size] if args.size else [10, 14]):
        for procs in ([args.procs] if args.procs else [1, cores]):
            r = run(size, procs, args.seconds)
            print(json.dumps(r), flush=True)
            res["runs"].append(r)
    with open(OUT, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", os.path.relpath(OUT, REPO))


if __name__ == "__main__":
    main()
