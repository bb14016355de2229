#!/usr/bin/env python3
"""Reference-generated reset statistics: what World.reset() of the REAL reference draws.

Runs only in the build container (needs /root/reference).  Imports the reference's Simulation package
unchanged (same stand-ins as gen_golden.py) and samples its two random generators many times:

  * reset_map(env, make_river=True)   environment.py:69-95   -> histograms of the river's start column, start row,
                                                               length, net column drift and per-column water count
  * get_agent_location(W, H)          utility.py:66-78        -> counts of the 36 possible spawn points

The device reset draws from a counter-based stream (numpy's legacy generator is not reproduced on a GPU), so
it can agree with the reference in DISTRIBUTION only; tests/test_reset_law.py compares the build's samples with these
histograms (two-sample chi-square).  Only histograms (data) are stored: tests/golden/ref_reset_stats_<size>.npz.

Rivers that cross the fire cell (the reference then holds a burning cell with infinite fire mobility: boundary code
13, WF_FIRE_BLOCKED) are part of the histograms like any other; `n_crossing` counts them, so that the rate at which
the build's generator crosses the fire cell can be compared too.
"""
import argparse
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
N_RIVERS = 40000
N_SPAWNS = 72000


def run_size(size):
    sys.path.insert(0, HERE)
    import gen_golden
    gen_golden._install_stand_ins()
    sys.path.insert(0, REF)
    from Simulation import constants
    constants.METADATA["width"] = size
    constants.METADATA["height"] = size
    from Simulation import environment as E              # the reference, unchanged
    from Simulation.utility import get_agent_location, get_fire_location, layer, types, circle_points
    import random
    np.random.seed(12345 + size)
    random.seed(12345 + size)
    W = H = size
    fx, fy = get_fire_location(W, H)
    env = E.create_map() if hasattr(E, "create_map") else None
    start_col = np.zeros(W, dtype=np.int64)
    start_row = np.zeros(H, dtype=np.int64)
    length = np.zeros(H + 1, dtype=np.int64)
    drift = np.zeros(2 * H + 1, dtype=np.int64)          # net drift + H
    col_cells = np.zeros(W, dtype=np.int64)
    n_cross = 0
    for _ in range(N_RIVERS):
        E.reset_map(env, True)
        water = env[:, :, layer["type"]] == types["water"]      # [x, y]
        if water[fx, fy]:
            n_cross += 1
        xs, ys = np.nonzero(water)
        order = np.argsort(ys)
        xs, ys = xs[order], ys[order]
        assert len(ys) > 0 and (np.diff(ys) == 1).all() and (np.abs(np.diff(xs)) == 1).all()
        start_col[xs[0]] += 1
        start_row[ys[0]] += 1
        length[len(ys)] += 1
        drift[xs[-1] - xs[0] + H] += 1
        np.add.at(col_cells, xs, 1)
    # spawn points in (radius, index) order of circle_points around the fire
    pts = []
    for r in (1, 2, 3):
        pts += [(int(x), int(y)) for x, y in circle_points(fx, fy, r)]
    index = {p: i for i, p in enumerate(pts)}
    assert len(index) == 36
    spawn = np.zeros(36, dtype=np.int64)
    for _ in range(N_SPAWNS):
        spawn[index[tuple(int(v) for v in get_agent_location(W, H))]] += 1
    out = os.path.join(HERE, f"ref_reset_stats_{size}.npz")
    np.savez_compressed(out, size=np.int32(size), n_rivers=np.int64(N_RIVERS), n_crossing=np.int64(n_cross),
                        start_col=start_col, start_row=start_row, length=length, drift=drift, col_cells=col_cells,
                        spawn=spawn, spawn_points=np.asarray(pts, dtype=np.int16))
    print(f"  wrote {os.path.basename(out)}: {N_RIVERS} rivers ({n_cross} through the fire cell), {N_SPAWNS} spawns")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=0)
    args = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit("reference not present: these statistics can only be regenerated in the build container")
    if args.size:
        run_size(args.size)
        return
    for size in (10, 14):
        print(f"size {size}:")
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--size", str(size)])


if __name__ == "__main__":
    main()
