#!/usr/bin/env python3
"""Golden-vector generator: drives the REAL reference implementation.

Runs only in the build container (it needs /root/reference, which never travels
to the GPU box).  It imports the reference's ``Simulation`` package unchanged,
read-only, from where it lies, drives ``ForestFire.step()/reset()`` with scripted
action sequences / poked fuel planes, and dumps small ``.npz`` fixtures next to
this file.  The fixtures are DATA (inputs + the reference's outputs); no
reference source text is stored.

Two third-party pieces of the reference's environment are absent in this image
 and are provided here, in this script only:

* PyPI ``colour`` (reference Simulation/utility.py:1): a stand-in exposing
  ``Color(name).red/.green/.blue`` for the five web colours the reference uses.
  It only feeds the cosmetic ``gray`` layer, never an integer plane recorded below.
* ``pyastar.astar_path``: the reference's wrapper (pyastar/pyastar.py:7-22)
  loads ``astar.so`` from its own (read-only) directory, so a loader with the
  same call contract is registered that binds the reference's own native code,
  compiled by ``oracle/Makefile`` from /root/reference/pyastar/astar.cpp into
  ``oracle/_ref/astar.so``.

Usage:  python tests/golden/gen_golden.py            (both sizes, subprocesses)
        python tests/golden/gen_golden.py --size 10  (one size, in-process)

Reference facts relied on (file:line under /root/reference):
  Simulation/environment.py:22-23   WIDTH/HEIGHT frozen at import -> one process per size
  Simulation/environment.py:38-50   env is float64 (W,H,9) indexed [x,y,layer]
  Simulation/forest_fire.py:30-49   step() -> [state, reward, done, {}]
"""
import argparse
import ctypes
import os
import subprocess
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
ASTAR_SO = os.path.join(REPO, "oracle", "_ref", "astar.so")

# plane order of the build's int8 (batch,H,W,4) state buffer
P_TYPE, P_FUEL, P_HEAT, P_AGENT = 0, 1, 2, 3
HEAT_FORCED = 25  # round((threshold+1)/0.162): set_fire_to() on a cold cell, environment.py:236-237
HEAT_IGNITED = 24  # wind fixtures: plane 2 only says how a cell was ignited (0 never / forced / by accumulated heat)
FIRE_BLOCKED = 13  # type fire with fire_mobility inf: a river drawn through the fire cell (environment.py:69-95, then :233-246)

# every astar_path call of the current step: (start x, start y, goal x, goal y, found).  The distinct start cells, in
# call order, are the burning cells get_reward popped from its set copy (environment.py:349-371) - the one piece of
# CPython-internal state (set iteration order) its result depends on when the fire is split.
ASTAR_LOG = []


def _install_stand_ins():
    sys.dont_write_bytecode = True
    # --- colour -----------------------------------------------------------
    rgb = {"green": (0.0, 128 / 255.0, 0.0), "red": (1.0, 0.0, 0.0),
           "black": (0.0, 0.0, 0.0), "brown": (165 / 255.0, 42 / 255.0, 42 / 255.0),
           "blue": (0.0, 0.0, 1.0)}

    class Color:
        def __init__(self, name):
            self.red, self.green, self.blue = rgb[name.lower()]

    m = types.ModuleType("colour")
    m.Color = Color
    sys.modules["colour"] = m

    # --- pyastar (binds the reference's own compiled astar.cpp) -------------
    lib = ctypes.cdll.LoadLibrary(ASTAR_SO)
    f32 = np.ctypeslib.ndpointer(dtype=np.float32, ndim=1, flags="C_CONTIGUOUS")
    i32 = np.ctypeslib.ndpointer(dtype=np.int32, ndim=1, flags="C_CONTIGUOUS")
    lib.astar.restype = ctypes.c_bool
    lib.astar.argtypes = [f32, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                          ctypes.c_int, ctypes.c_bool, i32]

    def astar_path(weights, start, goal, allow_diagonal=False):
        # contract of reference pyastar/pyastar.py:25-64: empty array <=> no path
        h, w = weights.shape
        s = int(start[0]) * w + int(start[1])
        g = int(goal[0]) * w + int(goal[1])
        paths = np.full(h * w, -1, dtype=np.int32)
        ok = lib.astar(np.ascontiguousarray(weights, dtype=np.float32).ravel(),
                       h, w, s, g, bool(allow_diagonal), paths)
        ASTAR_LOG.append((int(start[0]), int(start[1]), int(goal[0]), int(goal[1]), bool(ok and s != g)))
        if not ok or s == g:
            return np.array([])
        out, p = [], g
        while p != s:
            out.append(divmod(p, w))
            p = int(paths[p])
        out.append(divmod(s, w))
        return np.array(out[::-1])

    pkg = types.ModuleType("pyastar")
    pkg.__path__ = []
    sub = types.ModuleType("pyastar.pyastar")
    sub.astar_path = astar_path
    pkg.pyastar = sub
    sys.modules["pyastar"] = pkg
    sys.modules["pyastar.pyastar"] = sub


class Recorder:
    """Collects one fixture file worth of episodes."""

    def __init__(self, sim, layer, size, wind=False):
        self.sim, self.layer, self.size, self.wind = sim, layer, size, wind
        self.winds, self.temps = [], []
        self.c = 0.54 * 0.3 * (0 + 1) ** (-1)  # environment.py:286-287 with wind (0,0)
        self.rows = {k: [] for k in ("planes", "agent", "reward", "done", "obs",
                                     "fab", "border_len", "action", "updated")}
        self.init_planes, self.init_agent, self.init_world, self.ep_start, self.seeds = [], [], [], [], []
        self.pop_cells, self.pop_start = [], [0]     # burning cells popped by get_reward, per step (ragged)

    def _seqsum(self, n):
        t = 0.0
        for _ in range(int(n)):
            t += self.c
        return t

    def _planes(self):
        W = self.sim.W
        env, L = W.env, self.layer
        temp = env[:, :, L["temp"]]
        if self.wind:
            # direction-dependent addends (environment.py:260-287): the temperature is no multiple of one quantum;
            # plane 2 records how the cell was ignited, the float temperatures go into their own array
            q = np.where(temp == 4.0, HEAT_FORCED, np.where(temp > 3.0, HEAT_IGNITED, 0))
        else:
            q = np.rint(temp / self.c).astype(np.int64)
            # the float64 temperature must be exactly the running sum of q equal addends
            # (order independence claim of SURVEY 8a) or the forced value threshold+1
            for (x, y), v in np.ndenumerate(q):
                t = temp[x, y]
                if v == HEAT_FORCED and t == 4.0:
                    continue
                assert t == self._seqsum(v), (x, y, t, v)
        p = np.zeros((self.size, self.size, 4), dtype=np.int8)  # [y, x, plane]
        code = env[:, :, L["type"]].astype(np.int64)         # [x, y]
        fm = env[:, :, L["fire_mobility"]]
        # Two states only allow_dig_toggle can produce (toggle_digging -> dig() on a burning cell,
        # environment.py:123-138): type dirt but still a member of burning_cells -> boundary code 5; and, once
        # its fuel is gone (reduce_fuel :297-307), type burnt with fire_mobility still inf -> code 6.
        # A third one only make_rivers can produce: the river drawn through the fire cell (reset_map :69-95) leaves it
        # type water with fire_mobility inf, then set_fire_to (:233-246) retypes it to fire: a burning cell that shows as
        # fire but blocks the search -> code 13; burnt out it is type burnt, still blocked -> code 6 again.
        for (x, y) in W.burning_cells:
            if code[x, y] == 3:
                code[x, y] = 5
            else:
                assert code[x, y] == 1, (x, y, code[x, y])
                if np.isinf(fm[x, y]):
                    code[x, y] = FIRE_BLOCKED
        code[(code == 2) & np.isinf(fm)] = 6
        assert set(zip(*np.nonzero((code == 1) | (code == 5) | (code == FIRE_BLOCKED)))) == \
            {(int(x), int(y)) for x, y in W.burning_cells}
        # fire_mobility is a pure function of the boundary code (environment.py:85,133)
        assert np.array_equal(np.isinf(fm), code >= 3)
        p[:, :, P_TYPE] = code.T
        p[:, :, P_FUEL] = env[:, :, L["fuel"]].T
        p[:, :, P_HEAT] = q.T
        p[:, :, P_AGENT] = env[:, :, L["agent_pos"]].T
        return p

    def _agent(self):
        W = self.sim.W
        if W.agents:
            a = W.agents[0]
            return [int(a.x), int(a.y), 1, int(a.digging), int(a.dead)]
        return [-1, -1, 0, 0, 1]

    def begin(self, seed=-1):
        self.seeds.append(seed)
        self.ep_start.append(len(self.rows["action"]))
        self.init_planes.append(self._planes())
        self.init_agent.append(self._agent())
        self.init_world.append([int(self.sim.W.fire_at_border), len(self.sim.W.border_points)])
        if self.wind:
            W = self.sim.W
            self.winds.append([float(W.wind_speed), float(W.wind_vector[0]), float(W.wind_vector[1])])

    def step(self, action):
        del ASTAR_LOG[:]
        tick_before = self.sim.METADATA["a_speed_iter"]
        state, reward, done, _ = self.sim.step(action)
        r = self.rows
        r["updated"].append(tick_before == 1)                # forest_fire.py:40-43: update() ran in this step
        last = None
        for (sx, sy, _gx, _gy, _found) in ASTAR_LOG:
            if (sx, sy) != last:
                self.pop_cells.append((sx, sy))
                last = (sx, sy)
        self.pop_start.append(len(self.pop_cells))
        r["planes"].append(self._planes())
        r["agent"].append(self._agent())
        r["reward"].append(float(reward))
        r["done"].append(bool(done))
        r["obs"].append(np.asarray(state).astype(np.int8))  # [x, y, k] as the reference returns it
        r["fab"].append(bool(self.sim.W.fire_at_border))
        r["border_len"].append(len(self.sim.W.border_points))
        r["action"].append(int(action))
        if self.wind:
            self.temps.append(self.sim.W.env[:, :, self.layer["temp"]].T.astype(np.float32))   # [y, x]
        return state, reward, done

    def save(self, path):
        r = self.rows
        extra = {}
        if getattr(self, "a_speed_iter0", None) is not None:      # METADATA['a_speed_iter'] at the start of every episode
            extra["a_speed_iter0"] = np.asarray(self.a_speed_iter0, dtype=np.int32)
        if self.wind:
            extra.update(wind=np.asarray(self.winds, dtype=np.float64),       # per episode: speed, wx, wy
                         temp=np.asarray(self.temps, dtype=np.float32))      # per step: World.env[:, :, temp] as [y, x]
        np.savez_compressed(
            path,
            **extra,
            size=np.int32(self.size),
            ep_start=np.asarray(self.ep_start + [len(r["action"])], dtype=np.int64),
            init_planes=np.asarray(self.init_planes, dtype=np.int8),
            init_agent=np.asarray(self.init_agent, dtype=np.int16),
            init_world=np.asarray(self.init_world, dtype=np.int32),   # [fire_at_border, len(border_points)]
            seeds=np.asarray(self.seeds, dtype=np.int64),             # np.random.seed/random.seed before reset (-1: n/a)
            action=np.asarray(r["action"], dtype=np.int8),
            planes=np.asarray(r["planes"], dtype=np.int8),
            agent=np.asarray(r["agent"], dtype=np.int16),
            reward=np.asarray(r["reward"], dtype=np.float64),
            done=np.asarray(r["done"], dtype=np.uint8),
            obs=np.asarray(r["obs"], dtype=np.int8),
            fire_at_border=np.asarray(r["fab"], dtype=np.uint8),
            border_len=np.asarray(r["border_len"], dtype=np.int32),
            updated=np.asarray(r["updated"], dtype=np.uint8),         # update() ran in this step (a_speed, forest_fire.py:40-43)
            pop_cells=np.asarray(self.pop_cells, dtype=np.int16).reshape(-1, 2),   # (x, y) get_reward searched from ...
            pop_start=np.asarray(self.pop_start, dtype=np.int64),     # ... step i: pop_cells[pop_start[i]:pop_start[i+1]]
        )
        print(f"  wrote {os.path.relpath(path, REPO)}: {len(self.ep_start)} episodes, "
              f"{len(r['action'])} steps, {os.path.getsize(path)} bytes")


def circling_policy(sim, rng, avoid_fire=True):
    """Own scripted policy (not the reference's): walk clockwise around the fire,
    refusing moves into a burning cell when another is available."""
    W = sim.W
    if not W.agents:
        return 0
    a = W.agents[0]
    mx, my = W.WIDTH // 2, W.HEIGHT // 2
    if a.x >= mx and a.y > my:
        opts = [1, 3]
    elif a.x > mx and a.y <= my:
        opts = [1, 2]
    elif a.x <= mx and a.y < my:
        opts = [0, 2]
    else:
        opts = [0, 3]
    for _ in range(8):
        act = int(opts[rng.integers(len(opts))])
        if not avoid_fire or not a.fire_in_direction(act):
            return act
    return act


def run_size(size):
    _install_stand_ins()
    sys.path.insert(0, REF)
    from Simulation import constants
    constants.METADATA["width"] = size
    constants.METADATA["height"] = size
    from Simulation.forest_fire import ForestFire  # noqa: E402  (reference, unchanged)
    from Simulation.utility import layer
    import random

    sim = ForestFire()
    assert sim.W.WIDTH == size and sim.W.HEIGHT == size
    out = lambda name: os.path.join(HERE, f"ref_{size}_{name}.npz")
    NOOP = 5

    def reset(seed):
        np.random.seed(seed)
        random.seed(seed)
        sim.reset()

    # 1. known-answer episode of SURVEY 8a: seed 0, no-op actions until done
    rec = Recorder(sim, layer, size)
    reset(0)
    rec.begin()
    done = False
    while not done:
        _, _, done = rec.step(NOOP)
    rec.save(out("noop"))

    # 2. uniformly random actions 0..3 with ~10% no-ops (digs, rejected moves, deaths)
    rec = Recorder(sim, layer, size)
    rng = np.random.default_rng(1000 + size)
    for ep in range(24):
        reset(100 + ep)
        rec.begin()
        done = False
        while not done:
            a = int(rng.integers(0, 4)) if rng.random() > 0.1 else NOOP
            _, _, done = rec.step(a)
    rec.save(out("random"))

    # 3. circling policy: digs a ring -> exercises the containment (+1000) reward
    rec = Recorder(sim, layer, size)
    rng = np.random.default_rng(2000 + size)
    for ep in range(16):
        reset(200 + ep)
        rec.begin()
        done = False
        while not done:
            _, _, done = rec.step(circling_policy(sim, rng, avoid_fire=(ep % 4 != 3)))
    rec.save(out("circle"))

    # 4. random fuel U{10..40} poked after reset (BASELINE configs 2-5), random + no-op actions
    rec = Recorder(sim, layer, size)
    rng = np.random.default_rng(3000 + size)
    for ep in range(12):
        reset(300 + ep)
        sim.W.env[:, :, layer["fuel"]] = rng.integers(10, 41, size=(size, size)).astype(np.float64)
        rec.begin()
        done = False
        while not done:
            if ep % 3 == 0:
                a = NOOP
            elif ep % 3 == 1:
                a = int(rng.integers(0, 4))
            else:
                a = circling_policy(sim, rng)
            _, _, done = rec.step(a)
    rec.save(out("randfuel"))

    # 5. several ignition points + low fuel (fronts merge, burn-outs next to ignitions)
    rec = Recorder(sim, layer, size)
    rng = np.random.default_rng(4000 + size)
    for ep in range(8):
        reset(400 + ep)
        sim.W.env[:, :, layer["fuel"]] = rng.integers(6, 31, size=(size, size)).astype(np.float64)
        for _ in range(3):
            x, y = int(rng.integers(0, size)), int(rng.integers(0, size))
            if sim.W.is_burnable((x, y)):
                sim.W.set_fire_to((x, y))
        rec.begin()
        done = False
        while not done:
            a = int(rng.integers(0, 4)) if ep % 2 else NOOP
            _, _, done = rec.step(a)
    rec.save(out("multifire"))

    # 6. rivers (reset_map environment.py:69-95): water is neither burnable nor traversable.  A river may run through
    #    the fire cell (8-16 % of them do): set_fire_to then retypes that water cell to fire and its fire_mobility stays
    #    inf - boundary code 13 (see Recorder._planes); those seeds are part of the set.
    rec = Recorder(sim, layer, size)
    rng = np.random.default_rng(5000 + size)
    constants.METADATA["make_rivers"] = True
    fx, fy = size // 2, size // 2
    crossing, seed = 0, 500
    for ep in range(16):
        if ep >= 12:                    # the last four episodes: the next seeds whose river crosses the fire cell
            while True:
                reset(seed)
                if sim.W.env[fx, fy, layer["fire_mobility"]] == np.inf:
                    break
                seed += 1
        else:
            reset(seed)
        crossing += int(sim.W.env[fx, fy, layer["fire_mobility"]] == np.inf)
        rec.begin(seed)
        seed += 1
        done = False
        while not done:
            a = int(rng.integers(0, 4)) if ep % 2 == 0 else circling_policy(sim, rng)
            _, _, done = rec.step(a)
    constants.METADATA["make_rivers"] = False
    assert crossing >= 4, crossing
    seen = np.asarray(rec.rows["planes"])[..., P_TYPE]
    assert (seen == FIRE_BLOCKED).any() and (seen == 6).any(), "river fixtures must hold a blocked fire cell and its burnt-out state"
    rec.save(out("river"))

    # 7. allow_dig_toggle (forest_fire.py:35-36): action 4 toggles dig mode.  Odd episodes are scripted so that the
    #    agent toggles digging ON while standing on a BURNING cell - the reference then holds a dirt cell that keeps
    #    burning (codes 5 / 6 above): dig mode off, walk next to the fire, wait until the own cell ignites, toggle.
    rec = Recorder(sim, layer, size)
    rng = np.random.default_rng(6000 + size)
    constants.METADATA["allow_dig_toggle"] = True
    zombies = 0
    for ep in range(20):
        reset(600 + ep)
        if ep % 4 == 3:
            sim.W.env[:, :, layer["fuel"]] = rng.integers(4, 31, size=(size, size)).astype(np.float64)
        rec.begin()
        done, phase, t = False, 0, 0
        while not done:
            ag = sim.W.agents[0] if sim.W.agents else None
            if ep % 2 == 0 or ag is None:
                a = int(rng.integers(0, 5))
            elif phase == 0:                       # dig mode off
                a, phase = 4, 1
            elif phase == 1:                       # approach: stand on a grass cell next to the fire
                fx, fy = size // 2, size // 2
                opts = [d for d in range(4) if not ag.fire_in_direction(d)]
                best = [d for d in opts
                        if abs(ag._direction_to_coords(d)[0] - fx) + abs(ag._direction_to_coords(d)[1] - fy)
                        < abs(ag.x - fx) + abs(ag.y - fy)]
                near = any(ag.fire_in_direction(d) for d in range(4))
                if near and sim.W.is_burnable((ag.x, ag.y)):
                    a, phase = NOOP, 2
                else:
                    a = int(rng.choice(best)) if best else (int(rng.choice(opts)) if opts else NOOP)
            elif phase == 2:                       # wait for the own cell to ignite, then toggle digging on
                if sim.W.is_burning((ag.x, ag.y)):
                    a, phase = 4, 3
                    zombies += 1
                else:
                    a = NOOP if sim.W.is_burnable((ag.x, ag.y)) else int(rng.integers(0, 4))
            else:                                  # afterwards: wander, toggling now and then
                a = int(rng.integers(0, 5)) if t % 3 else NOOP
            _, _, done = rec.step(a)
            t += 1
    constants.METADATA["allow_dig_toggle"] = False
    assert zombies >= 4, zombies
    seen = np.asarray(rec.rows["planes"])[..., P_TYPE]
    assert (seen == 5).any() and (seen == 6).any(), "the toggle fixture must contain burning-dirt and blocked-burnt cells"
    rec.save(out("toggle"))

    # 8. METADATA['wind'] = "random" (environment.py:188-190): per-episode wind speed in {0, 0.7, 0.85} and vector in
    #    {-1,0,1}^2; the heat addend then depends on the direction burning cell -> neighbour (:260-287)
    rec = Recorder(sim, layer, size, wind=True)
    rng = np.random.default_rng(7000 + size)
    constants.METADATA["wind"] = "random"
    for ep in range(40):
        reset(700 + ep)
        if ep % 5 == 4:
            sim.W.env[:, :, layer["fuel"]] = rng.integers(10, 41, size=(size, size)).astype(np.float64)
        rec.begin(700 + ep)
        done = False
        while not done:
            if ep % 3 == 0:
                a = NOOP
            elif ep % 3 == 1:
                a = int(rng.integers(0, 4)) if rng.random() > 0.3 else NOOP
            else:
                a = circling_policy(sim, rng)
            _, _, done = rec.step(a)
    constants.METADATA["wind"] = [0.54, (0, 0)]
    seen = {tuple(w) for w in rec.winds}
    assert len({w[0] for w in seen}) == 3 and len({w[1:] for w in seen}) >= 8, seen
    rec.save(out("wind"))

    # 9. split fire (get_reward's containment search, environment.py:345-377, when the burning cells do not all lie in
    #    one region of passable cells): scripted agents that dig BURNT cells until a wall is complete, poked dirt walls,
    #    several ignition points.  The reference pops the burning cells from a copy of its set - the order is CPython's
    #    (recorded in pop_cells) - and searches from the first one against the border deque it has left from earlier steps.
    import Simulation.environment as E
    from Simulation.utility import dirt as ref_dirt, grass as ref_grass, types as ref_types
    real_spawn = E.get_agent_location
    c = size // 2

    def spawn_at(x, y):
        E.get_agent_location = lambda w, h: (x, y)

    def poke(cells, kind):
        for (x, y) in cells:
            if kind == "dirt":                                   # what Agent.dig writes (environment.py:127-133)
                sim.W.env[x, y, layer["type"]] = ref_types["dirt"]
                sim.W.env[x, y, layer["gray"]] = ref_dirt["gray"]
                sim.W.env[x, y, layer["fire_mobility"]] = np.inf
            else:                                                # what reduce_fuel writes (environment.py:302-304)
                sim.W.env[x, y, layer["type"]] = ref_types["burnt"]
                sim.W.env[x, y, layer["gray"]] = ref_grass["gray_burnt"]
                sim.W.env[x, y, layer["fuel"]] = 0

    def run(script, until_done=True, wait_burnt=None, tail=NOOP, max_steps=400):
        """script: actions; "wait" = no-ops until cell wait_burnt is burnt.  Then `tail` actions until done."""
        done, n = False, 0
        for a in script:
            if a == "wait":
                while not done and sim.W.env[wait_burnt[0], wait_burnt[1], layer["type"]] != ref_types["burnt"]:
                    _, _, done = rec.step(NOOP)
                    n += 1
                continue
            if done:
                break
            _, _, done = rec.step(a)
            n += 1
        while until_done and not done and n < max_steps:
            _, _, done = rec.step(tail if not callable(tail) else tail())
            n += 1

    Nn, Ss, Ee, Ww = 0, 1, 2, 3
    rec = Recorder(sim, layer, size)
    # 9a. a wall along the fire's row, closed by digging the burnt-out centre: the cell north of the centre burns on in
    #     a one-cell pocket, the rest of the fire south of the wall
    arc = [Nn, Nn, Ww, Ww, Ss, Ss]                               # around the centre and the cell north of it
    west, east = c - 1, size - 1 - (c + 1)
    wall_script = arc + [Ww] * west + [Ee] * west + [Nn, Nn, Ee, Ee, Ss, Ss] + [Ee] * east + [Ww] * east + ["wait", Ww]
    for variant in range(3):
        spawn_at(c + 1, c)
        reset(900 + variant)
        if variant == 1:                                         # the pocket cell burns for a long time
            sim.W.env[c, c - 1, layer["fuel"]] = 90
        if variant == 2:                                         # ... and so does the fire south of the wall
            sim.W.env[:, :, layer["fuel"]] = np.random.default_rng(9000 + size).integers(15, 61, size=(size, size)).astype(np.float64)
        rec.begin(900 + variant)
        run(wall_script, wait_burnt=(c, c))
    # 9b. the same wall inside a poked dirt box with one gap; the agent leaves along the wall and closes the gap from
    #     outside while BOTH parts of the split fire still burn -> containment is paid with the fire in two regions
    box = [(c + dx, c + dy) for dx in range(-3, 4) for dy in range(-3, 4) if max(abs(dx), abs(dy)) == 3 and (dx, dy) != (0, 3)]
    for variant in range(2):
        spawn_at(c + 1, c)
        reset(910 + variant)
        poke(box, "dirt")
        if variant == 1:
            sim.W.env[c, c - 1, layer["fuel"]] = 70
            sim.W.env[c, c + 1, layer["fuel"]] = 70
        rec.begin(910 + variant)
        script = [Ee, Ww] + arc + [Ww, Ee, "wait", Ee,                       # wall (c-2..c+2, c) inside the box
                                   Ww, Ww, Ww, Ss, Ss, Ss, Ee, Ee, Ee]       # out along the wall, round the corner, into the gap
        run(script, wait_burnt=(c, c))
    # 9c. a poked dirt wall from border to border, one ignition on each side; random walk on the agent's side
    rng = np.random.default_rng(9100 + size)
    for variant in range(3):
        spawn_at(c + 2, c)
        reset(920 + variant)
        poke([(c - 2, y) for y in range(size)], "dirt")
        sim.W.set_fire_to((1 + variant % 2, c - 1 + variant))
        sim.W.env[:, :, layer["fuel"]] = rng.integers(8, 41, size=(size, size)).astype(np.float64)
        rec.begin(920 + variant)
        run([], tail=(lambda: int(rng.integers(0, 4))) if variant else NOOP)
    # 9d. the reference's answer depends on the pop order: a short-lived fire in a chamber whose only way out is border
    #     point [1, 0] (index 2 of the deque), and a lone fire cell among burnt cells that reaches every other border
    #     point.  Whenever the chamber's cell is searched first the deque shrinks to 3 entries; should the lone cell
    #     then be searched first against those, it "fails" and containment is paid although it reaches the border.
    for variant in range(4):
        spawn_at(c + 2, c + (variant % 2))
        reset(930 + variant)
        ax = 1 if variant < 2 else size - 2                       # chamber on the left / on the right
        poke([(ax - 1, 0), (ax + 1, 0), (ax - 1, 1), (ax + 1, 1), (ax, 2), (0, size - 1)], "dirt")
        poke([(c + 1, c), (c - 1, c), (c, c + 1), (c, c - 1)], "burnt")
        sim.W.env[c, c, layer["fuel"]] = 60
        sim.W.set_fire_to((ax, 1))
        sim.W.env[ax, 1, layer["fuel"]] = 6 + variant
        rec.begin(930 + variant)
        run([], max_steps=80)
    E.get_agent_location = real_spawn
    n_multi = int((np.diff(rec.pop_start) > 1).sum())
    assert n_multi > 20, n_multi                                  # steps in which more than one burning cell was searched from
    rec.save(out("splitfire"))

    # 10. a_speed = 2 (forest_fire.py:40-43): update() runs on every second step, counted by METADATA['a_speed_iter'],
    #     which is module-global and is NOT touched by reset() (:52-54): an episode with an odd number of steps leaves
    #     the next one out of phase.  Several episodes back to back; the counter is only ever written by step().
    rec = Recorder(sim, layer, size)
    rng = np.random.default_rng(9500 + size)
    constants.METADATA["a_speed"] = 2
    iters = []
    for ep in range(10):
        reset(950 + ep)
        iters.append(int(constants.METADATA["a_speed_iter"]))
        rec.begin(950 + ep)
        done = False
        while not done:
            a = int(rng.integers(0, 4)) if ep % 2 == 0 else circling_policy(sim, rng)
            _, _, done = rec.step(a)
    constants.METADATA["a_speed"] = 1
    constants.METADATA["a_speed_iter"] = 1
    assert len(set(iters)) == 2, iters                            # both phases occur at an episode start
    rec.a_speed_iter0 = iters
    rec.save(out("aspeed2"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=0)
    args = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit("reference not present: fixtures can only be regenerated in the build container")
    if not os.path.exists(ASTAR_SO):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "ref"])
    if args.size:
        run_size(args.size)
        return
    for size in (10, 14):
        print(f"size {size}:")
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--size", str(size)])


if __name__ == "__main__":
    main()
