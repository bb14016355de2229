"""ForestFire: single-environment adapter with the attribute surface the reference's
learners touch, backed by a batch-of-one context on the GPU.

    sim.step(a) -> [state (W,H,3) float64, reward, done, {}]   forest_fire.py:30-49
    sim.reset() -> state                                        forest_fire.py:52-54
    sim.render() -> str                                         forest_fire.py:57-82
    sim.METADATA / DEBUG / n_actions / layer / get_name / width / height
    sim.W.WIDTH/HEIGHT/DEPTH, sim.W.agents[0].x/.y/.fire_in_direction(a),
    sim.W.RUNNING, sim.W.wind_speed/.wind_vector, sim.W.border_points,
    sim.W.get_state(), sim.W.inspect/show_layer/print_info/print_metadata

DQN.py / DQN_SARSA.py style loops run against it unchanged.  reset() draws the
spawn point from numpy's legacy global stream in the reference's own order, so
`np.random.seed(s); sim.reset()` spawns the agent exactly where the reference does.
"""
import pprint
from collections import deque

import numpy as np

from . import _native as N
from .batched import BatchedForestFire, encode_action
from .utility import get_name, grass, layer, temp_from_count, type2ascii, type2gray

# look-up tables indexed by the boundary's type code (5 = burning dirt, 6 = blocked burnt, 13 = blocked fire:
# include/wildfire.h; 7..12 are not used)
_REF_TYPE = np.array([0, 1, 2, 3, 4, 3, 2, 0, 0, 0, 0, 0, 0, 1], dtype=np.int64)    # the type the reference's World.env would hold
# World.get_state() per type code: (agent_pos filled in later, type == fire, fire_mobility != inf)
_STATE_LUT = np.array([[0.0, float(c in (N.FIRE, N.FIRE_BLOCKED)), float(c < N.DIRT)] for c in range(14)])
_ASCII = np.array([type2ascii[int(t)] for t in _REF_TYPE])
_GRAY = np.array([type2gray[int(t)] for t in _REF_TYPE])


class Agent:
    """View of the environment's agent (environment.py:99-171)."""

    def __init__(self, world, x, y, digging, dead):
        self.W = world
        self.x, self.y = int(x), int(y)
        self.digging = bool(digging)
        self.dead = bool(dead)

    def _direction_to_coords(self, direction):                  # environment.py:163-171
        if direction in ["N", 0]:
            return (self.x, self.y - 1)
        if direction in ["S", 1]:
            return (self.x, self.y + 1)
        if direction in ["E", 2]:
            return (self.x + 1, self.y)
        if direction in ["W", 3]:
            return (self.x - 1, self.y)

    def fire_in_direction(self, direction):                     # environment.py:158-160
        nx, ny = self._direction_to_coords(direction)
        return self.W.inbounds(nx, ny) and self.W.is_burning((nx, ny))

    def is_dead(self):
        return self.dead or self.W.is_burning((self.x, self.y))


class World:
    """Host-side view of one environment's device state (environment.py:174-471)."""

    def __init__(self, sim):
        self._sim = sim
        self.WIDTH, self.HEIGHT = sim.width, sim.height
        self.DEPTH = 3                                           # get_state().shape[2], :181
        self.wind_speed = sim._b.wind_speed                      # per episode with wind "random": set by _refresh
        self.wind_vector = sim._b.wind_vector
        self._counts, self._addends = None, None
        self._last_reward = 0
        self._planes = np.zeros((self.HEIGHT, self.WIDTH, 4), dtype=np.int8)
        self._env = np.zeros((), dtype=N.ENV_DTYPE)
        self.agents = []
        self.RUNNING = False
        self._burning_cells, self._border_points, self._contained = None, None, False

    # -- refreshed after every reset/step
    def _refresh(self, state=None):
        planes, envs = state if state is not None else self._sim._b.download()
        self._planes, self._env = planes[0], envs[0]
        b = self._sim._b
        if b.wind_mode != N.WIND_CALM:                           # environment.py:187-193
            speed, vec = b.winds(envs)
            self.wind_speed, self.wind_vector = float(speed[0]), (int(vec[0][0]), int(vec[0][1]))
            self._counts, self._addends = b.download_counts()[0], b.wind_addends(envs)[0]
        fl = int(self._env["flags"])
        self.RUNNING = bool(fl & N.F_RUNNING)
        self.fire_at_border = bool(fl & N.F_BORDER)
        if fl & N.F_ALIVE:
            self.agents = [Agent(self, self._env["x"], self._env["y"], fl & N.F_DIGGING, fl & N.F_DEAD)]
        else:
            self.agents = []
        self._burning_cells = None               # built on first use (the learners' hot loop reads neither)
        self._contained = bool(fl & N.F_CONTAINED)
        self._border_points = None

    @property
    def burning_cells(self):
        """World.burning_cells (environment.py:202): the set of (x, y) of every cell that burns."""
        if self._burning_cells is None:
            ys, xs = np.nonzero((self._planes[:, :, N.P_TYPE] & 3) == 1)      # fire (1, 13), or dirt dug under a fire (5)
            self._burning_cells = set(zip(xs.tolist(), ys.tolist()))
        return self._burning_cells

    @property
    def border_points(self):
        """The reference's deque shrinks while it searches; only "empty <=> containment already paid" is observable
        to callers (environment.py:345, misc.py:39)."""
        if self._border_points is None:
            self._border_points = deque()
            if not self._contained:
                self.reset_border_points()
        return self._border_points

    @border_points.setter
    def border_points(self, value):
        self._border_points = value

    def reset_border_points(self):
        """All border coordinates ([x, y] pairs, corners twice like the reference's deque, environment.py:215-222).
        Only its emptiness is ever observed from outside (misc.py:39)."""
        xs, ys = np.arange(self.WIDTH), np.arange(self.HEIGHT)
        top_bottom = np.stack([np.repeat(xs, 2), np.tile([0, self.HEIGHT - 1], self.WIDTH)], axis=1)
        left_right = np.stack([np.tile([0, self.HEIGHT - 1], self.HEIGHT), np.repeat(ys, 2)], axis=1)   # (sic) HEIGHT-1 as an x
        self.border_points = deque(np.concatenate([top_bottom, left_right]).tolist())

    def reset(self):
        self._sim.reset()

    @property
    def env(self):
        """float64 (W, H, 9) array laid out like the reference's World.env
        (environment.py:38-50), rebuilt from the int8 planes; read-only view."""
        p = self._planes
        code = p[:, :, N.P_TYPE].T
        t = _REF_TYPE[code].astype(np.float64)                  # codes 5 / 6 read dirt / burnt in the reference
        out = np.zeros((self.WIDTH, self.HEIGHT, 9))
        out[:, :, layer["type"]] = t
        out[:, :, layer["gray"]] = _GRAY[code]
        out[:, :, layer["temp"]] = self._temp().T
        out[:, :, layer["heat"]] = grass["heat"]
        out[:, :, layer["fuel"]] = p[:, :, N.P_FUEL].T
        out[:, :, layer["threshold"]] = grass["threshold"]
        out[:, :, layer["agent_pos"]] = p[:, :, N.P_AGENT].T
        out[:, :, layer["fire_mobility"]] = np.where(code >= N.DIRT, np.inf, 1.0)
        # (the river cell under the fire keeps the agent_mobility reset_map gave it, environment.py:87)
        out[:, :, layer["agent_mobility"]] = np.where((code == N.WATER) | (code == N.FIRE_BLOCKED), np.inf, 1.0)
        out.setflags(write=False)
        return out

    def _temp(self):
        """float64 (H, W): the temp layer.  Wind vector (0,0): the running sum of equal addends the heat plane counts.
        Wind with a direction: the sum of the counted per-direction addends of a grass cell; threshold + 1 where
        set_fire_to forced it (environment.py:236-237); threshold + 0.5 - a finite stand-in above the threshold - for a
        cell that accumulated heat ignited (the reference keeps the sum that crossed the threshold; the build keeps no
        temperature for a cell that can never take heat again, and the temperature of a burning or burnt cell is
        never read by the rule)."""
        heat = self._planes[:, :, N.P_HEAT]
        if self._counts is None:
            return np.vectorize(lambda n: temp_from_count(n, self.wind_speed))(heat)
        t = (self._counts.astype(np.float64) * self._addends).sum(axis=-1)
        t[heat == N.HEAT_FORCED] = grass["threshold"] + 1
        t[heat == N.HEAT_IGNITED] = grass["threshold"] + 0.5
        return t

    def inbounds(self, x, y):                                   # environment.py:225-226
        return 0 <= x < self.WIDTH and 0 <= y < self.HEIGHT

    def traversable(self, x, y):                                # environment.py:229-230
        return self._planes[y, x, N.P_TYPE] != N.WATER

    def is_burning(self, cell):                                 # environment.py:249-251
        x, y = cell
        return self._planes[y, x, N.P_TYPE] in (N.FIRE, N.FIRE_BLOCKED)

    def is_burnable(self, cell):                                # environment.py:254-257
        x, y = cell
        return self._planes[y, x, N.P_TYPE] == N.GRASS

    def get_state(self):                                        # environment.py:399-402
        p = self._planes
        out = _STATE_LUT[p[:, :, N.P_TYPE].T]                    # float64 (W, H, 3), one table look-up per cell
        out[:, :, 0] = p[:, :, N.P_AGENT].T
        return out

    def get_reward(self):
        """Reward of the last step (the reference recomputes it with side effects on
        its border deque, environment.py:342-390; here the step kernel produced it)."""
        return self._last_reward

    # ---- debug printers of misc.run_human (misc.py:61-72): same text as the reference prints, built from the
    # int8 planes with table look-ups instead of per-cell Python branches
    def print_state(self, state):                               # what the learner sees (environment.py:405-420)
        st = np.asarray(state)[0]                               # [x, y, k]
        sym = np.where(st[..., 0] != 0, "A", np.where(st[..., 1] != 0, "@", np.where(st[..., 2] == 0, "0", "+")))
        print("\n".join("".join(row) for row in sym.T) + "\n\n", end="")

    def inspect(self, cell):                                    # one cell (environment.py:423-436)
        x, y = cell
        code, fuel, heat, _ = (int(v) for v in self._planes[y, x])
        here = any((a.x, a.y) == (x, y) for a in self.agents[:1])
        lines = [("\n[Color] Grayscale:", _GRAY[code], "Ascii:", _ASCII[code]),
                 ("[Temperature] ", float(self._temp()[y, x])),
                 ("[Heat Power] ", grass["heat"]),
                 ("[Fuel Level] ", float(fuel)),
                 ("[Ignition Threshold], ", float(grass["threshold"])),
                 ("[Cell contains an Agent] ", here, "\n")]
        for ln in lines:
            print(*ln)

    def print_info(self, total_reward):                         # counts per cell type (environment.py:439-453)
        n_of = np.bincount(_REF_TYPE[self._planes[:, :, N.P_TYPE]].ravel(), minlength=5)
        cells = self.WIDTH * self.HEIGHT
        burnt, burning, dug, healthy = (int(n_of[k]) for k in (N.BURNT, N.FIRE, N.DIRT, N.GRASS))
        for label, value in (("[# of Burnt Cells] ", burnt), ("[# of Burning Cells] ", burning),
                             ("[# of Dug Cells] ", dug), ("[# of Healthy Cells] ", healthy),
                             ("[# of Damaged Cells] ", dug + burnt + burning), ("[Percent Burnt] ", burnt / cells),
                             ("[Percent Damaged] ", (burnt + dug + burning) / cells), ("[Total Reward] ", total_reward)):
            print(label, value)
        print("[Current Reward] ", self.get_reward(), "\n")

    def print_metadata(self):                                   # environment.py:456-460
        pprint.pprint(self._sim.METADATA)
        print("")

    def show_layer(self, layer_num=None):                       # environment.py:463-470
        if layer_num is None:
            pprint.pprint(layer)
            print("")
        else:
            print(self.env[:, :, layer[layer_num]].T)


class ForestFire:
    def __init__(self, device=0, metadata=None, size=None):
        """With neither `metadata` nor `size` the object works on the package's module-global METADATA dict
        (constants.METADATA), exactly like the reference (forest_fire.py:20; constants.py:30-57): edit that dict before
        constructing, and - the reference's global-state quirk - every such object of the process shares the a_speed
        counter step() mutates (forest_fire.py:40-43).  With arguments it owns a private copy."""
        from .constants import METADATA as shared
        if metadata is None and size is None:
            metadata = shared
        self._b = BatchedForestFire(1, size=size, device=device, metadata=metadata)
        self.METADATA = self._b.METADATA                        # forest_fire.py:21
        self.DEBUG = self.METADATA["debug"]
        self.layer = layer
        self.get_name = get_name
        self.width = self.METADATA["width"]
        self.height = self.METADATA["height"]
        self.n_actions = self.METADATA["n_actions"]
        self.W = World(self)
        self.reset()                                            # World.__init__ resets, :183

    def step(self, action):                                     # forest_fire.py:30-49
        if not self.W.RUNNING and not self.W.agents and encode_action(action) < 4:
            # the reference indexes W.agents[0] here and raises IndexError (forest_fire.py:33)
            raise IndexError("list index out of range")
        # the step and the state it leaves, one synchronisation, into buffers the batched object keeps
        reward, _, planes, envs = self._b.step_download_reuse(encode_action(action))
        r = float(reward[0])
        self.W._last_reward = int(r) if r == int(r) and r in (
            self.METADATA["default_reward"], self.METADATA["death_penalty"],
            self.METADATA["contained_bonus"]) else r
        self.W._refresh((planes, envs))
        return [self.W.get_state(), self.W._last_reward, not self.W.RUNNING, {}]

    def reset(self):                                            # forest_fire.py:52-54
        self._b.reset_host(rng=np.random)
        self.W._refresh()
        return self.W.get_state()

    def render(self):                                           # forest_fire.py:57-82
        """Prints the ascii map with index markers and returns it as one string (DQN.py:130,245 store it)."""
        sym = _ASCII[self.W._planes[:, :, N.P_TYPE]]            # [y, x]
        for a in self.W.agents[:1]:
            sym[a.y, a.x] = "A"
        rows = ["".join(r) for r in sym]
        print(" " + "".join(str(x % 10) for x in range(self.width)))
        for y, r in enumerate(rows):
            print(f"{y % 10}{r}")
        print("")
        return "\n" + "".join(r + "\n" for r in rows)

    def close(self):
        self._b.close()
