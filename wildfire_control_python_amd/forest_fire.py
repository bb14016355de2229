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
from collections import deque

import numpy as np

from . import _native as N
from .batched import BatchedForestFire, encode_action
from .utility import (color2ascii, get_name, grass, layer, temp_from_count, type2ascii,
                      type2gray, types)


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
        self.wind_speed = sim._b.wind_speed
        self.wind_vector = sim._b.wind_vector
        self._last_reward = 0
        self._planes = np.zeros((self.HEIGHT, self.WIDTH, 4), dtype=np.int8)
        self._env = np.zeros((), dtype=N.ENV_DTYPE)
        self.agents = []
        self.RUNNING = False

    # -- refreshed after every reset/step
    def _refresh(self):
        planes, envs = self._sim._b.download()
        self._planes, self._env = planes[0], envs[0]
        fl = int(self._env["flags"])
        self.RUNNING = bool(fl & N.F_RUNNING)
        self.fire_at_border = bool(fl & N.F_BORDER)
        if fl & N.F_ALIVE:
            self.agents = [Agent(self, self._env["x"], self._env["y"], fl & N.F_DIGGING, fl & N.F_DEAD)]
        else:
            self.agents = []
        ys, xs = np.nonzero(self._planes[:, :, N.P_TYPE] == N.FIRE)
        self.burning_cells = set(zip(xs.tolist(), ys.tolist()))
        # the reference's deque shrinks while it searches; only "empty <=> containment
        # already paid" is observable to callers (environment.py:345, misc.py:39)
        self.border_points = deque()
        if not (fl & N.F_CONTAINED):
            self.reset_border_points()

    def reset_border_points(self):                              # environment.py:215-222
        self.border_points = deque()
        for x in range(self.WIDTH):
            self.border_points.append([x, 0])
            self.border_points.append([x, self.HEIGHT - 1])
        for y in range(self.HEIGHT):
            self.border_points.append([0, y])
            self.border_points.append([self.HEIGHT - 1, y])

    def reset(self):
        self._sim.reset()

    @property
    def env(self):
        """float64 (W, H, 9) array laid out like the reference's World.env
        (environment.py:38-50), rebuilt from the int8 planes; read-only view."""
        p = self._planes
        t = p[:, :, N.P_TYPE].T.astype(np.float64)
        out = np.zeros((self.WIDTH, self.HEIGHT, 9))
        out[:, :, layer["type"]] = t
        out[:, :, layer["gray"]] = np.vectorize(type2gray.get)(p[:, :, N.P_TYPE].T)
        out[:, :, layer["temp"]] = np.vectorize(lambda n: temp_from_count(n, self.wind_speed))(
            p[:, :, N.P_HEAT].T)
        out[:, :, layer["heat"]] = grass["heat"]
        out[:, :, layer["fuel"]] = p[:, :, N.P_FUEL].T
        out[:, :, layer["threshold"]] = grass["threshold"]
        out[:, :, layer["agent_pos"]] = p[:, :, N.P_AGENT].T
        out[:, :, layer["fire_mobility"]] = np.where(t >= N.DIRT, np.inf, 1.0)
        out[:, :, layer["agent_mobility"]] = np.where(t == N.WATER, np.inf, 1.0)
        out.setflags(write=False)
        return out

    def inbounds(self, x, y):                                   # environment.py:225-226
        return 0 <= x < self.WIDTH and 0 <= y < self.HEIGHT

    def traversable(self, x, y):                                # environment.py:229-230
        return self._planes[y, x, N.P_TYPE] != N.WATER

    def is_burning(self, cell):                                 # environment.py:249-251
        x, y = cell
        return self._planes[y, x, N.P_TYPE] == N.FIRE

    def is_burnable(self, cell):                                # environment.py:254-257
        x, y = cell
        return self._planes[y, x, N.P_TYPE] == N.GRASS

    def get_state(self):                                        # environment.py:399-402
        p = self._planes
        t = p[:, :, N.P_TYPE].T
        return np.dstack((p[:, :, N.P_AGENT].T.astype(np.float64), t == N.FIRE, t < N.DIRT))

    def get_reward(self):
        """Reward of the last step (the reference recomputes it with side effects on
        its border deque, environment.py:342-390; here the step kernel produced it)."""
        return self._last_reward

    def print_state(self, state):                               # environment.py:405-420
        for y in range(state.shape[1]):
            for x in range(state.shape[1]):
                if state[0, x, y, 0]:
                    print("A", end="")
                elif state[0, x, y, 1]:
                    print("@", end="")
                elif not state[0, x, y, 2]:
                    print("0", end="")
                else:
                    print("+", end="")
            print("")
        print("")

    def inspect(self, cell):                                    # environment.py:423-436
        x, y = cell
        info = self.env[x, y, :]
        gray = info[layer["gray"]]
        print("\n[Color] Grayscale:", gray, "Ascii:", color2ascii[gray])
        print("[Temperature] ", info[layer["temp"]])
        print("[Heat Power] ", info[layer["heat"]])
        print("[Fuel Level] ", info[layer["fuel"]])
        print("[Ignition Threshold], ", info[layer["threshold"]])
        here = bool(self.agents) and (self.agents[0].x, self.agents[0].y) == (x, y)
        print("[Cell contains an Agent] ", here, "\n")

    def print_info(self, total_reward):                         # environment.py:439-453
        t = self._planes[:, :, N.P_TYPE]
        n = self.WIDTH * self.HEIGHT
        burnt, burning = int((t == N.BURNT).sum()), int((t == N.FIRE).sum())
        dug, healthy = int((t == N.DIRT).sum()), int((t == N.GRASS).sum())
        print("[# of Burnt Cells] ", burnt)
        print("[# of Burning Cells] ", burning)
        print("[# of Dug Cells] ", dug)
        print("[# of Healthy Cells] ", healthy)
        print("[# of Damaged Cells] ", dug + burnt + burning)
        print("[Percent Burnt] ", burnt / n)
        print("[Percent Damaged] ", (burnt + dug + burning) / n)
        print("[Total Reward] ", total_reward)
        print("[Current Reward] ", self.get_reward(), "\n")

    def print_metadata(self):                                   # environment.py:456-460
        import pprint
        pprint.PrettyPrinter().pprint(self._sim.METADATA)
        print("")

    def show_layer(self, layer_num=None):                       # environment.py:463-470
        if layer_num is None:
            import pprint
            pprint.PrettyPrinter().pprint(layer)
            print("")
        else:
            print(self.env[:, :, layer[layer_num]].T)


class ForestFire:
    def __init__(self, device=0, metadata=None, size=None):
        self._b = BatchedForestFire(1, size=size, device=device, metadata=metadata)
        self.METADATA = self._b.METADATA                        # forest_fire.py:21 (own copy here)
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
        a = np.array([encode_action(action)], dtype=np.int8)
        reward, _ = self._b.step_many(a, 1)
        r = float(reward[0, 0])
        self.W._last_reward = int(r) if r == int(r) and r in (
            self.METADATA["default_reward"], self.METADATA["death_penalty"],
            self.METADATA["contained_bonus"]) else r
        self.W._refresh()
        return [self.W.get_state(), self.W._last_reward, not self.W.RUNNING, {}]

    def reset(self):                                            # forest_fire.py:52-54
        self._b.reset_host(rng=np.random)
        self.W._refresh()
        return self.W.get_state()

    def render(self):                                           # forest_fire.py:57-82
        print(" ", end="")
        for x in range(self.width):
            print(x % 10, end="")
        print("")
        return_map = "\n"
        t = self.W._planes[:, :, N.P_TYPE]
        for y in range(self.height):
            print(y % 10, end="")
            for x in range(self.width):
                if self.W.agents and (self.W.agents[0].x, self.W.agents[0].y) == (x, y):
                    symbol = "A"
                else:
                    symbol = type2ascii[int(t[y, x])]
                return_map += symbol
                print(symbol, end="")
            return_map += "\n"
            print("")
        print("")
        return return_map

    def close(self):
        self._b.close()
