"""BatchedForestFire: many independent forest-fire environments stepped in lock-step
on one MI355X through the C-ABI of include/wildfire.h.

Same verbs as the reference environment facade (Simulation/forest_fire.py:18-54):
``reset() -> state``, ``step(action) -> [state, reward, done, {}]``, ``get_state()``,
with a leading batch axis on everything.  State lives in HBM as one int8
(batch, H, W, 4) buffer; nothing here computes a cell update on the CPU.
"""
import ctypes

import numpy as np

from . import _native as N
from .constants import METADATA as _DEFAULT_METADATA
from .utility import (get_agent_location, get_fire_location, get_name, grass, ignite_count,
                      layer)

_ACTION_CODES = {"N": 0, "S": 1, "E": 2, "W": 3, "D": 4}
NOOP = 5


def encode_action(a):
    """Reference action vocabulary (forest_fire.py:32-36): "N","S","E","W" or 0..3 move,
    "D" or 4 toggles digging (when allowed); anything else does nothing."""
    if isinstance(a, str):
        return _ACTION_CODES.get(a, NOOP)
    try:
        if a in range(5):
            return int(a)
    except TypeError:
        pass
    return NOOP


class BatchedForestFire:
    def __init__(self, batch, size=None, *, width=None, height=None, device=0, metadata=None,
                 seed=0, env_id0=0, auto_reset=False, fuel_range=None, ignite_ppm=0,
                 containment=None):
        if metadata is _DEFAULT_METADATA:
            # the module-global dict itself (the single-environment adapter with no arguments): shared by every such
            # object of the process, like the reference's METADATA (forest_fire.py:20, constants.py:30-57)
            self.METADATA = metadata
        else:
            self.METADATA = dict(_DEFAULT_METADATA)
            if metadata:
                self.METADATA.update(metadata)
                if "a_speed" in metadata and "a_speed_iter" not in metadata:
                    self.METADATA["a_speed_iter"] = self.METADATA["a_speed"]    # constants.py:42-43: both are A_SPEED
        if size is not None:
            width = height = size
        if width is not None:
            self.METADATA["width"] = int(width)
        if height is not None:
            self.METADATA["height"] = int(height)
        self.width, self.height = self.METADATA["width"], self.METADATA["height"]
        self.batch = int(batch)
        self.device = int(device)
        self.DEBUG = self.METADATA["debug"]
        self.layer = layer
        self.get_name = get_name
        self.n_actions = self.METADATA["n_actions"]
        self.seed = int(seed)
        self._reset_count = 0
        self._staging = None

        # World.reset, environment.py:187-193.  Wind vector (0,0): every heat addend is the same quantum and the rule
        # is an integer count (ignite_count).  A vector with a direction - fixed, or drawn per episode with "random" -
        # makes the addend depend on the direction burning cell -> neighbour: the state gains per-direction counts
        # (upload/download_counts) and the thread-per-cell kernels serve every shape.
        p = N.default_params()
        wind = self.METADATA["wind"]
        self.wind_random = wind == "random"
        if self.wind_random:
            p.wind_mode = N.WIND_RANDOM
            self.wind_speeds = (0, 0.7, 0.85)                                  # environment.py:189
            self.wind_speed, self.wind_vector = None, None                     # per environment: see winds()
        else:
            self.wind_speed, self.wind_vector = wind[0], tuple(int(v) for v in wind[1])
            self.wind_speeds = (self.wind_speed, 0.0, 0.0)
            if self.wind_vector != (0, 0):
                p.wind_mode = N.WIND_FIXED
                p.wind_code = N.wind_code(*self.wind_vector, 0)
        self.wind_mode = int(p.wind_mode)
        for i, v in enumerate(self.wind_speeds):
            p.wind_speed[i] = float(v)
        p.heat, p.threshold = float(grass["heat"]), float(grass["threshold"])
        if self.wind_mode == N.WIND_CALM:
            p.ignite_count = ignite_count(self.wind_speed)
        p.allow_dig_toggle = int(bool(self.METADATA["allow_dig_toggle"]))
        # containment test of get_reward: built for the one-wave grid family; None = "if available"
        p.containment = 1 if containment is None else int(bool(containment))
        p.auto_reset = int(bool(auto_reset))
        p.death_penalty = float(self.METADATA["death_penalty"])
        p.contained_bonus = float(self.METADATA["contained_bonus"])
        p.default_reward = float(self.METADATA["default_reward"])
        lo, hi = fuel_range if fuel_range is not None else (grass["fuel"], grass["fuel"])
        p.fuel_lo, p.fuel_hi = int(lo), int(hi)
        p.ignite_ppm = int(ignite_ppm)
        p.make_rivers = int(bool(self.METADATA["make_rivers"]))            # constants.py:44
        p.seed = self.seed
        p.env_id0 = int(env_id0)
        self.params = p

        self._lib = N.lib()
        h = ctypes.c_void_p()
        rc = self._lib.wf_create(ctypes.byref(h), self.device, self.batch, self.height,
                                 self.width, ctypes.byref(p))
        if rc == -2 and containment is None:      # WF_ERR_UNSUPPORTED: big-grid family, no containment yet
            p.containment = 0
            rc = self._lib.wf_create(ctypes.byref(h), self.device, self.batch, self.height,
                                     self.width, ctypes.byref(p))
        self.containment = bool(p.containment)
        if rc != N.WF_OK:
            raise N.WildfireError(rc, f"wf_create(batch={batch}, H={self.height}, W={self.width})")
        self._h = h
        self.kernel_name = self._lib.wf_kernel_name(self._h).decode()
        self.step_kernel = self._lib.wf_step_kernel_name(self._h).decode()      # behind a single step
        self.step_launches = int(self._lib.wf_step_launches(self._h))          # concurrent launches per step

    # ------------------------------------------------------------------ plumbing
    def _ck(self, rc, where):
        if rc != N.WF_OK:
            raise N.WildfireError(rc, where, self._lib.wf_last_hip_error(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.wf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        """Run on a caller-owned HIP stream (e.g. torch.cuda.current_stream().cuda_stream)."""
        self._ck(self._lib.wf_set_stream(self._h, ctypes.c_void_p(int(hip_stream))), "wf_set_stream")

    def sync(self):
        self._ck(self._lib.wf_sync(self._h), "wf_sync")

    def reload_env(self):
        """Experiments / tests: re-read the WF_* environment switches (they are cached at construction)."""
        self._ck(self._lib.wf_reload_env(self._h), "wf_reload_env")
        self.step_kernel = self._lib.wf_step_kernel_name(self._h).decode()
        self.step_launches = int(self._lib.wf_step_launches(self._h))

    @property
    def planes_ptr(self):
        return self._lib.wf_planes_device(self._h)

    @property
    def envs_ptr(self):
        return self._lib.wf_envs_device(self._h)

    # ------------------------------------------------------------- state exchange
    def upload(self, planes=None, envs=None, counts=None):
        """Replace device state.  planes int8 (batch,H,W,4); envs _native.ENV_DTYPE (batch,).
        When envs is omitted it is derived from the planes the way Agent.__init__ does
        (environment.py:103-109): the agent is where the agent_pos plane says.
        counts (wind with a direction only): uint8 (batch,H,W,4) heat addends received per direction; planes without
        counts start from zero heat (the old grid's counts are never kept)."""
        if counts is not None:
            counts = np.ascontiguousarray(counts, dtype=np.uint8)
            assert counts.shape == (self.batch, self.height, self.width, 4), counts.shape
        if planes is not None:
            planes = np.ascontiguousarray(planes, dtype=np.int8)
            assert planes.shape == (self.batch, self.height, self.width, 4), planes.shape
            if envs is None:
                envs = self.envs_from_planes(planes)
        if envs is not None:
            envs = np.ascontiguousarray(envs, dtype=N.ENV_DTYPE)
            assert envs.shape == (self.batch,)
        if planes is not None or envs is not None:
            self._ck(self._lib.wf_upload(self._h, N.ptr(planes), N.ptr(envs)), "wf_upload")
        # planes and counts form ONE state: wf_upload of planes zeroes the counts (a new grid has received no heat),
        # so the counts of a snapshot go up after its planes
        if counts is not None:
            self._ck(self._lib.wf_wind_counts_upload(self._h, N.ptr(counts)), "wf_wind_counts_upload")

    def download(self):
        planes = np.empty((self.batch, self.height, self.width, 4), dtype=np.int8)
        envs = np.empty(self.batch, dtype=N.ENV_DTYPE)
        self._ck(self._lib.wf_download(self._h, N.ptr(planes), N.ptr(envs)), "wf_download")
        return planes, envs

    def download_counts(self):
        """Wind with a direction: uint8 (batch,H,W,4), how many heat addends every grass cell has received from a
        burning cell below / above / left / right of it (the vector to it pointing N / S / E / W); the reference's
        temp layer is the sum of those addends (wind_addends())."""
        counts = np.empty((self.batch, self.height, self.width, 4), dtype=np.uint8)
        self._ck(self._lib.wf_wind_counts_download(self._h, N.ptr(counts)), "wf_wind_counts_download")
        return counts

    def winds(self, envs=None):
        """(speed float64 (batch,), vector int (batch, 2)) = World.wind_speed / wind_vector of every environment."""
        if self.wind_mode == N.WIND_CALM:
            return np.full(self.batch, float(self.wind_speed)), np.zeros((self.batch, 2), dtype=int)
        code = (self.download_envs() if envs is None else envs)["wind"].astype(int)
        speed = np.asarray(self.wind_speeds + (0.0,) * (4 - len(self.wind_speeds)))[(code >> 4) & 3]
        return speed, np.stack([(code & 3) - 1, ((code >> 2) & 3) - 1], axis=1)

    def wind_addends(self, envs=None):
        """float64 (batch, 4): the heat a burning cell hands to its neighbour in direction N, S, E, W per tick
        (apply_heat_from_to, environment.py:278-287)."""
        import math
        speed, vec = self.winds(envs)
        # the factor (angle + distance)**-1 of every wind vector and direction: 9 x 4 values, then a table look-up
        factor = np.empty((3, 3, 4))
        for wx in (-1, 0, 1):
            for wy in (-1, 0, 1):
                for k, (cx, cy) in enumerate(((0, -1), (0, 1), (1, 0), (-1, 0))):
                    angle = abs(math.atan2(wx * cy - wy * cx, wx * cx + wy * cy))
                    factor[wx + 1, wy + 1, k] = (angle + 1) ** (-1)
        return (speed * grass["heat"])[:, None] * factor[vec[:, 0] + 1, vec[:, 1] + 1]

    def download_envs(self):
        """Only the per-environment records (16 B each): agent position, flags, episode, tick."""
        envs = np.empty(self.batch, dtype=N.ENV_DTYPE)
        self._ck(self._lib.wf_download(self._h, None, N.ptr(envs)), "wf_download")
        return envs

    def envs_from_planes(self, planes):
        envs = np.zeros(self.batch, dtype=N.ENV_DTYPE)
        ag = planes[..., N.P_AGENT] != 0
        has = ag.reshape(self.batch, -1).any(axis=1)
        idx = ag.reshape(self.batch, -1).argmax(axis=1)
        envs["x"] = np.where(has, idx % self.width, 0)
        envs["y"] = np.where(has, idx // self.width, 0)
        fire = ((planes[..., N.P_TYPE] & 3) == 1).reshape(self.batch, -1).any(axis=1)     # burning_cells non-empty
        envs["flags"] = (has * (N.F_ALIVE | N.F_VISIBLE | N.F_DIGGING)) | ((has & fire) * N.F_RUNNING)
        return envs

    # ----------------------------------------------------------------------- reset
    def reset(self, mask=None, seed=None):
        """World.reset() (environment.py:186-212) on the device for every env (or the
        masked ones), spawn points from the counter-based stream.  Returns the state."""
        if seed is None:
            seed = self.seed + 0x9E3779B9 * self._reset_count
        self._reset_count += 1
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
            assert mask.shape == (self.batch,)
        self._ck(self._lib.wf_reset(self._h, N.ptr(mask), ctypes.c_uint64(int(seed) & (2**64 - 1))),
                 "wf_reset")
        # (METADATA['a_speed_iter'] is NOT re-armed: the reference's reset() never touches it, forest_fire.py:52-54 -
        # with a_speed > 1 an episode that ends mid-cycle leaves the next one out of phase, there as here)
        return self.get_state()

    def reset_host(self, spawns=None, rng=np.random, pyrng=None):
        """World.reset() built on the host and uploaded: spawn points either given
        ((batch,2) x,y) or drawn from numpy's legacy stream in the reference's own
        order (utility.py:66-78), so a seeded run spawns where the reference does.
        With wind "random" the speed comes from `rng` and the vector from `pyrng` (Python's random module by
        default), first of all draws, as in World.reset (environment.py:188-190)."""
        if pyrng is None:
            import random as pyrng
        B, H, W = self.batch, self.height, self.width
        planes = np.zeros((B, H, W, 4), dtype=np.int8)
        planes[..., N.P_FUEL] = grass["fuel"]                              # reset_map :63
        fx, fy = get_fire_location(W, H)
        planes[:, fy, fx, N.P_TYPE] = N.FIRE                               # set_fire_to :233-246
        planes[:, fy, fx, N.P_HEAT] = N.HEAT_FORCED
        envs = np.zeros(B, dtype=N.ENV_DTYPE)
        if self.wind_mode == N.WIND_FIXED:
            envs["wind"] = N.wind_code(*self.wind_vector, 0)
        for b in range(B):
            if self.wind_random:
                si = list(self.wind_speeds).index(rng.choice(list(self.wind_speeds)))
                envs[b]["wind"] = N.wind_code(pyrng.randint(-1, 1), pyrng.randint(-1, 1), si)
            if self.METADATA["make_rivers"]:
                self._make_river(planes[b], rng)          # drawn BEFORE the spawn point, as reset_map does
                # set_fire_to retypes the centre afterwards; under a river its fire_mobility stays inf (:85, :241)
                planes[b, fy, fx, N.P_TYPE] = N.FIRE_BLOCKED if planes[b, fy, fx, N.P_TYPE] == N.WATER else N.FIRE
            sx, sy = spawns[b] if spawns is not None else get_agent_location(W, H, rng)
            planes[b, sy, sx, N.P_AGENT] = 1                               # Agent.__init__ :108-113
            planes[b, sy, sx, N.P_TYPE] = N.DIRT
            envs[b]["x"], envs[b]["y"] = sx, sy
        envs["flags"] = N.F_ALIVE | N.F_VISIBLE | N.F_DIGGING | N.F_RUNNING
        self.upload(planes, envs, np.zeros(planes.shape, np.uint8) if self.wind_mode != N.WIND_CALM else None)
        return self.get_state()               # (a_speed_iter survives a reset, as in the reference: see reset())

    def _make_river(self, p, rng):
        """The reference's river generator (reset_map, environment.py:69-95), draw for draw on
        numpy's legacy stream: a water cell per row from a start near the top, drifting one
        column left or right.  p: int8 (H, W, 4) planes of one environment."""
        W, H = self.width, self.height
        d = [1, 2, 3]
        fx, fy = get_fire_location(W, H)
        river_x = rng.choice(list(range(W)))
        river_y = rng.choice(d)
        while river_y < (H - rng.choice(d)):
            p[river_y, river_x, N.P_TYPE] = N.WATER                       # :81 (numpy wraps a -1 index; so do we)
            new_y = river_y + 1
            new_x = river_x + rng.choice([1, -1])
            # `not a <= x < b and not at_fire`: a chained comparison draws `b` only when a <= x holds
            while True:
                ok = rng.choice(d) <= new_x
                if ok:
                    ok = new_x < (W - rng.choice(d))
                if ok or (new_x, new_y) == (fx, fy):
                    break
                new_x = river_x + rng.choice([1, -1])
            river_x, river_y = new_x, new_y

    # ------------------------------------------------------------------------ step
    def _update_mask(self, n_steps):
        m = 0
        md = self.METADATA
        for s in range(n_steps):                                           # forest_fire.py:40-43
            md["a_speed_iter"] -= 1
            if md["a_speed_iter"] == 0:
                m |= 1 << s
                md["a_speed_iter"] = md["a_speed"]
        return m

    def host_staging(self, n_steps=1):
        """(actions int8, reward float64, done uint8) numpy VIEWS, each (n_steps, batch), of the pinned host buffers the
        library lends (wf_host_staging): step_many(..., borrow=True) works on them in place, without host copies."""
        n_steps = int(n_steps)
        if self._staging is None or self._staging[0] != n_steps:
            pa, pr, pd = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
            self._ck(self._lib.wf_host_staging(self._h, n_steps, ctypes.byref(pa), ctypes.byref(pr), ctypes.byref(pd)), "wf_host_staging")
            shape = (n_steps, self.batch)
            view = lambda p, ct, dt: np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ct)), shape=shape).view(dt)
            self._staging = (n_steps, view(pa, ctypes.c_int8, np.int8), view(pr, ctypes.c_double, np.float64), view(pd, ctypes.c_uint8, np.uint8))
        return self._staging[1:]

    def step_many(self, actions=None, n_steps=None, borrow=False, update_mask=None):
        """n_steps ForestFire.step() calls in one API call (small batches: ONE kernel launch, the grid stays in
        registers between ticks).  actions int8 (n_steps, batch) or None for no-ops.
        Returns (reward float64 (n_steps,batch), done bool (n_steps,batch)).
        borrow=True: reward and done (uint8) are views of the library's pinned buffers, valid until the next step call -
        no host copies (at 65 536 environments they cost as much as the step); actions may be the lent action view."""
        if actions is not None:
            actions = np.ascontiguousarray(actions, dtype=np.int8)
            if actions.ndim == 1:
                actions = actions[None, :]
            assert actions.shape[1] == self.batch
            n_steps = actions.shape[0]
        n_steps = int(n_steps or 1)
        mask = self._update_mask(n_steps) if update_mask is None else int(update_mask)
        if self._staging is not None and n_steps > self._staging[0]:
            self._staging = None                      # a longer call re-allocates the lent buffers: the old views dangle
        if borrow:
            _, reward, done = self.host_staging(n_steps)
            self._ck(self._lib.wf_step(self._h, N.ptr(actions), n_steps, ctypes.c_uint64(mask), N.ptr(reward), N.ptr(done)), "wf_step")
            return reward, done
        reward = np.empty((n_steps, self.batch), dtype=np.float64)
        done = np.empty((n_steps, self.batch), dtype=np.uint8)
        self._ck(self._lib.wf_step(self._h, N.ptr(actions), n_steps, ctypes.c_uint64(mask),
                                   N.ptr(reward), N.ptr(done)), "wf_step")
        return reward, done.astype(bool)

    def step_download(self, actions=None):
        """One step and the resulting state under ONE synchronisation (wf_step_download):
        (reward float64 (batch,), done bool (batch,), planes int8 (batch,H,W,4), envs ENV_DTYPE (batch,)).
        For host-side views of small contexts (the single-environment adapter); large batches belong on the device API."""
        if actions is not None:
            actions = np.ascontiguousarray(actions, dtype=np.int8)
            assert actions.shape == (self.batch,)
        reward = np.empty(self.batch, dtype=np.float64)
        done = np.empty(self.batch, dtype=np.uint8)
        planes = np.empty((self.batch, self.height, self.width, 4), dtype=np.int8)
        envs = np.empty(self.batch, dtype=N.ENV_DTYPE)
        self._ck(self._lib.wf_step_download(self._h, N.ptr(actions), 1, ctypes.c_uint64(self._update_mask(1)),
                                            N.ptr(reward), N.ptr(done), N.ptr(planes), N.ptr(envs)), "wf_step_download")
        return reward, done.astype(bool), planes, envs

    def step_download_reuse(self, code):
        """step_download() for per-step host loops on SMALL contexts (the single-environment adapter): the same action code
        for every environment, and the outputs land in buffers the object keeps - (reward, done uint8, planes, envs) are
        VIEWS that the next call overwrites.  Spares five pointer conversions and five allocations per step (about 9 of the
        adapter's 32 us)."""
        sd = getattr(self, "_sd", None)
        if sd is None:
            a = np.empty(self.batch, dtype=np.int8)
            r = np.empty(self.batch, dtype=np.float64)
            d = np.empty(self.batch, dtype=np.uint8)
            p = np.empty((self.batch, self.height, self.width, 4), dtype=np.int8)
            e = np.empty(self.batch, dtype=N.ENV_DTYPE)
            sd = self._sd = (a, r, d, p, e, N.ptr(a), N.ptr(r), N.ptr(d), N.ptr(p), N.ptr(e), ctypes.c_uint64(0))
        a, r, d, p, e, pa, pr, pd, pp, pe, mask = sd
        a.fill(code)
        mask.value = self._update_mask(1)
        self._ck(self._lib.wf_step_download(self._h, pa, 1, mask, pr, pd, pp, pe), "wf_step_download")
        return r, d, p, e

    def step(self, actions=None, update_mask=None):
        """[state (batch,W,H,3) int8, reward float64 (batch,), done bool (batch,), {}]
        (forest_fire.py:30-49 with a batch axis).  update_mask: bit 0 = run update() in this step (default: the a_speed
        counter of METADATA decides, forest_fire.py:40-43)."""
        if actions is not None and not isinstance(actions, np.ndarray):
            actions = np.array([encode_action(a) for a in actions], dtype=np.int8)
        if actions is not None:
            actions = np.ascontiguousarray(actions, dtype=np.int8)
            assert actions.shape == (self.batch,)
        reward = np.empty(self.batch, dtype=np.float64)
        done = np.empty(self.batch, dtype=np.uint8)
        obs = np.empty((self.batch, self.width, self.height, 3), dtype=np.int8)
        # one launch: the observation is written by the step kernel itself
        mask = self._update_mask(1) if update_mask is None else int(update_mask)
        self._ck(self._lib.wf_step_obs(self._h, N.ptr(actions), 1, ctypes.c_uint64(mask),
                                       N.ptr(reward), N.ptr(done), N.ptr(obs)), "wf_step_obs")
        return [obs, reward, done.astype(bool), {}]

    def step_device(self, d_actions=None, n_steps=1, d_reward=None, d_done=None, update_mask=None, d_obs=None):
        """Asynchronous step on device buffers (torch tensors or raw pointers); with d_obs the
        observation of the resulting state is written by the same launch."""
        if update_mask is None:
            update_mask = self._update_mask(n_steps)
        self._ck(self._lib.wf_step_obs_device(self._h, N.ptr(d_actions), int(n_steps),
                                              ctypes.c_uint64(update_mask), N.ptr(d_reward), N.ptr(d_done),
                                              N.ptr(d_obs)), "wf_step_obs_device")

    # ----------------------------------------------------------------- observation
    def get_state(self):
        """World.get_state() (environment.py:399-402): int8 (batch, W, H, 3), [b, x, y, k]."""
        obs = np.empty((self.batch, self.width, self.height, 3), dtype=np.int8)
        self._ck(self._lib.wf_get_state(self._h, N.ptr(obs)), "wf_get_state")
        return obs

    def get_state_device(self, d_obs):
        self._ck(self._lib.wf_get_state_device(self._h, N.ptr(d_obs)), "wf_get_state_device")

    # ---------------------------------------------------------------------- timing
    def timer_start(self):
        self._ck(self._lib.wf_timer_start(self._h), "wf_timer_start")

    def timer_stop(self):
        ms = ctypes.c_float()
        self._ck(self._lib.wf_timer_stop(self._h, ctypes.byref(ms)), "wf_timer_stop")
        return float(ms.value)
