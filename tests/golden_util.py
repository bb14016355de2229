"""Helpers shared by the parity tests: load the reference-generated fixtures
(tests/golden/ref_*.npz, produced by tests/golden/gen_golden.py from the real
reference) and lay their episodes out as a batch of environments."""
import glob
import os

import numpy as np

from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
WIND_SPEEDS = [0.0, 0.7, 0.85]        # environment.py:189


def fixture_files(size=None):
    pat = f"ref_{size}_*.npz" if size else "ref_[0-9]*_*.npz"      # episode fixtures (not ref_reset_stats_*.npz)
    return sorted(glob.glob(os.path.join(GOLDEN, pat)))


class Episodes:
    """All episodes of one fixture file as a batch: env b = episode b."""

    def __init__(self, path):
        d = np.load(path)
        self.name = os.path.basename(path)
        self.size = int(d["size"])
        self.ep_start = d["ep_start"]
        self.B = len(self.ep_start) - 1
        self.lengths = np.diff(self.ep_start)
        self.T = int(self.lengths.max())
        self.d = d
        self.init_planes = np.ascontiguousarray(d["init_planes"])
        ia = d["init_agent"]
        e = np.zeros(self.B, dtype=O.ENV_DTYPE)
        e["x"], e["y"] = ia[:, 0], ia[:, 1]
        iw = d["init_world"]
        e["flags"] = (O.F_ALIVE | O.F_VISIBLE | O.F_DIGGING | O.F_RUNNING) \
            | (iw[:, 0] != 0) * O.F_BORDER | (iw[:, 1] == 0) * O.F_CONTAINED
        # wind fixtures (METADATA['wind'] = "random"): the episode's wind as the env's wind code, speed index into
        # WIND_SPEEDS; the per-direction heat counts start at zero (a fresh grid has received no heat)
        self.wind = "wind" in d.files
        if self.wind:
            w = d["wind"]
            si = np.array([WIND_SPEEDS.index(float(v)) for v in w[:, 0]])
            e["wind"] = (w[:, 1].astype(int) + 1) | (w[:, 2].astype(int) + 1) << 2 | si << 4
        self.init_envs = e
        # a_speed fixtures: update() runs only in the steps the reference's global counter says (forest_fire.py:40-43)
        self.a_speed_iter0 = d["a_speed_iter0"] if "a_speed_iter0" in d.files else None
        self._sigma_star()

    def _sigma_star(self):
        """Expected containment results.  The reference's search depends on CPython's set order once the fire is split
        over several regions of passable cells (tests/ref_reward_model.py); the build follows the order SIGMA_STAR.
        For the split-fire fixtures the expected `contained` flag and reward are therefore the model's under
        SIGMA_STAR; the steps where that differs from what the reference recorded are listed in self.divergent
        (and enumerated by tests/test_oracle_golden.py).  Every other fixture: the reference's record, untouched."""
        d = self.d
        self.exp_reward = d["reward"].copy()
        self.exp_contained = d["border_len"] == 0
        self.divergent = []            # (episode, step, kind) kind: "deque" (only len(border_points) differs) / "reward"
        if "splitfire" not in self.name:
            return
        from tests.ref_reward_model import replay_episode
        cells = self.size * self.size
        for e in range(self.B):
            lo, hi = int(self.ep_start[e]), int(self.ep_start[e + 1])
            types = d["planes"][lo:hi, :, :, 0]
            c, L, split, ran = replay_episode(types, d["fire_at_border"][lo:hi], d["init_world"][e, 1])
            bl = d["border_len"][lo:hi]
            prev = np.concatenate([[d["init_world"][e, 1]], bl[:-1]])
            ref_c = (bl == 0) & (prev > 0)
            for t in range(hi - lo):
                self.exp_contained[lo + t] = L[t] == 0
                if bool(c[t]) != bool(ref_c[t]):
                    if c[t]:
                        r = 1000.0                                        # environment.py:367
                    elif not d["agent"][lo + t, 2]:
                        r = -1000.0                                       # :380-381
                    elif not ((types[t] & 3) == 1).any():
                        r = 1000.0 * (int((types[t] == 0).sum()) / cells)  # :384-387
                    else:
                        r = -1.0                                          # :390
                    self.exp_reward[lo + t] = r
                    self.divergent.append((e, t, "reward"))
                elif L[t] != bl[t]:
                    self.divergent.append((e, t, "deque"))

    def oracle_params(self, **kw):
        """Oracle parameters as the generator configured the reference for this file."""
        if "toggle" in self.name:
            kw.setdefault("allow_dig_toggle", 1)           # constants.py:46
        if "river" in self.name:
            kw.setdefault("make_rivers", 1)                # constants.py:44
        if self.wind:
            kw.update(wind_mode=1, wind_speeds=WIND_SPEEDS + [0.0])
        return O.default_params(**kw)

    def wind_addends(self):
        """float64 (B, 4): the heat addend per direction k of every episode, computed like
        environment.py:260-287 (direction k = vector from the burning cell to the heated one: N, S, E, W)."""
        import math
        out = np.zeros((self.B, 4))
        for b, (sp, wx, wy) in enumerate(self.d["wind"]):
            for k, (cx, cy) in enumerate(((0, -1), (0, 1), (1, 0), (-1, 0))):
                angle = abs(math.atan2(wx * cy - wy * cx, wx * cx + wy * cy))
                out[b, k] = sp * 0.3 * (angle + 1) ** (-1)
        return out

    def check_temp(self, t, planes, counts, what="oracle"):
        """Wind fixtures: the recorded float temperature of every GRASS cell is the sum of its counted addends."""
        exp, live = self.expected_at(t, "temp")
        got = (counts[live].astype(np.float64) * self.wind_addends()[live][:, None, None, :]).sum(axis=-1)
        grass = planes[live][..., 0] == 0
        np.testing.assert_allclose(got[grass], exp[grass], rtol=0, atol=2e-6, err_msg=f"{self.name} step {t} ({what}) temp")
        assert not counts[live][~grass].any(), "counts of cells that are not grass are zero"

    def actions_at(self, t):
        """int8 (B,) actions of step t (no-op 5 where the episode is over)."""
        a = np.full(self.B, 5, dtype=np.int8)
        live = self.lengths > t
        a[live] = self.d["action"][self.ep_start[:-1][live] + t]
        return a, live

    def updates_at(self, t):
        """bool (B,): update() runs in step t of the episode (all True unless the fixture plays with a_speed)."""
        u = np.ones(self.B, dtype=bool)
        live = self.lengths > t
        u[live] = self.d["updated"][self.ep_start[:-1][live] + t] != 0
        return u

    def oracle_step(self, t, planes, envs, op, counts=None):
        """One oracle step of every episode, update() only where the fixture's a_speed counter ran it."""
        actions, _ = self.actions_at(t)
        upd = self.updates_at(t)
        if upd.all():
            return O.step(planes, envs, actions, op, counts=counts)
        reward, done = np.zeros(self.B), np.zeros(self.B, dtype=np.uint8)
        for flag in (True, False):
            idx = np.nonzero(upd == flag)[0]
            if len(idx) == 0:
                continue
            p, e = np.ascontiguousarray(planes[idx]), envs[idx].copy()
            cn = None if counts is None else np.ascontiguousarray(counts[idx])
            r, dn = O.step(p, e, actions[idx], op, do_update=flag, counts=cn)
            planes[idx], envs[idx], reward[idx], done[idx] = p, e, r, dn
            if counts is not None:
                counts[idx] = cn
        return reward, done

    def expected_at(self, t, key):
        live = self.lengths > t
        return self.d[key][self.ep_start[:-1][live] + t], live

    def check_step(self, t, planes, envs, reward, done, obs=None, what="oracle"):
        """Compare one batched step against the reference rows; returns live mask."""
        tag = f"{self.name} step {t} ({what})"
        exp_planes, live = self.expected_at(t, "planes")
        np.testing.assert_array_equal(planes[live], exp_planes, err_msg=tag + " planes")
        agent, _ = self.expected_at(t, "agent")
        alive = (envs["flags"][live] & O.F_ALIVE) != 0
        np.testing.assert_array_equal(alive, agent[:, 2] != 0, err_msg=tag + " alive")
        np.testing.assert_array_equal(envs["x"][live][alive], agent[alive, 0], err_msg=tag + " x")
        np.testing.assert_array_equal(envs["y"][live][alive], agent[alive, 1], err_msg=tag + " y")
        exp_r = self.exp_reward[self.ep_start[:-1][live] + t]
        np.testing.assert_array_equal(reward[live], exp_r, err_msg=tag + " reward")
        exp_d, _ = self.expected_at(t, "done")
        np.testing.assert_array_equal(done[live], exp_d, err_msg=tag + " done")
        fab, _ = self.expected_at(t, "fire_at_border")
        np.testing.assert_array_equal((envs["flags"][live] & O.F_BORDER) != 0, fab != 0,
                                      err_msg=tag + " fire_at_border")
        np.testing.assert_array_equal((envs["flags"][live] & O.F_CONTAINED) != 0,
                                      self.exp_contained[self.ep_start[:-1][live] + t], err_msg=tag + " contained")
        if obs is not None:
            exp_o, _ = self.expected_at(t, "obs")
            np.testing.assert_array_equal(obs[live], exp_o, err_msg=tag + " obs")
        return live


def build_native_example(tmp_path):
    """gcc examples/native_step.c (plain C client of include/wildfire.h) against the in-tree library."""
    import os
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "native_step")
    libdir = os.path.join(repo, "wildfire_control_python_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(repo, "include"),
                           os.path.join(repo, "examples", "native_step.c"), "-o", exe,
                           "-L" + libdir, "-lwildfire_hip", "-Wl,-rpath," + libdir])
    return exe
