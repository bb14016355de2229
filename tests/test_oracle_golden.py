"""Pin the CPU oracle (oracle/wf_oracle.c) against the reference's own outputs.

The fixtures were produced by running dashdeckers/Wildfire-Control-Python
unchanged (tests/golden/gen_golden.py): every plane, agent scalar, reward, done
flag and observation of every step must be reproduced bit-exactly."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.golden_util import Episodes, fixture_files


@pytest.mark.parametrize("path", fixture_files(), ids=lambda p: p.split("/")[-1])
def test_oracle_reproduces_reference(path):
    ep = Episodes(path)
    planes = ep.init_planes.copy()
    envs = ep.init_envs.copy()
    n_contained = n_burning_dirt = 0
    op = ep.oracle_params()                  # toggle / wind as the generator set them (constants.py:39,46)
    counts = np.zeros(planes.shape, np.uint8) if ep.wind else None
    for t in range(ep.T):
        actions, _ = ep.actions_at(t)
        reward, done = O.step(planes, envs, actions, op, counts=counts)
        obs = O.get_state(planes)
        live = ep.check_step(t, planes, envs, reward, done, obs)
        if ep.wind:
            ep.check_temp(t, planes, counts)
        n_contained += int((reward[live] == 1000.0).sum())
        n_burning_dirt += int((planes[live][..., 0] == 5).sum())
    if "circle" in path:
        assert n_contained > 0, "circle fixtures must exercise the containment reward"
    if "toggle" in path:
        assert n_burning_dirt > 20, "toggle fixtures must hold dirt cells that keep burning (dig on a burning cell)"


def test_known_answer_noop_10():
    """Known-answer episode (SURVEY 8a rule): centre burns 20 ticks, 4-neighbours
    ignite on the 19th tick, every later ring follows from the >= 19 quanta rule."""
    ep = Episodes(fixture_files(10)[[f.endswith("ref_10_noop.npz") for f in fixture_files(10)].index(True)])
    p = ep.d["planes"]
    assert (p[18, 4, 5, 0], p[18, 5, 4, 0], p[18, 6, 5, 0], p[18, 5, 6, 0]) == (1, 1, 1, 1)
    assert p[17, 4, 5, 0] == 0 and p[17, 4, 5, 2] == 18
    assert p[19, 5, 5, 0] == 2          # centre burnt out after 20 ticks of fuel
    assert ep.d["reward"][-1] == 0.0 and ep.d["done"][-1] == 1


def test_frozen_after_done():
    ep = Episodes(fixture_files(10)[0])
    planes, envs = ep.init_planes.copy(), ep.init_envs.copy()
    for t in range(ep.T):
        O.step(planes, envs, ep.actions_at(t)[0])
    assert not (envs["flags"] & O.F_RUNNING).any()
    before = planes.copy()
    r, d = O.step(planes, envs, np.zeros(ep.B, dtype=np.int8))
    np.testing.assert_array_equal(planes, before)
    assert (r == 0).all() and (d == 1).all()


def test_circle_points_match_reference_counts():
    # SURVEY 8a [probed]: r=1 -> 8 points incl. diagonals, r=2 -> 12, r=3 -> 16, no duplicates
    for r, n in ((1, 8), (2, 12), (3, 16)):
        pts = O.circle_points(r)
        assert len(pts) == n and len(set(pts)) == n
    assert O.circle_points(1)[:4] == [(1, 0), (-1, 0), (0, -1), (0, 1)]


def test_reset_default_law():
    planes, envs = O.new_state(64, 14, 14)
    O.reset(planes, envs, seed=7)
    assert (planes[:, 7, 7, 0] == 1).all() and (planes[:, 7, 7, 2] == O.HEAT_FORCED).all()
    assert (planes[..., 1] == 20).all()
    assert ((planes[..., 0] == 3).sum(axis=(1, 2)) == 1).all()      # spawn cell dug
    assert ((planes[..., 3] == 1).sum(axis=(1, 2)) == 1).all()
    y, x = envs["y"].astype(int), envs["x"].astype(int)
    assert (planes[np.arange(64), y, x, 0] == 3).all() and (planes[np.arange(64), y, x, 3] == 1).all()
    d2 = (x - 7) ** 2 + (y - 7) ** 2
    assert d2.min() >= 1 and d2.max() <= 10 and len(set(zip(x, y))) > 10
    assert (envs["episode"] == 1).all() and (envs["flags"] == (1 | 2 | 4 | 16)).all()
    # deterministic and env-id keyed: shard [32:64) alone reproduces the same envs
    p2, e2 = O.new_state(32, 14, 14)
    O.reset(p2, e2, seed=7, env_id0=32)
    np.testing.assert_array_equal(p2, planes[32:])
    np.testing.assert_array_equal(e2, envs[32:])


@pytest.mark.parametrize("size", [10, 14])
def test_oracle_reproduces_the_published_baseline_reward(size):
    """The reference publishes the mean episode reward of its clockwise random-walk baseline policy
    (DQN.py:286-389): 1130.2 at 10x10 and 1151.4 at 14x14 over 10 x 10 000 episodes
    (Plots/results_first.txt:2,20).  The oracle under the same policy (tests/policy_util.py), default
    fuel 20, containment on, reset law of wfo_reset, must land on it within 4 standard errors."""
    from tests.policy_util import (REFERENCE_BASELINE_MEAN, baseline_tolerance, run_baseline_episodes)
    B = 40960
    op = O.default_params(containment=1)
    planes, envs = O.new_state(B, size, size)
    rng = np.random.default_rng(size)
    total, died, steps = run_baseline_episodes(
        lambda: O.reset(planes, envs, seed=2024 + size, params=op),
        lambda a: (lambda r, d: (r, d.astype(bool)))(*O.step(planes, envs, a, op)),
        lambda: (envs["x"], envs["y"], (envs["flags"] & O.F_ALIVE) != 0, planes[..., 0] == 1),
        B, rng)
    tol = baseline_tolerance(total.std(), B)
    assert abs(total.mean() - REFERENCE_BASELINE_MEAN[size]) < tol, (total.mean(), tol)
    assert 0.01 < died.mean() < 0.15 and 60 < steps.mean() < 200
