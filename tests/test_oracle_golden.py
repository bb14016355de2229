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
        reward, done = ep.oracle_step(t, planes, envs, op, counts=counts)
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
    if "river" in path:
        assert (ep.d["planes"][..., 0] == 13).any(), "river fixtures must hold a river through the fire cell (code 13)"
    if "aspeed2" in path:
        assert len(set(ep.a_speed_iter0.tolist())) == 2, "episodes must start in both phases of the a_speed counter"


def _episode_rows(ep, e):
    d = ep.d
    lo, hi = int(ep.ep_start[e]), int(ep.ep_start[e + 1])
    ps, pc = d["pop_start"], d["pop_cells"]
    orders = [pc[ps[i]:ps[i + 1]] for i in range(lo, hi)]
    bl = d["border_len"][lo:hi]
    prev = np.concatenate([[d["init_world"][e, 1]], bl[:-1]])
    return d["planes"][lo:hi, :, :, 0], d["fire_at_border"][lo:hi], d["init_world"][e, 1], orders, bl, (bl == 0) & (prev > 0)


@pytest.mark.parametrize("path", fixture_files(), ids=lambda p: p.split("/")[-1])
def test_reward_model_with_the_recorded_pop_order_is_the_reference(path):
    """tests/ref_reward_model.py restates get_reward's containment search (environment.py:345-377) with the pop order of
    the burning cells as an input.  Fed with the order the reference actually used (recorded from its astar_path
    calls), it must reproduce len(border_points) and every containment payment of every step of every fixture -
    split fires included.  This pins the model the split-fire expectations below are derived from."""
    from tests.ref_reward_model import replay_episode
    ep = Episodes(path)
    for e in range(ep.B):
        types, fab, L0, orders, bl, ref_c = _episode_rows(ep, e)
        c, L, _, _ = replay_episode(types, fab, L0, orders)
        np.testing.assert_array_equal(L, bl, err_msg=f"{ep.name} episode {e}: len(border_points)")
        np.testing.assert_array_equal(c, ref_c, err_msg=f"{ep.name} episode {e}: containment paid")


# Steps of the split-fire fixtures where the reference's record differs from the build's rule, (episode, first step,
# number of steps with a different deque length, steps with a different reward).  Episodes 5-7: a dirt wall from border
# to border with a fire on each side - the reference's deque shrinks to what the cell it happens to pop first reaches.
# Episodes 8-9 (and 10-11 at 14x14 for the deque): the contrived state of gen_golden.py 9d - the reference pops the
# chamber's cell first, keeps 3 border points, and later pays the containment bonus for a fire that reaches the border.
SPLIT_DIVERGENCES = {
    "ref_10_splitfire.npz": {6: (9, 97, []), 7: (0, 14, []), 8: (0, 60, [5]), 9: (0, 60, [6])},
    "ref_14_splitfire.npz": {5: (0, 209, []), 7: (0, 90, []), 8: (0, 60, [5]), 9: (0, 60, [6]), 10: (0, 60, []), 11: (0, 60, [])},
}


@pytest.mark.parametrize("path", fixture_files(), ids=lambda p: p.split("/")[-1])
def test_build_rule_is_the_reference_under_sigma_star(path):
    """The build's containment rule ("no burning cell reaches a passable border cell") is the reference's search under
    the pop order SIGMA_STAR.  It must equal the reference's RECORD on every step of every fixture, except from a
    step on where the fire is split over several regions of passable cells - there the reference's result depends on
    CPython's set order.  Those steps are enumerated (SPLIT_DIVERGENCES) and each episode's first one is checked to be
    a split-fire step."""
    from tests.ref_reward_model import replay_episode
    ep = Episodes(path)
    found = {}
    for e in range(ep.B):
        types, fab, L0, _, bl, ref_c = _episode_rows(ep, e)
        c, L, split, _ = replay_episode(types, fab, L0)
        dl = np.nonzero(np.asarray(L) != bl)[0]
        dc = np.nonzero(np.asarray(c) != ref_c)[0]
        if len(dl) or len(dc):
            first = int(min(dl.tolist() + dc.tolist()))
            assert split[first], f"{ep.name} episode {e} step {first}: differs from the reference although the fire is in one region"
            found[e] = (first, int(len(dl)), [int(v) for v in dc])
    assert found == SPLIT_DIVERGENCES.get(ep.name, {}), (ep.name, found)
    # what Episodes.check_step expects is exactly that: the reference's record, the model's result on those steps
    assert sorted({(e, t) for e, t, kind in ep.divergent if kind == "reward"}) == \
        sorted((e, t) for e, v in found.items() for t in v[2])


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
    ep = Episodes([f for f in fixture_files(10) if f.endswith("ref_10_circle.npz")][0])
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


def test_search_model_reaches_exactly_what_the_reference_astar_finds():
    """oracle/_ref/astar.so is the reference's OWN native helper, compiled from pyastar/astar.cpp where it lies
    (oracle/Makefile).  On random maps of passable / blocked cells, `SearchState.reaches` (tests/ref_reward_model.py) must say
    "route" exactly when the reference's A* does - blocked start cells (their weight is never charged), blocked goals,
    start == goal and all."""
    import ctypes
    import os
    from tests.ref_reward_model import SearchState
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "astar.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/astar.so not built (needs /root/reference)")
    lib = ctypes.cdll.LoadLibrary(so)
    f32 = np.ctypeslib.ndpointer(dtype=np.float32, ndim=1, flags="C_CONTIGUOUS")
    i32 = np.ctypeslib.ndpointer(dtype=np.int32, ndim=1, flags="C_CONTIGUOUS")
    lib.astar.restype = ctypes.c_bool
    lib.astar.argtypes = [f32, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_bool, i32]
    rng = np.random.default_rng(7)
    checked = routes = 0
    for size in (10, 14):
        for _ in range(60):
            blocked = rng.random((size, size)) < rng.choice([0.2, 0.35, 0.5])         # [y, x]
            types = np.where(blocked, 3, 0)
            st = SearchState(types)
            # the reference hands astar the fire_mobility layer indexed [x, y] (environment.py:353-356)
            weights = np.where(blocked.T, np.inf, 1.0).astype(np.float32)              # [x, y]
            for _ in range(40):
                sx, sy, gx, gy = (int(v) for v in rng.integers(0, size, size=4))
                paths = np.full(size * size, -1, dtype=np.int32)
                found = bool(lib.astar(weights.ravel(), size, size, sx * size + sy, gx * size + gy, False, paths))
                found = found and (sx, sy) != (gx, gy)            # pyastar.py:52-64: an empty path for start == goal
                assert st.reaches((sx, sy), (gx, gy)) == found, (size, sx, sy, gx, gy)
                checked += 1
                routes += found
    assert checked == 4800 and 500 < routes < 4300
