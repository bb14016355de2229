"""The random parts of World.reset() (environment.py:186-212) in DISTRIBUTION.

The device reset draws from a counter-based stream, not numpy's legacy generator, so against the reference it
can only agree in law.  tests/golden/ref_reset_stats_<size>.npz hold histograms sampled from the REAL reference
(tests/golden/gen_reset_stats.py): the river of reset_map (environment.py:69-95) and the two-stage spawn draw of
get_agent_location (utility.py:66-78: radius uniform in {1,2,3}, then a uniform point of that circle).  The
build's generators (oracle here, HIP kernels under -m gpu; the two are compared bit for bit elsewhere) must
pass a two-sample chi-square test against them.
"""
import os

import numpy as np
import pytest
from scipy.stats import chi2_contingency

from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
P_MIN = 1e-4          # seeded samples: a deterministic verdict, not a flaky one


def ref_stats(size):
    return np.load(os.path.join(GOLDEN, f"ref_reset_stats_{size}.npz"))


def same_law(ours, ref, what):
    """Two-sample chi-square on the bins either sample populates (>= 10 counts in total, sparse tails pooled)."""
    ours, ref = np.asarray(ours, dtype=np.int64), np.asarray(ref, dtype=np.int64)
    assert ours.shape == ref.shape, what
    keep = (ours + ref) >= 10
    table = np.stack([np.append(ours[keep], ours[~keep].sum()), np.append(ref[keep], ref[~keep].sum())])
    table = table[:, table.sum(axis=0) > 0]
    p = chi2_contingency(table)[1]
    assert p > P_MIN, f"{what}: chi-square p = {p:.3g}\nours {ours}\nref  {ref}"


def river_histograms(water):
    """water: bool (N, H, W) masks [n, y, x] -> the histograms of gen_reset_stats.py."""
    N, H, W = water.shape
    start_col, start_row = np.zeros(W, np.int64), np.zeros(H, np.int64)
    length, drift, col_cells = np.zeros(H + 1, np.int64), np.zeros(2 * H + 1, np.int64), np.zeros(W, np.int64)
    for w in water:
        ys, xs = np.nonzero(w)                      # row-major: sorted by y
        assert len(ys) and (np.diff(ys) == 1).all() and (np.abs(np.diff(xs)) == 1).all(), "one cell per row, drifting by one"
        start_col[xs[0]] += 1
        start_row[ys[0]] += 1
        length[len(ys)] += 1
        drift[xs[-1] - xs[0] + H] += 1
        np.add.at(col_cells, xs, 1)
    return dict(start_col=start_col, start_row=start_row, length=length, drift=drift, col_cells=col_cells)


def check_rivers(water, size):
    ref = ref_stats(size)
    for k, h in river_histograms(water).items():
        same_law(h, ref[k], f"river {k} ({size}x{size})")
    # rivers through the fire cell (the reference: 8-16 %): same rate
    cross = int(water[:, size // 2, size // 2].sum())
    same_law([cross, len(water) - cross], [int(ref["n_crossing"]), int(ref["n_rivers"]) - int(ref["n_crossing"])],
             f"rivers through the fire cell ({size}x{size})")


def spawn_counts(x, y, size):
    pts = [tuple(p) for p in ref_stats(size)["spawn_points"].tolist()]
    index = {p: i for i, p in enumerate(pts)}
    return np.bincount([index[(int(a), int(b))] for a, b in zip(x, y)], minlength=36)


@pytest.mark.parametrize("size", [10, 14])
def test_oracle_river_law_matches_the_reference_generator(size):
    N = 30000
    water = np.stack([O.river(size, size, seed=77, env_id=i, episode=i % 3) for i in range(N)]).astype(bool)
    check_rivers(water, size)


@pytest.mark.parametrize("size", [10, 14])
def test_oracle_spawn_law_matches_the_reference_two_stage_draw(size):
    B = 72000
    planes, envs = O.new_state(B, size, size)
    O.reset(planes, envs, seed=5)
    ref = ref_stats(size)
    counts = spawn_counts(envs["x"], envs["y"], size)
    same_law(counts, ref["spawn"], f"spawn point ({size}x{size})")
    # two stages: the radius is uniform over {1,2,3} (8 / 12 / 16 points), NOT the point over all 36
    radius = np.array([counts[:8].sum(), counts[8:20].sum(), counts[20:].sum()])
    same_law(radius, [ref["spawn"][:8].sum(), ref["spawn"][8:20].sum(), ref["spawn"][20:].sum()], "spawn radius")
    assert abs(radius / B - 1 / 3).max() < 0.01


def test_reset_with_rivers_oracle_layout():
    """Order of World.reset with make_rivers: river, then the fire cell, then the agent digs its spawn cell (a spawn
    on a river cell becomes dirt: Agent.__init__ -> dig(), environment.py:111-113,123-133)."""
    B, S = 4096, 14
    planes, envs = O.new_state(B, S, S)
    O.reset(planes, envs, seed=9, params=O.default_params(make_rivers=1))
    t = planes[..., 0]
    full = np.stack([O.river(S, S, seed=9, env_id=i, episode=0) for i in range(B)]).astype(bool)
    crossed = full[:, S // 2, S // 2]
    assert 0.04 < crossed.mean() < 0.2
    # the fire cell: plain fire, or - under a river - fire with infinite fire mobility (set_fire_to after reset_map)
    np.testing.assert_array_equal(t[:, S // 2, S // 2], np.where(crossed, 13, 1))
    y, x = envs["y"].astype(int), envs["x"].astype(int)
    assert (t[np.arange(B), y, x] == 3).all()
    n_water = ((t == 4) | (t == 13)).reshape(B, -1).sum(axis=1)
    assert n_water.min() >= S - 7 and n_water.max() <= S - 2
    water = (t == 4) | (t == 13)
    # spawn on the river: the river shows a one-cell gap there
    for i in range(64):
        gap = full[i] & ~water[i]
        assert gap.sum() == int(full[i, y[i], x[i]]) and (gap.sum() == 0 or gap[y[i], x[i]])
    # the observation shows a blocked fire cell as fire, and as blocked (environment.py:399-402)
    obs = O.get_state(planes)
    i = int(np.nonzero(crossed)[0][0])
    assert obs[i, S // 2, S // 2, 1] == 1 and obs[i, S // 2, S // 2, 2] == 0


# ----------------------------------------------------------------------------- HIP
@pytest.mark.gpu
@pytest.mark.parametrize("W,H,B", [(14, 14, 20000), (10, 10, 20000), (12, 12, 3000), (20, 10, 3000), (24, 16, 600), (13, 11, 600), (256, 64, 8), (42, 30, 520), (64, 48, 200), (100, 150, 12)])
def test_device_reset_with_rivers_equals_oracle(monkeypatch, W, H, B):
    """make_rivers on the device: wf_reset and the in-kernel auto-reset draw the river from the counter-based stream,
    bit-equal to the oracle in every kernel family (rows / group / fused wave kernels, tiled, generic)."""
    import wildfire_control_python_amd as wf
    if (W, H) == (24, 16):
        monkeypatch.setenv("WF_FAMILY", "tiled")           # (a grid this small goes to the generic family by itself)
    env = wf.BatchedForestFire(B, width=W, height=H, seed=31, auto_reset=True, containment=True, fuel_range=(2, 6),
                               metadata={"make_rivers": True}, env_id0=1000)
    env.reset(seed=31)
    op = O.default_params(make_rivers=1, containment=1, fuel_lo=2, fuel_hi=6)
    planes, envs = O.new_state(B, H, W)
    O.reset(planes, envs, seed=31, env_id0=1000, params=op)
    gp, ge = env.download()
    np.testing.assert_array_equal(gp, planes, err_msg="reset planes")
    assert (gp[..., 0] == 4).any()
    if B >= 500:
        assert (gp[..., 0] == 13).any(), "some river must run through the fire cell"
    rng = np.random.default_rng(W * H)
    resets = paid = 0
    vis = np.uint16(~np.uint16(128 | 256))
    T = 60 if B > 1000 else 90
    for t in range(T):
        n = 3 if (W, H) == (12, 12) and t % 2 else 1           # the fused multi-step kernel resets in its step loop too
        a = rng.integers(0, 5, size=(n, B)).astype(np.int8)
        r, d = env.step_many(a)
        for k in range(n):
            stopped = (envs["flags"] & O.F_RUNNING) == 0
            ro, do = O.step(planes, envs, a[k], op)
            if stopped.any():
                O.reset(planes, envs, seed=31, mask=stopped, env_id0=1000, params=op)
                ro[stopped], do[stopped] = 0.0, 0
                resets += int(stopped.sum())
            np.testing.assert_array_equal(r[k], ro, err_msg=f"reward t={t}")
            np.testing.assert_array_equal(d[k], do.astype(bool), err_msg=f"done t={t}")
            paid += int((ro == 1000.0).sum())
        if t % 6 == 5 or t == T - 1:
            gp, ge = env.download()
            np.testing.assert_array_equal(gp, planes, err_msg=f"planes t={t}")
            ge["flags"] &= vis
            np.testing.assert_array_equal(ge, envs, err_msg=f"envs t={t}")
    assert resets > B // 2
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("size", [10, 14])
def test_device_reset_laws_match_the_reference(size):
    """Device spawn points and device rivers against the reference-generated histograms."""
    import wildfire_control_python_amd as wf
    B = 72000
    env = wf.BatchedForestFire(B, size=size, seed=8, metadata={"make_rivers": True})
    env.reset(seed=8)
    planes, envs = env.download()
    ref = ref_stats(size)
    same_law(spawn_counts(envs["x"], envs["y"], size), ref["spawn"], f"device spawn point ({size}x{size})")
    # the river, spawn cell put back (the agent dug it): compare with the stand-alone river of the same stream
    water = (planes[:30000, :, :, 0] == 4) | (planes[:30000, :, :, 0] == 13)
    y, x = envs["y"][:30000].astype(int), envs["x"][:30000].astype(int)
    full = water.copy()
    for i in range(len(full)):
        w = O.river(size, size, seed=8, env_id=i, episode=0).astype(bool)
        assert (w & ~water[i]).sum() <= 1 and not (water[i] & ~w).any()
        full[i] = w
    check_rivers(full, size)
    env.close()


def test_oracle_wind_law_is_the_reference_draw():
    """METADATA['wind'] = "random" (environment.py:188-190): speed = np.random.choice([0, 0.7, 0.85]) and the vector
    (r.randint(-1, 1), r.randint(-1, 1)) - three independent uniform draws, 27 equally likely winds per episode.
    (The device reset equals the oracle's bit for bit: test_wind_step_and_reset_match_oracle.)"""
    from scipy.stats import chisquare
    B = 54000
    planes, envs = O.new_state(B, 10, 10)
    O.reset(planes, envs, seed=99, params=O.default_params(wind_mode=2))
    code = envs["wind"].astype(int)
    wx, wy, si = code & 3, (code >> 2) & 3, code >> 4
    assert wx.max() == 2 and wy.max() == 2 and si.max() == 2
    cells = np.bincount(si * 9 + wy * 3 + wx, minlength=27)
    assert chisquare(cells).pvalue > P_MIN, cells
    # and a new episode draws anew
    O.reset(planes, envs, seed=99, params=O.default_params(wind_mode=2))
    assert 0.9 < (envs["wind"] != code).mean() / (26 / 27) < 1.1
