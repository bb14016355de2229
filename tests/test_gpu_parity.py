"""GPU parity tests (run with `pytest -m gpu` on an MI355X).

Everything goes through the C-ABI (libwildfire_hip.so via ctypes).  The HIP path is
compared bit-exactly (integer planes, scalars, float64 rewards) with
  * the reference's own outputs (tests/golden/ref_*.npz), and
  * the CPU oracle on the same seeded inputs,
and at BASELINE.json's full sizes through size-independent properties.
"""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.golden_util import Episodes, fixture_files

pytestmark = pytest.mark.gpu

VIS = np.uint16(~np.uint16(128 | 256))  # drop WF_F_REACH_VALID / _ROWS (internal cache flags) before comparing


@pytest.fixture(scope="module")
def wf():
    import torch
    torch.cuda.init()   # torch brings its own copy of the HIP runtime: let it find the device before this library's copy is busy
    import wildfire_control_python_amd as pkg
    from wildfire_control_python_amd import _native
    _native.lib()   # fail loudly if the HIP library is missing
    return pkg


def _envs(ge):
    e = ge.copy()
    e["flags"] &= VIS
    return e


# --------------------------------------------------------------- reference fixtures
@pytest.mark.parametrize("path", fixture_files(), ids=lambda p: p.split("/")[-1])
def test_hip_reproduces_reference(wf, path):
    """Every fixture generated from the real reference, replayed on the device.  The split-fire fixtures carry the
    build's rule where the reference's answer depends on CPython's set order (Episodes._sigma_star; the steps are
    enumerated in tests/test_oracle_golden.py); everything else is the reference's record bit for bit."""
    ep = Episodes(path)
    md = {}
    if "toggle" in path:
        md = {"allow_dig_toggle": True, "n_actions": 5}                               # constants.py:46 as the generator set it
    if "river" in path:
        md = {"make_rivers": True}                                                    # constants.py:44 (code 13 cells)
    if ep.wind:
        md = {"wind": "random"}                                                       # constants.py:39
    # a_speed fixtures: the reference's global counter starts every episode in one of two phases (it survives reset());
    # env b = episode b, so the episodes of each phase form one context whose counter starts where theirs did
    groups = [(np.arange(ep.B), dict(md))]
    if ep.a_speed_iter0 is not None:
        groups = [(np.nonzero(ep.a_speed_iter0 == it)[0], dict(md, a_speed=2, a_speed_iter=int(it)))
                  for it in sorted(set(ep.a_speed_iter0.tolist()))]
    envs_ = []
    for idx, m in groups:
        e = wf.BatchedForestFire(len(idx), size=ep.size, metadata=m or None)
        e.upload(np.ascontiguousarray(ep.init_planes[idx]), ep.init_envs[idx],
                 counts=np.zeros((len(idx),) + ep.init_planes.shape[1:], np.uint8) if ep.wind else None)
        envs_.append(e)
    n_contained = 0
    planes, envs = ep.init_planes.copy(), ep.init_envs.copy()
    obs = np.zeros((ep.B, ep.size, ep.size, 3), np.int8)
    reward, done = np.zeros(ep.B), np.zeros(ep.B, np.uint8)
    counts = np.zeros(ep.init_planes.shape, np.uint8) if ep.wind else None
    for t in range(ep.T):
        actions, _ = ep.actions_at(t)
        for (idx, _), e in zip(groups, envs_):
            obs[idx], reward[idx], d, _ = e.step(actions[idx])
            done[idx] = d
            planes[idx], envs[idx] = e.download()
            if ep.wind:
                counts[idx] = e.download_counts()
        live = ep.check_step(t, planes, envs, reward, done, obs, what="hip")
        if ep.wind:
            ep.check_temp(t, planes, counts, what="hip")
        n_contained += int((reward[live] == 1000.0).sum())
    if "circle" in path:
        assert n_contained > 0
    if "river" in path:
        assert (ep.d["planes"][..., 0] == 13).any()
    for e in envs_:
        e.close()


@pytest.mark.parametrize("variant", ["group", "rows", "fused", "cells"])
@pytest.mark.parametrize("size", [10, 14])
def test_split_fire_fixtures_in_every_kernel_family(wf, monkeypatch, variant, size):
    """The split-fire fixtures (scripted agents that dig burnt cells until a wall is complete, poked walls, several
    ignition points; gen_golden.py section 9) through each kernel family that serves a 10x10 / 14x14 grid."""
    for k in ("WF_NO_ROWS", "WF_GROUP", "WF_SHARED", "WF_ROWS", "WF_FAMILY"):
        monkeypatch.delenv(k, raising=False)
    if variant == "group":
        monkeypatch.setenv("WF_NO_ROWS", "1")
    if variant == "rows":
        monkeypatch.setenv("WF_GROUP", "2")
    if variant == "cells":
        monkeypatch.setenv("WF_FAMILY", "cells")
    ep = Episodes([f for f in fixture_files(size) if f.endswith("_splitfire.npz")][0])
    env = wf.BatchedForestFire(ep.B, size=size)
    if variant in ("group", "rows"):
        assert env.step_kernel.startswith("wf_step_" + variant), env.step_kernel
    if variant == "cells":
        assert env.step_kernel == "wf_step_env"
    env.upload(ep.init_planes, ep.init_envs)
    t = 0
    while t < ep.T:
        n = min(3, ep.T - t) if variant == "fused" else 1
        acts = np.stack([ep.actions_at(t + k)[0] for k in range(n)])
        if variant == "fused":
            r, d = env.step_many(acts)
        else:
            _, r1, d1, _ = env.step(acts[0])
            r, d = r1[None], d1[None]
        for k in range(n):
            live = ep.lengths > t + k
            np.testing.assert_array_equal(r[k][live], ep.exp_reward[ep.ep_start[:-1][live] + t + k], err_msg=f"reward t={t + k}")
            np.testing.assert_array_equal(d[k][live], ep.expected_at(t + k, "done")[0] != 0, err_msg=f"done t={t + k}")
        t += n
        planes, envs = env.download()
        live = ep.lengths > t - 1
        np.testing.assert_array_equal(planes[live], ep.expected_at(t - 1, "planes")[0], err_msg=f"planes t={t - 1}")
        np.testing.assert_array_equal((envs["flags"][live] & O.F_CONTAINED) != 0,
                                      ep.exp_contained[ep.ep_start[:-1][live] + t - 1], err_msg=f"contained t={t - 1}")
    assert len(ep.divergent) > 0
    env.close()


# ------------------------------------------------------------------ HIP vs oracle
SHAPES = [(14, 14), (10, 10), (8, 8), (12, 12), (16, 16), (16, 8), (8, 16), (20, 10), (10, 20)]


@pytest.mark.parametrize("W,H", SHAPES)
def test_step_matches_oracle_random_actions(wf, W, H):
    B, T = 512, 160
    env = wf.BatchedForestFire(B, width=W, height=H, seed=5, fuel_range=(6, 30), ignite_ppm=20000)
    env.reset(seed=77)
    op = O.default_params(fuel_lo=6, fuel_hi=30, ignite_ppm=20000)
    planes, envs = O.new_state(B, H, W)
    O.reset(planes, envs, seed=77, params=op)
    gp, ge = env.download()
    np.testing.assert_array_equal(gp, planes, err_msg="reset planes")
    np.testing.assert_array_equal(_envs(ge), envs, err_msg="reset envs")
    rng = np.random.default_rng(W * 100 + H)
    n_done = 0
    for t in range(T):
        a = rng.integers(0, 6, size=B).astype(np.int8)      # includes no-ops (4 is a no-op here, 5 too)
        obs, r, d, _ = env.step(a)
        ro, do = O.step(planes, envs, a, op)
        gp, ge = env.download()
        np.testing.assert_array_equal(gp, planes, err_msg=f"planes t={t}")
        np.testing.assert_array_equal(_envs(ge), envs, err_msg=f"envs t={t}")
        np.testing.assert_array_equal(r, ro, err_msg=f"reward t={t}")
        np.testing.assert_array_equal(d, do.astype(bool), err_msg=f"done t={t}")
        np.testing.assert_array_equal(obs, O.get_state(planes), err_msg=f"obs t={t}")
        n_done = int(do.sum())
    assert n_done > B // 4, "episodes should finish within the horizon"
    env.close()


# wind with a direction (environment.py:187-193, 260-287): "random" draws speed and vector per episode, a fixed
# [speed, (wx, wy)] gives every episode the same; per-direction heat counts ride along with the planes
@pytest.mark.parametrize("W,H,wind,auto,B", [(14, 14, "random", True, 384), (10, 10, "random", False, 384), (14, 14, [0.85, (1, -1)], False, 384),
                                             (24, 16, "random", False, 384), (13, 11, [0.7, (0, 1)], True, 384),
                                             (64, 40, "random", False, 96),        # few big grids: thread-per-cell sequence
                                             (42, 30, "random", True, 512)])       # a workgroup per environment
def test_wind_step_and_reset_match_oracle(wf, W, H, wind, auto, B):
    from tests.golden_util import WIND_SPEEDS
    T = 150
    env = wf.BatchedForestFire(B, width=W, height=H, seed=5, fuel_range=(6, 30), ignite_ppm=8000, auto_reset=auto,
                               metadata={"wind": wind})
    if (W, H) in SHAPES:            # the wave family's group-layout kernel, WIND instantiation
        assert env.step_kernel == f"wf_step_group<{W},{H},wind>" and env.containment, env.step_kernel
    else:
        assert "cells" in env.kernel_name and env.containment
    if wind == "random":
        op = O.default_params(fuel_lo=6, fuel_hi=30, ignite_ppm=8000, wind_mode=2, wind_speeds=WIND_SPEEDS + [0.0])
    else:
        op = O.default_params(fuel_lo=6, fuel_hi=30, ignite_ppm=8000, wind_mode=1, wind_code=O.wind_code(*wind[1], 0),
                              wind_speeds=[wind[0], 0.0, 0.0, 0.0])
    env.reset(seed=77)
    planes, envs = O.new_state(B, H, W)
    counts = np.zeros(planes.shape, np.uint8)
    O.reset(planes, envs, seed=77, params=op, counts=counts)
    gp, ge = env.download()
    np.testing.assert_array_equal(gp, planes, err_msg="reset planes")
    np.testing.assert_array_equal(_envs(ge), envs, err_msg="reset envs")
    if wind == "random" and B >= 384:
        assert len(set(envs["wind"].tolist())) == 27          # every speed x vector is drawn
    rng = np.random.default_rng(W * 100 + H)
    n_done = n_ign = 0
    for t in range(T):
        a = rng.integers(0, 6, size=B).astype(np.int8)
        obs, r, d, _ = env.step(a)
        stopped = (envs["flags"] & O.F_RUNNING) == 0
        ro, do = O.step(planes, envs, a, op, counts=counts)
        if auto and stopped.any():
            O.reset(planes, envs, seed=5, mask=stopped, params=op, counts=counts)
            ro[stopped], do[stopped] = 0.0, 0
        gp, ge = env.download()
        np.testing.assert_array_equal(gp, planes, err_msg=f"planes t={t}")
        np.testing.assert_array_equal(_envs(ge), envs, err_msg=f"envs t={t}")
        np.testing.assert_array_equal(env.download_counts(), counts, err_msg=f"counts t={t}")
        np.testing.assert_array_equal(r, ro, err_msg=f"reward t={t}")
        np.testing.assert_array_equal(d, do.astype(bool), err_msg=f"done t={t}")
        np.testing.assert_array_equal(obs, O.get_state(planes), err_msg=f"obs t={t}")
        n_done += int(do.sum())
        n_ign += int((planes[..., 2] == O.HEAT_IGNITED).sum())
    assert n_done > B // 4 and n_ign > 1000
    # an n-step call (wind has single-step kernels only: the library issues n launches)
    a3 = rng.integers(0, 6, size=(3, B)).astype(np.int8)
    r3, d3 = env.step_many(a3)
    for k in range(3):
        stopped = (envs["flags"] & O.F_RUNNING) == 0
        ro, do = O.step(planes, envs, a3[k], op, counts=counts)
        if auto and stopped.any():
            O.reset(planes, envs, seed=5, mask=stopped, params=op, counts=counts)
            ro[stopped], do[stopped] = 0.0, 0
        np.testing.assert_array_equal(r3[k], ro, err_msg=f"3-step call, step {k}: reward")
        np.testing.assert_array_equal(d3[k], do.astype(bool), err_msg=f"3-step call, step {k}: done")
    gp, ge = env.download()
    np.testing.assert_array_equal(gp, planes, err_msg="planes after the 3-step call")
    np.testing.assert_array_equal(env.download_counts(), counts, err_msg="counts after the 3-step call")
    env.close()


@pytest.mark.parametrize("W,H", [(14, 12), (14, 14), (10, 10)])
def test_wind_with_rivers_and_dig_toggle(wf, W, H):
    """All three off-by-default options of the reference at once (constants.py:39,44,46): random wind, rivers, dig
    toggle, plus auto-reset and containment - generic family (14x12) and the wave family's wind kernel against the
    oracle, every step."""
    from tests.golden_util import WIND_SPEEDS
    from wildfire_control_python_amd import _native as N
    B, T = 640, 120
    env = wf.BatchedForestFire(B, width=W, height=H, seed=9, fuel_range=(5, 20), auto_reset=True,
                               metadata={"wind": "random", "make_rivers": True, "allow_dig_toggle": True, "n_actions": 5})
    op = O.default_params(fuel_lo=5, fuel_hi=20, wind_mode=2, wind_speeds=WIND_SPEEDS + [0.0], make_rivers=1, allow_dig_toggle=1)
    env.reset(seed=3)
    planes, envs = O.new_state(B, H, W)
    counts = np.zeros(planes.shape, np.uint8)
    O.reset(planes, envs, seed=3, params=op, counts=counts)
    rng = np.random.default_rng(1)
    seen5 = 0
    for t in range(T):
        a = rng.integers(0, 6, size=B).astype(np.int8)
        obs, r, d, _ = env.step(a)
        stopped = (envs["flags"] & O.F_RUNNING) == 0
        ro, do = O.step(planes, envs, a, op, counts=counts)
        if stopped.any():
            O.reset(planes, envs, seed=9, mask=stopped, params=op, counts=counts)
            ro[stopped], do[stopped] = 0.0, 0
        gp, ge = env.download()
        np.testing.assert_array_equal(gp, planes, err_msg=f"planes t={t}")
        np.testing.assert_array_equal(_envs(ge), envs, err_msg=f"envs t={t}")
        np.testing.assert_array_equal(env.download_counts(), counts, err_msg=f"counts t={t}")
        np.testing.assert_array_equal(r, ro, err_msg=f"reward t={t}")
        np.testing.assert_array_equal(d, do.astype(bool), err_msg=f"done t={t}")
        np.testing.assert_array_equal(obs, O.get_state(planes), err_msg=f"obs t={t}")
        seen5 += int((planes[..., 0] == 5).sum())
    assert (planes[..., 0] == 4).any() and (seen5 > 0 or (W, H) != (14, 12))   # (random agents: burning dirt shows up at 14x12)
    env.close()
    # the count plane exists only with a wind that has a direction
    calm = wf.BatchedForestFire(4, size=10)
    buf = np.zeros((4, 10, 10, 4), np.uint8)
    assert calm._lib.wf_wind_counts_download(calm._h, N.ptr(buf)) == -2 and calm._lib.wf_wind_counts_upload(calm._h, N.ptr(buf)) == -2
    calm.close()


def test_wind_adapter_draws_like_the_reference(wf):
    """METADATA['wind'] = "random" through the single-environment adapter: a seeded reset draws the wind speed from
    numpy's stream and the vector from Python's, before the spawn point, as World.reset does (environment.py:188-190);
    the episode then replays the reference's, and World.env shows the reference's temperatures."""
    import random
    for size in (10, 14):
        ep = Episodes([f for f in fixture_files(size) if f.endswith("_wind.npz")][0])
        sim = wf.ForestFire(size=size, metadata={"wind": "random"})
        for i in (0, 3, 6, 12, 15):                            # no-op episodes of the generator (ep % 3 == 0) with default fuel (ep % 5 != 4)
            seed = int(ep.d["seeds"][i])
            np.random.seed(seed)
            random.seed(seed)
            sim.reset()
            assert (sim.W.wind_speed, *sim.W.wind_vector) == tuple(ep.d["wind"][i]), (size, i)
            assert (sim.W.agents[0].x, sim.W.agents[0].y) == tuple(ep.d["init_agent"][i, :2])
            lo = int(ep.ep_start[i])
            for t in range(int(ep.lengths[i])):
                state, reward, done, _ = sim.step(" ")
                np.testing.assert_array_equal(state, ep.d["obs"][lo + t].astype(np.float64))
                assert reward == ep.d["reward"][lo + t] and done == bool(ep.d["done"][lo + t])
                if t % 16 == 0:
                    temp = sim.W.env[:, :, sim.layer["temp"]].T
                    known = temp != 3.5          # (a cell ignited by accumulated heat reads the stand-in threshold + 0.5)
                    np.testing.assert_allclose(temp[known], ep.d["temp"][lo + t][known], rtol=0, atol=2e-6)
            assert done
        sim.close()


# big-grid families: row-strip tiles with LDS halo masks (W % 4 == 0, W <= 256) and the generic
# thread-per-cell kernel (any other shape).
BIG_SHAPES = [(24, 24, 96), (32, 20, 64), (64, 64, 24), (100, 37, 16), (256, 70, 6), (256, 256, 3),
              (11, 11, 128), (13, 10, 128), (30, 23, 48), (260, 9, 8),
              # generic family: a wave per environment (up to 1 024 cells), a workgroup per environment (more cells, 512+
              # environments), the thread-per-cell sequence (few huge grids, or WF_CELLS_LEGACY=1: marked by a negative B)
              (50, 42, 520), (11, 11, -128), (30, 23, -48), (15, 15, 700),
              # small grids the tiled family can serve go to the generic one (faster below 32x32); WF_FAMILY=tiled keeps
              # the tiled kernels covered there (marked by B + 10000); packed strips: 4 per wave up to W = 64, 2 up to 128
              (24, 24, 10096), (32, 20, 10064), (48, 48, 24), (64, 100, 12), (128, 128, 6), (36, 30, 40),
              (64, 300, 6), (128, 200, 4), (100, 140, 4),          # packed strips AND several workgroups per environment
              # the tiled family's step is ONE launch (wf_step_tiled<solo>: a workgroup of up to 16 waves per environment)
              # unless the observation is asked for and its tile does not fit a launch's LDS (256x256 here: every second
              # step below goes without) or the job has thousands of waves in workgroups of more than four (not here);
              # WF_NO_SOLO=1 keeps the two-launch form covered everywhere (B + 20000)
              (64, 64, 20024), (128, 128, 20006), (64, 100, 20012), (32, 20, 30064), (64, 300, 20006), (100, 140, 20004),
              (256, 256, 20003),
              (28, 28, 64), (28, 27, 64), (24, 23, 64),              # around the smallest grid the tiled family takes by itself (24x24, above)
              (32, 200, 12), (20, 60, 30), (32, 32, 300), (32, 64, 20040)]   # up to 32 cells wide: eight strips side by side per wave (one launch)


@pytest.mark.parametrize("W,H,B", BIG_SHAPES)
def test_big_grid_step_matches_oracle(wf, monkeypatch, W, H, B):
    monkeypatch.delenv("WF_TILED_ROWS", raising=False)
    _run_big_grid(wf, monkeypatch, W, H, B)


@pytest.mark.parametrize("W,H,rows", [(64, 64, 16), (64, 64, 4), (32, 32, 16), (32, 32, 2), (128, 128, 16), (48, 40, 5), (128, 60, 9),
                                        (32, 32, 1), (256, 16, 1), (12, 80, 0)])
def test_one_launch_tiled_step_with_any_strip_height(wf, monkeypatch, W, H, rows):
    """The one-launch tiled step takes strips of fewer than 16 rows where that suits the job (wf_tiled_solo_rows: small
    test batches get the short ones); WF_TILED_ROWS picks the height, here the full 16 and odd ones."""
    monkeypatch.setenv("WF_TILED_ROWS", str(rows))                # (0: the default, here for a grid three quads wide)
    _run_big_grid(wf, monkeypatch, W, H, 20)


@pytest.mark.parametrize("W,H,B,n", [(64, 64, 48, 16), (32, 32, 70, 16), (128, 64, 20, 7), (100, 40, 30, 33), (28, 28, 64, 64),
                                     (256, 256, 3, 5), (64, 64, 2100, 6), (32, 32, 9000, 4)])
def test_tiled_step_loop_matches_oracle(wf, monkeypatch, W, H, B, n):
    """n-step calls of the tiled family's one-launch step are ONE launch per share with the step loop inside the kernel
    (wf_step_tiled<.., SOLO, MULTI>): the workgroup that owns the environment runs its n steps back to back - agent
    logic, tick, flood, reward, auto-reset - with the caches between the steps dropped.  Against the oracle: rewards and
    done flags of every step, the state after every call; every other call through the device API with the fused
    observation of the last step (n - 1 looped steps + one launch that writes it); half of the agents close rings
    (floods and containment payments inside the loop).  A WF_NO_SOLO_LOOP=1 twin (n launches per call) must agree too.
    2 100 x 64x64 is past the size where a call goes out as three shares on three streams; 9 000 x 32x32 is a job of more
    waves than the chip holds at once, which steps with n launches by default (here: WF_SOLO_LOOP=1)."""
    import torch
    for k in ("WF_NO_SOLO", "WF_NO_SOLO_LOOP", "WF_SOLO_LOOP", "WF_TILED_ROWS", "WF_FAMILY"):
        monkeypatch.delenv(k, raising=False)
    kw = dict(width=W, height=H, seed=9, fuel_range=(3, 12), ignite_ppm=8000, auto_reset=True, containment=True)
    if B >= 9000:
        plain = wf.BatchedForestFire(B, **kw)
        assert plain._lib.wf_call_launches(plain._h, n) == 3 * n
        plain.close()
        monkeypatch.setenv("WF_SOLO_LOOP", "1")
    env = wf.BatchedForestFire(B, **kw)
    monkeypatch.delenv("WF_SOLO_LOOP", raising=False)
    monkeypatch.setenv("WF_NO_SOLO_LOOP", "1")
    twin = wf.BatchedForestFire(B, **kw)
    monkeypatch.delenv("WF_NO_SOLO_LOOP", raising=False)
    parts = 3 if B * W * H >= 8 << 20 else 1
    assert env.step_kernel == "wf_step_tiled<solo>"
    assert env._lib.wf_call_launches(env._h, n) == parts and twin._lib.wf_call_launches(twin._h, n) == n * parts
    env.reset(seed=31)
    twin.reset(seed=31)
    op = O.default_params(fuel_lo=3, fuel_hi=12, ignite_ppm=8000, containment=1)
    planes, envs = O.new_state(B, H, W)
    O.reset(planes, envs, seed=31, params=op)
    rng = np.random.default_rng(W + H + n)
    dev = torch.device("cuda", 0)
    d_r = torch.empty((n, B), dtype=torch.float64, device=dev)
    d_d = torch.empty((n, B), dtype=torch.uint8, device=dev)
    d_obs = torch.empty((B, W, H, 3), dtype=torch.int8, device=dev)
    fused_obs = H % 4 == 0 and H * W * 3 < 48 * 1024
    paid = resets = 0
    for call in range(3 if B > 1000 else 6):
        a = rng.integers(0, 6, size=(n, B)).astype(np.int8)
        a[0, :B // 2] = _circle_actions(envs[:B // 2], W, H, rng)
        a[1:, :B // 2] = a[0, :B // 2]                       # (stale after the first step: still ring-ish diggers)
        if call % 2:
            d_a = torch.from_numpy(a).to(dev)
            torch.cuda.synchronize()
            env.step_device(d_a, n, d_r, d_d, d_obs=d_obs if fused_obs else None)
            env.sync()
            r, d = d_r.cpu().numpy(), d_d.cpu().numpy().astype(bool)
        else:
            r, d = env.step_many(a)
        rt, dt = twin.step_many(a)
        for k in range(n):
            stopped = (envs["flags"] & O.F_RUNNING) == 0
            ro, do = O.step(planes, envs, a[k], op)
            if stopped.any():
                O.reset(planes, envs, seed=9, mask=stopped, params=op)
                ro[stopped], do[stopped] = 0.0, 0
                resets += int(stopped.sum())
            paid += int((ro == 1000.0).sum())
            np.testing.assert_array_equal(r[k], ro, err_msg=f"call {call} step {k}: reward")
            np.testing.assert_array_equal(d[k], do.astype(bool), err_msg=f"call {call} step {k}: done")
        np.testing.assert_array_equal(rt, r, err_msg=f"call {call}: reward, n launches against one")
        np.testing.assert_array_equal(dt, d, err_msg=f"call {call}: done, n launches against one")
        gp, ge = env.download()
        np.testing.assert_array_equal(gp, planes, err_msg=f"planes after call {call}")
        np.testing.assert_array_equal(_envs(ge), envs, err_msg=f"envs after call {call}")
        if call % 2 and fused_obs:
            np.testing.assert_array_equal(d_obs.cpu().numpy(), O.get_state(planes), err_msg=f"fused observation of call {call}")
    tp, te = twin.download()
    np.testing.assert_array_equal(tp, planes)
    np.testing.assert_array_equal(_envs(te), envs)
    assert resets > 0 or n * 6 < 40
    env.close()
    twin.close()


def _run_big_grid(wf, monkeypatch, W, H, B):
    T = 70
    monkeypatch.delenv("WF_FAMILY", raising=False)
    if B < 0:
        monkeypatch.setenv("WF_CELLS_LEGACY", "1")
        B = -B
    monkeypatch.delenv("WF_NO_SOLO", raising=False)
    no_solo = B > 20000
    if no_solo:
        monkeypatch.setenv("WF_NO_SOLO", "1")
        B -= 20000
    if B > 10000:
        monkeypatch.setenv("WF_FAMILY", "tiled")
        B -= 10000
    forced_tiled = "WF_FAMILY" in __import__("os").environ
    auto = W in (24, 15, 50) or (W == 30 and B == 48)
    env = wf.BatchedForestFire(B, width=W, height=H, seed=5, fuel_range=(3, 12), ignite_ppm=15000,
                               auto_reset=auto)
    assert "tiled" in env.kernel_name or "cells" in env.kernel_name
    assert ("tiled" in env.kernel_name) == (W % 4 == 0 and W <= 256 and (W * H > 575 or forced_tiled))
    if "cells" in env.kernel_name:
        legacy = "WF_CELLS_LEGACY" in __import__("os").environ or (W * H > 1024 and B < 512)
        assert env.step_kernel == ("wf_step_cells" if legacy else "wf_step_env")
    if "tiled" in env.kernel_name:
        assert env.step_kernel == ("wf_step_tiled" if no_solo else "wf_step_tiled<solo>"), env.step_kernel
    assert env.containment                                       # both big-grid families test containment
    env.reset(seed=123)
    op = O.default_params(fuel_lo=3, fuel_hi=12, ignite_ppm=15000, containment=int(env.containment))
    planes, envs = O.new_state(B, H, W)
    O.reset(planes, envs, seed=123, params=op)
    gp, ge = env.download()
    np.testing.assert_array_equal(gp, planes, err_msg="reset planes")
    np.testing.assert_array_equal(_envs(ge), envs, err_msg="reset envs")
    rng = np.random.default_rng(W * 1000 + H)
    for t in range(T):
        a = rng.integers(0, 6, size=B).astype(np.int8)
        if t % 2 and H * W * 3 > 48 * 1024:  # a step without the observation: one launch whatever the grid's size
            r, d = (x[0] for x in env.step_many(a))
            obs = env.get_state()
        else:
            obs, r, d, _ = env.step(a)
        stopped = (envs["flags"] & O.F_RUNNING) == 0
        ro, do = O.step(planes, envs, a, op)
        if auto and stopped.any():         # auto-reset shapes
            O.reset(planes, envs, seed=5, mask=stopped, params=op)
            ro[stopped], do[stopped] = 0.0, 0
        gp, ge = env.download()
        np.testing.assert_array_equal(gp, planes, err_msg=f"planes t={t}")
        np.testing.assert_array_equal(_envs(ge), envs, err_msg=f"envs t={t}")
        np.testing.assert_array_equal(r, ro, err_msg=f"reward t={t}")
        np.testing.assert_array_equal(d, do.astype(bool), err_msg=f"done t={t}")
        # World.get_state(): written by the tiled step kernel itself when H % 4 == 0, else by the observation kernel
        np.testing.assert_array_equal(obs, O.get_state(planes), err_msg=f"obs t={t}")
    env.close()


def test_big_grid_split_equals_single_launch_and_oracle(wf, monkeypatch):
    """Big-grid jobs (>= 8 Mi cells) on the context's own streams run as shares of the batch on as many streams (three
    by default); the result must not depend on it (reset streams are keyed by the context-level env index)."""
    W = H = 64
    B, T = 2100, 24
    kw = dict(width=W, height=H, seed=9, auto_reset=True, containment=True, fuel_range=(2, 5), ignite_ppm=30000)
    monkeypatch.delenv("WF_NO_SPLIT", raising=False)
    a_env = wf.BatchedForestFire(B, **kw)
    monkeypatch.setenv("WF_NO_SPLIT", "1")               # the WF_* switches are read when a context is created
    b_env = wf.BatchedForestFire(B, **kw)
    monkeypatch.delenv("WF_NO_SPLIT", raising=False)
    assert a_env.step_launches == 3 and b_env.step_launches == 1        # big-grid families: three shares of the batch
    a_env.reset(seed=9)
    b_env.reset(seed=9)
    op = O.default_params(containment=1, fuel_lo=2, fuel_hi=5, ignite_ppm=30000)
    planes, envs = O.new_state(B, H, W)
    O.reset(planes, envs, seed=9, params=op)
    rng = np.random.default_rng(1)
    resets = 0
    for t in range(T):
        a = rng.integers(0, 5, size=B).astype(np.int8)
        obs_a, r_a, d_a, _ = a_env.step(a)
        obs_b, r_b, d_b, _ = b_env.step(a)
        stopped = (envs["flags"] & O.F_RUNNING) == 0
        ro, do = O.step(planes, envs, a, op)
        if stopped.any():
            O.reset(planes, envs, seed=9, mask=stopped, params=op)
            ro[stopped], do[stopped] = 0.0, 0
            resets += int(stopped.sum())
        np.testing.assert_array_equal(r_a, r_b, err_msg=f"reward t={t}")
        np.testing.assert_array_equal(d_a, d_b, err_msg=f"done t={t}")
        np.testing.assert_array_equal(obs_a, obs_b, err_msg=f"obs t={t}")
        np.testing.assert_array_equal(r_a, ro, err_msg=f"reward vs oracle t={t}")
        np.testing.assert_array_equal(d_a, do.astype(bool), err_msg=f"done vs oracle t={t}")
    pa, ea = a_env.download()
    pb, eb = b_env.download()
    np.testing.assert_array_equal(pa, pb)
    np.testing.assert_array_equal(ea, eb)
    np.testing.assert_array_equal(pa, planes)
    np.testing.assert_array_equal(_envs(ea), envs)
    assert resets > 0
    a_env.close()
    b_env.close()


def test_big_grid_containment_is_refused_not_ignored(wf):
    """Where the containment test is not built (generic family beyond the flood kernel's LDS board, tiled
    family beyond 1024 rows) asking for it is an error, never a silent no-op."""
    with pytest.raises(Exception):
        wf.BatchedForestFire(1, width=301, height=301, containment=True)     # generic family, 90 601 cells
    with pytest.raises(Exception):
        wf.BatchedForestFire(1, width=8, height=1100, containment=True)      # tiled family, H > 1024
    env = wf.BatchedForestFire(1, width=301, height=301)                     # default: falls back to "off", loudly visible
    assert env.containment is False
    env.close()


def _circle_actions(ge, W, H, rng):
    """Scripted policy (host side, test only): walk clockwise around the grid centre."""
    x, y = ge["x"].astype(int), ge["y"].astype(int)
    mx, my = W // 2, H // 2
    a = np.empty(len(x), dtype=np.int8)
    for i in range(len(x)):
        if x[i] >= mx and y[i] > my:
            opts = (1, 3)
        elif x[i] > mx and y[i] <= my:
            opts = (1, 2)
        elif x[i] <= mx and y[i] < my:
            opts = (0, 2)
        else:
            opts = (0, 3)
        a[i] = opts[rng.integers(2)]
    return a


@pytest.mark.parametrize("W,H,family", [(24, 24, "tiled"), (32, 20, "tiled"), (14, 14, None), (13, 13, None), (30, 23, None), (11, 14, None),
                                        (24, 24, None), (60, 44, None), (51, 47, None),
                                        # the tiled family by itself: one launch per step (flood behind the sweep, from the masks the
                                        # sweep collected; one to eight waves per environment), and its two-launch form
                                        (28, 28, None), (32, 32, None), (64, 64, None), (100, 40, None), (128, 128, None),
                                        (64, 64, "two-launch"), (100, 40, "two-launch")])
def test_containment_rewards_with_circling_agents(wf, monkeypatch, W, H, family):
    """Agents that dig a ring around the fire: the +1000 containment reward (get_reward's A* part,
    environment.py:345-377) must appear exactly when and where the oracle pays it - tiled family
    (flood by the environment's workgroup or in the prologue + per-row board test), generic family (byte board in LDS)
    and wave family alike."""
    monkeypatch.delenv("WF_FAMILY", raising=False)
    monkeypatch.delenv("WF_NO_SOLO", raising=False)
    if family == "two-launch":
        monkeypatch.setenv("WF_NO_SOLO", "1")
    elif family:
        monkeypatch.setenv("WF_FAMILY", family)
    B, T = 96, 150
    env = wf.BatchedForestFire(B, width=W, height=H, seed=2, containment=True)
    env.reset(seed=77)
    op = O.default_params(containment=1)
    planes, envs = O.new_state(B, H, W)
    O.reset(planes, envs, seed=77, params=op)
    rng = np.random.default_rng(5)
    paid = 0
    for t in range(T):
        a = _circle_actions(envs, W, H, rng)
        _, r, d, _ = env.step(a)
        ro, do = O.step(planes, envs, a, op)
        gp, ge = env.download()
        np.testing.assert_array_equal(r, ro, err_msg=f"reward t={t}")
        np.testing.assert_array_equal(_envs(ge), envs, err_msg=f"envs t={t}")
        np.testing.assert_array_equal(gp, planes, err_msg=f"planes t={t}")
        paid += int((ro == 1000.0).sum())
    assert paid > B // 4, f"only {paid} containment rewards: the scenario does not exercise the test"
    env.close()


# The single-step path of the 14x14 and 10x10 grids has four kernels behind it, picked by batch size: the group
# layout (one environment per wave iteration) and the rows layout (two per iteration), each with
# private or workgroup-shared agent phases.  Force each one, ragged groups included.
KERNEL_VARIANTS = {
    "group-g1": {"WF_NO_ROWS": "1", "WF_GROUP": "1"},
    "group-g4": {"WF_NO_ROWS": "1", "WF_GROUP": "4"},
    "group-shared": {"WF_NO_ROWS": "1", "WF_GROUP": "8", "WF_SHARED": "1"},
    "rows-g2": {"WF_GROUP": "2"},
    "rows-g6-ragged": {"WF_GROUP": "6"},
    "rows-g8": {"WF_GROUP": "8"},
    "rows-g32": {"WF_GROUP": "32", "WF_ROWS": "1"},
    "rows-g22-ragged": {"WF_GROUP": "22", "WF_ROWS": "1"},
    "rows-shared-g8": {"WF_GROUP": "8", "WF_SHARED": "1"},
    "rows-shared-g16": {"WF_GROUP": "16", "WF_SHARED": "1"},
    "rows-shared4-g4": {"WF_GROUP": "4", "WF_SHARED": "4"},          # (round 4: the shape of a one-generation step on 14x14)
    "rows-shared2-g4": {"WF_GROUP": "4", "WF_SHARED": "2"},
    "rows-shared8-g2": {"WF_GROUP": "2", "WF_SHARED": "8"},
}


@pytest.mark.parametrize("S", [14, 10, 12])
@pytest.mark.parametrize("variant", list(KERNEL_VARIANTS))
def test_single_step_kernel_variants_match_oracle(wf, monkeypatch, variant, S):
    for k in ("WF_NO_ROWS", "WF_GROUP", "WF_SHARED", "WF_ROWS"):
        monkeypatch.delenv(k, raising=False)
    for k, v in KERNEL_VARIANTS[variant].items():
        monkeypatch.setenv(k, v)
    B, T = 4096 + 37, 90                   # ragged last group / last workgroup
    env = wf.BatchedForestFire(B, size=S, seed=11, auto_reset=True, containment=True, fuel_range=(4, 12))
    env.reset(seed=11)
    op = O.default_params(containment=1, fuel_lo=4, fuel_hi=12)
    planes, envs = O.new_state(B, S, S)
    O.reset(planes, envs, seed=11, params=op)
    rng = np.random.default_rng(3)
    paid = resets = 0
    half = B // 2
    for t in range(T):
        a = rng.integers(0, 6, size=B).astype(np.int8)
        a[:half] = _circle_actions(envs[:half], S, S, rng)       # diggers that close rings: floods + containment
        obs, r, d, _ = env.step(a)
        stopped = (envs["flags"] & O.F_RUNNING) == 0
        ro, do = O.step(planes, envs, a, op)
        if stopped.any():
            O.reset(planes, envs, seed=11, mask=stopped, params=op)
            ro[stopped], do[stopped] = 0.0, 0
            resets += int(stopped.sum())
        gp, ge = env.download()
        np.testing.assert_array_equal(r, ro, err_msg=f"reward t={t}")
        np.testing.assert_array_equal(d, do.astype(bool), err_msg=f"done t={t}")
        np.testing.assert_array_equal(_envs(ge), envs, err_msg=f"envs t={t}")
        np.testing.assert_array_equal(gp, planes, err_msg=f"planes t={t}")
        np.testing.assert_array_equal(obs, O.get_state(planes), err_msg=f"obs t={t}")
        paid += int((ro == 1000.0).sum())
    assert paid > 50 and resets > B, (paid, resets)
    env.close()


def test_fused_steps_equal_single_steps(wf):
    B, T = 256, 48
    a = np.random.default_rng(3).integers(0, 5, size=(T, B)).astype(np.int8)
    e1 = wf.BatchedForestFire(B, size=14, fuel_range=(10, 40))
    e2 = wf.BatchedForestFire(B, size=14, fuel_range=(10, 40))
    e1.reset(seed=9)
    e2.reset(seed=9)
    r1 = np.stack([e1.step(a[t])[1] for t in range(T)])
    r2a, d2a = e2.step_many(a[:16])
    r2b, d2b = e2.step_many(a[16:])
    np.testing.assert_array_equal(np.concatenate([r2a, r2b]), r1)
    p1, s1 = e1.download()
    p2, s2 = e2.download()
    np.testing.assert_array_equal(p1, p2)
    np.testing.assert_array_equal(_envs(s1), _envs(s2))
    e1.close()
    e2.close()


def test_multi_step_call_in_one_launch_matches_single_steps_and_oracle(wf, monkeypatch):
    """step_many on a large batch of a rows-layout shape is ONE launch per stream whose waves step their
    environments n times.  Same results as n single-step calls (WF_NO_MULTI=1) and as the oracle, with
    auto-resets, containment floods (boards rewritten and re-read inside the launch) and a_speed-style masks."""
    import ctypes
    from wildfire_control_python_amd import _native as N
    B, S, n, calls = 8192 + 64, 14, 7, 6
    kw = dict(size=S, seed=21, auto_reset=True, containment=True, fuel_range=(3, 10))
    monkeypatch.delenv("WF_NO_MULTI", raising=False)
    a_env = wf.BatchedForestFire(B, **kw)
    monkeypatch.setenv("WF_NO_MULTI", "1")               # read at construction
    b_env = wf.BatchedForestFire(B, **kw)
    monkeypatch.delenv("WF_NO_MULTI", raising=False)
    assert a_env._lib.wf_call_launches(a_env._h, n) == 2 and b_env._lib.wf_call_launches(b_env._h, n) == 2 * n
    a_env.reset(seed=21)
    b_env.reset(seed=21)
    op = O.default_params(containment=1, fuel_lo=3, fuel_hi=10)
    planes, envs = O.new_state(B, S, S)
    O.reset(planes, envs, seed=21, params=op)
    rng = np.random.default_rng(4)
    paid = resets = 0
    for call in range(calls):
        a = rng.integers(0, 6, size=(n, B)).astype(np.int8)
        a[:, :B // 2] = 5 if call == 0 else a[:, :B // 2]
        for k in range(n):
            a[k, :B // 3] = _circle_actions(envs[:B // 3], S, S, rng) if k == 0 else a[k, :B // 3]
        mask = (1 << n) - 1 if call % 2 == 0 else 0b1011011          # some steps without update() (a_speed > 1)
        ra, da = np.empty((n, B)), np.empty((n, B), dtype=np.uint8)
        rb, db = np.empty((n, B)), np.empty((n, B), dtype=np.uint8)
        obs_a = np.empty((B, S, S, 3), dtype=np.int8)
        obs_b = np.empty_like(obs_a)
        assert a_env._lib.wf_step_obs(a_env._h, N.ptr(a), n, ctypes.c_uint64(mask), N.ptr(ra), N.ptr(da), N.ptr(obs_a)) == 0
        assert b_env._lib.wf_step_obs(b_env._h, N.ptr(a), n, ctypes.c_uint64(mask), N.ptr(rb), N.ptr(db), N.ptr(obs_b)) == 0
        np.testing.assert_array_equal(ra, rb, err_msg=f"reward call {call}")
        np.testing.assert_array_equal(da, db, err_msg=f"done call {call}")
        np.testing.assert_array_equal(obs_a, obs_b, err_msg=f"obs call {call}")
        for k in range(n):
            stopped = (envs["flags"] & O.F_RUNNING) == 0
            ro, do = O.step(planes, envs, a[k], op, do_update=bool((mask >> k) & 1))
            if stopped.any():
                O.reset(planes, envs, seed=21, mask=stopped, params=op)
                ro[stopped], do[stopped] = 0.0, 0
                resets += int(stopped.sum())
            np.testing.assert_array_equal(ra[k], ro, err_msg=f"reward vs oracle call {call} step {k}")
            np.testing.assert_array_equal(da[k], do, err_msg=f"done vs oracle call {call} step {k}")
            paid += int((ro == 1000.0).sum())
        np.testing.assert_array_equal(obs_a, O.get_state(planes), err_msg=f"obs vs oracle call {call}")
    gp, ge = a_env.download()
    np.testing.assert_array_equal(gp, planes)
    np.testing.assert_array_equal(_envs(ge), envs)
    assert resets > B // 8 and paid > 0, (resets, paid)
    a_env.close()
    b_env.close()


@pytest.mark.parametrize("W,H,B,md", [(10, 10, 4096, None), (10, 10, 4096 + 37, None), (14, 14, 4096 + 3, {"allow_dig_toggle": True, "n_actions": 5}),
                                      (8, 8, 4096 + 11, None), (16, 8, 4100, {"make_rivers": True}), (12, 12, 9000, {"make_rivers": True, "allow_dig_toggle": True, "n_actions": 5}),
                                      (14, 14, 16384 + 77, None)])
def test_resident_step_loop_matches_plain_loop_and_oracle(wf, monkeypatch, W, H, B, md):
    """n-step calls on the rows kernel's RESIDENT step loop (cells in an LDS mirror for the whole call, records handed
    from step to step through LDS): small batches that are not split, ragged last waves, shapes with 2 / 3 / 7
    environments per wave iteration, dig toggle and rivers, auto-resets and containment floods inside the loop - equal to
    the loop through L2 and to the oracle, observation of the final state included."""
    import ctypes
    from wildfire_control_python_amd import _native as N
    n, calls = 16, 5
    toggle = bool(md and md.get("allow_dig_toggle"))
    kw = dict(width=W, height=H, seed=33, auto_reset=True, containment=True, fuel_range=(3, 9), metadata=md)
    monkeypatch.setenv("WF_RESIDENT", "1")                    # (an opt-in experiment: measured no faster than the loop through L2)
    monkeypatch.setenv("WF_MULTI_SMALL", "1")                 # batches that are not split take the step loop too
    a_env = wf.BatchedForestFire(B, **kw)                     # both read at construction
    monkeypatch.delenv("WF_RESIDENT", raising=False)
    b_env = wf.BatchedForestFire(B, **kw)
    monkeypatch.delenv("WF_MULTI_SMALL", raising=False)
    name = a_env._lib.wf_call_kernel_name(a_env._h, n).decode()
    assert "LDS-resident" in name and "LDS-resident" not in b_env._lib.wf_call_kernel_name(b_env._h, n).decode(), name
    a_env.reset(seed=33)
    b_env.reset(seed=33)
    op = O.default_params(containment=1, fuel_lo=3, fuel_hi=9, allow_dig_toggle=int(toggle), make_rivers=int(bool(md and md.get("make_rivers"))))
    planes, envs = O.new_state(B, H, W)
    O.reset(planes, envs, seed=33, params=op)
    rng = np.random.default_rng(B)
    paid = resets = 0
    for call in range(calls):
        a = rng.integers(0, 6, size=(n, B)).astype(np.int8)
        for k in range(0, n, 2):
            a[k, :B // 2] = _circle_actions(envs[:B // 2], W, H, rng)        # (stale after the first step: still ring-ish diggers)
        mask = (1 << n) - 1 if call % 2 == 0 else 0b1011011101101101
        ra, da = np.empty((n, B)), np.empty((n, B), dtype=np.uint8)
        rb, db = np.empty((n, B)), np.empty((n, B), dtype=np.uint8)
        obs_a = np.empty((B, W, H, 3), dtype=np.int8)
        obs_b = np.empty_like(obs_a)
        assert a_env._lib.wf_step_obs(a_env._h, N.ptr(a), n, ctypes.c_uint64(mask), N.ptr(ra), N.ptr(da), N.ptr(obs_a)) == 0
        assert b_env._lib.wf_step_obs(b_env._h, N.ptr(a), n, ctypes.c_uint64(mask), N.ptr(rb), N.ptr(db), N.ptr(obs_b)) == 0
        for k in range(n):
            stopped = (envs["flags"] & O.F_RUNNING) == 0
            ro, do = O.step(planes, envs, a[k], op, do_update=bool((mask >> k) & 1))
            if stopped.any():
                O.reset(planes, envs, seed=33, mask=stopped, params=op)
                ro[stopped], do[stopped] = 0.0, 0
                resets += int(stopped.sum())
            np.testing.assert_array_equal(ra[k], ro, err_msg=f"reward vs oracle call {call} step {k}")
            np.testing.assert_array_equal(da[k], do, err_msg=f"done vs oracle call {call} step {k}")
            paid += int((ro == 1000.0).sum())
        np.testing.assert_array_equal(ra, rb, err_msg=f"reward call {call}")
        np.testing.assert_array_equal(da, db, err_msg=f"done call {call}")
        np.testing.assert_array_equal(obs_a, O.get_state(planes), err_msg=f"obs vs oracle call {call}")
        np.testing.assert_array_equal(obs_a, obs_b, err_msg=f"obs call {call}")
        gp, ge = a_env.download()
        np.testing.assert_array_equal(gp, planes, err_msg=f"planes call {call}")
        np.testing.assert_array_equal(_envs(ge), envs, err_msg=f"envs call {call}")
        # a single step in between: the single-step kernels must find the boards and records the loop left
        a1 = rng.integers(0, 6, size=B).astype(np.int8)
        _, r1, d1, _ = a_env.step(a1)
        b_env.step(a1)
        stopped = (envs["flags"] & O.F_RUNNING) == 0
        ro, do = O.step(planes, envs, a1, op)
        if stopped.any():
            O.reset(planes, envs, seed=33, mask=stopped, params=op)
            ro[stopped], do[stopped] = 0.0, 0
        np.testing.assert_array_equal(r1, ro, err_msg=f"single step after call {call}")
    assert resets > B // 8, resets
    a_env.close()
    b_env.close()


def test_fused_multi_step_observation(wf):
    """wf_step_obs with n_steps > 1: the observation is that of the final state."""
    import ctypes
    from wildfire_control_python_amd import _native as N
    B = 200
    env = wf.BatchedForestFire(B, size=12, fuel_range=(4, 9), auto_reset=True, seed=8)
    env.reset(seed=8)
    a = np.random.default_rng(1).integers(0, 5, size=(24, B)).astype(np.int8)
    for t0 in (0, 8, 16):
        reward = np.empty((8, B)); done = np.empty((8, B), dtype=np.uint8)
        obs = np.empty((B, 12, 12, 3), dtype=np.int8)
        rc = env._lib.wf_step_obs(env._h, N.ptr(np.ascontiguousarray(a[t0:t0 + 8])), 8, ctypes.c_uint64(255),
                                  N.ptr(reward), N.ptr(done), N.ptr(obs))
        assert rc == 0
        np.testing.assert_array_equal(obs, O.get_state(env.download()[0]))
    env.close()


def test_a_speed_two_updates_every_other_step(wf):
    B = 128
    env = wf.BatchedForestFire(B, size=10, metadata={"a_speed": 2, "a_speed_iter": 2})
    env.reset(seed=4)
    op = O.default_params()
    planes, envs = O.new_state(B, 10, 10)
    O.reset(planes, envs, seed=4, params=op)
    rng = np.random.default_rng(8)
    for t in range(60):
        a = rng.integers(0, 4, size=B).astype(np.int8)
        _, r, d, _ = env.step(a)
        ro, do = O.step(planes, envs, a, op, do_update=(t % 2 == 1))   # forest_fire.py:40-43
        gp, ge = env.download()
        np.testing.assert_array_equal(gp, planes, err_msg=f"planes t={t}")
        np.testing.assert_array_equal(_envs(ge), envs, err_msg=f"envs t={t}")
        np.testing.assert_array_equal(r, ro)
    env.close()


def _toggle_script_actions(planes, envs, phase, rng):
    """Host-side scripted agents (test only) that provoke dig() on a BURNING cell: dig mode off, walk next to
    the fire, wait on a grass cell until it ignites under the agent, then toggle digging on (action 4)."""
    B, H, W, _ = planes.shape
    a = np.full(B, 5, dtype=np.int8)
    DX, DY = (0, 0, 1, -1), (-1, 1, 0, 0)
    for i in range(B):
        fl, x, y = int(envs["flags"][i]), int(envs["x"][i]), int(envs["y"][i])
        if phase[i] == 4 and (fl & O.F_ALIVE):               # after the dig on fire: close a ring around the fire
            a[i] = _circle_actions(envs[i:i + 1], W, H, rng)[0] if rng.random() < 0.9 else rng.integers(0, 6)
            continue
        if (fl & (O.F_ALIVE | O.F_RUNNING)) != (O.F_ALIVE | O.F_RUNNING) or phase[i] >= 3:
            a[i] = rng.integers(0, 6)
            continue
        t = planes[i, :, :, 0]
        burning_nb = [d for d in range(4) if 0 <= x + DX[d] < W and 0 <= y + DY[d] < H and t[y + DY[d], x + DX[d]] == 1]
        if phase[i] == 0:
            a[i], phase[i] = (4, 1) if fl & O.F_DIGGING else (5, 1)
        elif t[y, x] == 1:                                   # the cell under the agent burns: dig it (toggle on)
            a[i], phase[i] = 4, 3 + (i % 2)
        elif t[y, x] == 0 and burning_nb:
            a[i] = 5                                         # wait for the own cell to ignite
        else:
            fy, fx = np.nonzero(t == 1)
            if len(fx) == 0:
                a[i] = rng.integers(0, 4)
                continue
            k = np.argmin(np.abs(fx - x) + np.abs(fy - y))
            opts = [d for d in range(4) if d not in burning_nb and 0 <= x + DX[d] < W and 0 <= y + DY[d] < H
                    and abs(fx[k] - x - DX[d]) + abs(fy[k] - y - DY[d]) < abs(fx[k] - x) + abs(fy[k] - y)]
            a[i] = opts[rng.integers(len(opts))] if opts else rng.integers(0, 4)
    return a


@pytest.mark.parametrize("variant", ["group", "rows", "fused", "tiled", "cells"])
def test_dig_toggle_matches_oracle_including_dig_on_a_burning_cell(wf, monkeypatch, variant):
    """allow_dig_toggle (forest_fire.py:35-36, Agent.toggle_digging environment.py:136-138), bit-exact in every
    kernel family - including the corner the reference reaches when digging is toggled ON over a burning cell: the
    cell becomes dirt but stays in burning_cells (boundary code 5), keeps burning and radiating, ends as blocked
    burnt (code 6).  Half of the agents act at random (actions 0..5), half are scripted to provoke it."""
    for k in ("WF_NO_ROWS", "WF_GROUP", "WF_SHARED", "WF_ROWS"):
        monkeypatch.delenv(k, raising=False)
    if variant == "rows":
        monkeypatch.setenv("WF_GROUP", "2")
    W, H = {"tiled": (24, 16), "cells": (11, 13)}.get(variant, (10, 10))
    if variant == "tiled":
        monkeypatch.setenv("WF_FAMILY", "tiled")          # (a grid this small goes to the generic family by itself)
    B, T = 384, 130
    env = wf.BatchedForestFire(B, width=W, height=H, seed=21, containment=True, fuel_range=(12, 30),
                               metadata={"allow_dig_toggle": True, "n_actions": 5})
    if variant in ("group", "rows", "tiled"):
        assert env.step_kernel.startswith("wf_step_" + variant), env.step_kernel
    if variant == "cells":
        assert env.step_kernel == "wf_step_env"
    env.reset(seed=21)
    op = O.default_params(allow_dig_toggle=1, containment=1, fuel_lo=12, fuel_hi=30)
    planes, envs = O.new_state(B, H, W)
    O.reset(planes, envs, seed=21, params=op)
    rng = np.random.default_rng(2)
    phase = np.zeros(B, dtype=np.int64)
    phase[B // 2:] = 3                                           # second half: random actions throughout
    seen5 = seen6 = paid = 0
    t = 0
    while t < T:
        n = 3 if variant == "fused" else 1
        acts = np.empty((n, B), dtype=np.int8)
        oracle_out = []
        for k in range(n):                                       # the script reads the ORACLE's state, step by step
            acts[k] = _toggle_script_actions(planes, envs, phase, rng)
            oracle_out.append(O.step(planes, envs, acts[k], op))
        if variant == "fused":
            r, d = env.step_many(acts)
        else:
            obs, r1, d1, _ = env.step(acts[0])
            r, d = r1[None], d1[None]
            np.testing.assert_array_equal(obs, O.get_state(planes), err_msg=f"obs t={t}")
        for k in range(n):
            np.testing.assert_array_equal(r[k], oracle_out[k][0], err_msg=f"reward t={t + k}")
            np.testing.assert_array_equal(d[k], oracle_out[k][1].astype(bool), err_msg=f"done t={t + k}")
        gp, ge = env.download()
        np.testing.assert_array_equal(gp, planes, err_msg=f"planes t={t}")
        np.testing.assert_array_equal(_envs(ge), envs, err_msg=f"envs t={t}")
        seen5 += int((planes[..., 0] == 5).sum())
        seen6 += int((planes[..., 0] == 6).sum())
        paid += int(sum((o[0] == 1000.0).sum() for o in oracle_out))
        t += n
    assert seen5 > 200 and seen6 > 200, (seen5, seen6)
    np.testing.assert_array_equal(env.get_state(), O.get_state(planes))
    env.close()


def test_auto_reset_matches_oracle_reset(wf):
    B = 256
    env = wf.BatchedForestFire(B, size=10, seed=31, auto_reset=True, fuel_range=(3, 9))
    env.reset(seed=31)
    op = O.default_params(fuel_lo=3, fuel_hi=9)
    planes, envs = O.new_state(B, 10, 10)
    O.reset(planes, envs, seed=31, params=op)
    rng = np.random.default_rng(6)
    resets = 0
    for t in range(80):
        a = rng.integers(0, 4, size=B).astype(np.int8)
        obs, r, d, _ = env.step(a)
        stopped = (envs["flags"] & O.F_RUNNING) == 0
        ro, do = O.step(planes, envs, a, op)
        if stopped.any():           # next-step autoreset: the step resets instead, reward 0, done 0
            O.reset(planes, envs, seed=31, mask=stopped, params=op)
            ro[stopped], do[stopped] = 0.0, 0
            resets += int(stopped.sum())
        gp, ge = env.download()
        np.testing.assert_array_equal(obs, O.get_state(planes), err_msg=f"fused obs t={t}")
        np.testing.assert_array_equal(env.get_state(), obs, err_msg=f"standalone obs t={t}")
        np.testing.assert_array_equal(gp, planes, err_msg=f"planes t={t}")
        np.testing.assert_array_equal(_envs(ge), envs, err_msg=f"envs t={t}")
        np.testing.assert_array_equal(r, ro)
        np.testing.assert_array_equal(d, do.astype(bool))
    assert resets > B
    env.close()


def test_split_launch_equals_single_launch_and_oracle(wf, monkeypatch):
    """Batches >= 16384 on the context's own streams are stepped as two halves on two streams.
    The split must be invisible: bit-identical to one launch (WF_NO_SPLIT=1) and to the oracle,
    including auto-resets in the second half (the stream is keyed by the env's context index)."""
    B, S, T = 16384 + 192, 14, 70          # ragged: the halves are 8320 and 8256 environments
    kw = dict(size=S, seed=77, auto_reset=True, fuel_range=(3, 9))
    monkeypatch.delenv("WF_NO_SPLIT", raising=False)
    a_env = wf.BatchedForestFire(B, **kw)
    monkeypatch.setenv("WF_NO_SPLIT", "1")               # read at construction
    b_env = wf.BatchedForestFire(B, **kw)
    monkeypatch.delenv("WF_NO_SPLIT", raising=False)
    assert a_env.step_launches == 2 and b_env.step_launches == 1
    a_env.reset(seed=77)
    b_env.reset(seed=77)
    op = O.default_params(fuel_lo=3, fuel_hi=9)
    idx = np.concatenate([np.arange(0, B, 97), np.arange(B - 40, B)])
    p0, e0 = a_env.download()
    rng = np.random.default_rng(9)
    # the oracle replays a sample under the sample's own global ids: one single-env oracle state each
    samples = []
    for i in idx:
        samples.append([np.ascontiguousarray(p0[i:i + 1]), np.ascontiguousarray(_envs(e0[i:i + 1])).astype(O.ENV_DTYPE), op])
    resets = 0
    for t in range(T):
        a = rng.integers(0, 5, size=B).astype(np.int8)
        obs_a, r_a, d_a, _ = a_env.step(a)
        obs_b, r_b, d_b, _ = b_env.step(a)
        np.testing.assert_array_equal(obs_a, obs_b, err_msg=f"obs t={t}")
        np.testing.assert_array_equal(r_a, r_b, err_msg=f"reward t={t}")
        np.testing.assert_array_equal(d_a, d_b, err_msg=f"done t={t}")
        for k, i in enumerate(idx):
            pl, en, op_i = samples[k]
            stopped = (en["flags"] & O.F_RUNNING) == 0
            ro, do = O.step(pl, en, a[i:i + 1], op_i)
            if stopped.any():
                O.reset(pl, en, seed=77, mask=stopped, env_id0=int(i), params=op)
                ro[:], do[:] = 0.0, 0
                resets += 1
            assert r_a[i] == ro[0] and bool(d_a[i]) == bool(do[0]), (t, i)
    pa, ea = a_env.download()
    pb, eb = b_env.download()
    np.testing.assert_array_equal(pa, pb)
    np.testing.assert_array_equal(ea, eb)
    for k, i in enumerate(idx):
        np.testing.assert_array_equal(pa[i], samples[k][0][0], err_msg=f"env {i}")
    assert resets > len(idx)
    a_env.close()
    b_env.close()


def test_frozen_after_done_and_reset_mask(wf):
    B = 64
    env = wf.BatchedForestFire(B, size=10, fuel_range=(2, 4))
    env.reset(seed=1)
    for _ in range(40):
        _, r, d, _ = env.step(None)
    assert d.all()
    p0, e0 = env.download()
    obs, r, d, _ = env.step(np.zeros(B, dtype=np.int8))
    p1, e1 = env.download()
    np.testing.assert_array_equal(p0, p1)
    np.testing.assert_array_equal(obs, O.get_state(p1), err_msg="fused obs of frozen envs")
    assert (r == 0).all() and d.all()
    mask = np.zeros(B, dtype=np.uint8)
    mask[::2] = 1
    env.reset(mask=mask, seed=2)
    p2, e2 = env.download()
    np.testing.assert_array_equal(p2[1::2], p1[1::2])
    assert ((e2["flags"][::2] & 16) != 0).all() and ((e2["flags"][1::2] & 16) == 0).all()
    assert (e2["episode"][::2] == 2).all() and (e2["episode"][1::2] == 1).all()
    env.close()


def test_edge_moves_rejected_and_agent_plane_blank(wf):
    """Rejected move blanks the agent_pos plane until the next successful move
    (environment.py:142); water is not traversable (:229-230)."""
    S = 10
    planes = np.zeros((1, S, S, 4), dtype=np.int8)
    planes[..., 1] = 20
    planes[0, 5, 5, 0], planes[0, 5, 5, 2] = 1, 25
    planes[0, 0, 0, 0], planes[0, 0, 0, 3] = 3, 1          # agent in the corner on dirt
    planes[0, 0, 1, 0] = 4                                  # water to the east
    env = wf.BatchedForestFire(1, size=S)
    env.upload(planes)
    o = planes.copy()
    oe = env.envs_from_planes(planes).astype(O.ENV_DTYPE)
    for i, a in enumerate((0, 3, 2, 1, 0)):   # N (oob), W (oob), E (water), S (ok), N (back)
        obs, r, d, _ = env.step(np.array([a], dtype=np.int8))
        O.step(o, oe, np.array([a], dtype=np.int8))
        gp, ge = env.download()
        np.testing.assert_array_equal(gp, o)
        np.testing.assert_array_equal(_envs(ge), oe)
        if i < 3:               # rejected: agent stays in the corner, plane blank
            assert (ge["x"][0], ge["y"][0]) == (0, 0) and gp[0, :, :, 3].sum() == 0
    assert (gp[0, :, :, 3].sum() == 1) and gp[0, 0, 0, 3] == 1
    env.close()


# ------------------------------------------------------- single-env adapter (drop-in)
def test_adapter_replays_reference_noop_episode(wf):
    """np.random.seed(0); sim.reset() spawns where the reference does, and the whole
    no-op episode (state, reward, done per step) matches the reference fixture."""
    import random
    for size in (10, 14):
        ep = Episodes([f for f in fixture_files(size) if f.endswith("_noop.npz")][0])
        sim = wf.ForestFire(size=size)
        np.random.seed(0)
        random.seed(0)
        state = sim.reset()
        assert (sim.W.agents[0].x, sim.W.agents[0].y) == tuple(ep.d["init_agent"][0, :2])
        assert state.shape == (size, size, 3) and state.dtype == np.float64
        total = 0
        for t in range(ep.T):
            state, reward, done, info = sim.step(" ")
            np.testing.assert_array_equal(state, ep.d["obs"][t].astype(np.float64))
            assert reward == ep.d["reward"][t] and done == bool(ep.d["done"][t]) and info == {}
            total += reward
        assert done and not sim.W.RUNNING and total == ep.d["reward"].sum()
        assert sim.W.WIDTH == size and sim.W.DEPTH == 3 and sim.n_actions == 4
        txt = sim.render()
        assert txt.count("\n") == size + 1 and "#" in txt
        sim.close()


def test_adapter_rivers_match_reference_draw_for_draw(wf):
    """make_rivers=True: seeded reset reproduces the reference's river AND spawn point
    (reset_map environment.py:69-95 consumes the numpy stream before get_agent_location)."""
    import random
    for size in (10, 14):
        ep = Episodes([f for f in fixture_files(size) if f.endswith("_river.npz")][0])
        sim = wf.ForestFire(size=size, metadata={"make_rivers": True})
        crossing = 0
        for i, seed in enumerate(ep.d["seeds"]):
            np.random.seed(int(seed))
            random.seed(int(seed))
            state = sim.reset()
            planes, envs = sim._b.download()
            np.testing.assert_array_equal(planes[0], ep.init_planes[i], err_msg=f"size {size} seed {seed}")
            assert (envs["x"][0], envs["y"][0]) == tuple(ep.d["init_agent"][i, :2])
            c = size // 2
            if planes[0, c, c, 0] == 13:
                # the river runs through the fire cell: it shows as fire, blocks (plane 2 = 0), and World.env reads
                # type fire with infinite fire mobility like the reference's (environment.py:85, :241)
                crossing += 1
                assert state[c, c, 1] == 1 and state[c, c, 2] == 0
                assert sim.W.env[c, c, sim.layer["type"]] == 1 and np.isinf(sim.W.env[c, c, sim.layer["fire_mobility"]])
                assert sim.W.is_burning((c, c)) and (c, c) in sim.W.burning_cells
            if i % 4 == 3:                    # whole episodes through the single-environment adapter, crossing ones included
                lo, hi = int(ep.ep_start[i]), int(ep.ep_start[i + 1])
                for t in range(lo, hi):
                    state, reward, done, _ = sim.step(int(ep.d["action"][t]))
                    np.testing.assert_array_equal(state, ep.d["obs"][t].astype(np.float64), err_msg=f"seed {seed} step {t - lo}")
                    assert reward == ep.d["reward"][t] and done == bool(ep.d["done"][t])
        assert crossing >= 4
        sim.close()


def test_adapter_a_speed_counter_survives_reset_like_the_reference(wf):
    """METADATA['a_speed_iter'] is module-global in the reference and reset() never touches it (forest_fire.py:40-43,
    :52-54): with a_speed = 2 an episode with an odd number of steps leaves the next one out of phase.  Ten reference
    episodes played back to back in ONE adapter object, spawn points from numpy's stream like the reference's."""
    import random
    for size in (10, 14):
        ep = Episodes([f for f in fixture_files(size) if f.endswith("_aspeed2.npz")][0])
        sim = wf.ForestFire(size=size, metadata={"a_speed": 2, "a_speed_iter": int(ep.a_speed_iter0[0])})
        phases = set()
        for i, seed in enumerate(ep.d["seeds"]):
            np.random.seed(int(seed))
            random.seed(int(seed))
            sim.reset()
            assert sim.METADATA["a_speed_iter"] == ep.a_speed_iter0[i], f"episode {i}: the counter must carry over"
            phases.add(int(sim.METADATA["a_speed_iter"]))
            planes, _ = sim._b.download()
            np.testing.assert_array_equal(planes[0], ep.init_planes[i])
            lo, hi = int(ep.ep_start[i]), int(ep.ep_start[i + 1])
            for t in range(lo, hi):
                state, reward, done, _ = sim.step(int(ep.d["action"][t]))
                np.testing.assert_array_equal(state, ep.d["obs"][t].astype(np.float64), err_msg=f"episode {i} step {t - lo}")
                assert reward == ep.d["reward"][t] and done == bool(ep.d["done"][t]), (i, t - lo)
            planes, _ = sim._b.download()
            np.testing.assert_array_equal(planes[0], ep.d["planes"][hi - 1])
        assert phases == {1, 2}
        sim.close()


def test_adapter_without_arguments_shares_the_module_metadata(wf):
    """ForestFire() works on the package's module-global METADATA like the reference's (forest_fire.py:20): the
    a_speed counter is shared by every such object; objects built with arguments own a copy."""
    from wildfire_control_python_amd import constants
    a, b = wf.ForestFire(), wf.ForestFire()
    assert a.METADATA is constants.METADATA and b.METADATA is constants.METADATA
    c = wf.ForestFire(size=10)
    assert c.METADATA is not constants.METADATA
    old = dict(constants.METADATA)
    try:
        constants.METADATA["a_speed"] = 2
        constants.METADATA["a_speed_iter"] = 2
        a.step(5)                                           # counter 2 -> 1: no update
        assert constants.METADATA["a_speed_iter"] == 1
        b.step(5)                                           # the OTHER object's step runs the update (counter 1 -> 0 -> 2)
        assert constants.METADATA["a_speed_iter"] == 2
        assert a._b.download_envs()["tick"][0] == 0 and b._b.download_envs()["tick"][0] == 1
    finally:
        constants.METADATA.clear()
        constants.METADATA.update(old)
    for s in (a, b, c):
        s.close()


def test_adapter_string_and_int_actions(wf):
    sim = wf.ForestFire(size=10)
    np.random.seed(5)
    sim.reset()
    x0, y0 = sim.W.agents[0].x, sim.W.agents[0].y
    sim.step("E")
    assert (sim.W.agents[0].x, sim.W.agents[0].y) == (x0 + 1, y0)
    sim.step(3)
    assert (sim.W.agents[0].x, sim.W.agents[0].y) == (x0, y0)
    sim.step(np.int64(1))
    assert (sim.W.agents[0].x, sim.W.agents[0].y) == (x0, y0 + 1)
    assert sim.W.env.shape == (10, 10, 9) and sim.W.env[x0, y0, 0] == 3
    assert len(sim.W.border_points) == 40
    sim.close()


# --------------------------------------------------------- full-size properties (C3)
def test_full_size_properties_65536x14x14(wf):
    """BASELINE config 3 at full size: (i) determinism - two contexts fed the same
    inputs agree bit for bit; (ii) shard independence - env i behaves the same inside
    a 65536-batch and inside a 4096-batch holding global ids [8192, 12288);
    (iii) conservation laws of the rule; (iv) a strided sample equals the oracle."""
    B, S, T = 65536, 14, 24
    a = np.random.default_rng(0).integers(0, 4, size=(T, B)).astype(np.int8)
    big = wf.BatchedForestFire(B, size=S, seed=3, fuel_range=(10, 40))
    big2 = wf.BatchedForestFire(B, size=S, seed=3, fuel_range=(10, 40))
    small = wf.BatchedForestFire(4096, size=S, seed=3, fuel_range=(10, 40), env_id0=8192)
    for e in (big, big2, small):
        e.reset(seed=3)
    p0, e0 = big.download()
    idx = np.arange(0, B, 512)
    op = O.default_params(fuel_lo=10, fuel_hi=40)
    sp, se = np.ascontiguousarray(p0[idx]), np.ascontiguousarray(e0[idx]).astype(O.ENV_DTYPE)
    se["flags"] &= VIS
    prev = p0
    for t0 in range(0, T, 8):
        rb, db = big.step_many(a[t0:t0 + 8])
        rb2, _ = big2.step_many(a[t0:t0 + 8])
        rs, _ = small.step_many(a[t0:t0 + 8, 8192:12288])
        np.testing.assert_array_equal(rb, rb2)
        np.testing.assert_array_equal(rb[:, 8192:12288], rs)
        for k in range(8):
            ro, _ = O.step(sp, se, a[t0 + k, idx], op)
            np.testing.assert_array_equal(rb[k, idx], ro)
        p, e = big.download()
        np.testing.assert_array_equal(p[idx], sp)
        # fuel never increases, heat never decreases on grass, burnt/dirt are absorbing
        assert (p[..., 1] <= prev[..., 1]).all()
        assert ((prev[..., 0] == 2) <= (p[..., 0] >= 2)).all()
        assert ((prev[..., 0] == 3) <= (p[..., 0] == 3)).all()
        assert (p[..., 2] >= prev[..., 2]).all()
        assert (p[..., 3].reshape(B, -1).sum(axis=1) <= 1).all()
        prev = p
    ps, es = small.download()
    np.testing.assert_array_equal(prev[8192:12288], ps)
    p2, _ = big2.download()
    np.testing.assert_array_equal(prev, p2)
    for e in (big, big2, small):
        e.close()


# ------------------------------------------------- BASELINE configs at their stated sizes
def test_config2_4096x10x10_noop_16_ticks_per_call(wf):
    """BASELINE configs[1] exactly: 4 096 batched 10x10 grids, random fuel U{10..40}, no agent action, 16 ticks per
    call (fused ticks, grids in registers between ticks; since round 4 as four part launches of wf_step_ticks on the
    context's streams): whole state, rewards and done flags vs the oracle."""
    B, S, n = 4096, 10, 16
    env = wf.BatchedForestFire(B, size=S, seed=1234, fuel_range=(10, 40))
    assert env._lib.wf_call_launches(env._h, n) == 4
    env.reset(seed=1234)
    op = O.default_params(fuel_lo=10, fuel_hi=40)
    planes, envs = O.new_state(B, S, S)
    O.reset(planes, envs, seed=1234, params=op)
    for call in range(21):                                       # 336 ticks: every episode ends (the last one at tick 323)
        r, d = env.step_many(None, n)
        for k in range(n):
            ro, do = O.step(planes, envs, None, op)
            np.testing.assert_array_equal(r[k], ro, err_msg=f"reward call {call} tick {k}")
            np.testing.assert_array_equal(d[k], do.astype(bool), err_msg=f"done call {call} tick {k}")
        gp, ge = env.download()
        np.testing.assert_array_equal(gp, planes, err_msg=f"planes call {call}")
        np.testing.assert_array_equal(_envs(ge), envs, err_msg=f"envs call {call}")
    assert d[-1].all()
    np.testing.assert_array_equal(env.get_state(), O.get_state(planes))
    env.close()


def test_config4_1024x256x256_full_size(wf, monkeypatch):
    """BASELINE configs[3] at its stated size: 1 024 batched 256x256 grids, ~1 % of the cells pre-ignited, fuel
    U{10..40}, as bench.py runs it: through the asynchronous device API the job is three shares of the batch on three
    streams, each wf_tiled_prologue -> wf_step_tiled per step (a job of this many 256x256 grids keeps the four-wave
    workgroups, four per environment; fewer such grids step in one launch of 16-wave workgroups: the big-grid shape
    table above).  A WF_NO_SPLIT twin stepped through the host API with the observation must agree bit for bit on
    every plane, a strided sample of 32 environments (all shares, the seams and the last env) is replayed by the oracle,
    and the monotonicity laws of the rule are checked on ALL cells.  30 no-op ticks (the configuration), then 12 steps
    with agents digging."""
    import torch
    B, S = 1024, 256
    kw = dict(size=S, seed=44, fuel_range=(10, 40), ignite_ppm=10000, containment=True)
    monkeypatch.delenv("WF_NO_SPLIT", raising=False)
    a_env = wf.BatchedForestFire(B, **kw)
    monkeypatch.setenv("WF_NO_SPLIT", "1")
    b_env = wf.BatchedForestFire(B, **kw)
    monkeypatch.delenv("WF_NO_SPLIT", raising=False)
    assert a_env.step_launches == 3 and b_env.step_launches == 1 and a_env.step_kernel == "wf_step_tiled"
    dev = torch.device("cuda", 0)
    d_r = torch.empty((1, B), dtype=torch.float64, device=dev)
    d_d = torch.empty((1, B), dtype=torch.uint8, device=dev)
    a_env.reset(seed=44)
    b_env.reset(seed=44)
    idx = np.unique(np.concatenate([np.arange(0, B, 37), [510, 511, 512, 513, B - 1]]))
    assert len(idx) >= 32
    op = O.default_params(fuel_lo=10, fuel_hi=40, ignite_ppm=10000, containment=1)
    p0, e0 = a_env.download()
    assert 0.008 < (p0[..., 0] == 1).mean() < 0.012               # ~1 % of the cells burn from tick 0
    sp, se = np.ascontiguousarray(p0[idx]), _envs(e0[idx]).astype(O.ENV_DTYPE)
    # the sample's reset itself: oracle reset of those global ids
    for k, i in enumerate(idx):
        pl, en = O.new_state(1, S, S)
        O.reset(pl, en, seed=44, env_id0=int(i), params=op)
        np.testing.assert_array_equal(pl[0], sp[k], err_msg=f"reset env {i}")
    rng = np.random.default_rng(44)
    prev = p0
    for t in range(42):
        a = None if t < 30 else rng.integers(0, 5, size=B).astype(np.int8)
        d_a = None if a is None else torch.from_numpy(a).to(dev)
        torch.cuda.synchronize()
        a_env.step_device(d_a, 1, d_r, d_d, update_mask=1)
        a_env.sync()
        ra, da = d_r.cpu().numpy()[0], d_d.cpu().numpy()[0].astype(bool)
        obs_b, rb, db, _ = b_env.step(a)
        np.testing.assert_array_equal(ra, rb, err_msg=f"split vs single: reward t={t}")
        np.testing.assert_array_equal(da, db, err_msg=f"split vs single: done t={t}")
        ro, do = O.step(sp, se, None if a is None else a[idx], op)
        np.testing.assert_array_equal(ra[idx], ro, err_msg=f"reward vs oracle t={t}")
        np.testing.assert_array_equal(da[idx], do.astype(bool), err_msg=f"done vs oracle t={t}")
        if t % 7 == 6 or t == 41:
            p, e = a_env.download()
            np.testing.assert_array_equal(p[idx], sp, err_msg=f"planes vs oracle t={t}")
            np.testing.assert_array_equal(_envs(e[idx]), se, err_msg=f"envs vs oracle t={t}")
            # fuel never increases, heat never decreases, burnt / dirt are absorbing, at most one agent cell
            assert (p[..., 1] <= prev[..., 1]).all()
            assert (p[..., 2] >= prev[..., 2]).all()
            assert ((prev[..., 0] == 2) <= (p[..., 0] >= 2)).all()
            assert ((prev[..., 0] == 3) <= (p[..., 0] == 3)).all()
            assert (p[..., 3].reshape(B, -1).sum(axis=1) <= 1).all()
            prev = p
    pb, eb = b_env.download()
    np.testing.assert_array_equal(obs_b[idx], O.get_state(sp), err_msg="observation of the last step")
    np.testing.assert_array_equal(prev, pb, err_msg="split vs single launch: planes")
    np.testing.assert_array_equal(e, eb, err_msg="split vs single launch: records")
    assert (prev[..., 0] == 1).mean() > 0.02                      # the fronts have spread
    a_env.close()
    b_env.close()


def test_config5_last_shard_125000_envs_global_ids(wf):
    """BASELINE configs[4], the LAST of the 8 shards: 125 000 environments holding global ids 875 000 .. 999 999
    (ragged halves of 62 528 / 62 472 environments), bench inputs (fuel U{10..40}, uniform N/S/E/W, containment,
    device auto-reset).  A strided sample is replayed by the oracle under the sample's global ids, and a
    65 536-environment context holding ids 900 000 .. 965 535 must hold exactly the same state."""
    B, S, T, ID0 = 125000, 14, 48, 875000
    OFF, BS = 25000, 65536
    kw = dict(size=S, seed=7, auto_reset=True, containment=True, fuel_range=(10, 40))
    big = wf.BatchedForestFire(B, env_id0=ID0, **kw)
    sub = wf.BatchedForestFire(BS, env_id0=ID0 + OFF, **kw)
    assert big.step_launches == 2 and big.step_kernel.startswith("wf_step_rows")
    big.reset(seed=7)
    sub.reset(seed=7)
    op = O.default_params(containment=1, fuel_lo=10, fuel_hi=40)
    idx = np.unique(np.concatenate([np.arange(0, B, 1009), [62527, 62528, 62529, B - 2, B - 1]]))
    p0, e0 = big.download()
    samples = []
    for i in idx:
        pl, en = O.new_state(1, S, S)
        O.reset(pl, en, seed=7, env_id0=ID0 + int(i), params=op)
        np.testing.assert_array_equal(pl[0], p0[i], err_msg=f"reset env {i}")
        samples.append((pl, en))
    rng = np.random.default_rng(7)
    resets = 0
    for t in range(T):
        a = rng.integers(0, 4, size=B).astype(np.int8)
        _, r, d, _ = big.step(a)
        _, rs, ds, _ = sub.step(a[OFF:OFF + BS])
        np.testing.assert_array_equal(r[OFF:OFF + BS], rs, err_msg=f"shard vs sub-context: reward t={t}")
        np.testing.assert_array_equal(d[OFF:OFF + BS], ds, err_msg=f"shard vs sub-context: done t={t}")
        for k, i in enumerate(idx):
            pl, en = samples[k]
            stopped = (en["flags"] & O.F_RUNNING) == 0
            ro, do = O.step(pl, en, a[i:i + 1], op)
            if stopped.any():
                O.reset(pl, en, seed=7, mask=stopped, env_id0=ID0 + int(i), params=op)
                ro[:], do[:] = 0.0, 0
                resets += 1
            assert r[i] == ro[0] and bool(d[i]) == bool(do[0]), (t, i, r[i], ro[0])
    p, e = big.download()
    ps, es = sub.download()
    np.testing.assert_array_equal(p[OFF:OFF + BS], ps)
    np.testing.assert_array_equal(_envs(e[OFF:OFF + BS]), _envs(es))
    for k, i in enumerate(idx):
        np.testing.assert_array_equal(p[i], samples[k][0][0], err_msg=f"env {i}")
        np.testing.assert_array_equal(_envs(e[i:i + 1]), samples[k][1], err_msg=f"record {i}")
    assert resets > len(idx) // 2
    big.close()
    sub.close()


@pytest.mark.parametrize("B,split", [(8193, None), (8255, None), (4 * 4096 + 130, "4")])
def test_multi_step_split_with_a_ragged_last_part(wf, monkeypatch, B, split):
    """A multi-step call of a split context whose LAST part is smaller than 4 096 environments (8 193 -> 4 160 +
    4 033; WF_SPLIT=4 with 16 514 -> 3 x 4 160 + 4 034): every part must run the same kernel with the context's
    row stride for actions / reward / done and the context-level environment index for the reset stream."""
    monkeypatch.delenv("WF_SPLIT", raising=False)
    if split:
        monkeypatch.setenv("WF_SPLIT", split)
    S, n = 14, 5
    env = wf.BatchedForestFire(B, size=S, seed=13, auto_reset=True, containment=True, fuel_range=(3, 8))
    monkeypatch.delenv("WF_SPLIT", raising=False)
    parts = int(split or 2)
    assert env.step_launches == parts and env._lib.wf_call_launches(env._h, n) == parts
    env.reset(seed=13)
    op = O.default_params(containment=1, fuel_lo=3, fuel_hi=8)
    planes, envs = O.new_state(B, S, S)
    O.reset(planes, envs, seed=13, params=op)
    rng = np.random.default_rng(B)
    resets = 0
    for call in range(8):
        a = rng.integers(0, 5, size=(n, B)).astype(np.int8)
        r, d = env.step_many(a)
        for k in range(n):
            stopped = (envs["flags"] & O.F_RUNNING) == 0
            ro, do = O.step(planes, envs, a[k], op)
            if stopped.any():
                O.reset(planes, envs, seed=13, mask=stopped, params=op)
                ro[stopped], do[stopped] = 0.0, 0
                resets += int(stopped.sum())
            np.testing.assert_array_equal(r[k], ro, err_msg=f"reward call {call} step {k}")
            np.testing.assert_array_equal(d[k], do.astype(bool), err_msg=f"done call {call} step {k}")
        gp, ge = env.download()
        np.testing.assert_array_equal(gp, planes, err_msg=f"planes call {call}")
        np.testing.assert_array_equal(_envs(ge), envs, err_msg=f"envs call {call}")
    assert resets > B // 2
    env.close()


# ------------------------------------------------- the reference's own published numbers
@pytest.mark.parametrize("size", [10, 14])
def test_published_baseline_reward_is_reproduced(wf, size):
    """Plots/results_first.txt:2,20: the clockwise random-walk baseline (DQN.py:286-389) earns 1130.2 (10x10) /
    1151.4 (14x14) per episode on average over the reference's 100 000 episodes.  >= 131 072 episodes of the same
    policy on the HIP environment (device reset, default fuel 20, containment on, episode ends on `done`) must
    land within 4 standard errors of it: reset spawn law, move / dig / death, containment payment and the
    burnt-out reward, end to end against numbers the reference's authors wrote down."""
    from tests.policy_util import (REFERENCE_BASELINE_MEAN, baseline_tolerance, run_baseline_episodes)
    B, rounds = 32768, 4
    env = wf.BatchedForestFire(B, size=size, seed=99, containment=True)
    rng = np.random.default_rng(size)
    obs = [None]

    def reset():
        obs[0] = env.reset()

    def state():
        e = env.download_envs()
        return e["x"], e["y"], (e["flags"] & 1) != 0, obs[0][..., 1].transpose(0, 2, 1) != 0

    def step(a):
        o, r, d, _ = env.step(a)
        obs[0] = o
        return r, d

    totals, died = [], []
    for _ in range(rounds):
        t, dd, _ = run_baseline_episodes(reset, step, state, B, rng)
        totals.append(t)
        died.append(dd)
    total, died = np.concatenate(totals), np.concatenate(died)
    tol = baseline_tolerance(total.std(), len(total))
    assert abs(total.mean() - REFERENCE_BASELINE_MEAN[size]) < tol, (total.mean(), REFERENCE_BASELINE_MEAN[size], tol)
    assert 0.01 < died.mean() < 0.15
    env.close()


def test_adapter_runs_the_reference_baseline_loop(wf):
    """The reference's baseline loop, statement for statement what DQN.collect_memories(perform_baseline=True) and
    choose_randomwalk_action do (DQN.py:286-389), against the single-environment ForestFire adapter: sim.reset(),
    sim.W.agents, .x/.y, agents[0].fire_in_direction(action), sim.step(action) -> [state, reward, done, {}]."""
    from tests.policy_util import REFERENCE_BASELINE_MEAN
    size = 10
    sim = wf.ForestFire(size=size)
    np.random.seed(3)
    key_map = {"N": 0, "S": 1, "E": 2, "W": 3}
    totals, deaths, avoided = [], 0, 0
    for episode in range(400):
        state = sim.reset()
        assert state.shape == (size, size, 3)
        done, total_reward = False, 0
        while not done:
            if not sim.W.agents:
                action = 0
            else:
                width, height = sim.W.WIDTH, sim.W.HEIGHT
                agent_x, agent_y = sim.W.agents[0].x, sim.W.agents[0].y
                mid_x, mid_y = int(width / 2), int(height / 2)
                count = 0
                while True:
                    if agent_x >= mid_x and agent_y > mid_y:
                        possible_actions = ["S", "W"]
                    if agent_x > mid_x and agent_y <= mid_y:
                        possible_actions = ["S", "E"]
                    if agent_x <= mid_x and agent_y < mid_y:
                        possible_actions = ["N", "E"]
                    if agent_x < mid_x and agent_y >= mid_y:
                        possible_actions = ["N", "W"]
                    action = key_map[np.random.choice(possible_actions)]
                    fire_at_loc = sim.W.agents[0].fire_in_direction(action)
                    if not fire_at_loc or count > 10:
                        break
                    count += 1
                    avoided += 1
            sprime, reward, done, _ = sim.step(action)
            total_reward += reward
        totals.append(total_reward)
        deaths += len(sim.W.agents) == 0
    totals = np.array(totals, dtype=np.float64)
    # 400 episodes: a loose check of the level (std ~ 285 -> 4 s.e. ~ 57) - the tight one is the batched test above
    assert abs(totals.mean() - REFERENCE_BASELINE_MEAN[size]) < 4 * totals.std() / np.sqrt(len(totals)) + 5
    assert avoided > 50, "fire_in_direction never reported fire: the avoidance branch was not exercised"
    assert 0 < deaths < 60
    sim.close()


def test_fire_in_direction_on_a_constructed_state(wf):
    """Agent.fire_in_direction (environment.py:158-160): burning neighbour yes / no, and off-grid is False."""
    sim = wf.ForestFire(size=10)
    planes = np.zeros((1, 10, 10, 4), dtype=np.int8)
    planes[..., 1] = 20
    planes[0, 0, 4, 0], planes[0, 0, 4, 3] = 3, 1               # agent at x=4, y=0 (top row) on dirt
    planes[0, 0, 5, 0], planes[0, 0, 5, 2] = 1, 25              # fire to the east (x=5, y=0)
    planes[0, 1, 4, 0] = 2                                      # burnt cell to the south
    sim._b.upload(planes)
    sim.W._refresh()
    ag = sim.W.agents[0]
    assert (ag.x, ag.y) == (4, 0)
    assert ag.fire_in_direction(2) and ag.fire_in_direction("E")
    assert not ag.fire_in_direction(3) and not ag.fire_in_direction("S")
    assert not ag.fire_in_direction(0)                           # off the grid
    sim.close()


# ---------------------------------------------------------------- consumers of the path
@pytest.mark.parametrize("onegen", [True, False])
def test_full_size_every_environment_matches_oracle(wf, monkeypatch, onegen):
    """BASELINE config 3 exactly as bench.py runs it: 65 536 x 14x14 through the asynchronous DEVICE API on the context's own
    streams, auto-reset, containment.  Since round 4 a single step of this size goes out as THREE part launches of the rows
    kernel with 4 environments per wave and the agent phases shared by a workgroup's 4 waves (onegen; WF_NO_ONEGEN=1: the
    two half-batch launches of rounds 1-3).  Every environment against the oracle: rewards and done flags of every step, the
    fused observation of some, a synchronous host-buffer step (one launch, issue priority by progress) in between, the whole
    state at the end."""
    import torch
    for k in ("WF_GROUP", "WF_SHARED", "WF_SPLIT", "WF_NO_SPLIT", "WF_NO_ONEGEN", "WF_NO_ROWS"):
        monkeypatch.delenv(k, raising=False)
    if not onegen:
        monkeypatch.setenv("WF_NO_ONEGEN", "1")
    B, S, T = 65536, 14, 40
    env = wf.BatchedForestFire(B, size=S, seed=5, auto_reset=True, containment=True, fuel_range=(3, 25), ignite_ppm=3000)
    monkeypatch.delenv("WF_NO_ONEGEN", raising=False)
    assert env.step_kernel == ("wf_step_rows<14,14,2,true>" if onegen else "wf_step_rows<14,14,2,false>"), env.step_kernel
    assert env.step_launches == (3 if onegen else 2)
    env.reset(seed=5)
    op = O.default_params(containment=1, fuel_lo=3, fuel_hi=25, ignite_ppm=3000)
    planes, envs = O.new_state(B, S, S)
    O.reset(planes, envs, seed=5, params=op)
    rng = np.random.default_rng(5)
    dev = torch.device("cuda", 0)
    acts = rng.integers(0, 6, size=(T, B)).astype(np.int8)
    d_act = torch.from_numpy(acts).to(dev)
    d_r = torch.empty((T, B), dtype=torch.float64, device=dev)
    d_d = torch.empty((T, B), dtype=torch.uint8, device=dev)
    d_obs = torch.empty((B, S, S, 3), dtype=torch.int8, device=dev)
    torch.cuda.synchronize()
    resets = 0
    exp = []
    host = {}
    for t in range(T):
        stopped = (envs["flags"] & O.F_RUNNING) == 0
        ro, do = O.step(planes, envs, acts[t], op)
        if stopped.any():
            O.reset(planes, envs, seed=5, mask=stopped, params=op)
            ro[stopped], do[stopped] = 0.0, 0
            resets += int(stopped.sum())
        exp.append((ro, do))
        if t == 17:                                  # synchronous host-buffer step: ONE launch over the whole batch
            _, r, d, _ = env.step(acts[t])
            host[t] = (r, d.astype(np.uint8))
        elif t % 9 == 4:                             # the step kernel writes World.get_state() itself
            env.step_device(d_act[t], 1, d_r[t:t + 1], d_d[t:t + 1], update_mask=1, d_obs=d_obs)
            env.sync()
            np.testing.assert_array_equal(d_obs.cpu().numpy(), O.get_state(planes), err_msg=f"obs t={t}")
        else:
            env.step_device(d_act[t], 1, d_r[t:t + 1], d_d[t:t + 1], update_mask=1)
    gp, ge = env.download()
    hr, hd = d_r.cpu().numpy(), d_d.cpu().numpy()
    for t in range(T):
        r, d = host.get(t, (hr[t], hd[t]))
        np.testing.assert_array_equal(r, exp[t][0], err_msg=f"reward t={t}")
        np.testing.assert_array_equal(d, exp[t][1], err_msg=f"done t={t}")
    np.testing.assert_array_equal(gp, planes)
    np.testing.assert_array_equal(_envs(ge), envs)
    np.testing.assert_array_equal(env.get_state(), O.get_state(planes))
    assert resets > B // 4
    env.close()


def test_full_size_wind_random_matches_oracle(wf):
    """The headline size with METADATA['wind'] = "random": 65 536 x 14x14 on the wave family's wind kernel, as two shares
    of the batch on two streams (12.8 M cells), auto-reset, containment: every environment against the oracle - rewards and
    done flags each step, planes, records, wind codes and per-direction heat counts at the end."""
    from tests.golden_util import WIND_SPEEDS
    B, S, T = 65536, 14, 30
    env = wf.BatchedForestFire(B, size=S, seed=5, auto_reset=True, containment=True, fuel_range=(3, 25), ignite_ppm=3000,
                               metadata={"wind": "random"})
    assert env.step_launches == 2 and env.step_kernel == "wf_step_group<14,14,wind>", env.step_kernel
    env.reset(seed=5)
    op = O.default_params(containment=1, fuel_lo=3, fuel_hi=25, ignite_ppm=3000, wind_mode=2, wind_speeds=WIND_SPEEDS + [0.0])
    planes, envs = O.new_state(B, S, S)
    counts = np.zeros(planes.shape, np.uint8)
    O.reset(planes, envs, seed=5, params=op, counts=counts)
    rng = np.random.default_rng(5)
    resets = 0
    for t in range(T):
        a = rng.integers(0, 6, size=B).astype(np.int8)
        obs, r, d, _ = env.step(a)
        stopped = (envs["flags"] & O.F_RUNNING) == 0
        ro, do = O.step(planes, envs, a, op, counts=counts)
        if stopped.any():
            O.reset(planes, envs, seed=5, mask=stopped, params=op, counts=counts)
            ro[stopped], do[stopped] = 0.0, 0
            resets += int(stopped.sum())
        np.testing.assert_array_equal(r, ro, err_msg=f"reward t={t}")
        np.testing.assert_array_equal(d, do.astype(bool), err_msg=f"done t={t}")
        if t % 10 == 9:          # (written by the step kernel itself)
            np.testing.assert_array_equal(obs, O.get_state(planes), err_msg=f"obs t={t}")
    gp, ge = env.download()
    np.testing.assert_array_equal(gp, planes)
    np.testing.assert_array_equal(_envs(ge), envs)
    np.testing.assert_array_equal(env.download_counts(), counts)
    assert resets > B // 4            # a third of the winds has speed 0: the fire dies where it started
    env.close()


@pytest.mark.parametrize("algo", ["DQN", "SARSA", "DDQN", "DSARSA"])
def test_batched_learner_runs_on_the_device_api(wf, algo):
    """The four value-based learners of the reference, batched, drive the env through
    wf_step_obs_device with torch tensors (no host round trip) and learn something finite."""
    import torch
    env = wf.BatchedForestFire(512, size=10, auto_reset=True, seed=3)
    L = wf.BatchedLearner(env, algo, replay_capacity=8192, batch_size=64, store_per_step=256)
    w0 = [p.detach().clone() for p in L.model.parameters()]
    logs = L.learn(60)
    assert len(logs["loss"]) > 40 and all(np.isfinite(logs["loss"]))
    assert any(not torch.equal(a, b.detach()) for a, b in zip(w0, L.model.parameters()))
    assert L.m_len > 1000 and logs["finished_episodes"] >= 0
    # the observations the learner saw are the environment's observations
    np.testing.assert_array_equal(L.obs.cpu().numpy(), env.get_state())
    env.close()


# ---------------------------------------------------------------- API edge cases
@pytest.mark.parametrize("variant", ["rows-g2", "group-g4"])
def test_upload_and_kernel_switch_invalidate_reach_boards(wf, monkeypatch, variant):
    """The border-reachability boards are caches in kernel-private layouts.  Editing the planes from the host
    (a ring of dirt dropped around the fire) and switching between the kernels that keep different copies
    (single step <-> fused multi-step) must never leave a stale board behind: containment is paid exactly when
    the oracle's flood pays it."""
    for k in ("WF_NO_ROWS", "WF_GROUP", "WF_SHARED", "WF_ROWS"):
        monkeypatch.delenv(k, raising=False)
    for k, v in KERNEL_VARIANTS[variant].items():
        monkeypatch.setenv(k, v)
    B, S = 96, 14
    env = wf.BatchedForestFire(B, size=S, seed=4, containment=True, fuel_range=(30, 40))
    env.reset(seed=4)
    op = O.default_params(containment=1, fuel_lo=30, fuel_hi=40)
    planes, envs = O.new_state(B, S, S)
    O.reset(planes, envs, seed=4, params=op)
    rng = np.random.default_rng(2)
    paid = 0

    def both(a):
        nonlocal paid
        _, r, d, _ = env.step(a)
        ro, do = O.step(planes, envs, a, op)
        np.testing.assert_array_equal(r, ro)
        np.testing.assert_array_equal(d, do.astype(bool))
        paid += int((ro == 1000.0).sum())

    for t in range(6):                                     # boards get computed / validated
        both(rng.integers(0, 4, size=B).astype(np.int8))
    # host edit: a closed ring of dirt at distance 4 around the centre of every second environment
    gp, ge = env.download()
    c = S // 2
    for e in range(0, B, 2):
        for k in range(-4, 5):
            for (yy, xx) in ((c - 4, c + k), (c + 4, c + k), (c + k, c - 4), (c + k, c + 4)):
                if gp[e, yy, xx, 0] in (0, 2):
                    gp[e, yy, xx, 0] = 3
                    planes[e, yy, xx, 0] = 3
    env.upload(gp, ge)        # records as downloaded: their internal board-valid bits must not survive the upload
    for t in range(10):
        both(np.full(B, 5, dtype=np.int8))                 # no-ops: agents inside the ring stay put
    assert paid >= B // 4, f"{paid}: the uploaded rings should have contained fires"
    # switch kernels back and forth: 3 fused steps (4-cells-per-lane boards), then single steps again
    for rep in range(3):
        a3 = rng.integers(0, 4, size=(3, B)).astype(np.int8)
        r3, d3 = env.step_many(a3)
        for k in range(3):
            ro, do = O.step(planes, envs, a3[k], op)
            np.testing.assert_array_equal(r3[k], ro)
            np.testing.assert_array_equal(d3[k], do.astype(bool))
        for t in range(4):
            both(rng.integers(0, 4, size=B).astype(np.int8))
    gp, ge = env.download()
    np.testing.assert_array_equal(gp, planes)
    np.testing.assert_array_equal(_envs(ge), envs)
    env.close()


@pytest.mark.parametrize("W,H,B", [(256, 256, 6), (15, 15, 300), (13, 11, 257), (30, 23, 64), (51, 47, 33), (64, 70, 17), (100, 150, 5),
                                   (1, 1, 100), (1, 6, 9), (6, 1, 9), (3, 3, 1), (2, 9, 50), (40, 3, 7), (5, 2000, 2), (24, 24, 40)])
def test_get_state_kernel_any_shape_any_alignment(wf, W, H, B):
    """World.get_state() through the LDS-tiled transpose kernel (wf_get_state_cols): shapes that are no multiple of
    anything, one column per workgroup up to 64, columns that span environments, and output pointers at every
    byte alignment (a split step hands the kernel `obs + lo * 3 * W * H`)."""
    import torch
    env = wf.BatchedForestFire(B, width=W, height=H, containment=False)
    rng = np.random.default_rng(W * 1000 + H)
    planes = np.zeros((B, H, W, 4), dtype=np.int8)
    planes[..., 0] = rng.choice([0, 1, 2, 3, 4], size=(B, H, W), p=[0.5, 0.15, 0.1, 0.15, 0.1])
    planes[..., 1] = rng.integers(1, 40, size=(B, H, W))
    planes[..., 3] = rng.random((B, H, W)) < 0.02
    envs = np.zeros(B, dtype=O.ENV_DTYPE)
    envs["flags"] = O.F_RUNNING
    env.upload(planes, envs)
    want = O.get_state(planes)
    np.testing.assert_array_equal(env.get_state(), want)
    n = want.size
    buf = torch.zeros(n + 16, dtype=torch.int8, device="cuda")
    for off in (1, 2, 3, 5):
        buf.fill_(77)
        torch.cuda.synchronize()
        env.get_state_device(buf.data_ptr() + off)
        env.sync()
        got = buf.cpu().numpy()
        np.testing.assert_array_equal(got[off:off + n].reshape(want.shape), want, err_msg=f"offset {off}")
        assert (got[:off] == 77).all() and (got[off + n:] == 77).all(), f"offset {off}: bytes outside the buffer were written"
    env.close()


@pytest.mark.parametrize("B,W,H", [(1, 14, 14), (7, 10, 10), (300, 13, 11), (5000, 14, 14), (40, 64, 64)])
def test_step_download_equals_step_then_download(wf, B, W, H):
    """wf_step_download: the step and the state it leaves under one synchronisation - small contexts through the
    copy-kernel path, larger ones by DMA; same results as wf_step followed by wf_download."""
    a_env = wf.BatchedForestFire(B, width=W, height=H, seed=4, auto_reset=True, fuel_range=(4, 12))
    b_env = wf.BatchedForestFire(B, width=W, height=H, seed=4, auto_reset=True, fuel_range=(4, 12))
    a_env.reset(seed=4)
    b_env.reset(seed=4)
    rng = np.random.default_rng(B)
    for t in range(25):
        a = rng.integers(0, 6, size=B).astype(np.int8)
        r1, d1, p1, e1 = a_env.step_download(a)
        r2, d2 = b_env.step_many(a)
        p2, e2 = b_env.download()
        np.testing.assert_array_equal(r1, r2[0])
        np.testing.assert_array_equal(d1, d2[0])
        np.testing.assert_array_equal(p1, p2, err_msg=f"planes t={t}")
        np.testing.assert_array_equal(e1, e2, err_msg=f"envs t={t}")
    a_env.close()
    b_env.close()


@pytest.mark.parametrize("B", [1, 300, 16384, 70000])
def test_lent_host_buffers_spare_the_copies_and_change_nothing(wf, B):
    """wf_host_staging lends the pinned buffers wf_step works on; a call that is handed them back reads and writes them in
    place.  Same rewards / done flags / state as caller-owned arrays, below and above the size from which actions and
    results travel through device mirrors (16 384 actions per call)."""
    a_env = wf.BatchedForestFire(B, size=14, seed=3, auto_reset=True, fuel_range=(4, 12))
    b_env = wf.BatchedForestFire(B, size=14, seed=3, auto_reset=True, fuel_range=(4, 12))
    a_env.reset(seed=3)
    b_env.reset(seed=3)
    rng = np.random.default_rng(B)
    for n in (1, 1, 3, 1):
        acts, rew, done = a_env.host_staging(n)
        assert acts.shape == rew.shape == done.shape == (n, B) and rew.dtype == np.float64
        a = rng.integers(0, 6, size=(n, B)).astype(np.int8)
        acts[:] = a                                           # written straight into the pinned action block
        r1, d1 = a_env.step_many(acts, borrow=True)
        assert r1 is rew and d1 is done
        r2, d2 = b_env.step_many(a)
        np.testing.assert_array_equal(r1, r2)
        np.testing.assert_array_equal(d1.astype(bool), d2)
    pa, ea = a_env.download()
    pb, eb = b_env.download()
    np.testing.assert_array_equal(pa, pb)
    np.testing.assert_array_equal(ea, eb)
    a_env.close()
    b_env.close()


@pytest.mark.parametrize("S", [14, 13])
def test_wind_count_bytes_saturate_at_255_like_the_oracle(wf, S):
    """A count byte that has reached 255 takes no further addend (the oracle's `if (cnt < 255) cnt += 1`): uploaded counts
    of 250..255 next to fires that never ignite anything (wind speed 0: a third of the winds "random" draws) - the wave
    family's wind kernel (14x14) and the generic family (13x13), 40 steps."""
    from tests.golden_util import WIND_SPEEDS
    B = 96
    env = wf.BatchedForestFire(B, size=S, seed=2, fuel_range=(30, 60), ignite_ppm=20000, metadata={"wind": "random"})
    env.reset(seed=2)
    op = O.default_params(fuel_lo=30, fuel_hi=60, ignite_ppm=20000, wind_mode=2, wind_speeds=WIND_SPEEDS + [0.0])
    planes, envs = O.new_state(B, S, S)
    counts = np.zeros(planes.shape, np.uint8)
    O.reset(planes, envs, seed=2, params=op, counts=counts)
    rng = np.random.default_rng(S)
    envs["wind"] = O.wind_code(1, -1, 0)                       # speed index 0 = no heat at all: nothing ever ignites
    counts[:] = rng.integers(250, 256, size=counts.shape).astype(np.uint8) * (planes[..., 0:1] == 0)
    env.upload(planes, envs, counts)
    for t in range(40):
        a = rng.integers(0, 6, size=B).astype(np.int8)
        _, r, d, _ = env.step(a)
        ro, do = O.step(planes, envs, a, op, counts=counts)
        np.testing.assert_array_equal(r, ro, err_msg=f"reward t={t}")
        np.testing.assert_array_equal(env.download_counts(), counts, err_msg=f"counts t={t}")
    gp, _ = env.download()
    np.testing.assert_array_equal(gp, planes)
    assert (counts == 255).sum() > (counts == 254).sum() > 0 and not (planes[..., 2] == O.HEAT_IGNITED).any()
    env.close()


def test_upload_of_planes_resets_the_wind_counts(wf):
    """Planes and per-direction heat counts form one state (include/wildfire.h: wf_upload): uploading planes alone starts
    from zero heat, it never keeps the counts of the grid that was there before."""
    B, S = 64, 10
    env = wf.BatchedForestFire(B, size=S, metadata={"wind": [0.85, (1, 0)]})
    env.reset(seed=3)
    for _ in range(30):
        env.step(np.full(B, 5, dtype=np.int8))
    counts = env.download_counts()
    assert counts.any()
    planes, envs = env.download()
    env.upload(planes, envs)                              # planes only
    assert not env.download_counts().any()
    env.upload(planes, envs, counts)                      # a full snapshot: counts restored after the planes
    np.testing.assert_array_equal(env.download_counts(), counts)
    # the temp layer of the adapter's World.env is finite everywhere (a cell ignited by accumulated heat reads above the threshold)
    sim = wf.ForestFire(size=S, metadata={"wind": [0.85, (1, 0)]})
    for _ in range(40):
        sim.step(5)
    temp = sim.W.env[:, :, sim.layer["temp"]]
    assert np.isfinite(temp).all() and (temp[sim.W.env[:, :, sim.layer["type"]] == 2] > 3).all()
    sim.close()
    env.close()


def test_api_edge_cases(wf):
    """Odd batch sizes, NULL optional buffers, 64 fused steps, partial uploads, error codes."""
    import ctypes
    from wildfire_control_python_amd import _native as N
    lib = N.lib()
    for B in (1, 3, 7, 61, 257):                       # not multiples of the 4-wave workgroup / group size
        env = wf.BatchedForestFire(B, size=10, seed=B, fuel_range=(5, 9))
        env.reset(seed=B)
        op = O.default_params(fuel_lo=5, fuel_hi=9)
        planes, envs = O.new_state(B, 10, 10)
        O.reset(planes, envs, seed=B, params=op)
        a = np.random.default_rng(B).integers(0, 4, size=(64, B)).astype(np.int8)
        # 64 fused steps with reward/done omitted (NULL), then compare state with 64 oracle steps
        assert lib.wf_step(env._h, N.ptr(np.ascontiguousarray(a)), 64, ctypes.c_uint64(2**64 - 1), None, None) == 0
        for t in range(64):
            O.step(planes, envs, a[t], op)
        gp, ge = env.download()
        np.testing.assert_array_equal(gp, planes, err_msg=f"B={B} fused 64")
        np.testing.assert_array_equal(_envs(ge), envs)
        # NULL actions = no-op for everyone
        r = np.empty(B); d = np.empty(B, dtype=np.uint8)
        assert lib.wf_step(env._h, None, 1, ctypes.c_uint64(1), N.ptr(r), N.ptr(d)) == 0
        ro, do = O.step(planes, envs, None, op)
        np.testing.assert_array_equal(r, ro)
        np.testing.assert_array_equal(d, do)
        # n_steps out of range, NULL context
        assert lib.wf_step(env._h, None, 0, ctypes.c_uint64(1), None, None) == -1
        assert lib.wf_step(env._h, None, 65, ctypes.c_uint64(1), None, None) == -1
        assert lib.wf_get_state(env._h, None) == -1
        # partial upload: scalars only (planes untouched)
        ge2 = ge.copy()
        ge2["flags"] &= np.uint16(~np.uint16(16))       # stop everything
        env.upload(None, ge2)
        _, r2, d2, _ = env.step(None)
        assert d2.all() and (r2 == 0).all()
        np.testing.assert_array_equal(env.download()[0], planes, err_msg="a scalars-only upload must leave the planes alone")
        env.close()
        env.close()                                      # idempotent
    # invalid device index / shapes
    h = ctypes.c_void_p()
    p = N.default_params()
    assert lib.wf_create(ctypes.byref(h), 99, 4, 10, 10, ctypes.byref(p)) == -1
    assert lib.wf_create(ctypes.byref(h), 0, 4, 0, 10, ctypes.byref(p)) == -1
    # a record pointing outside the grid is refused, nothing is uploaded
    env = wf.BatchedForestFire(5, size=10)
    env.reset(seed=1)
    gp, ge = env.download()
    bad = ge.copy()
    bad["x"][3] = 10
    assert lib.wf_upload(env._h, None, N.ptr(bad)) == -1
    bad["x"][3], bad["y"][0] = 0, 200
    assert lib.wf_upload(env._h, N.ptr(np.zeros_like(gp)), N.ptr(bad)) == -1
    weird = gp.copy()
    weird[2, 3, 4, 0] = 5                                # burning dirt: a code only allow_dig_toggle knows
    assert lib.wf_upload(env._h, N.ptr(weird), None) == -1
    weird[2, 3, 4, 0] = -1
    assert lib.wf_upload(env._h, N.ptr(weird), N.ptr(ge)) == -1
    weird[2, 3, 4, 0] = gp[2, 3, 4, 0]
    weird[4, 7, 7, 3] = 2                                # agent_pos is 0 or 1 (utility.py:121)
    assert lib.wf_upload(env._h, N.ptr(weird), None) == -1
    gp2, ge2 = env.download()
    np.testing.assert_array_equal(gp2, gp)
    np.testing.assert_array_equal(ge2, ge)
    env.close()
    # a grid that cannot hold the spawn circle (radius 3 around the centre): steps, but neither resets nor auto-resets
    assert lib.wf_create(ctypes.byref(h), 0, 4, 6, 9, ctypes.byref(p)) == 0
    assert lib.wf_reset(h, None, ctypes.c_uint64(1)) == -2
    assert lib.wf_step(h, None, 1, ctypes.c_uint64(1), None, None) == 0
    p.auto_reset = 1
    assert lib.wf_set_params(h, ctypes.byref(p)) == -2
    assert lib.wf_destroy(h) == 0
    assert lib.wf_create(ctypes.byref(h), 0, 4, 9, 6, ctypes.byref(p)) == -2


def test_external_stream_and_timer(wf):
    """wf_set_stream with a torch stream: env work is ordered with torch work on that stream."""
    import torch
    env = wf.BatchedForestFire(1024, size=14, seed=1)
    env.reset(seed=1)
    st = torch.cuda.Stream()
    env.set_stream(st.cuda_stream)
    with torch.cuda.stream(st):
        act = torch.randint(0, 4, (1024,), dtype=torch.int8, device="cuda")
        rew = torch.empty((1, 1024), dtype=torch.float64, device="cuda")
        done = torch.empty((1, 1024), dtype=torch.uint8, device="cuda")
        obs = torch.empty((1024, 14, 14, 3), dtype=torch.int8, device="cuda")
        env.timer_start()
        for _ in range(5):
            env.step_device(act, 1, rew, done, d_obs=obs)
        ms = env.timer_stop()
        total = rew.sum()                               # torch op on the same stream sees the results
    st.synchronize()
    r = rew.cpu().numpy()[0]
    assert ms > 0 and float(total.item()) == float(r.sum()) and set(np.unique(r)) <= {-1000.0, -1.0, 0.0, 1000.0}
    assert (r != 0).sum() > 512                         # most agents are still alive and were stepped
    np.testing.assert_array_equal(obs.cpu().numpy(), env.get_state())
    env.close()


def test_plain_c_client_matches_python_path(wf, tmp_path):
    """The same run through examples/native_step.c (plain C against include/wildfire.h) and through the Python
    mirror: identical final state (FNV-1a checksum of the planes) and reward sum."""
    import subprocess
    from tests.golden_util import build_native_example
    exe = build_native_example(tmp_path)
    B, T = 1000, 60
    r = subprocess.run([exe, str(B), str(T)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    env = wf.BatchedForestFire(B, size=14, auto_reset=True, fuel_range=(10, 40))
    env.reset(seed=1234)
    lcg = np.uint32(12345)
    reward_sum = 0.0
    with np.errstate(over="ignore"):
        for t in range(T):
            a = np.empty(B, dtype=np.int8)
            for i in range(B):
                lcg = np.uint32(lcg * np.uint32(1664525) + np.uint32(1013904223))
                a[i] = int(lcg) >> 30
            _, rew, _, _ = env.step(a)
            reward_sum += float(rew.sum())
    planes, _ = env.download()
    h = 1469598103934665603
    for b in planes.astype(np.uint8).ravel().tolist():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert f"state checksum {h:016x}" in out, out
    assert f"reward sum {reward_sum:.1f}" in out, out
    env.close()


def test_bench_contract_single_and_two_ranks(wf):
    """bench.py as the driver runs it: one JSON line with the contract's keys, extra config legs with their own
    rooflines; `--gpus 2` on this one-GPU box is refused loudly; the N = 2 launch through torch.distributed.run
    with --share-gpu (two ranks on this box's one GPU -> gloo carries the barrier, labelled shared_card) shards
    the global batch and reports the whole-job rate; a WORLD_SIZE that contradicts --gpus is an error."""
    import json
    import os
    import subprocess
    import sys
    import torch
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(repo, "bench.py"), "--steps", "30", "--warmup", "5", "--batch", "16384", "--no-cpu-baseline"]
    r1 = subprocess.run(base + ["--gpus", "1", "--legs", "c2,c4"], capture_output=True, text=True, timeout=300, cwd=repo)
    assert r1.returncode == 0, r1.stderr[-2000:]
    lines = [l for l in r1.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r1.stdout
    d1 = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d1, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d1["roofline"], k
    assert d1["n_gpus"] == 1 and d1["scaling"] == "weak" and d1["vs_baseline"] is None
    # 30 steps are far below the minimum duration of the timed region: repeated, and said so
    assert d1["steps_requested"] == 30 and d1["steps"] % 30 == 0 and d1["steps"] * d1["ms_per_step"] >= 49.0
    assert abs(d1["roofline"]["frac"] - d1["roofline"]["achieved"] / d1["roofline"]["peak"]) < 1e-9
    assert "custom size" in d1["config"]["workload"] and "configs[2]" not in d1["config"]["workload"]
    for leg, tag in (("c2", "configs[1]"), ("c4", "configs[3]")):
        assert tag in d1["configs"][leg]["workload"] and 0 < d1["configs"][leg]["roofline"]["frac"] < 1.2, d1["configs"][leg]
    assert d1["configs"]["c4"]["roofline"]["kernel"] == "wf_step_tiled"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None)
    if torch.cuda.device_count() < 2:
        r = subprocess.run(base + ["--gpus", "2"], capture_output=True, text=True, timeout=120, cwd=repo, env=env)
        assert r.returncode != 0 and "GPU(s) visible" in r.stderr and not r.stdout.strip(), (r.returncode, r.stderr[-500:])
    r = subprocess.run(base + ["--gpus", "2"], capture_output=True, text=True, timeout=120, cwd=repo,
                       env=dict(env, WORLD_SIZE="4", RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", "29533"] + base[1:] + ["--gpus", "2", "--share-gpu", "--no-extras"],
                        capture_output=True, text=True, timeout=400, cwd=repo, env=env)
    assert r2.returncode == 0, r2.stderr[-2000:]
    lines = [l for l in r2.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r2.stdout
    d2 = json.loads(lines[0])
    assert d2["n_gpus"] == 2 and d2["config"]["global_batch"] == 2 * 16384 and d2["value"] > 0
    assert d2["shared_card"] is True and d2["rccl_ranks"] is None
    # the same through bench.py's own per-GPU launcher (no torchrun)
    r3 = subprocess.run(base + ["--gpus", "2", "--share-gpu", "--no-extras"], capture_output=True, text=True, timeout=400, cwd=repo, env=env)
    assert r3.returncode == 0, r3.stderr[-2000:]
    d3 = json.loads([l for l in r3.stdout.splitlines() if l.strip().startswith("{")][0])
    assert d3["n_gpus"] == 2 and d3["config"]["global_batch"] == 2 * 16384


def test_bench_rehearsal_of_config5_shards_and_a_parent_without_torch(wf):
    """BASELINE configs[4] is 8 shards of 125 000 environments, one process per GPU.  On this one-GPU box: bench.py's own
    launcher with 4 rank processes of that shard size on the one card (the pool allows 6 GPU processes at a time and this
    test runner is one of them), gloo for the barrier.  The parent never imports torch (-X importtime)."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "-X", "importtime", os.path.join(repo, "bench.py"), "--gpus", "4", "--share-gpu", "--batch", "125000",
                        "--no-extras", "--no-cpu-baseline", "--steps", "20", "--warmup", "5", "--min-ms", "5"],
                       capture_output=True, text=True, timeout=600, cwd=repo, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["config"]["global_batch"] == 500000 and d["shared_card"] is True and d["value"] > 0
    assert "configs[4]" in d["config"]["workload"]
    # -X importtime is inherited by nobody: the lines on stderr are the PARENT's imports
    assert not [l for l in r.stderr.splitlines() if "import time:" in l and "| torch" in l]


def test_sharded_forest_fire_equals_one_context(wf):
    """ShardedForestFire: one process, one context per shard, asynchronous device calls, no exchange between shards.
    Four shards (all on device 0 here) of a 10 007-environment batch equal ONE context of the whole batch bit for bit:
    reset streams are keyed by the global environment id."""
    import torch
    G, S = 10007, 14
    # (a_speed = 2: ONE counter drives every shard, whichever front end - host or device - issues the step)
    kw = dict(seed=5, auto_reset=True, fuel_range=(6, 14), containment=True, metadata={"a_speed": 2})
    sh = wf.ShardedForestFire(G, size=S, devices=(0, 0, 0, 0), **kw)
    one = wf.BatchedForestFire(G, size=S, **kw)
    np.testing.assert_array_equal(sh.reset(seed=9), one.reset(seed=9))
    rng = np.random.default_rng(1)
    dev = torch.device("cuda", 0)
    d_act = [torch.empty(hi - lo, dtype=torch.int8, device=dev) for lo, hi in sh.bounds]
    d_rew = [torch.empty(hi - lo, dtype=torch.float64, device=dev) for lo, hi in sh.bounds]
    d_done = [torch.empty(hi - lo, dtype=torch.uint8, device=dev) for lo, hi in sh.bounds]
    d_obs = [torch.empty((hi - lo, S, S, 3), dtype=torch.int8, device=dev) for lo, hi in sh.bounds]
    for t in range(70):
        a = rng.integers(0, 5, size=G).astype(np.int8)
        obs1, r1, dn1, _ = one.step(a)
        if t % 2:
            obs, r, dn, _ = sh.step(a)
        else:                                                     # the asynchronous device path
            for (lo, hi), da in zip(sh.bounds, d_act):
                da.copy_(torch.from_numpy(a[lo:hi]))
            torch.cuda.synchronize()
            sh.step_device(d_act, 1, d_rew, d_done, d_obs)
            sh.sync()
            obs = np.concatenate([o.cpu().numpy() for o in d_obs])
            r = np.concatenate([x.cpu().numpy() for x in d_rew])
            dn = np.concatenate([x.cpu().numpy() for x in d_done]).astype(bool)
        np.testing.assert_array_equal(r, r1, err_msg=f"reward t={t}")
        np.testing.assert_array_equal(dn, dn1, err_msg=f"done t={t}")
        np.testing.assert_array_equal(obs, obs1, err_msg=f"obs t={t}")
    p, e = sh.download()
    p1, e1 = one.download()
    np.testing.assert_array_equal(p, p1)
    np.testing.assert_array_equal(_envs(e), _envs(e1))
    assert (e["episode"] > 1).any()
    sh.close()
    one.close()


# ------------------------------------------------------------------ arbitrary uploaded states
def _random_state(B, H, W, rng, toggle, rivers=False):
    """Arbitrary (valid) boundary states, not the ones a reset produces: scattered fire with any fuel (0 and 1
    included: burn-out on the first tick), grass at every heat level below ignition, burnt / dirt / water patches,
    with the toggle the two extra codes, agents anywhere (visible or not, digging or not)."""
    planes = np.zeros((B, H, W, 4), dtype=np.int8)
    kinds = [0, 1, 2, 3, 4] + ([5] if toggle else []) + ([6] if toggle or rivers else []) + ([13] if rivers else [])
    probs = np.array([0.62, 0.10, 0.08, 0.08, 0.06] + ([0.03] if toggle else []) + ([0.03] if toggle or rivers else []) + ([0.03] if rivers else []))
    t = rng.choice(kinds, size=(B, H, W), p=probs / probs.sum()).astype(np.int8)
    t[:, 0, 0] = np.where(t[:, 0, 0] == 4, 0, t[:, 0, 0])          # (a place for the agent even on a 1x1 grid)
    fuel = rng.integers(1, 45, size=(B, H, W))
    burning = (t == 1) | (t == 5) | (t == 13)
    fuel[burning] = rng.integers(-3, 30, size=int(burning.sum()))     # <= 0: poked states only, burn out at once
    heat = rng.integers(0, 19, size=(B, H, W))
    heat[(t == 1) | (t == 13)] = rng.choice([19, 25], size=int(((t == 1) | (t == 13)).sum()))
    planes[..., 0], planes[..., 1], planes[..., 2] = t, fuel, heat
    envs = np.zeros(B, dtype=O.ENV_DTYPE)
    for b in range(B):
        ok = np.argwhere(t[b] != 4)
        y, x = ok[rng.integers(len(ok))]
        envs[b]["x"], envs[b]["y"] = x, y
        fl = O.F_ALIVE | O.F_RUNNING | (O.F_DIGGING if rng.random() < 0.6 else 0)
        if rng.random() < 0.8:
            fl |= O.F_VISIBLE
            planes[b, y, x, 3] = 1
        border = np.concatenate([burning[b, 0], burning[b, -1], burning[b, :, 0], burning[b, :, -1]]).any()
        if border or rng.random() < 0.1:
            fl |= O.F_BORDER
        envs[b]["flags"] = fl
    return planes, envs


@pytest.mark.parametrize("W,H,toggle", [(14, 14, False), (10, 10, True), (12, 12, True), (16, 8, False), (24, 16, True),
                                        (64, 70, False), (11, 13, True),
                                        # grids too small for World.reset()'s spawn circle: uploaded states only
                                        (5, 4, True), (3, 3, False), (1, 6, False), (6, 1, True), (2, 9, False), (1, 1, False),
                                        (4, 4, True), (40, 3, False), (36, 30, True), (100, 70, True), (51, 47, False),
                                        # make_rivers contexts: blocked fire cells (code 13) anywhere, with and without the toggle
                                        (14, 14, "rivers"), (10, 10, "both"), (12, 12, "rivers"), (16, 16, "both"), (32, 32, "rivers"),
                                        (11, 13, "both"), (64, 70, "rivers"), (100, 70, "both")])
def test_arbitrary_uploaded_states_match_oracle(wf, W, H, toggle):
    """Parity on states no reset would produce (uploaded through wf_upload): every kernel family, with and without
    the dig toggle / rivers, 40 steps of random actions from each of several random states."""
    B = 320
    rivers = toggle in ("rivers", "both")
    toggle = toggle is True or toggle == "both"
    md = {}
    if toggle:
        md.update({"allow_dig_toggle": True, "n_actions": 5})
    if rivers:
        md["make_rivers"] = True
    env = wf.BatchedForestFire(B, width=W, height=H, containment=True, metadata=md or None)
    op = O.default_params(allow_dig_toggle=int(toggle), containment=1, make_rivers=int(rivers))
    rng = np.random.default_rng(W * 131 + H)
    for rep in range(3):
        planes, envs = _random_state(B, H, W, rng, toggle, rivers)
        env.upload(planes, envs)
        for t in range(40):
            a = rng.integers(0, 6, size=B).astype(np.int8)
            if t % 9 == 8:                                       # a fused multi-step call in between
                a2 = rng.integers(0, 6, size=(3, B)).astype(np.int8)
                r3, d3 = env.step_many(a2)
                for k in range(3):
                    ro, do = O.step(planes, envs, a2[k], op)
                    np.testing.assert_array_equal(r3[k], ro, err_msg=f"rep {rep} t={t} fused step {k}: reward")
                    np.testing.assert_array_equal(d3[k], do.astype(bool), err_msg=f"rep {rep} t={t} fused step {k}: done")
            else:
                obs, r, d, _ = env.step(a)
                ro, do = O.step(planes, envs, a, op)
                np.testing.assert_array_equal(r, ro, err_msg=f"rep {rep} t={t}: reward")
                np.testing.assert_array_equal(d, do.astype(bool), err_msg=f"rep {rep} t={t}: done")
                np.testing.assert_array_equal(obs, O.get_state(planes), err_msg=f"rep {rep} t={t}: obs")
            gp, ge = env.download()
            np.testing.assert_array_equal(gp, planes, err_msg=f"rep {rep} t={t}: planes")
            np.testing.assert_array_equal(_envs(ge), envs, err_msg=f"rep {rep} t={t}: envs")
    env.close()


# ------------------------------------------------ the n-step kernel of one environment per wave (wf_step_ticks)
@pytest.mark.parametrize("W,H", [(8, 8), (10, 10), (12, 12), (14, 14), (16, 16), (16, 8), (8, 16), (20, 10), (10, 20)])
@pytest.mark.parametrize("md", [None, {"make_rivers": True, "allow_dig_toggle": True, "n_actions": 5}], ids=["default", "rivers+toggle"])
def test_n_step_kernel_every_shape_and_variant(wf, monkeypatch, W, H, md):
    """wf_step_ticks (packed cells, agent / flags / reward on the scalar unit) in its four loop variants - with and
    without actions, every step updating or not (a_speed-style masks) - on every wave-family shape: calls of 2..64
    steps (blocks of 16 steps: actions loaded, rewards / done flags stored per block; an auto-reset that falls on a
    block's first or last step), auto-reset and frozen environments, containment floods and payments, rivers and the
    dig toggle, step by step against the oracle; the round-1 kernel (WF_FUSED_LEGACY=1) leaves the same state."""
    for k in ("WF_FUSED_LEGACY", "WF_MULTI_SMALL", "WF_NO_MULTI", "WF_GROUP"):
        monkeypatch.delenv(k, raising=False)
    B = 200
    rare = md is not None
    paid_total = 0
    for auto in (True, False):
        kw = dict(width=W, height=H, seed=31, auto_reset=auto, containment=True, fuel_range=(2, 7), metadata=md)
        env = wf.BatchedForestFire(B, **kw)
        assert env._lib.wf_call_kernel_name(env._h, 5).decode().startswith("wf_step_ticks"), env._lib.wf_call_kernel_name(env._h, 5)
        monkeypatch.setenv("WF_FUSED_LEGACY", "1")
        old = wf.BatchedForestFire(B, **kw)
        monkeypatch.delenv("WF_FUSED_LEGACY")
        assert old._lib.wf_call_kernel_name(old._h, 5).decode().startswith("wf_step_wave")
        env.reset(seed=31)
        old.reset(seed=31)
        op = O.default_params(containment=1, fuel_lo=2, fuel_hi=7, allow_dig_toggle=int(rare), make_rivers=int(rare))
        planes, envs = O.new_state(B, H, W)
        O.reset(planes, envs, seed=31, params=op)
        rng = np.random.default_rng(W * 11 + H + int(rare))
        paid = resets = frozen = 0
        for call, n in enumerate((2, 16, 17, 5, 33, 64, 15, 1, 32, 48)):
            noop = call % 4 == 3                            # every fourth call carries no actions
            mask = (1 << n) - 1 if call % 3 != 1 else int(rng.integers(1, 1 << min(n, 62))) | 1
            acts = None if noop else np.empty((n, B), dtype=np.int8)
            outs = []
            for k in range(n):
                a = None
                if not noop:
                    a = rng.integers(0, 6, size=B).astype(np.int8)
                    a[:B // 2] = _circle_actions(envs[:B // 2], W, H, rng)
                    acts[k] = a
                stopped = (envs["flags"] & O.F_RUNNING) == 0
                ro, do = O.step(planes, envs, a, op, do_update=bool((mask >> k) & 1))
                if stopped.any() and auto:
                    O.reset(planes, envs, seed=31, mask=stopped, params=op)
                    ro[stopped], do[stopped] = 0.0, 0
                    resets += int(stopped.sum())
                frozen += int(stopped.sum()) if not auto else 0
                outs.append((ro, do))
                paid += int((ro == 1000.0).sum())
            r, d = env.step_many(acts, n, update_mask=mask) if noop else env.step_many(acts, update_mask=mask)
            r0, d0 = old.step_many(acts, n, update_mask=mask) if noop else old.step_many(acts, update_mask=mask)
            for k in range(n):
                np.testing.assert_array_equal(r[k], outs[k][0], err_msg=f"auto={auto} call {call} (n={n}) step {k}: reward")
                np.testing.assert_array_equal(d[k], outs[k][1].astype(bool), err_msg=f"auto={auto} call {call} (n={n}) step {k}: done")
            np.testing.assert_array_equal(r0, r)
            np.testing.assert_array_equal(d0, d)
            gp, ge = env.download()
            np.testing.assert_array_equal(gp, planes, err_msg=f"auto={auto} call {call}: planes")
            np.testing.assert_array_equal(_envs(ge), envs, err_msg=f"auto={auto} call {call}: envs")
            gp0, ge0 = old.download()
            np.testing.assert_array_equal(gp0, gp)
            np.testing.assert_array_equal(_envs(ge0), _envs(ge))
        np.testing.assert_array_equal(env.get_state(), O.get_state(planes))
        assert resets > B if auto else frozen > B, (auto, paid, resets, frozen)
        paid_total += paid
        env.close()
        old.close()
    if W * H <= 144:                                         # (the circling agents close their rings in time on the small grids)
        assert paid_total > 0


@pytest.mark.parametrize("W,H,B,noop", [(16, 16, 8192 + 17, False), (20, 10, 16384, False), (14, 14, 8192 + 5, True), (10, 10, 16384, True)])
def test_n_step_calls_of_split_contexts_on_the_one_wave_kernel(wf, monkeypatch, W, H, B, noop):
    """Batches that are stepped as two part launches per single step: an n-step call goes to ONE wf_step_ticks launch where
    that was measured faster - shapes without a rows-layout step loop, and calls without actions - with results equal
    to the oracle's, and single steps in between (the part launches) continue from the state it leaves."""
    for k in ("WF_FUSED_LEGACY", "WF_MULTI_SMALL", "WF_NO_MULTI", "WF_GROUP", "WF_NO_SPLIT"):
        monkeypatch.delenv(k, raising=False)
    env = wf.BatchedForestFire(B, width=W, height=H, seed=41, auto_reset=True, containment=True, fuel_range=(2, 6))
    assert env.step_launches == 2
    rows = env.step_kernel.startswith("wf_step_rows")
    assert env._lib.wf_call_launches(env._h, 8) == (2 if rows else 4)      # (the introspection speaks of calls with actions; wf_step_ticks: four parts)
    assert rows == noop
    env.reset(seed=41)
    op = O.default_params(containment=1, fuel_lo=2, fuel_hi=6)
    planes, envs = O.new_state(B, H, W)
    O.reset(planes, envs, seed=41, params=op)
    rng = np.random.default_rng(W + H)
    resets = 0
    for call, n in enumerate((8, 1, 20, 1, 1, 16)):
        acts = None if noop and n > 1 else np.empty((n, B), dtype=np.int8)
        outs = []
        for k in range(n):
            a = None
            if acts is not None:
                a = rng.integers(0, 6, size=B).astype(np.int8)
                a[:B // 2] = _circle_actions(envs[:B // 2], W, H, rng)
                acts[k] = a
            stopped = (envs["flags"] & O.F_RUNNING) == 0
            ro, do = O.step(planes, envs, a, op)
            if stopped.any():
                O.reset(planes, envs, seed=41, mask=stopped, params=op)
                ro[stopped], do[stopped] = 0.0, 0
                resets += int(stopped.sum())
            outs.append((ro, do))
        r, d = env.step_many(acts, n)
        for k in range(n):
            np.testing.assert_array_equal(r[k], outs[k][0], err_msg=f"call {call} (n={n}) step {k}: reward")
            np.testing.assert_array_equal(d[k], outs[k][1].astype(bool), err_msg=f"call {call} (n={n}) step {k}: done")
        gp, ge = env.download()
        np.testing.assert_array_equal(gp, planes, err_msg=f"call {call}: planes")
        np.testing.assert_array_equal(_envs(ge), envs, err_msg=f"call {call}: envs")
    assert resets > B // 2
    env.close()


@pytest.mark.parametrize("W,H,B,split", [(10, 10, 8192 + 70, None), (14, 14, 4096 + 3, None), (10, 10, 2048 + 9, None), (8, 8, 9000, "3")])
def test_device_calls_with_changing_part_counts(wf, monkeypatch, W, H, B, split):
    """Asynchronous device-API calls on the context's own streams: single steps of a large batch go out as two part launches,
    n-step calls without actions as two or four wf_step_ticks launches (round 4) - another cut of the batch over the streams,
    so a call with another number of parts must wait for everything the call before it put on any stream.  Groups of calls
    of both kinds go out back to back, without a synchronisation in between; then the state and every call's outputs are
    checked against the oracle."""
    import torch
    for k in ("WF_FUSED_LEGACY", "WF_MULTI_SMALL", "WF_NO_MULTI", "WF_GROUP", "WF_NO_SPLIT", "WF_SPLIT", "WF_TICKS_SPLIT"):
        monkeypatch.delenv(k, raising=False)
    if split:
        monkeypatch.setenv("WF_SPLIT", split)
    env = wf.BatchedForestFire(B, width=W, height=H, seed=43, auto_reset=True, containment=True, fuel_range=(2, 6))
    monkeypatch.delenv("WF_SPLIT", raising=False)
    want = int(split) if split else (4 if B >= 4096 else 2)
    if env.step_launches > 1 and env.step_kernel.startswith("wf_step_rows"):
        want = env.step_launches      # (the introspection speaks of calls WITH actions: the rows kernel's step loop, one launch per part)
    assert env._lib.wf_call_launches(env._h, 8) == want, env._lib.wf_call_launches(env._h, 8)
    env.reset(seed=43)
    op = O.default_params(containment=1, fuel_lo=2, fuel_hi=6)
    planes, envs = O.new_state(B, H, W)
    O.reset(planes, envs, seed=43, params=op)
    rng = np.random.default_rng(W + B)
    dev = torch.device("cuda", 0)
    resets = 0
    schedule = ((8, 1, 1), (16, 1, 5), (5, 1, 1, 1, 12), (1, 3), (7,))
    for g, group in enumerate(schedule):
        issued = []
        for n in group:                                  # the whole group goes out without a synchronisation in between
            acts = None if n > 1 else rng.integers(0, 5, size=(1, B)).astype(np.int8)
            outs = []
            for k in range(n):
                stopped = (envs["flags"] & O.F_RUNNING) == 0
                ro, do = O.step(planes, envs, None if acts is None else acts[k], op)
                if stopped.any():
                    O.reset(planes, envs, seed=43, mask=stopped, params=op)
                    ro[stopped], do[stopped] = 0.0, 0
                    resets += int(stopped.sum())
                outs.append((ro, do))
            d_act = None if acts is None else torch.from_numpy(acts).to(dev)
            d_r = torch.full((n, B), -7.0, dtype=torch.float64, device=dev)
            d_d = torch.full((n, B), 9, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()                     # (torch's stream: the inputs exist; the context's streams run on)
            env.step_device(d_act, n, d_r, d_d, update_mask=(1 << n) - 1)
            issued.append((n, outs, d_r, d_d, d_act))
        gp, ge = env.download()                          # joins the context's streams
        np.testing.assert_array_equal(gp, planes, err_msg=f"group {g}: planes")
        np.testing.assert_array_equal(_envs(ge), envs, err_msg=f"group {g}: envs")
        for i, (n, outs, d_r, d_d, _) in enumerate(issued):
            r, d = d_r.cpu().numpy(), d_d.cpu().numpy()
            for k in range(n):
                np.testing.assert_array_equal(r[k], outs[k][0], err_msg=f"group {g} call {i} (n={n}) step {k}: reward")
                np.testing.assert_array_equal(d[k], outs[k][1], err_msg=f"group {g} call {i} (n={n}) step {k}: done")
    assert resets > B // 4
    env.close()


def test_alternating_sweeps_of_the_rows_kernel_change_nothing(wf, monkeypatch):
    """A state past the Infinity Cache (300 000 x 14x14 = 235 MB): single steps walk the batch forwards and backwards in
    turn (round 4; a launch argument of the rows kernel).  The order in which a launch visits its environments must not
    show: the same steps with WF_SWEEP_ALT=0, and the oracle on the first and last 2 048 environments."""
    B, S, T = 300000, 14, 7
    rng = np.random.default_rng(5)
    acts = rng.integers(0, 5, size=(T, B)).astype(np.int8)
    out = []
    for alt in ("1", "0"):
        monkeypatch.setenv("WF_SWEEP_ALT", alt)
        env = wf.BatchedForestFire(B, size=S, seed=77, auto_reset=True, containment=True, fuel_range=(2, 5))
        assert env.step_kernel.startswith("wf_step_rows") and env.step_launches == 2
        env.reset(seed=77)
        import torch
        dev = torch.device("cuda", 0)
        d_act = torch.from_numpy(acts).to(dev)
        d_r = torch.empty((T, B), dtype=torch.float64, device=dev)
        d_d = torch.empty((T, B), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        rs = []
        for t in range(T):
            if t == 3:                                   # a synchronous host-buffer step in between: ONE launch over the whole batch
                r, d = env.step_many(acts[t:t + 1])
                rs.append((r[0].copy(), d[0].astype(np.uint8)))
            else:                                        # device API: two part launches on the context's streams
                env.step_device(d_act[t], 1, d_r[t:t + 1], d_d[t:t + 1], update_mask=1)
                rs.append(None)
        planes_envs = env.download()
        hr, hd = d_r.cpu().numpy(), d_d.cpu().numpy()
        rs = [x if x is not None else (hr[t], hd[t]) for t, x in enumerate(rs)]
        out.append((planes_envs, rs))
        env.close()
        del d_act, d_r, d_d
    monkeypatch.delenv("WF_SWEEP_ALT", raising=False)
    (pa, ea), ra = out[0]
    (pb, eb), rb = out[1]
    np.testing.assert_array_equal(pa, pb)
    np.testing.assert_array_equal(_envs(ea), _envs(eb))
    for t in range(T):
        np.testing.assert_array_equal(ra[t][0], rb[t][0], err_msg=f"reward step {t}")
        np.testing.assert_array_equal(ra[t][1], rb[t][1], err_msg=f"done step {t}")
    # the oracle on both ends of the batch (streams keyed by the environment's index in the context)
    op = O.default_params(containment=1, fuel_lo=2, fuel_hi=5)
    for lo in (0, B - 2048):
        planes, envs = O.new_state(2048, S, S)
        O.reset(planes, envs, seed=77, env_id0=lo, params=op)
        for t in range(T):
            stopped = (envs["flags"] & O.F_RUNNING) == 0
            ro, do = O.step(planes, envs, acts[t, lo:lo + 2048], op)
            if stopped.any():
                O.reset(planes, envs, seed=77, mask=stopped, env_id0=lo, params=op)
                ro[stopped], do[stopped] = 0.0, 0
            np.testing.assert_array_equal(ra[t][0][lo:lo + 2048], ro, err_msg=f"oracle reward step {t} envs {lo}..")
        np.testing.assert_array_equal(pa[lo:lo + 2048], planes, err_msg=f"oracle planes envs {lo}..")


def test_torch_can_be_imported_after_the_library():
    """PyTorch's ROCm wheels bundle their own HIP runtime; two copies in one process do not share the device, and the one that
    initialises second finds no GPU.  The loader therefore maps torch's copy first when torch is installed (without importing
    torch): a fresh process that creates a context, steps it and only THEN imports torch and uses the GPU through it."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "import_order_check.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "torch after the library: 8.0" in out.stdout, out.stdout


# ------------------------------------------------ rows layout with an odd number of environments per iteration
@pytest.mark.parametrize("W,H,group,B", [(8, 8, "7", 4096 + 37), (8, 8, "14", 4096 + 37), (16, 8, "6", 4096 + 37), (8, 16, "3", 4096 + 37),
                                         (8, 8, None, 40000), (8, 16, None, 36000), (16, 8, "12", 9000 + 11)])
def test_rows_layout_with_odd_environments_per_iteration(wf, monkeypatch, W, H, group, B):
    """8x8 puts 7 environments side by side in a wave iteration, 16x8 / 8x16 put 3: groups are multiples of
    that, and split launches cut the batch at multiples of 64 x that.  Ragged last groups, auto-resets, floods and
    containment payments, single steps and multi-step calls, against the oracle."""
    for k in ("WF_NO_ROWS", "WF_GROUP", "WF_SHARED", "WF_ROWS"):
        monkeypatch.delenv(k, raising=False)
    if group:
        monkeypatch.setenv("WF_GROUP", group)
    env = wf.BatchedForestFire(B, width=W, height=H, seed=17, auto_reset=True, containment=True, fuel_range=(3, 9))
    assert env.step_kernel.startswith(f"wf_step_rows<{W},{H},"), env.step_kernel
    if B >= 8192:
        assert env.step_launches == 2
    env.reset(seed=17)
    op = O.default_params(containment=1, fuel_lo=3, fuel_hi=9)
    planes, envs = O.new_state(B, H, W)
    O.reset(planes, envs, seed=17, params=op)
    rng = np.random.default_rng(W * 7 + H)
    paid = resets = 0
    half = B // 2
    T = 45 if B > 20000 else 70
    t = 0
    while t < T:
        n = 4 if (t % 10 == 9 and B >= 8192) else 1
        acts = np.empty((n, B), dtype=np.int8)
        outs = []
        for k in range(n):
            a = rng.integers(0, 6, size=B).astype(np.int8)
            a[:half] = _circle_actions(envs[:half], W, H, rng)
            acts[k] = a
            stopped = (envs["flags"] & O.F_RUNNING) == 0
            ro, do = O.step(planes, envs, a, op)
            if stopped.any():
                O.reset(planes, envs, seed=17, mask=stopped, params=op)
                ro[stopped], do[stopped] = 0.0, 0
                resets += int(stopped.sum())
            outs.append((ro, do))
            paid += int((ro == 1000.0).sum())
        if n == 1:
            obs, r1, d1, _ = env.step(acts[0])
            r, d = r1[None], d1[None]
            np.testing.assert_array_equal(obs, O.get_state(planes), err_msg=f"obs t={t}")
        else:
            r, d = env.step_many(acts)
        for k in range(n):
            np.testing.assert_array_equal(r[k], outs[k][0], err_msg=f"reward t={t + k}")
            np.testing.assert_array_equal(d[k], outs[k][1].astype(bool), err_msg=f"done t={t + k}")
        if t % 5 == 4 or n > 1:
            gp, ge = env.download()
            np.testing.assert_array_equal(gp, planes, err_msg=f"planes t={t}")
            np.testing.assert_array_equal(_envs(ge), envs, err_msg=f"envs t={t}")
        t += n
    assert resets > B // 2 and paid > 0, (resets, paid)
    env.close()
