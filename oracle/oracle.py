"""ctypes loader for the CPU parity oracle (oracle/wf_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, as the checker.  The product package never imports it.
Parity status: pinned by tests/golden/ref_*.npz (tests/test_oracle_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libwf_oracle.so")

ENV_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("flags", "<u2"), ("wind", "<u2"),
                      ("episode", "<u4"), ("tick", "<u4")])
assert ENV_DTYPE.itemsize == 16

F_ALIVE, F_VISIBLE, F_DIGGING, F_DEAD, F_RUNNING, F_BORDER, F_CONTAINED = 1, 2, 4, 8, 16, 32, 64
HEAT_FORCED = 25
HEAT_IGNITED = 24        # wind_mode != 0: plane 2 of a cell ignited by accumulated heat


class Params(ctypes.Structure):
    _fields_ = [("wind_speed", ctypes.c_double), ("wind_x", ctypes.c_double),
                ("wind_y", ctypes.c_double), ("heat", ctypes.c_double),
                ("threshold", ctypes.c_double), ("death_penalty", ctypes.c_double),
                ("contained_bonus", ctypes.c_double), ("default_reward", ctypes.c_double),
                ("allow_dig_toggle", ctypes.c_int32), ("containment", ctypes.c_int32),
                ("fuel_lo", ctypes.c_int32), ("fuel_hi", ctypes.c_int32),
                ("ignite_ppm", ctypes.c_uint32), ("make_rivers", ctypes.c_int32),
                ("wind_mode", ctypes.c_int32), ("wind_code", ctypes.c_int32),
                ("wind_speeds", ctypes.c_double * 4)]


def wind_code(wx, wy, speed_index=0):
    """wfo_env.wind of a wind vector (wx, wy) in {-1,0,1}^2 and an index into Params.wind_speeds."""
    return (int(wx) + 1) | (int(wy) + 1) << 2 | int(speed_index) << 4


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "wf_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", HERE, LIB_PATH])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB_PATH)
        i8p = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")
        envp = np.ctypeslib.ndpointer(dtype=ENV_DTYPE, ndim=1, flags="C_CONTIGUOUS")
        L.wfo_default_params.argtypes = [ctypes.POINTER(Params)]
        L.wfo_default_params.restype = None
        L.wfo_step.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, i8p, envp,
                               ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(Params),
                               ctypes.c_void_p, ctypes.c_void_p]
        L.wfo_step.restype = ctypes.c_int
        L.wfo_step_wind.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, i8p, envp, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(Params),
                                    ctypes.c_void_p, ctypes.c_void_p]
        L.wfo_step_wind.restype = ctypes.c_int
        L.wfo_reset.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, i8p, envp,
                                ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64,
                                ctypes.POINTER(Params)]
        L.wfo_reset.restype = ctypes.c_int
        L.wfo_get_state.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, i8p, i8p]
        L.wfo_get_state.restype = ctypes.c_int
        L.wfo_rng_u32.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32]
        L.wfo_rng_u32.restype = ctypes.c_uint32
        L.wfo_circle_points.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                        ctypes.POINTER(ctypes.c_int)]
        L.wfo_circle_points.restype = ctypes.c_int
        L.wfo_river.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                                np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")]
        L.wfo_river.restype = ctypes.c_int
        _lib = L
    return _lib


def default_params(**kw):
    p = Params()
    lib().wfo_default_params(ctypes.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        if k == "wind_speeds":
            for i, x in enumerate(v):
                p.wind_speeds[i] = x
            continue
        setattr(p, k, v)
    return p


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def step(planes, envs, actions=None, params=None, do_update=True, counts=None):
    """One reference step per env, in place.  planes int8 (B,H,W,4); envs ENV_DTYPE (B,);
    counts uint8 (B,H,W,4) with params.wind_mode != 0 (see wf_oracle.h)."""
    B, H, W, P = planes.shape
    assert P == 4 and planes.dtype == np.int8 and planes.flags.c_contiguous
    if actions is not None:
        actions = np.ascontiguousarray(actions, dtype=np.int8)
        assert actions.shape == (B,)
    reward = np.zeros(B, dtype=np.float64)
    done = np.zeros(B, dtype=np.uint8)
    p = params if params is not None else default_params()
    if counts is not None:
        assert counts.shape == planes.shape and counts.dtype == np.uint8 and counts.flags.c_contiguous
    rc = lib().wfo_step_wind(B, H, W, planes, envs, _ptr(counts), _ptr(actions), int(bool(do_update)),
                             ctypes.byref(p), _ptr(reward), _ptr(done))
    assert rc == 0
    return reward, done


def reset(planes, envs, seed, mask=None, env_id0=0, params=None, counts=None):
    B, H, W, P = planes.shape
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
    if counts is not None:        # a fresh grid has received no heat
        counts[slice(None) if mask is None else mask != 0] = 0
    p = params if params is not None else default_params()
    rc = lib().wfo_reset(B, H, W, planes, envs, _ptr(mask), int(seed), int(env_id0),
                         ctypes.byref(p))
    assert rc == 0


def get_state(planes):
    B, H, W, P = planes.shape
    obs = np.zeros((B, W, H, 3), dtype=np.int8)
    rc = lib().wfo_get_state(B, H, W, planes, obs)
    assert rc == 0
    return obs


def rng_u32(seed, env_id, episode, draw):
    return int(lib().wfo_rng_u32(int(seed), int(env_id), int(episode), int(draw)))


def circle_points(r):
    dx = (ctypes.c_int * 32)()
    dy = (ctypes.c_int * 32)()
    n = lib().wfo_circle_points(int(r), dx, dy)
    return [(dx[i], dy[i]) for i in range(n)]


def river(H, W, seed, env_id=0, episode=0):
    """uint8 (H, W) water mask of the river wfo_reset would draw for (seed, env_id, episode)."""
    water = np.zeros((H, W), dtype=np.uint8)
    lib().wfo_river(int(H), int(W), int(seed), int(env_id), int(episode), water)
    return water


def new_state(batch, H, W):
    return (np.zeros((batch, H, W, 4), dtype=np.int8), np.zeros(batch, dtype=ENV_DTYPE))


def envs_from_agent_rows(rows, running=True):
    """Fixture 'agent' rows [x, y, alive, digging, dead] -> ENV_DTYPE array (flags partly)."""
    rows = np.atleast_2d(rows)
    e = np.zeros(len(rows), dtype=ENV_DTYPE)
    alive = rows[:, 2] != 0
    e["x"] = np.where(alive, rows[:, 0], 0)
    e["y"] = np.where(alive, rows[:, 1], 0)
    e["flags"] = (alive * F_ALIVE) | ((rows[:, 3] != 0) * F_DIGGING) | (running * F_RUNNING)
    return e
