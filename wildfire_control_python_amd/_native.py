"""ctypes binding of the C-ABI in include/wildfire.h (libwildfire_hip.so).

Mirrors the reference's only FFI convention (pyastar/pyastar.py:7-22): the shared
library sits next to this module, is loaded with ctypes, takes C-contiguous numpy
buffers owned by the caller.  Unlike the reference's loader (which prints and
calls exit()), a missing library raises ImportError: there is no CPU fallback.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WILDFIRE_LIB") or os.path.join(HERE, "libwildfire_hip.so")   # override: A/B builds

ENV_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("flags", "<u2"), ("wind", "<u2"),
                      ("episode", "<u4"), ("tick", "<u4")])
assert ENV_DTYPE.itemsize == 16

P_TYPE, P_FUEL, P_HEAT, P_AGENT = 0, 1, 2, 3
GRASS, FIRE, BURNT, DIRT, WATER = 0, 1, 2, 3, 4
DIRT_BURNING, BURNT_BLOCKED = 5, 6   # allow_dig_toggle only: dig() on a burning cell (include/wildfire.h)
FIRE_BLOCKED = 13                    # make_rivers only: the river under the fire cell (type fire, fire_mobility inf)
HEAT_FORCED = 25
HEAT_IGNITED = 24          # wind with a direction: plane 2 of a cell ignited by accumulated heat
WIND_CALM, WIND_FIXED, WIND_RANDOM = 0, 1, 2
F_ALIVE, F_VISIBLE, F_DIGGING, F_DEAD = 1, 2, 4, 8
F_RUNNING, F_BORDER, F_CONTAINED, F_REACH_VALID, F_REACH_ROWS = 16, 32, 64, 128, 256

WF_OK = 0
ABI_VERSION = 2


class Params(ctypes.Structure):
    _fields_ = [("ignite_count", ctypes.c_int32), ("allow_dig_toggle", ctypes.c_int32),
                ("containment", ctypes.c_int32), ("auto_reset", ctypes.c_int32),
                ("death_penalty", ctypes.c_double), ("contained_bonus", ctypes.c_double),
                ("default_reward", ctypes.c_double),
                ("fuel_lo", ctypes.c_int32), ("fuel_hi", ctypes.c_int32),
                ("ignite_ppm", ctypes.c_uint32), ("make_rivers", ctypes.c_int32),
                ("seed", ctypes.c_uint64), ("env_id0", ctypes.c_uint64),
                ("wind_mode", ctypes.c_int32), ("wind_code", ctypes.c_int32),
                ("wind_speed", ctypes.c_double * 4),
                ("heat", ctypes.c_double), ("threshold", ctypes.c_double)]


def wind_code(wx, wy, speed_index=0):
    """WF_WIND_CODE of include/wildfire.h: wind vector (wx, wy) in {-1,0,1}^2 + index into Params.wind_speed."""
    if wx not in (-1, 0, 1) or wy not in (-1, 0, 1) or not 0 <= speed_index <= 3:
        raise ValueError("wind vector components must be -1, 0 or 1 (environment.py:190)")
    return (int(wx) + 1) | (int(wy) + 1) << 2 | int(speed_index) << 4


class WildfireError(RuntimeError):
    def __init__(self, status, where, hip=0):
        self.status, self.hip = status, hip
        msg = lib().wf_strerror(status).decode() if _lib is not None else str(status)
        super().__init__(f"{where}: {msg} (status {status}, hip error {hip})")


# every symbol include/wildfire.h declares: (name, restype, argtypes)
_vp, _i, _u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64
SYMBOLS = [
    ("wf_abi_version", _i, []),
    ("wf_default_params", None, [ctypes.POINTER(Params)]),
    ("wf_strerror", ctypes.c_char_p, [_i]),
    ("wf_last_hip_error", _i, [_vp]),
    ("wf_create", _i, [ctypes.POINTER(_vp), _i, _i, _i, _i, ctypes.POINTER(Params)]),
    ("wf_destroy", _i, [_vp]),
    ("wf_set_params", _i, [_vp, ctypes.POINTER(Params)]),
    ("wf_set_stream", _i, [_vp, _vp]),
    ("wf_upload", _i, [_vp, _vp, _vp]),
    ("wf_download", _i, [_vp, _vp, _vp]),
    ("wf_wind_counts_upload", _i, [_vp, _vp]),
    ("wf_wind_counts_download", _i, [_vp, _vp]),
    ("wf_reset", _i, [_vp, _vp, _u64]),
    ("wf_reset_device", _i, [_vp, _vp, _u64]),
    ("wf_step", _i, [_vp, _vp, _i, _u64, _vp, _vp]),
    ("wf_step_device", _i, [_vp, _vp, _i, _u64, _vp, _vp]),
    ("wf_step_obs", _i, [_vp, _vp, _i, _u64, _vp, _vp, _vp]),
    ("wf_step_download", _i, [_vp, _vp, _i, _u64, _vp, _vp, _vp, _vp]),
    ("wf_host_staging", _i, [_vp, _i, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    ("wf_step_obs_device", _i, [_vp, _vp, _i, _u64, _vp, _vp, _vp]),
    ("wf_get_state", _i, [_vp, _vp]),
    ("wf_get_state_device", _i, [_vp, _vp]),
    ("wf_sync", _i, [_vp]),
    ("wf_planes_device", _vp, [_vp]),
    ("wf_envs_device", _vp, [_vp]),
    ("wf_timer_start", _i, [_vp]),
    ("wf_timer_stop", _i, [_vp, ctypes.POINTER(ctypes.c_float)]),
    ("wf_kernel_name", ctypes.c_char_p, [_vp]),
    ("wf_step_kernel_name", ctypes.c_char_p, [_vp]),
    ("wf_step_launches", _i, [_vp]),
    ("wf_call_launches", _i, [_vp, _i]),
    ("wf_step_plan", _i, [_i, _i, _i] + [ctypes.POINTER(_i)] * 4),
    ("wf_call_kernel_name", ctypes.c_char_p, [_vp, _i]),
    ("wf_reload_env", _i, [_vp]),
]

_lib = None


def _share_torch_hip_runtime():
    """PyTorch's ROCm wheels bundle their own HIP / HSA runtime under torch/lib.  Two copies of the runtime in one process do not
    share the device: whichever initialises second finds "No HIP GPUs".  When torch is installed but not imported yet, its copy
    is loaded here (located with find_spec: torch itself is NOT imported), so that this library's libamdhip64 dependency
    resolves to the copy torch will use later and `import torch` after this package works as it does before it.  Without
    torch (a plain C client, a torch-free Python process) the system runtime under /opt/rocm is used as linked."""
    import sys
    if "torch" in sys.modules or os.environ.get("WILDFIRE_NO_TORCH_RUNTIME"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.origin:
            cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
            if os.path.exists(cand):
                ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except Exception:
        pass


def lib():
    """Load libwildfire_hip.so (built by wildfire_control_python_amd/csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `make -C {os.path.join(HERE, 'csrc')}` "
                "(or __graft_entry__.build()).  The wildfire step has no CPU fallback.")
        _share_torch_hip_runtime()
        L = ctypes.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)          # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if L.wf_abi_version() != ABI_VERSION:
            raise ImportError(f"{LIB_PATH}: ABI version {L.wf_abi_version()} != {ABI_VERSION}")
        _lib = L
    return _lib


def default_params():
    p = Params()
    lib().wf_default_params(ctypes.byref(p))
    return p


def ptr(a):
    """void* of a numpy array / torch tensor / raw int / None."""
    if a is None:
        return None
    if isinstance(a, int):
        return ctypes.c_void_p(a)
    if isinstance(a, np.ndarray):
        return ctypes.c_void_p(a.ctypes.data)          # (half the cost of .ctypes.data_as(): 1.2 against 2.3 us, five per step call)
    if hasattr(a, "data_ptr"):
        return ctypes.c_void_p(a.data_ptr())
    raise TypeError(type(a))
