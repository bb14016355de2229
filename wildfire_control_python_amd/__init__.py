"""wildfire_control_python_amd - MI355X-native batched wildfire cellular-automaton step.

One hot path of dashdeckers/Wildfire-Control-Python (ForestFire.step / update and the
World/Agent code under it) rebuilt as hand-written gfx950 HIP kernels behind a C-ABI
(include/wildfire.h), with the reference's own step()/reset()/get_state() surface on top.

    from wildfire_control_python_amd import ForestFire, BatchedForestFire

Importing the package does not touch the GPU; constructing an environment does, and
fails loudly (ImportError / WildfireError) when the HIP library or a device is missing.
"""
from .constants import METADATA  # noqa: F401
from .utility import layer, types, grass  # noqa: F401
from .batched import BatchedForestFire, encode_action, NOOP  # noqa: F401
from .forest_fire import ForestFire, World, Agent  # noqa: F401
from .launcher import shard_bounds, make_shard, visible_gpu_count, ShardedForestFire  # noqa: F401
from .learner import BatchedLearner, ALGORITHMS  # noqa: F401

__all__ = ["ForestFire", "BatchedForestFire", "World", "Agent", "METADATA", "layer", "types",
           "grass", "encode_action", "NOOP", "shard_bounds", "make_shard", "visible_gpu_count", "ShardedForestFire",
           "BatchedLearner", "ALGORITHMS"]
