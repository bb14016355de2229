"""Per-GPU launcher: environments never interact (one World per env,
environment.py:174-212), so a global batch shards contiguously over the GPUs of a
node with NO collective on the step path.  Env i of the global batch lives on rank
i // per_rank; its reset stream is keyed by the GLOBAL id (params.env_id0 + local
index), so results are independent of the number of GPUs."""
import os


def shard_bounds(global_batch, world_size, rank):
    """Contiguous [lo, hi) slice of the global batch owned by `rank`."""
    per = -(-global_batch // world_size)
    lo = min(rank * per, global_batch)
    hi = min(lo + per, global_batch)
    return lo, hi


def rank_info():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    return rank, world, local


def make_shard(global_batch, size, rank=None, world_size=None, device=None, **kw):
    """BatchedForestFire holding this rank's slice of a global batch."""
    from .batched import BatchedForestFire
    r, w, local = rank_info()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    device = local if device is None else device
    lo, hi = shard_bounds(global_batch, world_size, rank)
    if hi <= lo:
        raise ValueError(f"rank {rank} of {world_size} owns no environment of {global_batch}")
    return BatchedForestFire(hi - lo, size=size, device=device, env_id0=lo, **kw)
