"""Per-GPU launcher: environments never interact (one World per env,
environment.py:174-212), so a global batch shards contiguously over the GPUs of a
node with NO collective on the step path.  Env i of the global batch lives on rank
i // per_rank; its reset stream is keyed by the GLOBAL id (params.env_id0 + local
index), so results are independent of the number of GPUs."""
import glob
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


def visible_gpu_count(topology="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs this process would see, counted WITHOUT loading the HIP runtime or torch: the KFD topology in sysfs
    (a node with SIMDs is a GPU), cut down by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when they
    are set.  For launchers: the parent of N rank processes has no business on the GPU (and on some pools a process that
    has initialised it may not exec or fork freely).  No KFD topology at all (no amdgpu driver) = 0 GPUs; None when
    it exists but cannot be read."""
    n = 0
    if not os.path.isdir(topology):
        return 0
    nodes = glob.glob(os.path.join(topology, "*", "properties"))
    if not nodes:
        return None
    for path in nodes:
        try:
            props = dict(line.split()[:2] for line in open(path) if len(line.split()) >= 2)
        except OSError:
            return None
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


class ShardedForestFire:
    """One process, several GPUs: a global batch of independent environments held as one contiguous shard per device
    (env i on shard i // per_shard, reset streams keyed by the global id - the same layout as one process per GPU,
    launcher.make_shard), every shard a BatchedForestFire context of its own driven through the asynchronous *_device
    entry points.  There is nothing to exchange between shards (no collective): step_device() issues every shard's
    launches back to back and returns; sync() waits for all of them.  `devices`: one device index per shard (an index
    may repeat - several contexts on one card - which is how the tests rehearse it on a one-GPU box).

    Buffers handed to step_device / get_state_device are per-shard lists of torch tensors (or raw device pointers) that
    live on the shard's device; the numpy-facing reset() / step() / download() concatenate over the shards."""

    def __init__(self, global_batch, size=None, devices=(0,), **kw):
        from .batched import BatchedForestFire
        self.global_batch = int(global_batch)
        self.devices = [int(d) for d in devices]
        self.bounds = [shard_bounds(self.global_batch, len(self.devices), r) for r in range(len(self.devices))]
        if any(hi <= lo for lo, hi in self.bounds):
            raise ValueError(f"{len(self.devices)} shards for {global_batch} environments: a shard would be empty")
        kw.pop("env_id0", None)
        self.shards = [BatchedForestFire(hi - lo, size=size, device=d, env_id0=lo, **kw)
                       for (lo, hi), d in zip(self.bounds, self.devices)]
        self.width, self.height = self.shards[0].width, self.shards[0].height
        self.METADATA = self.shards[0].METADATA

    def close(self):
        for s in self.shards:
            s.close()

    def sync(self):
        for s in self.shards:
            s.sync()

    # ---- device path: asynchronous, one list entry per shard
    def step_device(self, d_actions=None, n_steps=1, d_reward=None, d_done=None, d_obs=None):
        none = [None] * len(self.shards)
        mask = self.shards[0]._update_mask(n_steps)              # one a_speed counter for the whole batch
        for s, a, r, d, o in zip(self.shards, d_actions or none, d_reward or none, d_done or none, d_obs or none):
            s.step_device(a, n_steps, r, d, update_mask=mask, d_obs=o)

    def get_state_device(self, d_obs):
        for s, o in zip(self.shards, d_obs):
            s.get_state_device(o)

    # ---- host path (numpy in / out, concatenated over the shards)
    def _split(self, a, axis=0):
        import numpy as np
        if a is None:
            return [None] * len(self.shards)
        return [np.ascontiguousarray(np.take(a, range(lo, hi), axis=axis)) for lo, hi in self.bounds]

    def reset(self, seed=None):
        import numpy as np
        return np.concatenate([s.reset(seed=seed) for s in self.shards])

    def step(self, actions=None):
        import numpy as np
        mask = self.shards[0]._update_mask(1)                     # ONE a_speed counter for the whole batch
        outs = [s.step(a, update_mask=mask) for s, a in zip(self.shards, self._split(actions))]
        return [np.concatenate([o[0] for o in outs]), np.concatenate([o[1] for o in outs]),
                np.concatenate([o[2] for o in outs]), {}]

    def step_many(self, actions=None, n_steps=None):
        import numpy as np
        n = int(n_steps or 1) if actions is None else (1 if np.ndim(actions) == 1 else np.shape(actions)[0])
        mask = self.shards[0]._update_mask(n)
        outs = [s.step_many(a, n_steps, update_mask=mask) for s, a in zip(self.shards, self._split(actions, axis=-1))]
        return np.concatenate([o[0] for o in outs], axis=1), np.concatenate([o[1] for o in outs], axis=1)

    def download(self):
        import numpy as np
        parts = [s.download() for s in self.shards]
        return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])

    def get_state(self):
        import numpy as np
        return np.concatenate([s.get_state() for s in self.shards])
