"""RCCL on this box: init a one-rank nccl group, one all_reduce and a barrier (what bench.py --gpus N does before and after its timed region)."""
import os
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
t = torch.ones(1, device="cuda")
dist.all_reduce(t)
dist.barrier()
m = torch.tensor([1.5, 2.5], dtype=torch.float64, device="cuda")
dist.all_reduce(m, op=dist.ReduceOp.MAX)
print("rccl ok:", t.item(), m.tolist(), torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
dist.destroy_process_group()
