"""Create / step / destroy contexts in a loop and watch free device memory (usage: python tools/leak_check.py [cycles])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wildfire_control_python_amd import BatchedForestFire
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
torch.cuda.init()
BatchedForestFire(64, size=14).close()
prev = torch.cuda.mem_get_info()[0]
for rnd in range(3):
    for i in range(N):
        for kw in (dict(size=14), dict(size=64), dict(width=13, height=11)):
            e = BatchedForestFire(2048, auto_reset=True, **kw)
            e.reset(seed=i)
            e.step(None)
            e.close()
    free = torch.cuda.mem_get_info()[0]
    print("round %d: free %.1f MB (delta %.2f MB over %d contexts)" % (rnd, free / 1e6, (prev - free) / 1e6, 3 * N), flush=True)
    prev = free
