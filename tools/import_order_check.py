"""A fresh process that loads the library and steps a context BEFORE it imports torch (tests/test_gpu_parity.py:
test_torch_can_be_imported_after_the_library; WILDFIRE_NO_TORCH_RUNTIME=1 shows the failure the loader avoids)."""
import os
import sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wildfire_control_python_amd as pkg
env = pkg.BatchedForestFire(64, size=14, seed=1)
env.reset(seed=1)
o, r, d, _ = env.step(np.zeros(64, dtype=np.int8))
print("library first: stepped", r[:3])
import torch
x = torch.ones(4, device="cuda") * 2
print("torch after the library:", x.sum().item(), torch.cuda.get_device_name(0))
