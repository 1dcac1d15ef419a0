"""Which libamdhip64 ends up mapped, and does torch see the GPU, when libfedrann_hip.so is loaded first?"""
import sys
sys.path.insert(0, ".")
from fedrann_amd import _lib
ctx = _lib.Context(0)
print("context ok:", ctx.device_info() if hasattr(ctx, "device_info") else "")
import torch
print("torch sees gpu:", torch.cuda.is_available(), torch.cuda.device_count())
libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l or "libhsa" in l})
print("\n".join(libs))
