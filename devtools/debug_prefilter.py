import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FDR_KNN_MODE"] = "prefilter"
from fedrann_amd import _lib
from oracle import oracle
rng = np.random.default_rng(12)
_ = rng.standard_normal((4000, 128)); _ = rng.random((4000, 128)); a = rng.integers(0, 4000, size=20000); b = rng.integers(0, 4000, size=20000)
base = rng.standard_normal((4, 128)).astype(np.float32)
T = base[rng.integers(0, 4, size=3000)] + 1e-4 * rng.standard_normal((3000, 128)).astype(np.float32)
ctx = _lib.Context(0)
gi, gd = ctx.knn(T, 20)
print("uncertified", ctx.last_uncertified())
wi, wd = oracle.knn(T, 20)
bad = np.flatnonzero((gi != wi).any(1))
print("bad queries", bad.size, bad[:10])
Eh, _, _ = oracle.normalize(T)
h = Eh.astype(np.float16).astype(np.float32)
for q in bad[:3]:
    s = (h[q][None, :] * h).sum(1)
    dt = np.clip(1 - s, 0, 1).astype(np.float32)
    o = np.argsort(dt, kind="stable")
    print("q", q, "dt[K-1]", dt[o[19]], "dt[KP-1]", dt[o[31]], "dt[40]", dt[o[40]])
    print(" got ", gi[q], gd[q])
    print(" want", wi[q], wd[q])
