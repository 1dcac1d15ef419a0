"""Workspace sizes the library asks for at the BASELINE configurations (planner sanity at sizes that are
never run in the tests).  usage: python devtools/plan_sizes.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fedrann_amd import _lib  # noqa: E402

ctx = _lib.Context(0)
for name, nq, nt, d, k in [("config 2", 100_000, 100_000, 128, 20), ("config 3", 1_000_000, 1_000_000, 128, 20),
                           ("config 4, one of 8 ranks", 1_250_000, 10_000_000, 128, 20),
                           ("config 5, one of 8 ranks", 2_500_000, 20_000_000, 256, 50),
                           ("beyond the prefilter's segment cap", 1000, 30_000_000, 128, 20),
                           ("tiny", 10, 10, 16, 5)]:
    t0 = time.perf_counter()
    b = ctx.knn_workspace_bytes(nq, nt, d, k)
    print("%-36s nq=%9d nt=%9d d=%3d k=%2d -> %8.2f GB  (%.2f s)" % (name, nq, nt, d, k, b / 1e9, time.perf_counter() - t0))
