"""Randomised cross-check of the k-NN code paths on the GPU: for random shapes and data (sparse rows,
duplicate classes, near-ties, zero rows) the prefilter mode, the exact mode and -- for the smaller cases --
the oracle must agree bit for bit.  usage: python devtools/soak.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fedrann_amd import _lib  # noqa: E402
from oracle import oracle as O  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = _lib.Context(0)
bad = 0
for c in range(cases):
    n = int(rng.choice([9000, 17000, 33000, 70000, 150000, 300000, 450000]))  # (>= 131073 rows: synchronised rounds, two queues)
    d = int(rng.choice([64, 128, 128, 200, 256, 256, 500]))
    k = int(rng.choice([5, 20, 20, 24, 33, 50, 50, 56, 64]))
    kind = int(rng.integers(0, 4))
    if kind == 0:    # sparse rows like real embeddings
        E = rng.standard_normal((n, d)).astype(np.float32)
        E[rng.random(E.shape) < rng.choice([0.8, 0.93, 0.97])] = 0
    elif kind == 1:  # duplicate classes
        u = rng.standard_normal((int(rng.integers(40, 3000)), d)).astype(np.float32)
        u[rng.random(u.shape) < 0.9] = 0
        E = u[rng.integers(0, u.shape[0], size=n)]
    elif kind == 2:  # tight clusters: near-ties everywhere
        base = rng.standard_normal((int(rng.integers(3, 60)), d)).astype(np.float32)
        E = base[rng.integers(0, base.shape[0], size=n)] + 1e-4 * rng.standard_normal((n, d)).astype(np.float32)
    else:            # few distinct magnitudes: exact distance ties
        E = rng.integers(-1, 2, size=(n, d)).astype(np.float32)
        E[rng.random(E.shape) < 0.9] = 0
    t0 = time.perf_counter()
    ctx.set_dedup_mode(str(rng.choice(["auto", "auto", "off", "force"])))
    ctx.set_knn_mode("prefilter")
    pi, pd = ctx.knn(E, k)
    unc, uniq = ctx.last_uncertified(), ctx.last_unique()
    launches = ctx.last_prefilter_launches()
    ctx.set_knn_mode("exact")
    xi, xd = ctx.knn(E, k)
    same = np.array_equal(pi, xi) and np.array_equal(pd.view(np.uint32), xd.view(np.uint32))
    note = ""
    if n <= 17000:
        wi, wd = O.knn(E, k)
        ok = np.array_equal(xi, wi) and np.array_equal(xd.view(np.uint32), wd.view(np.uint32))
        same = same and ok
        note = " oracle=%s" % ok
    bad += not same
    if not np.array_equal(pi, xi) or not np.array_equal(pd.view(np.uint32), xd.view(np.uint32)):
        rows = np.flatnonzero((pi != xi).any(1) | (pd.view(np.uint32) != xd.view(np.uint32)).any(1))
        print("   prefilter != exact in %d rows, first %s" % (rows.size, rows[:5]))
        for r in rows[:2]:
            print("   row", r, "nnz", int((E[r] != 0).sum()), "\n    P", pi[r], pd[r], "\n    X", xi[r], xd[r])
    print("case %2d n=%6d d=%3d k=%2d kind=%d uncertified=%6d unique=%s launches=%s  %s%s  %.1fs"
          % (c, n, d, k, kind, unc, uniq, launches, "OK" if same else "MISMATCH", note, time.perf_counter() - t0), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
