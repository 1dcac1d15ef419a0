"""Development: how many normalised embedding rows are bitwise duplicates?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fedrann_amd.synth import synth
from fedrann_amd.precompute import build_precompute_matrix
from oracle import oracle as O
for R in [int(a) for a in sys.argv[1:]] or [100_000, 1_000_000]:
    d = 128
    s = synth(R, seed=602)
    P = build_precompute_matrix(s["counts"], d)
    E = O.embed(s["indptr"], s["indices"], (P.indptr, P.indices, P.data), s["n_features"], d)
    Eh, _, zero = O.normalize(E)
    t0 = time.time()
    v = np.ascontiguousarray(Eh).view(np.dtype((np.void, 4 * d))).ravel()
    u, inv, cnt = np.unique(v, return_inverse=True, return_counts=True)
    nnz = (Eh != 0).sum(1)
    csize = cnt[inv]
    print("reads %d: unique rows %d (%.1f %%), zero rows %d, rows in classes >= 20: %.1f %%, >= 1000: %.1f %%, max class %d  (%.0fs)"
          % (R, u.size, 100.0 * u.size / R, int(zero.sum()), 100.0 * (csize >= 20).mean(), 100.0 * (csize >= 1000).mean(), cnt.max(), time.time() - t0))
    for k in range(0, 6):
        m = nnz == k
        if m.any():
            print("   rows with %d non-zeros: %.1f %% of rows, mean class size %.0f" % (k, 100.0 * m.mean(), csize[m].mean()))
