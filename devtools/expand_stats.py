"""How many queries of the bench workload meet a multi-member class among their k neighbours (the slow path of
expand_classes_kernel), and how many members such a query has to sort."""
import sys

import numpy as np

from fedrann_amd import _lib
from fedrann_amd.precompute import build_precompute_matrix
from fedrann_amd.synth import synth

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d, k = 128, 20
s = synth(R, seed=0)
P = build_precompute_matrix(s["counts"], d)
ctx = _lib.Context(0)
ctx.projection_load(P.indptr, P.indices, P.data, s["n_features"], d)
idx, dist, E = ctx.embed_knn(s["indptr"], s["indices"], k, return_embedding=True)
v = np.ascontiguousarray(E).view(np.uint64)
mult = np.random.default_rng(1).integers(1, 2**63, size=v.shape[1], dtype=np.uint64) | np.uint64(1)
h = (v * mult).sum(axis=1, dtype=np.uint64)
_, inv, cnt = np.unique(h, return_inverse=True, return_counts=True)
size = cnt[inv]
print("rows", R, "unique", cnt.size, "rows in multi-member classes", int((size > 1).sum()))
ns = size[idx]
slow = (ns > 1).any(axis=1)
print("queries with a multi-member class in their list: %.4f" % slow.mean())
tot = np.minimum(ns, k).sum(axis=1)
for lim in (20, 64, 128, 400):
    print("  members to sort <= %d: %.4f" % (lim, (tot <= lim).mean()))
