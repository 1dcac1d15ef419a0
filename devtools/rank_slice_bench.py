"""Time what ONE rank of an N-rank run does after the all-gather, for N = 1, 2, 4, 8, on a single GPU -- a model
of the strong-scaling curve without RCCL.  Two variants: the k-NN of the rank's ROW block against all rows
(fdr_knn_dev), and what distributed.ShardedPipeline does when the rows repeat: classes of all rows, k-NN of the
rank's share of the UNIQUE rows, expansion of its own rows (the exchange of the shares is not timed).
usage: python devtools/rank_slice_bench.py [reads] [N,N,...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fedrann_amd import _lib  # noqa: E402
from fedrann_amd.distributed import HipEngine, shard_rows  # noqa: E402
from fedrann_amd.precompute import build_precompute_matrix  # noqa: E402
from fedrann_amd.synth import synth  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d, k = 128, 20
s = synth(R, seed=602)
P = build_precompute_matrix(s["counts"], d)
dev = torch.device("cuda", 0)
ctx = _lib.Context(0)
ctx.projection_load(P.indptr, P.indices, P.data, s["n_features"], d)
eng = HipEngine(ctx, dev)
n = len(s["indptr"]) - 1
E = eng.embed(torch.from_numpy(np.ascontiguousarray(s["indptr"], np.int64)).to(dev),
              torch.from_numpy(np.ascontiguousarray(s["indices"], np.int32)).to(dev), n, d)
Ehat = torch.zeros((n, eng.padded_dim(d)), dtype=torch.float32, device=dev)
zero = torch.zeros((n,), dtype=torch.uint8, device=dev)
eng.normalize(E, Ehat, zero)
base = None
for G in ([int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (1, 2, 4, 8)):
    S, blocks = shard_rows(n, G)
    lo, hi = blocks[0]
    for _ in range(2):
        eng.knn(Ehat[lo:hi], zero[lo:hi], hi - lo, Ehat, zero, n, d, k)
    torch.cuda.synchronize(dev)
    ctx.timing(True)
    t0 = time.perf_counter()
    for _ in range(5):
        eng.knn(Ehat[lo:hi], zero[lo:hi], hi - lo, Ehat, zero, n, d, k)
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / 5 * 1e3
    kinds = {name: round(ctx.timing_read(i)[1] / 5, 3) for i, name in enumerate(_lib.KERNELS)}
    ctx.timing(False)
    base = base or ms
    print("N=%d: %6d query rows per rank, k-NN %.3f ms  (x%.2f vs N=1, ideal x%d)  %s"
          % (G, hi - lo, ms, base / ms, G, {a: b for a, b in kinds.items() if b}))
    if G > 1:
        nq_max = -(-n // G)
        nu = eng.knn_classes(Ehat, zero, n, d, k, nq_max)
        if nu > 0:
            Su = -(-nu // G)
            iu = torch.zeros((Su, k), dtype=torch.int32, device=dev)
            du = torch.zeros((Su, k), dtype=torch.float32, device=dev)
            pu = torch.zeros((Su * G, 2 * k), dtype=torch.int32, device=dev)

            def unique_step():
                eng.knn_classes(Ehat, zero, n, d, k, nq_max)
                eng.knn_unique(0, min(nu, Su), k, iu, du)
                pu[:Su, :k] = iu
                pu[:Su, k:] = du.view(torch.int32)
                return eng.knn_expand(lo, hi - lo, k, pu)
            unique_step()
            torch.cuda.synchronize(dev)
            ctx.timing(True)
            t0 = time.perf_counter()
            for _ in range(5):
                unique_step()
            torch.cuda.synchronize(dev)
            ms2 = (time.perf_counter() - t0) / 5 * 1e3
            kinds = {name: round(ctx.timing_read(i)[1] / 5, 3) for i, name in enumerate(_lib.KERNELS)}
            ctx.timing(False)
            print("     unique-row split: %d of %d unique rows per rank, classes + k-NN + expansion %.3f ms  (x%.2f vs N=1)  %s"
                  % (min(nu, Su), nu, ms2, base / ms2, {a: b for a, b in kinds.items() if b}))
