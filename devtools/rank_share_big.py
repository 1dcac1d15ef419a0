"""Time ONE rank's share of an 8-GPU run at sizes whose CSR the host cannot synthesise quickly: device-made
embeddings (tests/test_gpu_configs.py's generator), queries = rows [0, n/8) against all n rows.
usage: python devtools/rank_share_big.py [rows=10000000] [dim=128] [k=20] [ranks=8]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fedrann_amd import _lib  # noqa: E402
from fedrann_amd.distributed import HipEngine, shard_rows  # noqa: E402
from test_gpu_configs import _device_embeddings  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
k = int(sys.argv[3]) if len(sys.argv) > 3 else 20
G = int(sys.argv[4]) if len(sys.argv) > 4 else 8
dev = torch.device("cuda", 0)
ctx = _lib.Context(0)
eng = HipEngine(ctx, dev)
E = _device_embeddings(n, d, nnz=6 if d <= 128 else 8, loci=max(1000, int(0.4 * n)), seed=4, doubling=d > 128)
Ehat = torch.zeros((n, eng.padded_dim(d)), dtype=torch.float32, device=dev)
zero = torch.zeros((n,), dtype=torch.uint8, device=dev)
eng.normalize(E, Ehat, zero)
del E
S, blocks = shard_rows(n, G)
lo, hi = blocks[0]
eng.knn(Ehat[lo:hi], zero[lo:hi], hi - lo, Ehat, zero, n, d, k)
torch.cuda.synchronize(dev)
ctx.timing(True)
t0 = time.perf_counter()
reps = 2
for _ in range(reps):
    eng.knn(Ehat[lo:hi], zero[lo:hi], hi - lo, Ehat, zero, n, d, k)
torch.cuda.synchronize(dev)
ms = (time.perf_counter() - t0) / reps * 1e3
kinds = {name: round(ctx.timing_read(i)[1] / reps, 2) for i, name in enumerate(_lib.KERNELS)}
ut, uq = ctx.last_unique()
print("rows=%d d=%d k=%d ranks=%d: %d query rows per rank, k-NN %.1f ms -> %.1f M read-pairs/s for the node "
      "(if every rank takes as long); unique targets %d, queries %d; uncertified %d; workspace %.1f GB; %s"
      % (n, d, k, G, hi - lo, ms, n * k / (ms * 1e-3) / 1e6, ut, uq, ctx.last_uncertified(),
         ctx.knn_workspace_bytes(hi - lo, n, d, k) / 1e9, {a: b for a, b in kinds.items() if b}))
