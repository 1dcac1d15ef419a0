"""One rank of tests/test_gpu_distributed.py: the row-sharded pipeline with the real HipEngine.  All
ranks share GPU 0 and exchange over gloo (a one-GPU rehearsal of the N > 1 path; RCCL needs one GPU
per rank).  usage: python _gpu_rank_worker.py OUTDIR READS DIM K   (RANK / WORLD_SIZE / MASTER_* in env)"""
import os
import sys

import numpy as np
import torch  # (first: fedrann_amd._lib then shares torch's HIP runtime either way)
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fedrann_amd import _lib  # noqa: E402
from fedrann_amd.distributed import HipEngine, ShardedPipeline, local_csr  # noqa: E402
from fedrann_amd.precompute import build_precompute_matrix  # noqa: E402
from fedrann_amd.synth import synth  # noqa: E402

outdir, R, d, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
s = synth(R, seed=23, m=60)
P = build_precompute_matrix(s["counts"], d)
n = len(s["indptr"]) - 1
ctx = _lib.Context(0)
ctx.projection_load(P.indptr, P.indices, P.data, s["n_features"], d)
pipe = ShardedPipeline(HipEngine(ctx, dev), n, d, k, rank=rank, world_size=world, device=dev)
ip, ix = local_csr(s["indptr"], s["indices"], pipe.lo, pipe.hi)
for _ in range(2):  # the second pass reuses every buffer
    idx, dst, E = pipe.step(torch.from_numpy(ip).to(dev), torch.from_numpy(ix).to(dev))
torch.cuda.synchronize(dev)
np.savez(os.path.join(outdir, "rank%d.npz" % rank), idx=idx.cpu().numpy(), dist=dst.cpu().numpy(),
         lo=pipe.lo, hi=pipe.hi, E=E.cpu().numpy())
dist.barrier()
dist.destroy_process_group()
ctx.close()
