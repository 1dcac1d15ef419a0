"""One rank of tests/test_gpu_distributed.py::test_sharded_unique_row_split: the all-gather + k-NN part of the
sharded pipeline on duplicate-heavy embeddings (the ranks split the unique rows).  All ranks share GPU 0 and
exchange over gloo.  usage: python _gpu_rank_worker_dup.py OUTDIR E.npy K"""
import os
import sys

import numpy as np
import torch  # (first: fedrann_amd._lib then shares torch's HIP runtime either way)
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fedrann_amd import _lib  # noqa: E402
from fedrann_amd.distributed import HipEngine, ShardedPipeline  # noqa: E402

outdir, path, k = sys.argv[1], sys.argv[2], int(sys.argv[3])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
E_all = np.load(path)
n, d = E_all.shape
ctx = _lib.Context(0)


class GivenEmbeddings(HipEngine):
    """The real engine, except that embed() returns the rank's rows of a given E (the test is about the
    exchange and the search, not about A . P)."""

    def embed(self, indptr, indices, n_rows, d):
        return torch.from_numpy(E_all[pipe.lo:pipe.hi]).to(self.device)


pipe = ShardedPipeline(GivenEmbeddings(ctx, dev), n, d, k, rank=rank, world_size=world, device=dev)
for _ in range(2):  # the second pass reuses every buffer
    idx, dst, _ = pipe.step(None, None)
torch.cuda.synchronize(dev)
ut, uq = ctx.last_unique()
np.savez(os.path.join(outdir, "rank%d.npz" % rank), idx=idx.cpu().numpy(), dist=dst.cpu().numpy(), lo=pipe.lo,
         hi=pipe.hi, unique_targets=ut, unique_queries=uq)
dist.barrier()
dist.destroy_process_group()
ctx.close()
