"""The N > 1 path (row sharding + one all-gather + per-rank k-NN) on CPU: world_size 2 and 3 over
gloo, with the device stages played by the CPU oracle.  Checks that the sharded result equals the
unsharded one bit for bit, including a ragged last shard and an empty rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fedrann_amd.distributed import ShardedPipeline, local_csr, shard_rows
from fedrann_amd.precompute import build_precompute_matrix
from fedrann_amd.synth import synth


class OracleEngine:
    """Same interface as fedrann_amd.distributed.HipEngine, arithmetic by the CPU oracle.  The
    'Ehat layout' here is simply the normalised rows padded to a multiple of 128 columns."""

    def __init__(self, P, n_features):
        from oracle import oracle
        self.O, self.P, self.F = oracle, P, n_features

    def padded_dim(self, d):
        return -(-d // 128) * 128

    def embed(self, indptr, indices, n_rows, d):
        E = self.O.embed(indptr.numpy(), indices.numpy().astype(np.int64),
                         (self.P.indptr, self.P.indices, self.P.data), self.F, d)
        return torch.from_numpy(E)

    def normalize(self, E, Ehat_out, zero_out):
        n, d = E.shape
        if n:
            Eh, _, z = self.O.normalize(E.numpy())
            Ehat_out[:n, :d] = torch.from_numpy(Eh)
            zero_out[:n] = torch.from_numpy(z)

    def knn(self, Qhat, qzero, nq, That, tzero, nt, d, k):
        if nq == 0:
            return torch.empty((0, k), dtype=torch.int32), torch.empty((0, k), dtype=torch.float32)
        i, dd = self.O.knn_normalized(Qhat.numpy()[:, :d], qzero.numpy(), That.numpy()[:nt, :d],
                                      tzero.numpy()[:nt], k)
        return torch.from_numpy(i), torch.from_numpy(dd)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, R, d, k, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = synth(R, seed=11, m=60)
    P = build_precompute_matrix(s["counts"], d)
    n = len(s["indptr"]) - 1
    pipe = ShardedPipeline(OracleEngine(P, s["n_features"]), n, d, k, rank=rank, world_size=world)
    ip, ix = local_csr(s["indptr"], s["indices"], pipe.lo, pipe.hi)
    idx, dst, E = pipe.step(torch.from_numpy(ip), torch.from_numpy(ix))
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), idx=idx.numpy(), dist=dst.numpy(), lo=pipe.lo,
             hi=pipe.hi, E=E.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_rows():
    S, b = shard_rows(100_000, 8)
    assert S == 12512 and b[0] == (0, 12512) and b[7] == (87584, 100_000)
    S, b = shard_rows(70, 3)
    assert S == 32 and b == [(0, 32), (32, 64), (64, 70)]
    S, b = shard_rows(40, 4)
    assert S == 32 and b == [(0, 32), (32, 40), (40, 40), (40, 40)]  # ranks 2, 3 hold no rows


@pytest.mark.parametrize("world,R", [(2, 700), (3, 330), (4, 40)])
def test_sharded_equals_unsharded(tmp_path, oracle, world, R):
    d, k = 128, 10
    mp.spawn(_worker, args=(world, _free_port(), R, d, k, str(tmp_path)), nprocs=world, join=True)
    s = synth(R, seed=11, m=60)
    P = build_precompute_matrix(s["counts"], d)
    E = oracle.embed(s["indptr"], s["indices"], (P.indptr, P.indices, P.data), s["n_features"], d)
    wi, wd = oracle.knn(E, k)
    got_i, got_d, got_E = [], [], []
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert z["idx"].shape[0] == int(z["hi"]) - int(z["lo"])
        got_i.append(z["idx"])
        got_d.append(z["dist"])
        got_E.append(z["E"])
    assert np.array_equal(np.concatenate(got_E).view(np.uint32), E.view(np.uint32))
    assert np.array_equal(np.concatenate(got_i), wi)
    assert np.array_equal(np.concatenate(got_d).view(np.uint32), wd.view(np.uint32))
