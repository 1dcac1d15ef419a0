"""Row-sharded embed + k-NN over the GPUs of one node (one process per GPU).

The path shards by read rows: rank g embeds and normalises rows [g*S, (g+1)*S) of the feature
matrix, the normalised embeddings (and their zero-row flags) are exchanged with ONE all-gather
(RCCL over xGMI when the process group's backend is "nccl"), and each rank then searches its own
rows against all N targets.  No merge step: a query's full top-k is produced on its owner rank.  When the
gathered rows repeat (>= 5 % duplicates; the sparse projections of overlapping reads often coincide), the
ranks split the UNIQUE rows instead and exchange those results with a second, small all-gather (see step()).

torch is used for device memory, streams and torch.distributed only; the arithmetic is in
libfedrann_hip.so (HipEngine).  The engine is injected so that the sharding / exchange logic can be
exercised on CPU with the gloo backend (tests/test_distributed.py drives it with an oracle-backed
engine); HipEngine itself refuses to run without a GPU.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import _lib


def shard_rows(n_rows, world_size, align=32):
    """Contiguous row blocks, equal size S (a multiple of `align`, so fwd/rev row pairs and the
    32-row tiles never straddle ranks); trailing ranks may hold fewer (or no) real rows."""
    S = -(-n_rows // world_size)
    S = -(-S // align) * align
    return S, [(min(n_rows, g * S), min(n_rows, (g + 1) * S)) for g in range(world_size)]


class HipEngine:
    """The three device stages on one GPU, on torch-owned HBM buffers and torch's current stream."""

    def __init__(self, context, device):
        if not torch.cuda.is_available():
            raise _lib.FedrannHipError("HipEngine needs a GPU: there is no CPU fallback")
        self.ctx = context
        self.device = torch.device(device)
        self._ws = None

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def padded_dim(self, d):
        return self.ctx.padded_dim(d)

    def embed(self, indptr, indices, n_rows, d):
        E = torch.empty((n_rows, d), dtype=torch.float32, device=self.device)
        if n_rows:
            self.ctx.embed_dev(n_rows, indptr.data_ptr(), indices.data_ptr(), E.data_ptr(), self._stream())
        return E

    def normalize(self, E, Ehat_out, zero_out):
        n, d = E.shape
        if n:
            self.ctx.normalize_dev(E.data_ptr(), n, d, Ehat_out.data_ptr(), zero_out.data_ptr(),
                                   self._stream())

    def knn(self, Qhat, qzero, nq, That, tzero, nt, d, k):
        idx = torch.empty((nq, k), dtype=torch.int32, device=self.device)
        dst = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        if nq == 0:
            return idx, dst
        need = self.ctx.knn_workspace_bytes(nq, nt, d, k)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        self.ctx.knn_dev(Qhat.data_ptr(), qzero.data_ptr(), nq, That.data_ptr(), tzero.data_ptr(), nt,
                         0, d, k, idx.data_ptr(), dst.data_ptr(), self._ws.data_ptr(),
                         self._ws.numel(), self._stream())
        return idx, dst

    # -- duplicate-row classes across ranks (fdr_knn_classes_dev / _unique_dev / _expand_dev) --------------
    def knn_classes(self, That, tzero, nt, d, k, nq_max):
        """Build the classes of the gathered target set; returns the number of unique rows, 0 = use knn()."""
        need = self.ctx.knn_workspace_bytes(nq_max, nt, d, k)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self.ctx.knn_classes_dev(That.data_ptr(), tzero.data_ptr(), nt, d, k, nq_max, self._ws.data_ptr(),
                                        self._ws.numel(), self._stream())

    def knn_unique(self, u_lo, u_hi, k, out_idx, out_dst):
        """k-NN of the unique rows [u_lo, u_hi) into the first u_hi - u_lo rows of out_idx / out_dst."""
        if u_hi > u_lo:
            self.ctx.knn_unique_dev(u_lo, u_hi, out_idx.data_ptr(), out_dst.data_ptr(), self._stream())

    def knn_expand(self, q0, nq, k, packed_u_all):
        """packed_u_all int32 [n_unique (padded), 2 k]: a unique row's k indices, then its k distance bit patterns."""
        idx = torch.empty((nq, k), dtype=torch.int32, device=self.device)
        dst = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        if nq:
            base = packed_u_all.data_ptr()
            self.ctx.knn_expand_dev(q0, nq, 0, base, base + 4 * k, idx.data_ptr(), dst.data_ptr(), self._stream(),
                                    u_row_stride=2 * k)
        return idx, dst


class ShardedPipeline:
    """embed -> normalise -> all-gather -> k-NN for this rank's row block."""

    def __init__(self, engine, n_rows_total, d, k, rank=0, world_size=1, group=None, device="cpu"):
        self.engine, self.n, self.d, self.k = engine, int(n_rows_total), int(d), int(k)
        self.rank, self.world, self.group = rank, world_size, group
        self.device = torch.device(device)
        self.S, self.blocks = shard_rows(self.n, world_size)
        self.lo, self.hi = self.blocks[rank]
        self.dp = engine.padded_dim(d)
        G, S = world_size, self.S
        # gathered buffers (all ranks' normalised rows, row i of the matrix at position i)
        self.Ehat_all = torch.zeros((G * S, self.dp), dtype=torch.float32, device=self.device)
        self.zero_all = torch.zeros((G * S,), dtype=torch.uint8, device=self.device)
        if G > 1:
            self.Ehat_loc = torch.zeros((S, self.dp), dtype=torch.float32, device=self.device)
            self.zero_loc = torch.zeros((S,), dtype=torch.uint8, device=self.device)
        else:
            self.Ehat_loc, self.zero_loc = self.Ehat_all, self.zero_all
        self._pu = self._pu_loc = self._iu_loc = self._du_loc = None  # unique-row results (sized on first use)

    def step(self, indptr_local, indices_local):
        """indptr_local / indices_local: this rank's CSR rows (indptr rebased to 0) as tensors on
        the pipeline's device.  Returns (idx int32 [rows, k], dist float32 [rows, k], E)."""
        nloc = self.hi - self.lo
        E = self.engine.embed(indptr_local, indices_local, nloc, self.d)
        self.engine.normalize(E, self.Ehat_loc, self.zero_loc)
        if self.world > 1:
            # two collectives: the rows (S x DP fp32 per rank; the class layer compares whole rows, so the search
            # cannot start on less) and the zero flags (S bytes; packing them behind a rank's rows would leave the
            # gathered rows non-contiguous, and the kernels address row i at i * DP)
            w = dist.all_gather_into_tensor(self.zero_all, self.zero_loc, group=self.group, async_op=True)
            dist.all_gather_into_tensor(self.Ehat_all, self.Ehat_loc, group=self.group)
            w.wait()
        q0 = self.rank * self.S
        if self.world > 1 and hasattr(self.engine, "knn_classes"):
            # Split the UNIQUE rows over the ranks instead of searching every duplicate query row on every
            # rank that holds a member of its class: all ranks build the same class tables from the gathered
            # rows (the decision to do so is a function of the exact unique count: the same on every rank), each
            # searches its share of the unique rows, ONE more all-gather exchanges the shares (a unique row's k
            # indices and k distance bit patterns side by side), and each rank expands its own rows.
            nq_max = -(-self.n // self.world)
            nu = self.engine.knn_classes(self.Ehat_all, self.zero_all, self.n, self.d, self.k, nq_max)
            if nu > 0:
                Su, k = -(-nu // self.world), self.k
                u_lo, u_hi = min(nu, self.rank * Su), min(nu, (self.rank + 1) * Su)
                if self._pu is None or self._pu.shape[0] != Su * self.world:
                    self._pu = torch.zeros((Su * self.world, 2 * k), dtype=torch.int32, device=self.device)
                    self._pu_loc = torch.zeros((Su, 2 * k), dtype=torch.int32, device=self.device)
                    self._iu_loc = torch.zeros((Su, k), dtype=torch.int32, device=self.device)
                    self._du_loc = torch.zeros((Su, k), dtype=torch.float32, device=self.device)
                self.engine.knn_unique(u_lo, u_hi, k, self._iu_loc, self._du_loc)
                self._pu_loc[:, :k] = self._iu_loc
                self._pu_loc[:, k:] = self._du_loc.view(torch.int32)
                dist.all_gather_into_tensor(self._pu, self._pu_loc, group=self.group)
                idx, dst = self.engine.knn_expand(q0, nloc, k, self._pu)  # (unique row u sits at row u)
                return idx, dst, E
        idx, dst = self.engine.knn(self.Ehat_all[q0:q0 + nloc], self.zero_all[q0:q0 + nloc], nloc,
                                   self.Ehat_all, self.zero_all, self.n, self.d, self.k)
        return idx, dst, E


def local_csr(indptr, indices, lo, hi):
    """Rows [lo, hi) of a host CSR, indptr rebased to 0 (numpy)."""
    ip = np.ascontiguousarray(indptr[lo:hi + 1] - indptr[lo], dtype=np.int64)
    ix = np.ascontiguousarray(indices[indptr[lo]:indptr[hi]], dtype=np.int32)
    return ip, ix
