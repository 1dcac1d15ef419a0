"""BASELINE.json configs 4 and 5 on the one GPU a test box has: ONE RANK'S SHARE of the 8-GPU runs through the
device API (fdr_embed_dev, fdr_normalize_dev + fdr_knn_dev: what distributed.ShardedPipeline calls).

  config 4: 10 M reads, d = 128, k = 20, rows sharded 1.25 M per GPU   -> queries = rows [0, 1.25 M) of 10 M targets
  config 5: 10 M reads doubled (20 M rows), d = 256, k = 50            -> queries = rows [0, 2.5 M) of 20 M targets

Both at FULL per-rank size from the real generator (tests/_rank_share.py: synth CSR, F = 25 M projection tables,
embed of every row on this GPU in pieces, then the rank's k-NN), checked against the CPU oracle with the targets
streamed back from HBM in pieces.  The smaller cases below use device-made embeddings (locus prototypes with a few
non-zero components + per-read drop-outs: sparse rows, near-duplicates, exact duplicates, all-zero rows) for shapes
the two configs do not cover.  Parity: a sample of query rows against the exact CPU oracle over ALL targets, bit
for bit, + whole-result properties."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _device_embeddings(n, d, nnz, loci, seed, doubling=False):
    import torch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    rows = n // 2 if doubling else n
    comp = torch.randint(0, d, (loci, nnz), device=dev, generator=g)
    val = torch.randint(1, 9, (loci, nnz), device=dev, generator=g).float() * 0.37
    val = val * (torch.randint(0, 2, (loci, nnz), device=dev, generator=g).float() * 2 - 1)
    E = torch.zeros((n, d), dtype=torch.float32, device=dev)
    step = 1 << 20
    for lo in range(0, rows, step):
        m = min(step, rows - lo)
        locus = torch.randint(0, loci, (m,), device=dev, generator=g)
        keep = (torch.rand((m, nnz), device=dev, generator=g) < 0.8).float()  # sequencing errors drop components
        blk = torch.zeros((m, d), dtype=torch.float32, device=dev)
        blk.scatter_add_(1, comp[locus], val[locus] * keep)
        if doubling:  # row 2i = the read, row 2i + 1 = its strand mirror (here: the components reversed)
            E[2 * lo:2 * (lo + m):2] = blk
            E[2 * lo + 1:2 * (lo + m):2] = blk.flip(1)
        else:
            E[lo:lo + m] = blk
    return E


def _check_rank_share(ctx, oracle, E, nq, k, sample=256):
    import torch
    from fedrann_amd.distributed import HipEngine
    dev = E.device
    n, d = E.shape
    eng = HipEngine(ctx, dev)
    dp = ctx.padded_dim(d)
    Ehat = torch.zeros((n, dp), dtype=torch.float32, device=dev)
    zero = torch.zeros((n,), dtype=torch.uint8, device=dev)
    eng.normalize(E, Ehat, zero)
    need = ctx.knn_workspace_bytes(nq, n, d, k)
    assert need < 200e9
    idx, dst = eng.knn(Ehat[:nq], zero[:nq], nq, Ehat, zero, n, d, k)
    torch.cuda.synchronize(dev)
    ut, uq = ctx.last_unique()
    idx, dst = idx.cpu().numpy(), dst.cpu().numpy()
    # whole-result properties: keys strictly ascending per row, indices in range, distances in [0, 1]
    key = dst.view(np.uint32).astype(np.uint64) << np.uint64(32) | idx.astype(np.uint64)
    assert np.all(key[:, 1:] > key[:, :-1])
    assert idx.min() >= 0 and idx.max() < n and dst.min() >= 0 and dst.max() <= 1
    # sampled exact oracle over ALL targets (incl. all-zero query rows and the block's first / last rows)
    Eh_host, _, z_host = oracle.normalize(E.cpu().numpy())
    rng = np.random.default_rng(1)
    zr = np.flatnonzero(z_host[:nq])[:16]
    rows = np.unique(np.concatenate([rng.choice(nq, size=sample, replace=False), zr, [0, 1, nq - 2, nq - 1]]))
    wi, wd = oracle.knn_normalized(Eh_host[rows], z_host[rows], Eh_host, z_host, k)
    assert np.array_equal(idx[rows], wi)
    assert np.array_equal(dst[rows].view(np.uint32), wd.view(np.uint32))
    return ut, uq


def _real_rank_share(ctx, oracle, tag, **kw):
    import json
    import os
    import sys
    from _rank_share import assert_rank_share, run_rank_share
    ctx.set_knn_mode("auto")
    ctx.set_dedup_mode("auto")
    info = run_rank_share(ctx, oracle, log=lambda *a: print("[%s]" % tag, *a, file=sys.stderr, flush=True), **kw)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):  # (the measured line of this run, for profiles/)
        with open(os.path.join(out, "%s_rank_share.json" % tag), "w") as f:
            f.write(json.dumps(info) + "\n")
    assert_rank_share(info)
    return info


def test_config4_real_generator_one_rank_share_full_size(ctx, oracle):
    """Config 4 as BASELINE states it, rank 0's share: synth(10 M reads) CSR -> projection tables at F = 25 M ->
    embed (the block's E against orc_embed, bit for bit; the other ranks' blocks embedded on this GPU too) ->
    1.25 M query rows against all 10 M rows, d = 128, k = 20."""
    info = _real_rank_share(ctx, oracle, "config4", R=10_000_000, d=128, k=20, doubling=False)
    assert info["n_features"] == 25_000_000 and info["query_rows"] == 1_250_016  # (blocks are multiples of 32 rows)
    assert 20 <= info["unique_targets"] <= 10_000_000 and info["unique_queries"] <= 1_250_000
    assert info["prefilter_queues"] == 2 and info["prefilter_launches"] > 30  # synchronised rounds, two queues


def test_config5_real_generator_one_rank_share_full_size(ctx, oracle):
    """Config 5 as BASELINE states it, one rank's share at full size: 10 M reads with fwd / rev doubling = 20 M
    rows (39 target segments of 48), d = 256, k = 50 (K' = 58), 2.5 M query rows."""
    info = _real_rank_share(ctx, oracle, "config5", R=10_000_000, d=256, k=50, doubling=True, sample=64)
    assert info["rows"] == 20_000_000 and info["query_rows"] == 2_500_000
    assert info["unique_targets"] <= 20_000_000


def test_config5_shape_doubled_rows_d256_k50(ctx, oracle):
    """Config 5's shape at 1 M doubled rows: d = 256 (DP = 256 kernels), k = 50 (K' = 58: the 2 x 32-key register
    lists), fwd / mirrored row pairs; queries = one rank's eighth."""
    ctx.set_knn_mode("auto")
    ctx.set_dedup_mode("auto")
    E = _device_embeddings(1_000_000, 256, nnz=8, loci=300_000, seed=5, doubling=True)
    _check_rank_share(ctx, oracle, E, 125_000, 50)
    _check_rank_share(ctx, oracle, E[:200_000].contiguous(), 200_000, 50, sample=128)  # and all-pairs at 200 k rows


def test_reference_default_dimension_500_in_rounds(ctx, oracle):
    """d = 500 (the reference's default -n; DP = 512 kernels, two workgroups per CU), k = 50, enough query blocks
    for the prefilter pass to run in synchronised rounds on two queues: a rank's quarter of 600 k rows."""
    ctx.set_knn_mode("auto")
    ctx.set_dedup_mode("auto")
    E = _device_embeddings(600_000, 500, nnz=8, loci=250_000, seed=6)
    _check_rank_share(ctx, oracle, E, 150_000, 50, sample=128)
    assert ctx.last_prefilter_launches()[1] == 2


def test_ping_pong_kernel_short_lists_d256_and_d500(ctx, oracle):
    """The ping-pong candidate pass with 2 x 16-key register lists (K' <= 32: k = 20), which no BASELINE config runs:
    d = 256 (knn_prefilter_pp_kernel<256, 8, 16>) on 200 k rows all pairs, d = 500 (<512, 8, 16>: eight-unit stages,
    two tiles of accumulators) on a rank's quarter of 600 k rows -- 512 query blocks and more, so that prefilter_shape()
    chooses it; sparse rows with exact duplicates and all-zero rows."""
    ctx.set_knn_mode("auto")
    ctx.set_dedup_mode("auto")
    E = _device_embeddings(400_000, 256, nnz=8, loci=150_000, seed=8, doubling=True)
    _check_rank_share(ctx, oracle, E[:200_000].contiguous(), 200_000, 20, sample=128)
    assert ctx.last_prefilter_launches()[1] == 2  # (launches of one workgroup per CU on two queues)
    del E
    E = _device_embeddings(600_000, 500, nnz=8, loci=250_000, seed=9)
    _check_rank_share(ctx, oracle, E, 150_000, 20, sample=128)
    assert ctx.last_prefilter_launches()[1] == 2


def test_rank_slice_of_one_million_rows_d128_in_one_launch(ctx, oracle):
    """One rank's share of 1 M rows at d = 128 / k = 20 (the bench default on eight or nine GPUs): between one and 1.8 query blocks
    per workgroup slot and >= 4 targets per query, where prefilter_shape() takes the eight-wave four-stage pass in ONE
    launch of guided segments -- a combination neither the all-pairs cases (rounds, or the four-wave shape) nor
    config 4's share (rounds) run."""
    ctx.set_knn_mode("auto")
    ctx.set_dedup_mode("auto")
    E = _device_embeddings(1_000_000, 128, nnz=8, loci=400_000, seed=11)
    ut, uq = _check_rank_share(ctx, oracle, E, 110_000, 20, sample=192)  # (a ninth: 430 query blocks of 256)
    assert 65_536 <= uq <= 117_900 and ut >= 4 * uq  # (256 .. 460 blocks of 256 unique queries on 256 CUs)
    assert ctx.last_prefilter_launches() == (1, 1)


def test_per_rank_workspace_of_configs_4_and_5_fits_hbm(ctx):
    """fdr_knn_workspace_bytes for one rank of config 4 (1.25 M x 10 M, d = 128, k = 20) and of config 5
    (2.5 M x 20 M, d = 256, k = 50), plus the gathered embeddings and the results, against 288 GB of HBM."""
    hbm = ctx.device_info()["hbm_bytes"]
    assert hbm > 250e9
    # ... and config 4's whole matrix on ONE GPU (10 M x 10 M: the N = 1 point of its scaling curve, profiles/r4_bench_10m.json)
    for nq, nt, d, k in ((1_250_000, 10_000_000, 128, 20), (2_500_000, 20_000_000, 256, 50), (10_000_000, 10_000_000, 128, 20)):
        ws = ctx.knn_workspace_bytes(nq, nt, d, k)
        total = ws + nt * ctx.padded_dim(d) * 4 + nt + nq * k * 8
        assert 0 < ws and total < 0.9 * hbm, (nq, nt, d, k, ws, total)
