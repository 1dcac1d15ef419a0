"""One rank's share of a row-sharded run AS BASELINE.json STATES IT, on the one GPU a test box has (shared by
tests/test_gpu_configs.py and devtools/rank_share_real.py):

  real generator   fedrann_amd.synth.synth(R, reads=...)  -> binary read x k-mer CSR
  projection       build_precompute_matrix(counts, d) at the config's F (25 M features at 10 M reads)
                   -> fdr_projection_load
  embed            fdr_csr_compact + fdr_embed_dev for EVERY row: the rank's own block, and the other ranks'
                   blocks in pieces on the same GPU (what the all-gather would deliver)
  k-NN             fdr_normalize_dev + fdr_knn_dev: queries = the rank's block, targets = all N rows

Parity: the rank's block of E against orc_embed bit for bit; query rows sampled PER EXECUTION PATH (tests/_strata.py:
fdr_last_query_paths) against the exact CPU oracle over ALL targets -- the targets are streamed back from HBM in pieces of <= 1 M rows and the per-piece
top-k lists merged on the host by (distance bits, index), so host memory stays around 1-2 GB; whole-result
properties on every row."""
import os
import time

import numpy as np


def host_threads(per_thread_gb=3.0):
    """Generator threads the host's free memory allows (a 100 k-read chunk holds ~2 GB of temporaries)."""
    n = min(16, len(os.sched_getaffinity(0)))
    try:
        import psutil
        n = max(2, min(n, int(psutil.virtual_memory().available / 2**30 / per_thread_gb)))
    except Exception:
        pass
    return n


def run_rank_share(ctx, oracle, R, d, k, doubling, ranks=8, rank=0, sample=64, piece_reads=1_600_000, seed=602,
                   reps=1, log=None, unique_split=False):
    import torch
    from fedrann_amd.distributed import HipEngine, shard_rows
    from fedrann_amd.precompute import build_precompute_matrix
    from fedrann_amd.synth import synth

    say = log or (lambda *a: None)
    dev = torch.device("cuda", 0)
    eng = HipEngine(ctx, dev)
    per = 2 if doubling else 1
    n = R * per
    S, blocks = shard_rows(n, ranks)
    lo, hi = blocks[rank]
    nq = hi - lo
    chunk = 100_000
    threads = host_threads()
    info = {"reads": R, "rows": n, "dim": d, "knn": k, "doubling": bool(doubling), "ranks": ranks, "rank": rank,
            "query_rows": nq, "host_threads": threads}

    # ---- the rank's own block: its CSR, E on the device, E from the oracle ----------------------------------
    t0 = time.perf_counter()
    r_lo, r_hi = lo // per, -(-hi // per)
    g_lo = r_lo // chunk * chunk
    s = synth(R, seed=seed, doubling=doubling, chunk=chunk, reads=(g_lo, r_hi), threads=threads)
    skip = lo - g_lo * per
    ip = np.ascontiguousarray(s["indptr"][skip:skip + nq + 1] - s["indptr"][skip])
    ix = np.ascontiguousarray(s["indices"][s["indptr"][skip]:s["indptr"][skip + nq]])
    F, counts = s["n_features"], s["counts"]
    del s
    info["synth_block_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    P = build_precompute_matrix(counts, d)
    ctx.projection_load(P.indptr, P.indices, P.data, F, d)
    info.update(n_features=int(F), projection_nnz=int(P.nnz), projection_s=time.perf_counter() - t0)
    say("block CSR %d rows, %d ids; F = %d, nnz(P) = %d" % (nq, ix.size, F, P.nnz))

    def embed_rows(ip_, ix_, out):
        cip, cix = ctx.csr_compact(ip_, ix_)
        d_ip, d_ix = torch.from_numpy(cip).to(dev), torch.from_numpy(cix).to(dev)
        m = ip_.size - 1
        ctx.embed_dev(m, d_ip.data_ptr(), d_ix.data_ptr(), out.data_ptr(), eng._stream())
        torch.cuda.synchronize(dev)
        return int(cix.size)

    E_all = torch.empty((n, d), dtype=torch.float32, device=dev)
    E_blk = torch.empty((nq, d), dtype=torch.float32, device=dev)
    ctx.timing(True)
    live = embed_rows(ip, ix, E_blk)
    info["embed_block_ms"] = ctx.timing_read(0)[1]
    ctx.timing(False)
    info.update(block_nnz=int(ix.size), block_nnz_with_projection_entries=live)
    t0 = time.perf_counter()
    want_E = oracle.embed(ip, ix, (P.indptr, P.indices, P.data), F, d)
    info["oracle_embed_s"] = time.perf_counter() - t0
    got_E = E_blk.cpu().numpy()
    info["embed_block_identical_to_oracle"] = bool(np.array_equal(got_E.view(np.uint32), want_E.view(np.uint32)))
    del want_E, got_E, ip, ix

    # ---- every row of the matrix, in pieces (the other ranks' blocks) -----------------------------------------
    t0 = time.perf_counter()
    nnz_all = 0
    for p0 in range(0, R, piece_reads):
        p1 = min(R, p0 + piece_reads)
        sp = synth(R, seed=seed, doubling=doubling, chunk=chunk, reads=(p0, p1), threads=threads)
        nnz_all += int(sp["indices"].size)
        embed_rows(sp["indptr"], sp["indices"], E_all[p0 * per:p1 * per])
        del sp
        say("embedded reads [%d, %d) of %d  (%.0f s)" % (p0, p1, R, time.perf_counter() - t0))
    info.update(nnz=nnz_all, synth_and_embed_all_s=time.perf_counter() - t0)
    info["block_equals_piecewise_rows"] = bool(torch.equal(E_blk, E_all[lo:hi]))
    del E_blk

    # ---- normalise, k-NN of the rank's block against all rows ------------------------------------------------
    dp = ctx.padded_dim(d)
    Ehat = torch.zeros((n, dp), dtype=torch.float32, device=dev)
    zero = torch.zeros((n,), dtype=torch.uint8, device=dev)
    eng.normalize(E_all, Ehat, zero)
    torch.cuda.synchronize(dev)
    info["workspace_gb"] = ctx.knn_workspace_bytes(nq, n, d, k) / 1e9
    best = None
    for rep in range(max(1, reps)):
        ctx.timing(True)
        t0 = time.perf_counter()
        idx, dst = eng.knn(Ehat[lo:hi], zero[lo:hi], nq, Ehat, zero, n, d, k)
        torch.cuda.synchronize(dev)
        sec = time.perf_counter() - t0
        kinds = {}
        from fedrann_amd import _lib
        for i, name in enumerate(_lib.KERNELS):
            ms = ctx.timing_read(i)[1]
            if ms:
                kinds[name] = ms
        ctx.timing(False)
        say("k-NN pass %d: %.2f s  %s" % (rep, sec, kinds))
        if best is None or sec < best:
            best, info["kernels_ms"] = sec, kinds
    ut, uq = ctx.last_unique()
    launches, queues = ctx.last_prefilter_launches()
    info.update(knn_seconds=best, unique_targets=ut, unique_queries=uq, uncertified=ctx.last_uncertified(),
                prefilter_launches=launches, prefilter_queues=queues,
                node_read_pairs_per_s_if_every_rank_takes_as_long=n * k / best,
                prefilter_pflops_on_unique_rows=(2.0 * ut * uq * d / (info["kernels_ms"]["knn_prefilter"] * 1e-3) / 1e15
                                                 if info["kernels_ms"].get("knn_prefilter") else None))
    paths = ctx.last_query_paths(nq)  # (before the workspace is used again)
    idx_keep, dst_keep = idx.clone(), dst.clone()
    # rows of the block that share their normalised row, bit for bit, with another row of the matrix -- counted
    # independently of the library (a 64-bit hash of the row's bits, torch on the device)
    g = torch.Generator(device=dev)
    g.manual_seed(12345)
    w = torch.randint(-(2 ** 62), 2 ** 62, (dp,), dtype=torch.int64, device=dev, generator=g) | 1
    hsh = torch.empty((n,), dtype=torch.int64, device=dev)
    for c0 in range(0, n, 1 << 20):
        c1 = min(n, c0 + (1 << 20))
        hsh[c0:c1] = (Ehat[c0:c1].view(torch.int32).to(torch.int64) * w).sum(1)
    _, inv, mult = torch.unique(hsh, return_inverse=True, return_counts=True)
    members_expected = int((mult[inv[lo:hi]] > 1).sum().item())
    del hsh, inv, mult
    # ---- what the rank does in distributed.ShardedPipeline when the rows repeat: classes of all rows, k-NN of ITS SHARE
    # of the unique rows (1 / ranks of them, not the unique rows of its block), expansion after the exchange -- timed
    # here without the exchange; its share's rows that are also unique rows of the block must equal the direct result
    if unique_split:
        nq_max = -(-n // ranks)
        nu = eng.knn_classes(Ehat, zero, n, d, k, nq_max)
        if nu > 0:
            Su = -(-nu // ranks)
            iu = torch.zeros((Su, k), dtype=torch.int32, device=dev)
            du = torch.zeros((Su, k), dtype=torch.float32, device=dev)
            best2 = None
            for rep in range(max(1, reps)):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                eng.knn_classes(Ehat, zero, n, d, k, nq_max)
                eng.knn_unique(rank * Su, min(nu, (rank + 1) * Su), k, iu, du)
                torch.cuda.synchronize(dev)
                sec2 = time.perf_counter() - t0
                best2 = sec2 if best2 is None else min(best2, sec2)
            info.update(unique_rows=int(nu), unique_rows_per_rank=int(min(nu, (rank + 1) * Su) - rank * Su),
                        unique_split_seconds=best2,
                        node_read_pairs_per_s_unique_split=n * k / best2)
            say("unique-row split: %d of %d unique rows, classes + k-NN %.3f s" % (info["unique_rows_per_rank"], nu, best2))
            del iu, du
    del Ehat
    idx_h, dst_h = idx_keep.cpu().numpy(), dst_keep.cpu().numpy()
    del idx, dst, idx_keep, dst_keep

    # ---- whole-result properties ----------------------------------------------------------------------------
    key = dst_h.view(np.uint32).astype(np.uint64) << np.uint64(32) | idx_h.astype(np.uint64)
    info["keys_strictly_ascending"] = bool(np.all(key[:, 1:] > key[:, :-1]))
    info["values_in_range"] = bool(idx_h.min() >= 0 and idx_h.max() < n and dst_h.min() >= 0 and dst_h.max() <= 1)
    del key

    # ---- sampled exact oracle over ALL targets, targets streamed back in pieces --------------------------------
    t0 = time.perf_counter()
    zero_h = zero.cpu().numpy()
    # the oracle's rows: up to `sample` from EVERY execution path the library reports for this call (tests/_strata.py:
    # certified, range pass, exact kernel, all-zero, member of a duplicate-row class, a plateau query inside a class,
    # the block's first / last rows) -- a uniform sample of a few hundred rows meets the rare paths by chance only
    from _strata import stratified_rows
    picked, path_counts, path_taken = stratified_rows(paths, per=sample, seed=1)
    rows = lo + picked
    info.update(path_counts=path_counts, path_rows_sampled=path_taken, path_sample_per_stratum=sample,
                path_zero_count_matches_flags=bool(path_counts["zero"] == int(zero_h[lo:hi].sum())),
                path_class_members_match_row_hashes=bool(path_counts["class_member"] == members_expected))
    Eq = E_all[torch.from_numpy(rows).to(dev)].cpu().numpy()
    Qh, _, qz = oracle.normalize(Eq)
    keys = []
    step = 1_000_000
    for t_lo in range(0, n, step):
        t_hi = min(n, t_lo + step)
        Th, _, tz = oracle.normalize(E_all[t_lo:t_hi].cpu().numpy())
        assert np.array_equal(tz, zero_h[t_lo:t_hi])
        if t_hi - t_lo < k:  # (never at these sizes; a last piece shorter than k would need padding)
            raise AssertionError("piece shorter than k")
        wi, wd = oracle.knn_normalized(Qh, qz, Th, tz, k, t_base=t_lo)
        keys.append(wd.view(np.uint32).astype(np.uint64) << np.uint64(32) | wi.astype(np.uint64))
    allk = np.sort(np.concatenate(keys, axis=1), axis=1)[:, :k]
    want_idx = (allk & np.uint64(0xffffffff)).astype(np.int32)
    want_bits = (allk >> np.uint64(32)).astype(np.uint32)
    got_idx, got_bits = idx_h[rows - lo], dst_h[rows - lo].view(np.uint32)
    info.update(oracle_rows=int(rows.size), oracle_s=time.perf_counter() - t0,
                identical_indices=bool(np.array_equal(got_idx, want_idx)),
                identical_distance_bits=bool(np.array_equal(got_bits, want_bits)),
                zero_row_fraction=float(zero_h.mean()))
    return info


def assert_rank_share(info):
    assert info["embed_block_identical_to_oracle"], "E of the rank's block differs from orc_embed"
    assert info["block_equals_piecewise_rows"], "the block's rows differ from the piecewise-embedded rows"
    assert info["keys_strictly_ascending"] and info["values_in_range"]
    assert info["identical_indices"], "neighbour indices differ from the oracle"
    assert info["identical_distance_bits"], "distance bits differ from the oracle"
    # the strata: the codes partition the rows (checked when they are drawn), agree with what is known independently
    # of the library, and every path this workload exercises gave the oracle its rows
    assert info["path_zero_count_matches_flags"] and info["path_class_members_match_row_hashes"]
    c, t = info["path_counts"], info["path_rows_sampled"]
    for name in c:
        assert t[name] == min(c[name], info["path_sample_per_stratum"]), (name, c, t)
    assert c["certified"] > 0 and c["range"] >= 64 and c["zero"] > 0 and c["class_member"] >= 64, c
