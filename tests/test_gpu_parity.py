"""Parity of the HIP path (through the C-ABI) with the CPU oracle and the reference's golden
vectors.  Integer / index results and fp32 bit patterns must be IDENTICAL; the k-NN distances are
additionally checked within 1e-5 as BASELINE.json's north_star words it."""
import numpy as np
import pytest

from conftest import golden_embed_case
from fedrann_amd import _lib
from fedrann_amd.precompute import build_precompute_matrix
from fedrann_amd.synth import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["exact", "prefilter", "exact+classes", "prefilter+classes"])
def knn_mode(request, ctx):
    """Every test runs under both k-NN modes.  "prefilter" = fp16 MFMA candidate pass + certificate +
    exact fp32 re-rank (exact kernel for uncertified queries); its results must be the same bits.
    "+classes" forces the duplicate-row class layer (search unique rows, expand) at every size -- by
    default it only engages from 8192 target rows, which the large tests below cover."""
    mode, _, classes = request.param.partition("+")
    ctx.set_knn_mode(mode)
    ctx.set_dedup_mode("force" if classes else "auto")
    yield request.param
    ctx.set_knn_mode("auto")
    ctx.set_dedup_mode("auto")


def _skip_forced_classes(knn_mode):
    if "+" in knn_mode:
        pytest.skip("the class layer engages by itself at this size")


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _assert_knn_equal(got, want):
    gi, gd = got
    wi, wd = want
    assert np.array_equal(gi, wi), "neighbour indices / ranks differ in %d of %d cells" % (
        int((gi != wi).sum()), gi.size)
    assert np.abs(gd - wd).max() <= 1e-5
    assert np.array_equal(_bits(gd), _bits(wd)), "distance bit patterns differ"


# ---- E = A . P -------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["tiny", "mid"])
def test_embed_matches_reference_golden(ctx, tag):
    indptr, indices, P, F, d, E_bits = golden_embed_case(tag)
    # rows arrive in kmer_searcher's arbitrary order; the ABI wants ascending columns per row
    from fedrann_amd.feature_extraction import canonical_csr
    ip, ix = canonical_csr(indptr, indices, F)
    ctx.projection_load(P[0], P[1], P[2], F, d)
    E = ctx.embed(ip, ix)
    assert np.array_equal(_bits(E), E_bits)


@pytest.mark.parametrize("tag", ["tiny", "mid"])
def test_embed_of_compacted_csr_matches_reference_golden(ctx, tag):
    """fdr_csr_compact drops the ids whose projection row is empty before the upload; E must stay the
    reference's bits (the surviving ids keep their order, so the sequential fp32 sums are unchanged)."""
    indptr, indices, P, F, d, E_bits = golden_embed_case(tag)
    from fedrann_amd.feature_extraction import canonical_csr
    ip, ix = canonical_csr(indptr, indices, F)
    ctx.projection_load(P[0], P[1], P[2], F, d)
    cip, cix = ctx.csr_compact(ip, ix)
    nonempty = np.diff(P[0]) > 0
    keep = nonempty[ix]
    rows = np.repeat(np.arange(ip.size - 1), np.diff(ip))
    assert np.array_equal(cix, ix[keep])
    assert np.array_equal(np.diff(cip), np.bincount(rows[keep], minlength=ip.size - 1))
    assert np.array_equal(_bits(ctx.embed(cip, cix)), E_bits)


def test_compacted_csr_same_embedding_and_neighbours_on_synthetic(ctx):
    s = synth(20_000, seed=11)
    P = build_precompute_matrix(s["counts"], 128)
    ctx.projection_load(P.indptr, P.indices, P.data, s["n_features"], 128)
    cip, cix = ctx.csr_compact(s["indptr"], s["indices"], n_threads=3)
    assert cix.size < 0.5 * s["indices"].size  # density 1/sqrt(F): most ids are dead (78 % at F = 267 k)
    a = ctx.embed_knn(s["indptr"], s["indices"], 20, return_embedding=True)
    b = ctx.embed_knn(cip, cix, 20, return_embedding=True)
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))


def test_development_knobs_do_not_reach_the_release_library(ctx, oracle, monkeypatch):
    """FDR_KNN_DEBUG=1 used to skip the top-k slow path (timing experiments, wrong results); the release
    build no longer reads it, nor any other knob (tests/test_abi.py checks the names are gone)."""
    for name, val in (("FDR_KNN_DEBUG", "1"), ("FDR_KNN_EXTRA", "2"), ("FDR_KNN_NSEG", "7"), ("FDR_KNN_RANGE", "0")):
        monkeypatch.setenv(name, val)
    rng = np.random.default_rng(8)
    E = rng.standard_normal((12_000, 96)).astype(np.float32)
    E[rng.random(E.shape) < 0.9] = 0
    _assert_knn_equal(ctx.knn(E, 20), oracle.knn(E, 20))


@pytest.mark.parametrize("R,d,m", [(3000, 128, 200), (1500, 64, 50), (800, 200, 120), (500, 256, 400),
                                   (600, 500, 300)])
def test_embed_matches_oracle_on_synthetic(ctx, oracle, R, d, m):
    s = synth(R, seed=R + d, m=m)
    P = build_precompute_matrix(s["counts"], d)
    ctx.projection_load(P.indptr, P.indices, P.data, s["n_features"], d)
    E = ctx.embed(s["indptr"], s["indices"])
    want = oracle.embed(s["indptr"], s["indices"], (P.indptr, P.indices, P.data), s["n_features"], d)
    assert np.array_equal(_bits(E), _bits(want))


def test_embed_empty_rows_and_long_rows(ctx, oracle):
    _, _, P, F, d, _ = golden_embed_case("mid")
    rng = np.random.default_rng(4)
    rows = [np.sort(rng.choice(F, size=n, replace=False)) for n in (0, 1, 63, 64, 65, 0, 5000, 129, 0)]
    indptr, indices = oracle.rows_to_csr(rows)
    ctx.projection_load(P[0], P[1], P[2], F, d)
    E = ctx.embed(indptr, indices.astype(np.int32))
    want = oracle.embed(indptr, indices, P, F, d)
    assert np.array_equal(_bits(E), _bits(want))
    assert not E[0].any() and not E[5].any() and not E[8].any()


@pytest.mark.parametrize("d", [128, 500])
def test_embed_short_rows_eight_per_wave(ctx, oracle, d):
    """Rows of at most 64 ids -- what a compacted CSR holds -- take the embed kernel's short-row path: eight rows per
    wave and turn, one chunk each.  Lengths at the chunk's edges (0, 1, 63, 64), a row count that leaves the last
    group partly filled, a projection whose rows hold several entries; then the same rows with one long row in every
    third group (those groups take the long-row path)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(21 + d)
    F = 8192
    P = sp.random(F, d, density=0.02, format="csr", dtype=np.float32, random_state=5,
                  data_rvs=lambda n: rng.standard_normal(n).astype(np.float32))
    P.sort_indices()
    lens = rng.choice([0, 1, 2, 17, 40, 63, 64], size=8 * 150 + 3)
    rows = [np.sort(rng.choice(F, size=int(n), replace=False)) for n in lens]
    ctx.projection_load(P.indptr, P.indices, P.data, F, d)
    for variant in range(2):
        if variant == 1:
            for g in range(0, len(rows) - 8, 24):
                rows[g + 5] = np.sort(rng.choice(F, size=65 + g % 400, replace=False))
        indptr, indices = oracle.rows_to_csr(rows)
        E = ctx.embed(indptr, indices.astype(np.int32))
        want = oracle.embed(indptr, indices, (P.indptr, P.indices, P.data), F, d)
        assert np.array_equal(_bits(E), _bits(want))


def test_embed_dense_projection_rows(ctx, oracle):
    # a projection where most features have several entries: exercises the multi-entry loop
    rng = np.random.default_rng(10)
    F, d = 4096, 128
    import scipy.sparse as sp
    P = sp.random(F, d, density=0.05, format="csr", dtype=np.float32, random_state=3,
                  data_rvs=lambda n: rng.standard_normal(n).astype(np.float32))
    P.sort_indices()
    rows = [np.sort(rng.choice(F, size=int(n), replace=False)) for n in rng.integers(1, 300, size=500)]
    indptr, indices = oracle.rows_to_csr(rows)
    ctx.projection_load(P.indptr, P.indices, P.data, F, d)
    E = ctx.embed(indptr, indices.astype(np.int32))
    want = oracle.embed(indptr, indices, (P.indptr, P.indices, P.data), F, d)
    assert np.array_equal(_bits(E), _bits(want))


def test_embed_host_upload_in_chunks_raw_and_compacted(ctx, oracle):
    """fdr_embed's pipelined upload (host_upload.inc: raw chunks from the front over PCIe, chunks compacted by host
    threads from the back) on inputs above its 1 M-id threshold: a sparse projection (the helpers' chunks fit the
    staging buffer), a dense one (every id survives: the staging buffer overflows and the chunks go raw after all), rows
    without ids and one very long row.  E = the oracle's bits either way."""
    import scipy.sparse as sp
    rng = np.random.default_rng(14)
    for F, dens, nrows in ((200_000, 0.0005, 30_000), (4096, 0.6, 24_000)):
        P = sp.random(F, 128, density=dens, format="csr", dtype=np.float32, random_state=5,
                      data_rvs=lambda n: rng.standard_normal(n).astype(np.float32))
        P.sort_indices()
        lens = rng.integers(0, 160, size=nrows)
        lens[7] = 0
        lens[nrows // 2] = min(F, 60_000)
        rows = [np.sort(rng.choice(F, size=int(n), replace=False)) for n in lens]
        indptr, indices = oracle.rows_to_csr(rows)
        assert indices.size > (1 << 20)
        ctx.projection_load(P.indptr, P.indices, P.data, F, 128)
        E = ctx.embed(indptr, indices.astype(np.int32))
        want = oracle.embed(indptr, indices, (P.indptr, P.indices, P.data), F, 128)
        assert np.array_equal(_bits(E), _bits(want))


# ---- k-NN ------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,d,k", [(2000, 128, 20), (777, 128, 20), (3000, 64, 20), (1200, 200, 50),
                                   (900, 256, 50), (300, 128, 1), (513, 100, 64), (64, 128, 64),
                                   (33, 7, 33), (1100, 500, 50), (700, 512, 20), (400, 300, 64)])
def test_knn_dense_random_matches_oracle(ctx, oracle, n, d, k):
    E = np.random.default_rng(n + d + k).standard_normal((n, d)).astype(np.float32)
    _assert_knn_equal(ctx.knn(E, k), oracle.knn(E, k))


def test_knn_multi_segment_matches_oracle(ctx, oracle):
    # 20 000 rows -> several target segments per query block + the merge kernel
    E = np.random.default_rng(77).standard_normal((20000, 128)).astype(np.float32)
    _assert_knn_equal(ctx.knn(E, 20), oracle.knn(E, 20))


def test_knn_on_reference_embedding_with_ties_and_zero_rows(ctx, oracle):
    *_, E_bits = golden_embed_case("mid")
    E = E_bits.view(np.float32)
    assert (np.abs(E).sum(1) == 0).sum() > 0  # the golden matrix has all-zero rows
    for k in (20, 50):
        _assert_knn_equal(ctx.knn(E, k), oracle.knn(E, k))


def test_knn_heavy_ties(ctx, oracle):
    rng = np.random.default_rng(9)
    base = rng.standard_normal((50, 128)).astype(np.float32)
    onehot = np.zeros((400, 128), np.float32)
    onehot[np.arange(400), rng.integers(0, 6, size=400)] = rng.choice([-2.0, 3.0], size=400)
    E = np.concatenate([base, base, np.zeros((70, 128), np.float32), onehot, base[:25],
                        np.zeros((3, 128), np.float32)])
    E = E[rng.permutation(E.shape[0])]
    for k in (20, 50):
        _assert_knn_equal(ctx.knn(E, k), oracle.knn(E, k))


def test_reference_default_shape_d500_k50(ctx, oracle):
    # the reference CLI defaults: -n 500, --nndescent-n-neighbors 50, fwd/rev doubled rows
    s = synth(2500, seed=7, doubling=True)
    P = build_precompute_matrix(s["counts"], 500)
    ctx.projection_load(P.indptr, P.indices, P.data, s["n_features"], 500)
    idx, dist, E = ctx.embed_knn(s["indptr"], s["indices"], 50, return_embedding=True)
    want_E = oracle.embed(s["indptr"], s["indices"], (P.indptr, P.indices, P.data), s["n_features"], 500)
    assert np.array_equal(_bits(E), _bits(want_E))
    _assert_knn_equal((idx, dist), oracle.knn(want_E, 50))


def test_knn_synthetic_pipeline_config2_shape_small(ctx, oracle):
    # the bench workload's generator at a size the oracle finishes in seconds
    s = synth(6000, seed=602)
    P = build_precompute_matrix(s["counts"], 128)
    ctx.projection_load(P.indptr, P.indices, P.data, s["n_features"], 128)
    idx, dist, E = ctx.embed_knn(s["indptr"], s["indices"], 20, return_embedding=True)
    want_E = oracle.embed(s["indptr"], s["indices"], (P.indptr, P.indices, P.data), s["n_features"], 128)
    assert np.array_equal(_bits(E), _bits(want_E))
    _assert_knn_equal((idx, dist), oracle.knn(want_E, 20))
    _assert_knn_equal(ctx.knn(E, 20), (idx, dist))  # fused == separate


def test_knn_full_size_properties_and_sampled_oracle(ctx, oracle, knn_mode):
    """BASELINE config 2 (100k rows, d=128, k=20): size-independent properties over the whole
    result + exact oracle agreement on a sample of query rows."""
    _skip_forced_classes(knn_mode)
    s = synth(100_000, seed=602)
    P = build_precompute_matrix(s["counts"], 128)
    ctx.projection_load(P.indptr, P.indices, P.data, s["n_features"], 128)
    idx, dist, E = ctx.embed_knn(s["indptr"], s["indices"], 20, return_embedding=True)
    n = E.shape[0]
    key = _bits(dist).astype(np.uint64) << np.uint64(32) | idx.astype(np.uint64)
    assert np.all(key[:, 1:] > key[:, :-1])  # strictly ascending (dist, idx), no repeats
    assert idx.min() >= 0 and idx.max() < n and dist.min() >= 0 and dist.max() <= 1
    nonzero = np.abs(E).sum(1) > 0
    self_found = (idx == np.arange(n)[:, None]).any(1)
    assert np.all(self_found[nonzero])  # a non-zero row always finds itself (distance ~ 0)
    assert np.all(dist[nonzero, 0] <= 1e-6)
    idx2, dist2 = ctx.knn(E, 20)  # idempotence: same answer from the separate entry point
    assert np.array_equal(idx, idx2) and np.array_equal(_bits(dist), _bits(dist2))
    rows = np.random.default_rng(1).choice(n, size=384, replace=False)
    Eh, _, zero = oracle.normalize(E)
    wi, wd = oracle.knn_normalized(Eh[rows], zero[rows], Eh, zero, 20)
    _assert_knn_equal((idx[rows], dist[rows]), (wi, wd))


def test_knn_one_million_rows_sampled_oracle(ctx, oracle, knn_mode):
    """BASELINE config 3 (1M rows, d=128, k=20, ~6 non-zeros per embedding row => heavy ties, ~2 %
    all-zero rows, 17 % duplicate rows): exact oracle agreement on a sample of query rows +
    whole-result properties."""
    _skip_forced_classes(knn_mode)
    s = synth(1_000_000, seed=602)
    P = build_precompute_matrix(s["counts"], 128)
    ctx.projection_load(P.indptr, P.indices, P.data, s["n_features"], 128)
    idx, dist, E = ctx.embed_knn(s["indptr"], s["indices"], 20, return_embedding=True)
    n = E.shape[0]
    key = _bits(dist).astype(np.uint64) << np.uint64(32) | idx.astype(np.uint64)
    assert np.all(key[:, 1:] > key[:, :-1])
    assert idx.min() >= 0 and idx.max() < n and dist.min() >= 0 and dist.max() <= 1
    want_E = oracle.embed(s["indptr"], s["indices"], (P.indptr, P.indices, P.data), s["n_features"], 128)
    assert np.array_equal(_bits(E), _bits(want_E))
    Eh, _, zero = oracle.normalize(E)
    # the oracle's rows: >= 64 from EVERY execution path the library reports for this call (tests/_strata.py)
    from _strata import stratified_rows
    paths = ctx.last_query_paths(n)
    rows, counts, taken = stratified_rows(paths, per=64, seed=3)
    # rows in a duplicate-row class of several rows, counted independently (bitwise-equal normalised rows)
    _, inv, mult = np.unique(np.ascontiguousarray(Eh).view(np.dtype((np.void, Eh.shape[1] * 4))).ravel(),
                             return_inverse=True, return_counts=True)
    assert counts["class_member"] == int((mult[inv] > 1).sum()) and counts["class_member"] > 64
    if knn_mode.startswith("prefilter"):
        assert counts["zero"] == int(zero.sum()) and counts["zero"] > 64
        assert counts["certified"] > 0.9 * (n - counts["zero"]) and counts["range"] >= 64 and counts["range_in_class"] > 0
        assert taken["certified"] == 64 and taken["range"] == 64 and taken["zero"] == 64 and taken["class_member"] == 64
    else:
        assert counts["exact"] == n  # (exact mode: every row, the all-zero ones too, goes through the fp32 kernel)
        zr = np.flatnonzero(zero)[:32]
        rows = np.unique(np.concatenate([rows, zr]))
    wi, wd = oracle.knn_normalized(Eh[rows], zero[rows], Eh, zero, 20)
    _assert_knn_equal((idx[rows], dist[rows]), (wi, wd))


def test_knn_tight_clusters_all_ties(ctx, oracle):
    """Thousands of rows at distance exactly 0 from each other: every list boundary is an index tie,
    and queues overflow in the middle of tiles (regression: a retried candidate with dist == tau and a
    smaller index than a row the partner lane had just inserted was dropped)."""
    rng = np.random.default_rng(21)
    base = rng.standard_normal((4, 128)).astype(np.float32)
    T = base[rng.integers(0, 4, size=3000)] + 1e-4 * rng.standard_normal((3000, 128)).astype(np.float32)
    for k in (20, 50):
        _assert_knn_equal(ctx.knn(T, k), oracle.knn(T, k))
    D = np.repeat(base, 700, axis=0)  # exact duplicates
    _assert_knn_equal(ctx.knn(D, 33), oracle.knn(D, 33))


def test_prefilter_error_bound_and_fallback_accounting(ctx, oracle, knn_mode):
    """The certificate's eps must bound |fp16 similarity - fp32 chain| (checked on sampled pairs with
    numpy's fp16), and an input made of near-ties must be routed through the exact kernel."""
    if not knn_mode.startswith("prefilter"):
        pytest.skip("prefilter mode only")
    rng = np.random.default_rng(12)
    E = rng.standard_normal((4000, 128)).astype(np.float32)
    E[rng.random(E.shape) < 0.9] = 0  # sparse rows like the real embeddings
    E[np.abs(E).sum(1) == 0, 0] = 1
    Eh, _, _ = oracle.normalize(E)
    a, b = rng.integers(0, 4000, size=20000), rng.integers(0, 4000, size=20000)
    exact = np.array([oracle.pair_dist(Eh[i], Eh[j]) for i, j in zip(a[:2000], b[:2000])], np.float32)
    h = Eh.astype(np.float16).astype(np.float64)
    approx = 1.0 - (h[a[:2000]] * h[b[:2000]]).sum(1)
    assert np.abs(np.clip(approx, 0, 1) - exact).max() < 0.00105
    _assert_knn_equal(ctx.knn(E, 20), oracle.knn(E, 20))
    # 3000 copies of 4 directions + noise at the 1e-4 level: every top-20 boundary is a near-tie
    base = rng.standard_normal((4, 128)).astype(np.float32)
    T = base[rng.integers(0, 4, size=3000)] + 1e-4 * rng.standard_normal((3000, 128)).astype(np.float32)
    _assert_knn_equal(ctx.knn(T, 20), oracle.knn(T, 20))
    assert ctx.last_uncertified() > 0


def _rows_with_duplicate_classes(rng, n, d, n_unique):
    """Sparse rows drawn from n_unique distinct ones with very uneven multiplicities (classes of 1 up to
    thousands, incl. the all-zero class and rows that differ only by a positive scale)."""
    U = rng.standard_normal((n_unique, d)).astype(np.float32)
    U[rng.random(U.shape) < 0.95] = 0
    U[:40] = 0
    U[40:48, :] = 0
    U[40:48, 7] = 1.0  # eight "distinct" slots that are the same row
    w = rng.pareto(0.7, size=n_unique) + 0.02
    pick = rng.choice(n_unique, size=n, p=w / w.sum())
    E = U[pick]
    scale = np.where(rng.random(n) < 0.3, np.float32(2.0), np.float32(1.0))  # exact in fp32
    return (E * scale[:, None]).astype(np.float32)


@pytest.mark.parametrize("n,d,k,n_unique", [(40_000, 128, 20, 5000), (30_000, 200, 50, 300),
                                            (20_000, 128, 64, 400),
                                            # (the expansion packs floor(64 / k) queries into a wave: 64, 21, 9, 1)
                                            (20_000, 128, 1, 400), (20_000, 128, 3, 2000), (24_000, 128, 7, 6000),
                                            (25_000, 128, 33, 3000)])
def test_knn_duplicate_row_classes(ctx, oracle, n, d, k, n_unique):
    """Inputs dominated by duplicate rows: the class layer must return the same bits as the plain
    all-pairs search (members of a class in ascending index order, lists cut in the middle of a class)."""
    rng = np.random.default_rng(n + k)
    E = _rows_with_duplicate_classes(rng, n, d, n_unique)
    got = ctx.knn(E, k)
    ut, uq = ctx.last_unique()
    assert k <= ut < n // 2 and uq == ut  # the layer engaged (all rows are queries here)
    _assert_knn_equal(got, oracle.knn(E, k))


def test_device_api_query_subset_with_duplicate_classes(ctx, oracle):
    """A rank's query block is a slice of the targets; its classes are a subset of the target classes."""
    import torch
    from fedrann_amd.distributed import HipEngine
    dev = torch.device("cuda", 0)
    n = 36_000
    E = _rows_with_duplicate_classes(np.random.default_rng(77), n, 128, 2500)
    eng = HipEngine(ctx, dev)
    dE = torch.from_numpy(E).to(dev)
    Ehat = torch.zeros((n, 128), dtype=torch.float32, device=dev)
    zero = torch.zeros((n,), dtype=torch.uint8, device=dev)
    eng.normalize(dE, Ehat, zero)
    wi, wd = oracle.knn(E, 20)
    for lo, hi in ((0, n), (18_000, 27_000), (n - 4099, n), (1, 2050)):
        idx, dst = eng.knn(Ehat[lo:hi], zero[lo:hi], hi - lo, Ehat, zero, n, 128, 20)
        torch.cuda.synchronize(dev)
        ut, uq = ctx.last_unique()
        assert uq <= ut < n // 2 and uq < hi - lo
        _assert_knn_equal((idx.cpu().numpy(), dst.cpu().numpy()), (wi[lo:hi], wd[lo:hi]))


def test_device_api_query_subset_like_a_rank(ctx, oracle):
    """fdr_normalize_dev + fdr_knn_dev on torch-owned buffers with the queries a SLICE of the targets
    (what one rank of the row-sharded pipeline does), incl. a ragged, non-32-aligned slice."""
    import torch
    from fedrann_amd.distributed import HipEngine
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(31)
    E = rng.standard_normal((9000, 128)).astype(np.float32)
    E[rng.random(E.shape) < 0.93] = 0  # sparse rows: ties and a few all-zero rows
    eng = HipEngine(ctx, dev)
    dE = torch.from_numpy(E).to(dev)
    Ehat = torch.zeros((9000, 128), dtype=torch.float32, device=dev)
    zero = torch.zeros((9000,), dtype=torch.uint8, device=dev)
    eng.normalize(dE, Ehat, zero)
    wi, wd = oracle.knn(E, 20)
    for lo, hi in ((0, 9000), (4500, 6750), (8967, 9000), (1, 130)):
        idx, dst = eng.knn(Ehat[lo:hi], zero[lo:hi], hi - lo, Ehat, zero, 9000, 128, 20)
        torch.cuda.synchronize(dev)
        _assert_knn_equal((idx.cpu().numpy(), dst.cpu().numpy()), (wi[lo:hi], wd[lo:hi]))


# ---- beyond the MFMA kernels' shapes: d > 512 or k > 64 on the generic kernel ------------------------------
@pytest.mark.parametrize("n,d,k", [(3000, 1000, 100), (2500, 600, 50), (4000, 128, 100), (700, 2048, 128),
                                   (1500, 513, 65), (130, 700, 128)])
def test_knn_generic_kernel_matches_oracle(ctx, oracle, n, d, k):
    """The reference accepts any -n / --nndescent-n-neighbors (__main__.py:128-146): sizes outside the MFMA
    kernels' shapes run on knn_generic_kernel with the same canonical arithmetic -- sparse rows with ties,
    duplicates and all-zero rows, indices and distance bits against the oracle."""
    rng = np.random.default_rng(n + d + k)
    E = np.zeros((n, d), np.float32)
    nnz = 6
    cols = rng.integers(0, d, size=(n, nnz))
    vals = (rng.integers(1, 5, size=(n, nnz)) * 0.37 * rng.choice([-1.0, 1.0], size=(n, nnz))).astype(np.float32)
    np.put_along_axis(E, cols, vals, axis=1)
    E[::17] = 0.0               # all-zero rows
    E[5::40] = E[3]             # exact duplicates (ties decided by the index)
    E[7::50] = 2.5 * E[4]       # scaled copies
    got = ctx.knn(E, k)
    _assert_knn_equal(got, oracle.knn(E, k))


def test_embed_knn_at_dimension_1000(ctx, oracle):
    """-n 1000: projection tables, embed (DP = 1024 accumulators), normalise and the generic k-NN, end to end
    against the oracle."""
    from fedrann_amd.precompute import build_precompute_matrix
    from fedrann_amd.synth import synth
    s = synth(1500, seed=31, m=120)
    P = build_precompute_matrix(s["counts"], 1000)
    ctx.projection_load(P.indptr, P.indices, P.data, s["n_features"], 1000)
    idx, dist, E = ctx.embed_knn(s["indptr"], s["indices"], 100, return_embedding=True)
    want_E = oracle.embed(s["indptr"], s["indices"], (P.indptr, P.indices, P.data), s["n_features"], 1000)
    assert np.array_equal(E.view(np.uint32), want_E.view(np.uint32))
    _assert_knn_equal((idx, dist), oracle.knn(want_E, 100))


# ---- error behaviour -----------------------------------------------------------------------------
def test_errors_are_raised_not_swallowed(ctx):
    E = np.zeros((10, 16), np.float32)
    with pytest.raises(_lib.FedrannHipError):
        ctx.knn(E, 11)  # k > n
    with pytest.raises(_lib.FedrannHipError):
        ctx.knn(E, 0)
    with pytest.raises(_lib.FedrannHipError):
        ctx.knn(np.zeros((100, 2049), np.float32), 5)  # d > FDR_MAX_DIM
    with pytest.raises(_lib.FedrannHipError):
        ctx.knn(np.zeros((200, 16), np.float32), 129)  # k > FDR_MAX_K
    fresh = _lib.Context(0)
    with pytest.raises(_lib.FedrannHipError):
        fresh.embed(np.array([0, 1], np.int64), np.array([0], np.int32))  # no projection loaded
    fresh.close()


# ---- the certificate's error bound on adversarial rows ---------------------------------------------
def _adversarial_rows(kind, n, d, rng):
    if kind == "fp16_midpoints":
        # components that sit (after normalisation, to within 1e-7) half way between two fp16 numbers:
        # (1 + (2 m + 1) 2^-11) 2^e -- every operand rounds by the full half ulp, in a direction numpy's and
        # the GPU's round-to-nearest-even must agree on
        nnz = 6
        E = np.zeros((n, d), np.float32)
        for i in range(n):
            c = rng.choice(d, size=nnz, replace=False)
            m = rng.integers(0, 512, size=nnz)
            e = rng.integers(-3, 0, size=nnz)
            E[i, c] = (1.0 + (2 * m + 1) * 2.0 ** -11) * 2.0 ** e * rng.choice([-1.0, 1.0], size=nnz)
        # rows drawn from a few component sets: many positive similarities, many near ties
        E[n // 2:] = E[rng.integers(0, n // 2, size=n - n // 2)] * rng.choice([1.0, 0.5, 2.0], size=(n - n // 2, 1))
        E[n // 2:, :] += (rng.random((n - n // 2, d)) < 0.01) * np.float32(0.25)
        return E.astype(np.float32)
    if kind == "fp16_subnormals":
        # one or two dominant components and dozens below fp16's smallest normal number (6.1e-5) after
        # normalisation: the fp16 copies lose them almost entirely
        E = (rng.random((n, d)) < 0.3) * rng.uniform(1e-6, 5e-5, size=(n, d)) * rng.choice([-1.0, 1.0], size=(n, d))
        for i in range(n):
            c = rng.choice(d, size=2, replace=False)
            E[i, c] = rng.choice([-1.0, 1.0], size=2) * rng.uniform(0.5, 1.0, size=2)
        return E.astype(np.float32)
    # dense rows: sum |x||y| = 1 for every pair, the fp32 accumulation term d 2^-24 is at its largest
    base = rng.standard_normal((8, d))
    E = base[rng.integers(0, 8, size=n)] + 0.3 * rng.standard_normal((n, d))
    return E.astype(np.float32)


@pytest.mark.parametrize("kind,d", [("fp16_midpoints", 128), ("fp16_subnormals", 128), ("dense", 512),
                                    ("fp16_midpoints", 512)])
def test_certificate_error_bound_on_adversarial_rows(ctx, oracle, knn_mode, kind, d):
    """FDR_PREFILTER_EPS = 0.00105 must bound |fp16 similarity - canonical similarity| on ALL pairs of a
    2000-row set built to stress it (knn_prefilter.inc's certificate rests on it), and the k-NN must still be
    the oracle's bits."""
    if not knn_mode.startswith("prefilter"):
        pytest.skip("prefilter mode only")
    rng = np.random.default_rng(d + len(kind))
    E = _adversarial_rows(kind, 2000, d, rng)
    Eh, _, zero = oracle.normalize(E)
    assert not zero.any()
    h = Eh.astype(np.float16).astype(np.float64)       # the operands the MFMA sees (round to nearest even)
    approx = h @ h.T                                    # (its fp32 accumulation adds <= d 2^-24 <= 3.1e-5)
    exact = Eh.astype(np.float64) @ Eh.astype(np.float64).T  # (the fp32 fma chain is within 1e-6 of this)
    err = np.abs(approx - exact).max()
    assert err + d * 2.0 ** -24 + 1e-6 < 0.00105, err
    if kind == "fp16_midpoints":
        assert err > 2e-4  # (the construction does push the error towards the bound)
    for k in (20, 50):
        _assert_knn_equal(ctx.knn(E, k), oracle.knn(E, k))


@pytest.mark.parametrize("k", [30, 31, 33, 34, 36])
def test_exact_mode_retry_pass_on_the_distance_one_plateau(ctx, oracle, knn_mode, k):
    """Rows whose similarities are mostly <= 0 (distance clamped to exactly 1.0): the k-th neighbour of many
    queries sits on the plateau, decided by the index alone, and with k in 30..36 (two-entry append queues)
    the queues overflow in the middle of a tile, so the exact kernel's retry pass must admit dist == tau == 1
    candidates whose similarity is far below the similarity-space gate."""
    rng = np.random.default_rng(100 + k)
    n, d = 3000, 96
    E = np.zeros((n, d), np.float32)
    for i in range(n):  # two or three components per row, signs mixed: most pairs have sim <= 0 or no overlap
        c = rng.choice(d, size=rng.integers(2, 4), replace=False)
        E[i, c] = rng.choice([-1.0, 1.0], size=c.size) * rng.uniform(0.5, 1.5, size=c.size)
    _assert_knn_equal(ctx.knn(E, k), oracle.knn(E, k))
