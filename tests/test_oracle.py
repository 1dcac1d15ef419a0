"""The CPU oracle against the golden vectors captured from the reference (CPU only)."""
import hashlib
import json

import numpy as np
import pytest

from conftest import golden, golden_embed_case


@pytest.mark.parametrize("tag", ["tiny", "mid"])
def test_precompute_matches_reference(oracle, tag):
    g = np.load(golden("precompute_%s.npz" % tag))
    ip, ix, v = oracle.precompute_matrix(g["counts"], int(g["d"]))
    assert np.array_equal(ip, g["indptr"])
    assert np.array_equal(ix, g["indices"])
    assert np.array_equal(v.view(np.uint32), g["data_bits"])


def test_precompute_survey_kat(oracle):
    # known-answer test recorded in SURVEY.md section 8c (numpy 1.26.4, F=12, d=4)
    ip, ix, v = oracle.precompute_matrix(np.array([2, 5, 3, 7, 2, 11]), 4)
    assert ip.tolist() == [0, 0, 3, 4, 6, 8, 10, 12, 12, 13, 14, 16, 17]
    assert ix.tolist() == [1, 2, 3, 2, 1, 3, 2, 3, 1, 3, 0, 3, 2, 3, 2, 3, 2]
    want = np.array([-0.81471545, 0.81471545, -0.81471545, -1.2900923, 0.5015928, -0.5015928,
                     -1.6674201, -1.6674201, -0.08097321, 0.08097321, -1.6674201, 1.6674201,
                     -1.2900923, 0.5015928, 1.6674201, 1.6674201, -0.08097321], dtype=np.float32)
    assert np.array_equal(v, want)


def test_precompute_big_digest(oracle):
    meta = json.load(open(golden("precompute_big.json")))
    counts = np.random.default_rng(11).integers(2, 61, size=meta["L"]).astype(np.int64)
    ip, ix, v = oracle.precompute_matrix(counts, meta["d"])
    h = hashlib.sha256()
    for a in (ip, ix, v.view(np.uint32)):
        h.update(np.ascontiguousarray(a).tobytes())
    assert ix.size == meta["nnz"]
    assert h.hexdigest() == meta["sha256_indptr_i64_indices_i32_data_f32"]


@pytest.mark.parametrize("tag", ["tiny", "mid"])
def test_embed_matches_reference(oracle, tag):
    indptr, indices, P, F, d, E_bits = golden_embed_case(tag)
    E = oracle.embed(indptr, indices, P, F, d)
    assert np.array_equal(E.view(np.uint32), E_bits)


def test_embed_survey_kat(oracle):
    indptr, indices, P, F, d, _ = golden_embed_case("tiny")
    E = oracle.embed(indptr, indices, P, F, d)
    want = np.array([[0, 1054301044, 0, 3201784692], [3218435590, 0, 3181761864, 1074450716],
                     [0, 3209728305, 1062244657, 3209728305], [0, 0, 0, 0], [0, 0, 3215270335, 0],
                     [0, 3209728305, 3203622042, 3209728305], [0, 3209728305, 3205393654, 3209728305],
                     [3218435590, 3181761864, 3215270335, 1071631194], [0, 0, 0, 1056991332],
                     [0, 1056991331, 0, 3204474980]], dtype=np.uint32)
    assert np.array_equal(E.view(np.uint32), want)


def test_embed_empty_and_unsorted_rows(oracle):
    _, _, P, F, d, _ = golden_embed_case("mid")
    rng = np.random.default_rng(3)
    rows = [rng.choice(F, size=n, replace=False) for n in (0, 5, 0, 300, 1)]
    indptr, indices = oracle.rows_to_csr(rows)
    E1 = oracle.embed(indptr, indices, P, F, d)
    indptr2, indices2 = oracle.rows_to_csr([np.sort(r) for r in rows])
    E2 = oracle.embed(indptr2, indices2, P, F, d)
    assert np.array_equal(E1.view(np.uint32), E2.view(np.uint32))
    assert not E1[0].any() and not E1[2].any()


def test_parse_output_bin_and_metadata(oracle, tmp_path):
    import struct
    meta = json.load(open(golden("metadata_tiny.json")))
    reads = [[0, 3, 5], [1], [7, 2], [11, 0, 1, 2], [4, 10, 9]]
    p = tmp_path / "output.bin"
    with open(p, "wb") as f:
        f.write(struct.pack("<4sB3sQ", b"KMER", 1, b"\0\0\0", len(reads)))
        for n, r in zip(meta["names"], reads):
            f.write(struct.pack("<H", len(n)) + n.encode() + struct.pack("<I", len(r)))
            f.write(struct.pack("<%dQ" % len(r), *r))
    names, strands, rows = oracle.parse_output_bin(str(p), 6)
    assert names == meta["read_names"] and strands == meta["strands"]
    assert rows[0] == [0, 3, 5] and rows[1] == [6, 9, 11] and rows[7] == [5, 6, 7, 8]


@pytest.mark.parametrize("tag", ["overlaps_edge", "overlaps_rand"])
def test_overlaps_tsv_matches_reference(oracle, tag):
    g = np.load(golden(tag + ".npz"))
    txt = oracle.overlaps_tsv(g["indices"], g["dist_bits"].view(np.float32), list(g["names"]),
                              [int(s) for s in g["strands"]])
    assert txt == open(golden(tag + ".tsv"), newline="").read()


def test_knn_scalar_chain_equals_blocked(oracle):
    rng = np.random.default_rng(5)
    E = rng.standard_normal((700, 96)).astype(np.float32)
    Eh, _, zero = oracle.normalize(E)
    idx, dist = oracle.knn(E, 7)
    for i in rng.integers(0, 700, size=200):
        for r in range(7):
            dd = np.float32(oracle.pair_dist(Eh[i], Eh[idx[i, r]]))
            assert dd.view(np.uint32) == dist[i, r].view(np.uint32)
    # ascending (dist, idx)
    key = dist.view(np.uint32).astype(np.uint64) << np.uint64(32) | idx.astype(np.uint64)
    assert np.all(key[:, 1:] > key[:, :-1])


def test_knn_agrees_with_sklearn_brute(oracle):
    from sklearn.neighbors import NearestNeighbors
    rng = np.random.default_rng(6)
    E = rng.standard_normal((1500, 128)).astype(np.float32)  # tie-free
    idx, dist = oracle.knn(E, 20)
    D, I = NearestNeighbors(n_neighbors=20, algorithm="brute", metric="cosine").fit(E).kneighbors(E)
    assert np.array_equal(I, idx)
    assert np.abs(D - dist).max() < 1e-6


def test_knn_ties_and_zero_rows(oracle):
    rng = np.random.default_rng(7)
    base = rng.standard_normal((40, 32)).astype(np.float32)
    E = np.concatenate([base, base, np.zeros((6, 32), np.float32), base[:10]])  # duplicates + zero rows
    idx, dist = oracle.knn(E, 12)
    n = E.shape[0]
    z = np.arange(80, 86)
    # a zero row: the zero rows first (distance 0, ascending index), then distance 1 by index
    for q in z:
        assert idx[q, :6].tolist() == z.tolist() and np.all(dist[q, :6] == 0)
        assert idx[q, 6:].tolist() == list(range(6)) and np.all(dist[q, 6:] == 1)
    # duplicates of row 3 are rows 3, 43, 83+3=89: identical distances, ascending index
    assert idx[3, :3].tolist() == [3, 43, 89]
    assert dist[3, 0] == dist[3, 1] == dist[3, 2]
    assert idx.min() >= 0 and idx.max() < n


def test_nndescent_restatement_approximates_the_exact_search(oracle):
    """oracle/nndescent.c (the reference's k-NN ALGORITHM restated: RP forest + NN-descent; bench.py's second CPU
    baseline): rows ascending by (distance, index), self at distance 0, and a tie-aware recall close to 1 against
    the exact oracle -- with the reference's forest (300 trees, leaves of 200) and with a small one, where the
    descent rounds have to do the work."""
    from fedrann_amd.precompute import build_precompute_matrix
    from fedrann_amd.synth import synth
    s = synth(6000, seed=11, m=120)
    P = build_precompute_matrix(s["counts"], 128)
    E = oracle.embed(s["indptr"], s["indices"], (P.indptr, P.indices, P.data), s["n_features"], 128)
    Eh, _, zero = oracle.normalize(E)
    k = 20
    _, wd = oracle.knn_normalized(Eh, zero, Eh, zero, k)
    for trees, leaf, floor in ((300, 200, 0.999), (6, 40, 0.9)):
        idx, dist, stats = oracle.nndescent(Eh, zero, k, n_trees=trees, leaf_size=leaf, seed=5)
        assert idx.shape == (6000, k) and idx.min() >= 0 and idx.max() < 6000
        assert np.all(np.diff(dist, axis=1) >= 0) and stats["rounds"] >= 1
        nz = zero == 0
        assert np.all((idx[nz] == np.arange(6000)[nz, None]).any(1) | (dist[nz, -1] == 0))  # self, unless k exact duplicates precede it
        recall = float((dist <= wd[:, k - 1:k]).mean())
        assert recall >= floor, (trees, leaf, recall)
