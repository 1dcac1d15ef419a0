"""Host-side logic of fedrann_amd (no GPU): projection builder, output.bin / npz readers, TSV
writer, synthetic generator -- each against the reference's golden vectors and/or the oracle."""
import hashlib
import io
import json
import struct

import numpy as np
import pytest

from conftest import golden, golden_embed_case
from fedrann_amd import feature_extraction as fx
from fedrann_amd import precompute as pc
from fedrann_amd.__main__ import build_parser, get_output_dataframe
from fedrann_amd.synth import synth


def _write_fasta(path, counts, crlf=False):
    nl = "\r\n" if crlf else "\n"
    with open(path, "w", newline="") as f:
        for c in counts:
            f.write(">%d%s%s%s" % (int(c), nl, "ACGTACGT", nl))


def _write_output_bin(path, names, rows):
    with open(path, "wb") as f:
        f.write(struct.pack("<4sB3sQ", b"KMER", 1, b"\0\0\0", len(names)))
        for n, r in zip(names, rows):
            nb = n if isinstance(n, bytes) else n.encode()
            f.write(struct.pack("<H", len(nb)) + nb + struct.pack("<I", len(r)))
            f.write(struct.pack("<%dQ" % len(r), *[int(i) for i in r]))


@pytest.mark.parametrize("tag", ["tiny", "mid"])
def test_get_precompute_matrix_matches_reference(tmp_path, tag):
    g = np.load(golden("precompute_%s.npz" % tag))
    fa = tmp_path / "lib.fasta"
    _write_fasta(fa, g["counts"])
    P, F = pc.get_precompute_matrix(n_components=int(g["d"]), counter_file=str(fa),
                                    n_features=2 * len(g["counts"]))
    assert F == 2 * len(g["counts"]) and P.shape == (F, int(g["d"])) and P.dtype == np.float32
    assert np.array_equal(P.indptr, g["indptr"])
    assert np.array_equal(P.indices, g["indices"])
    assert np.array_equal(P.data.view(np.uint32), g["data_bits"])


def test_precompute_big_digest_and_oracle(oracle):
    meta = json.load(open(golden("precompute_big.json")))
    counts = np.random.default_rng(11).integers(2, 61, size=meta["L"]).astype(np.int64)
    P = pc.build_precompute_matrix(counts, meta["d"])
    h = hashlib.sha256()
    for a in (P.indptr.astype(np.int64), P.indices.astype(np.int32), P.data.view(np.uint32)):
        h.update(np.ascontiguousarray(a).tobytes())
    assert h.hexdigest() == meta["sha256_indptr_i64_indices_i32_data_f32"]


def test_read_kmer_counts_variants(tmp_path):
    counts = [2, 5, 31, 1000000007, 7]
    a = tmp_path / "a.fasta"
    _write_fasta(a, counts)
    assert pc.read_kmer_counts(str(a)).tolist() == counts
    b = tmp_path / "b.fasta"
    _write_fasta(b, counts, crlf=True)
    assert pc.read_kmer_counts(str(b)).tolist() == counts
    c = tmp_path / "c.fasta"  # no trailing newline, header followed by two k-mers
    c.write_text(">3\nAAAA\nCCCC\n>9\nGGGG")
    assert pc.read_kmer_counts(str(c)).tolist() == [3, 3, 9]
    gen = list(pc.kmer_count_generator(str(c), 3))
    assert gen == [(0, 3), (3, 3), (1, 3), (4, 3), (2, 9), (5, 9)]
    bad = tmp_path / "bad.fasta"
    bad.write_text(">x1\nAAAA\n")
    with pytest.raises(ValueError):
        pc.read_kmer_counts(str(bad))
    (tmp_path / "e.fasta").write_text("")
    assert pc.read_kmer_counts(str(tmp_path / "e.fasta")).size == 0


def test_output_bin_reader_matches_oracle(tmp_path, oracle):
    rng = np.random.default_rng(2)
    L = 500
    names = ["r%d/x" % i for i in range(60)] + [b"bad\xff\xfename", "", "dup", "dup"]
    rows = [rng.choice(2 * L, size=int(rng.integers(0, 40)), replace=False) for _ in names]
    p = tmp_path / "output.bin"
    _write_output_bin(p, names, rows)
    o_names, o_strands, o_rows = oracle.parse_output_bin(str(p), L)
    n2, s2 = fx.get_metadata(str(p), 2 * L)
    assert n2 == o_names and s2 == o_strands
    gen = list(fx.parse_kmer_searcher_output(str(p), L))
    assert [g[0] for g in gen] == o_names and [g[2] for g in gen] == o_strands
    assert [list(g[1]) for g in gen] == o_rows
    indptr, indices, n3, s3 = fx.build_feature_csr(str(p), 2 * L)
    assert n3 == o_names and s3 == o_strands
    for r, want in enumerate(o_rows):
        assert indices[indptr[r]:indptr[r + 1]].tolist() == sorted(want)


def test_output_bin_errors(tmp_path):
    p = tmp_path / "x.bin"
    p.write_bytes(b"KME")
    with pytest.raises(ValueError):
        fx.read_kmer_searcher_output(str(p))
    p.write_bytes(struct.pack("<4sB3sQ", b"XXXX", 1, b"\0\0\0", 0))
    with pytest.raises(ValueError):
        fx.read_kmer_searcher_output(str(p))
    p.write_bytes(struct.pack("<4sB3sQ", b"KMER", 2, b"\0\0\0", 0))
    with pytest.raises(ValueError):
        fx.read_kmer_searcher_output(str(p))
    p.write_bytes(struct.pack("<4sB3sQ", b"KMER", 1, b"\0\0\0", 1) + struct.pack("<H", 2) + b"ab"
                  + struct.pack("<I", 3) + struct.pack("<2Q", 1, 2))
    with pytest.raises(ValueError):
        fx.read_kmer_searcher_output(str(p))
    p.write_bytes(struct.pack("<4sB3sQ", b"KMER", 1, b"\0\0\0", 0))
    names, indptr, idx = fx.read_kmer_searcher_output(str(p))
    assert names == [] and indptr.tolist() == [0] and idx.size == 0


def test_native_output_bin_loader_errors(tmp_path):
    """fdr_kmer_output_load: the reference's format errors (feature_extraction.py:111-119) as ValueError,
    plus truncation, out-of-range and repeated indices; a missing file is an I/O error of the library."""
    from fedrann_amd import _lib
    p = tmp_path / "x.bin"
    hdr = lambda magic, ver, n: struct.pack("<4sB3sQ", magic, ver, b"\0\0\0", n)
    rec = lambda name, idx: (struct.pack("<H", len(name)) + name + struct.pack("<I", len(idx))
                             + struct.pack("<%dQ" % len(idx), *idx))
    cases = [b"KME", hdr(b"XXXX", 1, 0), hdr(b"KMER", 2, 0),
             hdr(b"KMER", 1, 1) + struct.pack("<H", 2) + b"ab" + struct.pack("<I", 3) + struct.pack("<2Q", 1, 2),
             hdr(b"KMER", 1, 2) + rec(b"a", [1, 2]),          # second record missing
             hdr(b"KMER", 1, 1) + rec(b"a", [1, 10]),         # index == n_features
             hdr(b"KMER", 1, 1) + rec(b"a", [3, 4, 3])]       # repeated index
    for blob in cases:
        p.write_bytes(blob)
        with pytest.raises(ValueError):
            fx.build_feature_csr(str(p), 10)
    with pytest.raises(_lib.FedrannHipError):
        fx.build_feature_csr(str(tmp_path / "missing.bin"), 10)
    p.write_bytes(hdr(b"KMER", 1, 1) + rec(b"a", [1]))
    with pytest.raises(_lib.FedrannHipError):
        fx.build_feature_csr(str(p), 9)  # F must be even (F = 2L)
    p.write_bytes(hdr(b"KMER", 1, 0))
    indptr, indices, names, strands = fx.build_feature_csr(str(p), 10)
    assert indptr.tolist() == [0] and indices.size == 0 and names == [] and strands == []


@pytest.mark.parametrize("threads", [1, 0, 5])
def test_native_output_bin_loader_large_random(tmp_path, oracle, threads):
    """20 k records (enough for the thread pool to engage), ragged incl. empty records, long ids."""
    rng = np.random.default_rng(5)
    L = 40_000
    R = 20_000
    names = ["read_%d_%s" % (i, "x" * int(rng.integers(0, 30))) for i in range(R)]
    names[17] = b"\xc3\x28 not utf8"
    names[18] = "caf\u00e9"
    lens = rng.integers(0, 60, size=R)
    lens[::97] = 0
    rows = [rng.choice(2 * L, size=int(n), replace=False) for n in lens]
    p = tmp_path / "output.bin"
    _write_output_bin(p, names, rows)
    o_names, o_strands, o_rows = oracle.parse_output_bin(str(p), L)
    indptr, indices, n3, s3 = fx.build_feature_csr(str(p), 2 * L, n_threads=threads)
    assert n3 == o_names and s3 == o_strands
    assert indptr.dtype == np.int64 and indices.dtype == np.int32
    want_ptr, want_idx = oracle.rows_to_csr([sorted(r) for r in o_rows])
    assert np.array_equal(indptr, want_ptr) and np.array_equal(indices, want_idx)


def test_canonical_csr_checks():
    ip, ix = fx.canonical_csr([0, 3, 3, 5], [9, 2, 4, 7, 1], 10)
    assert ix.tolist() == [2, 4, 9, 1, 7] and ix.dtype == np.int32 and ip.dtype == np.int64
    with pytest.raises(ValueError):
        fx.canonical_csr([0, 2], [3, 10], 10)
    with pytest.raises(ValueError):
        fx.canonical_csr([0, 2], [3, 3], 10)
    with pytest.raises(ValueError):
        fx.canonical_csr([0, 3], [1, 2], 10)


def test_feature_matrix_npz_roundtrip_is_scipy_format(tmp_path):
    import scipy.sparse as sp
    s = synth(300, seed=5, m=40)
    p = tmp_path / "feature_matrix.npz"
    fx.save_feature_matrix_npz(str(p), s["indptr"], s["indices"], s["n_features"])
    A = sp.csr_matrix((np.ones(s["indices"].size, np.int8), s["indices"], s["indptr"].astype(np.int32)),
                      shape=(300, s["n_features"]))
    q = tmp_path / "ref.npz"
    sp.save_npz(str(q), A)
    assert p.read_bytes() == q.read_bytes()  # byte-identical to scipy.sparse.save_npz
    ip, ix, F = fx.load_feature_matrix_npz(str(p))
    assert F == s["n_features"] and np.array_equal(ip, s["indptr"]) and np.array_equal(ix, s["indices"])


@pytest.mark.parametrize("tag", ["overlaps_edge", "overlaps_rand"])
def test_overlaps_writer_is_byte_identical(tag):
    g = np.load(golden(tag + ".npz"))
    df = get_output_dataframe(g["indices"], g["dist_bits"].view(np.float32), list(g["names"]),
                              [int(s) for s in g["strands"]])
    buf = io.StringIO()
    df.to_csv(buf, sep="\t", index=False)
    assert buf.getvalue() == open(golden(tag + ".tsv"), newline="").read()
    assert df["distance"].dtype == np.float32 and df["neighbor_rank"].dtype == np.int64


@pytest.mark.parametrize("tag", ["overlaps_edge", "overlaps_rand"])
def test_native_overlaps_writer_is_byte_identical_to_reference_output(tmp_path, tag):
    """fdr_overlaps_write (no DataFrame, no pandas) against the TSVs the reference's own get_output_dataframe
    + to_csv produced (tests/golden/make_golden.py): -1 aliasing, inf, 1.1920929e-07, 1.0, rank gaps."""
    from fedrann_amd.__main__ import write_overlaps
    g = np.load(golden(tag + ".npz"))
    p = tmp_path / "o.tsv"
    lines = write_overlaps(str(p), g["indices"], g["dist_bits"].view(np.float32), list(g["names"]),
                           [int(s) for s in g["strands"]])
    want = open(golden(tag + ".tsv"), "rb").read()
    assert p.read_bytes() == want and lines == want.count(b"\n") - 1


@pytest.mark.parametrize("threads", [1, 4])
def test_native_overlaps_writer_equals_pandas_on_random_graphs(tmp_path, threads):
    """Large enough for the thread pool; float32 distances over the whole range incl. values below 1e-4
    (scientific layout), exact 0 / 1, inf; rows written whole and as two appended blocks."""
    from fedrann_amd import _lib, global_variables
    from fedrann_amd.__main__ import write_overlaps
    rng = np.random.default_rng(4)
    n, k = 9000, 12
    idx = rng.integers(0, n, size=(n, k)).astype(np.int32)
    idx[::3, 0] = np.arange(n)[::3]
    idx[rng.random((n, k)) < 0.01] = -1
    dist = rng.integers(0, 0x3f800001, size=(n, k), dtype=np.uint32).view(np.float32).copy()
    dist[rng.random((n, k)) < 0.02] = np.inf
    dist[0, :4] = [0.0, 1.0, 1e-4, 9.9999e-05]
    names = ["r%d/%s" % (i // 2, "ab"[i % 2] * (i % 5)) for i in range(n)]
    strands = [i % 2 for i in range(n)]
    df = get_output_dataframe(idx, dist, names, strands)
    buf = io.StringIO()
    df.to_csv(buf, sep="\t", index=False)
    want = buf.getvalue().encode()
    global_variables.threads = threads
    try:
        p = tmp_path / "o.tsv"
        assert write_overlaps(str(p), idx, dist, names, strands) == df.shape[0]
        assert p.read_bytes() == want
        q = tmp_path / "parts.tsv"
        off, buf8 = _lib.pack_names(names)
        a = _lib.overlaps_write(str(q), idx[:4000], dist[:4000], off, buf8, np.array(strands, np.uint8), row0=0)
        b = _lib.overlaps_write(str(q), idx[4000:], dist[4000:], off, buf8, np.array(strands, np.uint8), row0=4000,
                                append=True, header=False)
        assert q.read_bytes() == want and a + b == df.shape[0]
    finally:
        global_variables.threads = 1
    with pytest.raises(_lib.FedrannHipError):
        bad = idx.copy()
        bad[5, 5] = n
        write_overlaps(str(tmp_path / "bad.tsv"), bad, dist, names, strands)


def test_fastq_ids_on_pipeline_paths(tmp_path):
    from fedrann_amd.kmer_search import read_sequences
    fq = tmp_path / "r.fastq"
    fq.write_bytes(b"@id1 runid=9 ch=3\nACGT\n+\nIIII\n@id2\tRG:Z:a\nGG\n+\nII\n@ only_description\nTT\n+\nII\n@id4\nA\n+\nI\n")
    ids, seqs, off = read_sequences(str(fq))  # the stand-alone tool keeps the whole header (kmer_searcher.cpp:186)
    assert ids == [b"id1 runid=9 ch=3", b"id2\tRG:Z:a", b" only_description", b"id4"]
    ids, seqs, off = read_sequences(str(fq), fastq_ids_as_fasta=True)  # pipeline: after seqkit fq2fa
    assert ids == [b"id1", b"id2", b"id4"] and bytes(seqs) == b"ACGTGGA" and off.tolist() == [0, 4, 6, 7]


def test_overlaps_writer_equals_oracle_loop(oracle):
    rng = np.random.default_rng(8)
    n, k = 57, 9
    idx = rng.integers(0, n, size=(n, k)).astype(np.int32)
    idx[::3, 0] = np.arange(n)[::3]
    dist = np.sort(rng.random((n, k)).astype(np.float32), axis=1)
    names = ["n%d" % (i // 2) for i in range(n)]
    strands = [i % 2 for i in range(n)]
    df = get_output_dataframe(idx, dist, names, strands)
    buf = io.StringIO()
    df.to_csv(buf, sep="\t", index=False)
    assert buf.getvalue() == oracle.overlaps_tsv(idx, dist, names, strands)


def test_synth_properties():
    s = synth(2000, seed=602, m=60)
    ip, ix, F = s["indptr"], s["indices"], s["n_features"]
    assert ip[0] == 0 and ip[-1] == ix.size and len(ip) == 2001
    assert np.all(np.diff(ip) >= 1)  # no empty row
    rows = np.repeat(np.arange(2000), np.diff(ip))
    same = rows[1:] == rows[:-1]
    assert np.all(ix[1:][same] > ix[:-1][same])  # strictly ascending inside a row
    assert ix.min() >= 0 and ix.max() < F and s["counts"].size == F // 2
    s2 = synth(2000, seed=602, m=60)
    assert np.array_equal(s2["indices"], ix)
    d = synth(50, seed=1, m=30, doubling=True)
    L = d["n_features"] // 2
    for i in range(50):
        a = d["indices"][d["indptr"][2 * i]:d["indptr"][2 * i + 1]].astype(np.int64)
        b = d["indices"][d["indptr"][2 * i + 1]:d["indptr"][2 * i + 2]].astype(np.int64)
        assert sorted(np.where(a < L, a + L, a - L).tolist()) == b.tolist()


def test_cli_flags_match_reference_defaults():
    a = build_parser().parse_args(["-o", "out", "--feature-matrix", "x.npz", "--kmer-counts", "c.npy"])
    assert (a.kmer_size, a.kmer_sample_fraction, a.kmer_min_multiplicity) == (16, 0.005, 2)
    assert (a.threads, a.chunk_size, a.embedding_dimension) == (1, 1000, 500)
    assert (a.nndescent_n_trees, a.nndescent_n_neighbors, a.seed) == (300, 50, 356115)
    assert not a.save_feature_matrix and not a.keep_intermediates and not a.mprof


def test_native_overlaps_writer_quotes_names_and_blanks_nan_like_pandas(tmp_path):
    """csv.QUOTE_MINIMAL as DataFrame.to_csv applies it: a name with the separator, a double quote or a line feed
    is quoted (quotes doubled), a carriage return alone is not; a NaN distance is an empty field."""
    from fedrann_amd.__main__ import write_overlaps
    names = ['a"b', "c\rd", "e\nf", "g\th", " lead", "x", "q,r", "s't", '""', "plain"]
    n, k = len(names), 4
    rng = np.random.default_rng(9)
    idx = rng.integers(0, n, size=(n, k)).astype(np.int32)
    dist = rng.random((n, k)).astype(np.float32)
    dist[1, 2] = np.nan
    dist[4, 0] = np.nan
    strands = [i % 2 for i in range(n)]
    df = get_output_dataframe(idx, dist, names, strands)
    buf = io.StringIO()
    df.to_csv(buf, sep="\t", index=False)
    p = tmp_path / "o.tsv"
    assert write_overlaps(str(p), idx, dist, names, strands) == df.shape[0]
    assert p.read_bytes() == buf.getvalue().encode()


@pytest.mark.parametrize("threads", [1, 3])
def test_native_output_bin_loader_record_range(tmp_path, oracle, threads):
    """fdr_kmer_output_load_range: one rank's block of rows (records [lo, hi)) equals the slice of the whole
    matrix, the names are those of all records, empty and out-of-range blocks are handled."""
    from fedrann_amd import _lib
    from fedrann_amd.__main__ import record_names
    rng = np.random.default_rng(11)
    L, R = 5000, 700
    path = tmp_path / "output.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<4sB3sQ", b"KMER", 1, b"\0\0\0", R))
        for r in range(R):
            name = b"read_%d" % r if r % 50 else b""
            cnt = 0 if r % 97 == 0 else int(rng.integers(1, 300))
            ids = rng.choice(2 * L, size=cnt, replace=False).astype("<u8")
            f.write(struct.pack("<H", len(name)) + name + struct.pack("<I", cnt) + ids.tobytes())
    ip, ix, noff, nbuf = _lib.kmer_output_load(str(path), 2 * L, threads)
    for lo, hi in ((0, R), (0, 1), (100, 356), (699, 700), (350, 350), (650, 9999)):
        total, bip, bix, boff, bbuf = _lib.kmer_output_load_range(str(path), 2 * L, lo, hi, n_threads=threads)
        hi = min(hi, R)
        assert total == R
        assert np.array_equal(bip, ip[2 * lo:2 * hi + 1] - ip[2 * lo])
        assert np.array_equal(bix, ix[ip[2 * lo]:ip[2 * hi]])
        assert np.array_equal(boff, noff) and np.array_equal(bbuf, nbuf)
    total, bip, bix, boff, bbuf = _lib.kmer_output_load_range(str(path), 2 * L, 10, 20, with_names=False)
    assert boff is None and bbuf is None and bip.size == 21
    assert _lib.kmer_output_records(str(path)) == R
    # the writer's view of the names: the records' own (valid UTF-8 ids pass through, nothing is doubled)
    off1, buf1 = record_names(noff, nbuf)
    assert off1 is noff and buf1 is nbuf


def test_record_names_follow_the_reference_rule_for_invalid_utf8_and_doubled_writer_mode(tmp_path):
    """Ids that are not valid UTF-8 are rewritten as the reference does (feature_extraction.py:125-128); the writer's
    doubled-rows mode (strands=None: row t = record t >> 1 on strand t & 1) gives the bytes of the per-row form."""
    from fedrann_amd import _lib
    from fedrann_amd.__main__ import record_names
    ids = [b"ok", b"caf\xc3\xa9", b"bad\xff\xfeid", b""]
    off = np.zeros(len(ids) + 1, dtype=np.int64)
    np.cumsum([len(b) for b in ids], out=off[1:])
    off1, buf1 = record_names(off, np.frombuffer(b"".join(ids), dtype=np.uint8))
    got = [bytes(buf1[off1[i]:off1[i + 1]]) for i in range(len(ids))]
    assert got == [b"ok", "café".encode(), b"bad__id", b""]
    # doubled-rows mode against the same rows spelled out
    rng = np.random.default_rng(3)
    R, k = len(ids), 5
    idx = rng.integers(-2, 2 * R, size=(2 * R, k)).astype(np.int32)
    dist = rng.random((2 * R, k)).astype(np.float32)
    off2, buf2 = _lib.pack_names([n.decode() for n in got for _ in (0, 1)])
    strands = np.tile(np.array([0, 1], dtype=np.uint8), R)
    a, b = tmp_path / "a.tsv", tmp_path / "b.tsv"
    na = _lib.overlaps_write(str(a), idx, dist, off2, buf2, strands)
    nb = _lib.overlaps_write(str(b), idx, dist, off1, buf1, None)
    assert na == nb and a.read_bytes() == b.read_bytes()
    # a block of rows (a rank's share) in doubled mode
    nc = _lib.overlaps_write(str(tmp_path / "c.tsv"), idx[2:6], dist[2:6], off1, buf1, None, row0=2, header=False)
    want = _lib.overlaps_write(str(tmp_path / "d.tsv"), idx[2:6], dist[2:6], off2, buf2, strands, row0=2, header=False)
    assert nc == want and (tmp_path / "c.tsv").read_bytes() == (tmp_path / "d.tsv").read_bytes()
