"""Host side of the k-mer search (no GPU): library and read parsers against the oracle's line-by-line
restatement of kmer_searcher.cpp, output.bin writer round trip through the native loader."""
import struct

import numpy as np
import pytest

from fedrann_amd import feature_extraction as fx
from fedrann_amd import kmer_search as ks


def test_library_loader_matches_oracle(oracle):
    rng = np.random.default_rng(3)
    toks = []
    for _ in range(4000):
        n = int(rng.choice([15, 15, 15, 15, 14, 16, 3]))
        t = "".join(rng.choice(list("ACGTacgt"), size=n))
        if rng.random() < 0.02:
            t = t[:5] + "N" + t[6:]
        toks.append(t)
    toks += toks[:50] + [">15", ">123456789012345", "ACGTACGTACGTACG", "acgtacgtacgtacg"]  # repeats, a 15-char header
    text = ("\n".join(toks[:2000]) + "\t \r\n" + " ".join(toks[2000:]) + "\n").encode()
    for k in (15, 3, 16):
        got = ks.load_kmer_library(text, k)
        assert np.array_equal(got, oracle.kmer_library(text, k))
    assert ks.load_kmer_library([b"ACG\n", b"CGT\nACG"], 3).tolist() == [6, 27]
    assert ks.load_kmer_library(b"", 5).size == 0
    with pytest.raises(ValueError):
        ks.load_kmer_library(b"ACGT", 32)


def _check_reader(path, oracle):
    ids, seqs, off = ks.read_sequences(str(path))
    o_ids, o_seqs = oracle.read_sequences(str(path))
    assert ids == o_ids
    assert [bytes(seqs[off[i]:off[i + 1]]) for i in range(len(ids))] == o_seqs


def test_read_sequences_matches_oracle(tmp_path, oracle):
    rng = np.random.default_rng(4)
    lines = [b"stray line before any header"]
    for i in range(300):
        head = b">r%d" % i
        if i % 7 == 0:
            head += b" description here"
        if i % 11 == 0:
            head += b"\tx"
        if i == 13:
            head = b">"          # empty id: the record is dropped
        if i == 14:
            head = b"> onlydesc"  # empty id too
        lines.append(head)
        for _ in range(int(rng.integers(0, 4))):
            lines.append(bytes(rng.choice(list(b"ACGTNacgt"), size=int(rng.integers(0, 80))).astype(np.uint8)))
        if i % 17 == 0:
            lines.append(b"ACGT\r")
    for tail in (b"\n", b"", b"\n\n"):
        p = tmp_path / "a.fa"
        p.write_bytes(b"\n".join(lines) + tail)
        _check_reader(p, oracle)
    fq = []
    for i in range(100):
        s = bytes(rng.choice(list(b"ACGTN"), size=int(rng.integers(0, 50))).astype(np.uint8))
        fq += [b"@q%d some text" % i, s, b"+", b"@" * len(s)]  # quality lines that look like headers
    p = tmp_path / "a.fq"
    p.write_bytes(b"\n".join(fq) + b"\n")
    _check_reader(p, oracle)
    p.write_bytes(b"")
    assert ks.read_sequences(str(p))[0] == []
    p.write_bytes(b"\n>x\nAC\n")  # an empty first line: FASTA
    _check_reader(p, oracle)


def test_output_bin_writer_round_trip(tmp_path):
    rng = np.random.default_rng(5)
    F = 2000
    ids = [b"read/%d" % i for i in range(200)]
    rows = [np.sort(rng.choice(F, size=int(rng.integers(0, 30)), replace=False)) for _ in ids]
    indptr = np.zeros(len(ids) + 1, dtype=np.int64)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    indices = np.concatenate(rows).astype(np.int32)
    p = tmp_path / "output.bin"
    ks.write_output_bin(str(p), ids, indptr, indices)
    ip2, ix2, names, strands = fx.build_feature_csr(str(p), F)
    assert names[0::2] == [x.decode() for x in ids] and strands[:2] == [0, 1]
    for r, want in enumerate(rows):
        assert ix2[ip2[2 * r]:ip2[2 * r + 1]].tolist() == want.tolist()
    # the native appender (what the streaming kmer_searcher uses) writes the same bytes, in any grouping
    from fedrann_amd import _lib
    q = tmp_path / "appended.bin"
    q.write_bytes(p.read_bytes()[:16])
    for lo, hi in ((0, 0), (0, 70), (70, 71), (71, 200)):
        _lib.kmer_output_append(str(q), ids[lo:hi], indptr[lo:hi + 1] - indptr[lo], indices[indptr[lo]:indptr[hi]])
    assert q.read_bytes() == p.read_bytes()
    size = q.stat().st_size
    for bad in ([b"bad\xffid"], [b"x" * 70000], [b"tab\tid"]):
        with pytest.raises(ValueError):
            _lib.kmer_output_append(str(q), bad, [0, 0], [])
    assert q.stat().st_size == size  # (nothing is written when a record is refused)
    with pytest.raises(ValueError):
        ks.write_output_bin(str(p), [b"bad\xffid"], [0, 0], [])
    ks.write_kmer_frequency_bin(str(tmp_path / "f.bin"), indices, F)
    freq = np.fromfile(str(tmp_path / "f.bin"), dtype="<u8").reshape(-1, 2)
    want = np.bincount(indices, minlength=F)
    assert np.array_equal(freq[:, 0], np.flatnonzero(want)) and np.array_equal(freq[:, 1], want[want > 0])


def test_kmer_search_oracle_quirks(oracle):
    """The restated quirks themselves (kmer_searcher.cpp:306-352): window after an invalid character,
    reads shorter than k, the empty read, lower case."""
    codes = oracle.kmer_library(b"ACG CGT TTT AAC AAA", 3)
    assert codes.tolist() == [6, 27, 63, 1, 0]
    ip, ix = oracle.kmer_search([b"ACGT", b"", b"AC", b"ANTTTA", b"acgt", b"NA"], codes, 3)
    assert ip.tolist() == [0, 2, 3, 4, 5, 7, 7]
    assert ix.tolist() == [0, 1, 4, 3, 2, 0, 1]


def _mixed_fasta(rng, n):
    out = [b"junk before the first header\n"]
    for i in range(n):
        name = b"" if i % 37 == 5 else b"r%d desc\tx" % i
        seq = bytes(rng.choice(list(b"ACGTN"), size=int(rng.integers(0, 400))).astype(np.uint8))
        out.append(b">" + name + b"\n")
        for j in range(0, len(seq), 70):
            out.append(seq[j:j + 70] + (b"\r\n" if i % 50 == 3 else b"\n"))
            if i % 11 == 0:
                out.append(b"\n")
    return b"".join(out)


def _mixed_fastq(rng, n):
    out = []
    for i in range(n):
        name = b"q%d extra" % i if i % 9 else b"q%d" % i
        seq = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(1, 300))).astype(np.uint8))
        out.append(b"@" + name + b"\n" + seq + b"\n+\n" + b"@" * len(seq) + b"\n")  # quality lines that start with '@'
        if i % 13 == 0:
            out.append(b"\n")
    return b"".join(out)[:-1]  # no newline at the end of the file


@pytest.mark.parametrize("kind", ["fasta", "fastq"])
@pytest.mark.parametrize("ids_as_fasta", [False, True])
def test_streaming_reader_equals_whole_file_reader(tmp_path, kind, ids_as_fasta):
    """iter_sequence_blocks (what the pipeline uses: the read set never sits in host memory) yields the records
    of read_sequences, whatever the piece size -- pieces smaller than a record, records with empty names, CRs,
    empty lines, text before the first header, FASTQ quality lines starting with '@', no final newline."""
    rng = np.random.default_rng(1)
    p = tmp_path / ("reads." + kind)
    p.write_bytes((_mixed_fasta if kind == "fasta" else _mixed_fastq)(rng, 500))
    ids, seqs, off = ks.read_sequences(str(p), fastq_ids_as_fasta=ids_as_fasta)
    assert len(ids) > 400
    for chunk in (64, 1000, 4096, 1 << 20):
        got_ids, got_seqs, got_off = [], [], [0]
        for a, b, c in ks.iter_sequence_blocks(str(p), fastq_ids_as_fasta=ids_as_fasta, chunk_bytes=chunk):
            got_ids += a
            got_seqs.append(b)
            got_off += (c[1:] + got_off[-1]).tolist()
        assert got_ids == ids
        assert np.array_equal(np.concatenate(got_seqs), seqs) and np.array_equal(np.array(got_off), off)
    empty = tmp_path / "empty.fa"
    empty.write_bytes(b"")
    assert list(ks.iter_sequence_blocks(str(empty))) == []


@pytest.mark.parametrize("kind", ["fasta", "fastq"])
def test_native_reader_equals_numpy_statement_on_truncated_pieces(kind):
    """fdr_reads_scan / fdr_reads_parse against the numpy statement of the rules (_complete_prefix +
    _parse_records) on pieces cut at arbitrary bytes: inside headers, sequences, quality lines, with and without
    the end-of-file flag (a FASTQ record missing its last lines at the end of the file keeps what is there)."""
    from fedrann_amd import _lib
    rng = np.random.default_rng(3)
    raw = (_mixed_fasta if kind == "fasta" else _mixed_fastq)(rng, 120)
    is_fastq = kind == "fastq"
    cuts = sorted(set(rng.integers(0, len(raw), size=150).tolist() + [0, 1, 2, len(raw) - 1, len(raw)]))
    for n in cuts:
        piece = raw[:n]
        buf = np.frombuffer(piece, dtype=np.uint8).copy() if n else np.zeros(1, dtype=np.uint8)
        for eof in (False, True):
            for flag in (False, True):
                used, ids, seqs, off = _lib.reads_parse(buf, n, is_fastq, flag, eof)
                want_used = ks._complete_prefix(piece, is_fastq, eof)
                w_ids, w_seqs, w_off = ks._parse_records(piece[:want_used], is_fastq, flag)
                assert used == want_used, (n, eof, flag)
                assert ids == w_ids and np.array_equal(seqs, w_seqs) and np.array_equal(off, w_off), (n, eof, flag)


@pytest.mark.parametrize("k", [5, 15, 31])
def test_vectorised_library_files_equal_the_line_by_line_ones(k):
    """run_kmer_searcher's library files and in-memory library: '>count' / k-mer lines built with array operations
    equal the per-record text (counts of 1 to 17 digits), the reverse library equals what `seqkit seq -r -p` makes
    of the forward one, and the codes handed to the search equal load_kmer_library of `cat fwd rev | grep -v '^>'`
    (a palindromic k-mer keeps its forward slot only)."""
    from fedrann_amd import count_kmers as ck
    rng = np.random.default_rng(k)
    codes = np.unique(rng.integers(0, 1 << min(2 * k, 62), size=5000).astype(np.uint64))
    counts = rng.integers(1, 5000, size=codes.size).astype(np.uint64)
    counts[:5] = [1, 9, 10, 99999, 12345678901234567]
    kk = ck.codes_to_kmers(codes, k)
    want = b"".join(b">%d\n" % c + row.tobytes() + b"\n" for c, row in zip(counts.tolist(), kk))
    assert ck.kmer_library_text(codes, counts, k).tobytes() == want
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    rc = ck.revcomp_codes(codes, k)
    want_rev = b"".join(b">%d\n" % c + row.tobytes().translate(comp)[::-1] + b"\n" for c, row in zip(counts.tolist(), kk))
    assert ck.kmer_library_text(rc, counts, k).tobytes() == want_rev
    texts = [b"\n".join(l for l in t.split(b"\n") if not l.startswith(b">")) + b"\n" for t in (want, want_rev)]
    assert np.array_equal(ks.load_kmer_library(texts, k), ks.unique_first(np.concatenate((codes, rc))))
    assert ck.kmer_library_text(codes[:0], counts[:0], k).size == 0


def test_library_of_palindromes_keeps_forward_slots():
    from fedrann_amd import count_kmers as ck
    k = 4
    codes = np.array(sorted({0b00011011, 0b00000000, 0b11100100, 0b01101001}), dtype=np.uint64)  # ACGT is its own reverse complement
    rc = ck.revcomp_codes(codes, k)
    acgt = np.uint64(0b00011011)
    assert rc[codes == acgt][0] == acgt
    lib = ks.unique_first(np.concatenate((codes, rc)))
    assert (lib == acgt).sum() == 1 and lib.size == np.unique(np.concatenate((codes, rc))).size
    assert np.array_equal(lib[:codes.size], codes)  # (forward slots first, in their order)
