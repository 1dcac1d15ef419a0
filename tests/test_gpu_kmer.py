"""fdr_kmer_search (GPU) against the oracle's restatement of kmer_searcher.cpp: identical per-read index
sets, incl. the reference's treatment of invalid characters, short and empty reads."""
import numpy as np
import pytest

from fedrann_amd import _lib
from fedrann_amd import feature_extraction as fx
from fedrann_amd import kmer_search as ks
from fedrann_amd.synth import synth_sequences

pytestmark = pytest.mark.gpu


def _reads(seqs, off):
    return [bytes(seqs[off[i]:off[i + 1]]) for i in range(off.size - 1)]


def _assert_same(ctx, oracle, seqs, off, codes, k):
    ip, ix = ctx.kmer_search(seqs, off, codes, k)
    wp, wx = oracle.kmer_search(_reads(seqs, off), codes, k)
    assert np.array_equal(ip, wp), "row pointers differ"
    assert np.array_equal(ix, wx), "library indices differ"
    return ip, ix


def test_kmer_search_handcrafted_quirks(ctx, oracle):
    codes = oracle.kmer_library(b"ACG CGT TTT AAC AAA", 3)
    reads = [b"ACGT", b"", b"AC", b"ANTTTA", b"acgt", b"NA", b"", b"TTTTTTTTTT", b"N", b"ACGNACG"]
    off = np.zeros(len(reads) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(r) for r in reads])
    seqs = np.frombuffer(b"".join(reads), dtype=np.uint8)
    _assert_same(ctx, oracle, seqs, off, codes, 3)
    _assert_same(ctx, oracle, seqs, off, codes[:0], 3)          # empty library
    _assert_same(ctx, oracle, seqs[:0], np.zeros(4, np.int64), codes, 3)  # only empty reads


@pytest.mark.parametrize("k,n_reads,mean_len", [(15, 3000, 3000), (21, 1500, 5000), (31, 800, 8000), (5, 400, 300)])
def test_kmer_search_synthetic_reads(ctx, oracle, k, n_reads, mean_len):
    s = synth_sequences(n_reads, genome_len=300_000, mean_len=mean_len, k=k, sample=0.05, n_rate=1e-3,
                        seed=100 + k)
    codes = ks.load_kmer_library([b"\n".join(s["fwd"]) + b"\n", b"\n".join(s["rev"]) + b"\n"], k)
    assert np.array_equal(codes, oracle.kmer_library(b"\n".join(s["fwd"] + s["rev"]) + b"\n", k))
    seqs, off = s["seqs"].copy(), s["seq_off"]
    seqs[::997] |= 0x20  # some lower case
    ip, ix = _assert_same(ctx, oracle, seqs, off, codes, k)
    assert ix.size > n_reads  # the library really is hit


def test_kmer_search_short_reads_and_chunk_boundaries(ctx, oracle):
    """Thousands of reads shorter than, equal to and just above k, so that read boundaries fall everywhere
    inside the 4 KiB chunks and the 16-position thread stretches."""
    rng = np.random.default_rng(9)
    k = 15
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=50_000)]
    lens = rng.integers(0, 40, size=20_000)
    starts = rng.integers(0, genome.size - 40, size=lens.size)
    pieces = [genome[a:a + n] for a, n in zip(starts, lens)]
    off = np.zeros(lens.size + 1, dtype=np.int64)
    off[1:] = np.cumsum(lens)
    seqs = np.concatenate(pieces).copy()
    seqs[rng.random(seqs.size) < 0.003] = ord("N")
    win = np.lib.stride_tricks.sliding_window_view(genome, k)[::7]
    text = b"\n".join(w.tobytes() for w in win) + b"\nAAAAAAAAAAAAAAA\n"  # + the code an empty read looks up
    codes = ks.load_kmer_library(text, k)
    _assert_same(ctx, oracle, seqs, off, codes, k)


def test_kmer_search_hit_buffer_regrow(ctx, oracle):
    """A library holding most 9-mers: ~80 % of the 3 M windows hit, more than the first guess of the hit
    buffer (max(2^20, positions / 4)), so the pass is repeated with the exact size."""
    rng = np.random.default_rng(10)
    k = 9
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=3_000_000)]
    text = b"\n".join(w.tobytes() for w in np.lib.stride_tricks.sliding_window_view(genome[:400_000], k)) + b"\n"
    codes = ks.load_kmer_library(text, k)
    off = np.arange(0, genome.size + 1, 10_000, dtype=np.int64)
    ip, ix = _assert_same(ctx, oracle, genome, off, codes, k)
    assert ix.size > 300 * 1000  # (unique per read; the raw hits are ~2.4 M)


def test_kmer_searcher_drop_in_writes_reference_files(ctx, oracle, tmp_path):
    """kmer_searcher(): library files + FASTA in, output.bin + kmer_frequency.bin out; output.bin goes
    through the same loader as the reference tool's file would."""
    s = synth_sequences(500, genome_len=80_000, mean_len=2000, k=15, seed=77)
    fwd, rev, fa = tmp_path / "fwd.fasta", tmp_path / "rev.fasta", tmp_path / "reads.fasta"
    fwd.write_bytes(b"".join(b">%d\n%s\n" % (3 + i % 5, x) for i, x in enumerate(s["fwd"])))
    rev.write_bytes(b"".join(b">%d\n%s\n" % (3 + i % 5, x) for i, x in enumerate(s["rev"])))
    reads = _reads(s["seqs"], s["seq_off"])
    fa.write_bytes(b"".join(b">%s extra\n%s\n" % (i, b"\n".join(r[j:j + 70] for j in range(0, len(r), 70)))
                            for i, r in zip(s["ids"], reads)))
    ids, ip, ix, n_lib = ks.kmer_searcher([str(fwd), str(rev)], str(fa), str(tmp_path / "out"), 15, context=ctx)
    assert ids == s["ids"] and n_lib == 2 * len(s["fwd"])
    codes = oracle.kmer_library(b"\n".join(s["fwd"] + s["rev"]), 15)
    wp, wx = oracle.kmer_search(reads, codes, 15)
    assert np.array_equal(ip, wp) and np.array_equal(ix, wx)
    F = 2 * len(s["fwd"])
    ip2, ix2, names, strands = fx.build_feature_csr(str(tmp_path / "out" / "output.bin"), F)
    assert names[0::2] == [x.decode() for x in ids]
    assert np.array_equal(ix2[ip2[0]:ip2[1]], ix[ip[0]:ip[1]])
    freq = np.fromfile(str(tmp_path / "out" / "kmer_frequency.bin"), dtype="<u8").reshape(-1, 2)
    assert int(freq[:, 1].sum()) == ix.size


@pytest.mark.parametrize("k,min_count", [(15, 2), (5, 1), (31, 1), (21, 3)])
def test_kmer_count_matches_numpy_restatement(ctx, oracle, k, min_count):
    """fdr_kmer_count (jellyfish count -C | dump -L): canonical codes and counts, windows with an invalid
    character or crossing a read boundary skipped, reads shorter than k contribute nothing."""
    s = synth_sequences(1500, genome_len=60_000, mean_len=1500, k=k, sample=0.01, n_rate=2e-3, seed=300 + k)
    seqs, off = s["seqs"].copy(), s["seq_off"]
    seqs[::501] |= 0x20
    codes, counts = ctx.kmer_count(seqs, off, k, min_count)
    wc, wn = oracle.kmer_count(_reads(seqs, off), k, min_count)
    assert np.array_equal(codes, wc) and np.array_equal(counts, wn)
    assert codes.size > 0 and int(counts.min()) >= min_count
    # short and empty reads, nothing valid at all
    reads = [b"ACGT", b"", b"NNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNN", b"AC"]
    o = np.zeros(5, dtype=np.int64)
    o[1:] = np.cumsum([len(r) for r in reads])
    c2, n2 = ctx.kmer_count(np.frombuffer(b"".join(reads), dtype=np.uint8), o, k, 1)
    w2, m2 = oracle.kmer_count(reads, k, 1)
    assert np.array_equal(c2, w2) and np.array_equal(n2, m2)


def test_kmer_search_in_blocks(ctx, oracle):
    """kmer_search.search splits read sets beyond block_chars into blocks of whole reads: same CSR."""
    s = synth_sequences(1200, genome_len=50_000, mean_len=1200, k=15, sample=0.02, n_rate=1e-3, seed=41)
    codes = oracle.kmer_library(b"\n".join(s["fwd"] + s["rev"]), 15)
    one = ks.search(s["seqs"], s["seq_off"], codes, 15, context=ctx)
    many = ks.search(s["seqs"], s["seq_off"], codes, 15, context=ctx, block_chars=100_000)
    tiny = ks.search(s["seqs"], s["seq_off"], codes, 15, context=ctx, block_chars=1)  # one read per call
    for ip, ix in (many, tiny):
        assert np.array_equal(ip, one[0]) and np.array_equal(ix, one[1])
    wp, wx = oracle.kmer_search(_reads(s["seqs"], s["seq_off"]), codes, 15)
    assert np.array_equal(one[0], wp) and np.array_equal(one[1], wx)


@pytest.mark.parametrize("k,min_count,block", [(15, 2, 200_000), (9, 3, 50_000), (31, 1, 700_000)])
def test_kmer_count_in_blocks(ctx, oracle, k, min_count, block):
    """Read sets beyond one block's characters (2^31 by default; here a few hundred thousand) are counted block
    by block and the runs merged on the device: the same table as in one piece and as the oracle's, thresholds
    applied to the TOTALS (a k-mer seen once in each of two blocks passes min_count = 2)."""
    s = synth_sequences(1500, genome_len=60_000, mean_len=1500, k=k, sample=0.01, n_rate=2e-3, seed=900 + k)
    seqs, off = s["seqs"], s["seq_off"]
    try:
        ctx.set_kmer_count_block(0)
        one = ctx.kmer_count(seqs, off, k, min_count)
        assert ctx.last_kmer_count_blocks() == 1
        ctx.set_kmer_count_block(block)
        many = ctx.kmer_count(seqs, off, k, min_count)
        assert ctx.last_kmer_count_blocks() >= 3
        # a read longer than a block is refused, loudly
        ctx.set_kmer_count_block(int(np.diff(off).max()))
        with pytest.raises(_lib.FedrannHipError, match="characters"):
            ctx.kmer_count(seqs, off, k, min_count)
    finally:
        ctx.set_kmer_count_block(0)
    wc, wn = oracle.kmer_count(_reads(seqs, off), k, min_count)
    assert np.array_equal(one[0], wc) and np.array_equal(one[1], wn)
    assert np.array_equal(many[0], wc) and np.array_equal(many[1], wn)


def test_streamed_counting_and_search_equal_the_one_piece_calls(ctx, oracle, tmp_path):
    """The pipeline streams the reads (iter_sequence_blocks): counting in pieces (fdr_kmer_count_begin / _add /
    _finish) equals the one-piece call and the oracle; kmer_searcher fed in pieces writes the bytes the one-piece
    writer writes and returns the same CSR."""
    s = synth_sequences(600, genome_len=60_000, mean_len=1500, k=15, sample=0.05, seed=91)
    reads = _reads(s["seqs"], s["seq_off"])
    fa = tmp_path / "reads.fasta"
    fa.write_bytes(b"".join(b">%s\n%s\n" % (i, r) for i, r in zip(s["ids"], reads)))
    k = 15
    one = ctx.kmer_count(s["seqs"], s["seq_off"], k, 2)
    ctx.kmer_count_begin(k)
    pieces = 0
    for _, seqs, off in ks.iter_sequence_blocks(str(fa), chunk_bytes=50_000):
        ctx.kmer_count_add(seqs, off)
        pieces += 1
    many = ctx.kmer_count_finish(2)
    assert pieces > 5
    wc, wn = oracle.kmer_count(reads, k, 2)
    assert np.array_equal(one[0], wc) and np.array_equal(one[1], wn)
    assert np.array_equal(many[0], wc) and np.array_equal(many[1], wn)
    with pytest.raises(_lib.FedrannHipError):
        ctx.kmer_count_add(s["seqs"], s["seq_off"])  # (no begin)
    lib = tmp_path / "lib.txt"
    lib.write_bytes(b"\n".join(s["fwd"] + s["rev"]) + b"\n")
    ids, ip, ix, n_lib = ks.kmer_searcher(str(lib), str(fa), str(tmp_path / "a"), k, context=ctx)
    n_reads, none, nnz, n_lib2 = ks.kmer_searcher(str(lib), str(fa), str(tmp_path / "b"), k, context=ctx, collect=False,
                                                chunk_bytes=40_000)
    assert (n_reads, none, nnz, n_lib2) == (600, None, ix.size, n_lib) and ids == s["ids"]
    ks.write_output_bin(str(tmp_path / "whole.bin"), ids, ip, ix)
    ks.write_kmer_frequency_bin(str(tmp_path / "whole_freq.bin"), ix, n_lib)
    for d in ("a", "b"):
        assert (tmp_path / d / "output.bin").read_bytes() == (tmp_path / "whole.bin").read_bytes()
        assert (tmp_path / d / "kmer_frequency.bin").read_bytes() == (tmp_path / "whole_freq.bin").read_bytes()
    wp, wx = oracle.kmer_search(reads, oracle.kmer_library(lib.read_bytes(), k), k)
    assert np.array_equal(ip, wp) and np.array_equal(ix, wx)


def test_run_kmer_searcher_from_reads_only(ctx, oracle, tmp_path):
    """count_kmers.run_kmer_searcher: FASTA in; library files, output.bin out; same return tuple as the
    reference.  The library equals the thresholded canonical counts filtered by the documented sampler."""
    from fedrann_amd import count_kmers as ck
    from fedrann_amd import global_variables as gv
    from fedrann_amd.precompute import read_kmer_counts
    s = synth_sequences(400, genome_len=50_000, mean_len=2000, k=15, seed=55)
    reads = _reads(s["seqs"], s["seq_off"])
    fa = tmp_path / "reads.fasta"
    fa.write_bytes(b"".join(b">%s\n%s\n" % (i, r) for i, r in zip(s["ids"], reads)))
    gv.temp_dir, gv.seed = str(tmp_path / "temp"), 1234
    (tmp_path / "temp").mkdir()
    out_bin, n_features, read_count = ck.run_kmer_searcher(str(fa), 15, 0.05, 2, context=ctx)
    assert read_count == 400 and n_features % 2 == 0
    wc, wn = oracle.kmer_count(reads, 15, 2)
    keep = ck.sample_kmers(wc.size, 0.05, 1234)
    assert n_features == 2 * keep.size
    assert np.array_equal(read_kmer_counts(str(tmp_path / "temp" / "fwd_kmer_library.fasta")), wn[keep].astype(np.int64))
    fwd_text = b"\n".join(r.tobytes() for r in ck.codes_to_kmers(wc[keep], 15)) + b"\n"
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    rev_text = b"\n".join(r.tobytes().translate(comp)[::-1] for r in ck.codes_to_kmers(wc[keep], 15)) + b"\n"
    lib = oracle.kmer_library(fwd_text + rev_text, 15)
    wp, wx = oracle.kmer_search(reads, lib, 15)
    ip2, ix2, names, strands = fx.build_feature_csr(out_bin, n_features)
    for r in (0, 7, 399):
        assert np.array_equal(ix2[ip2[2 * r]:ip2[2 * r + 1]], wx[wp[r]:wp[r + 1]])
