"""End-to-end drop-in check on the GPU: the fedrann command line (python -m fedrann_amd) from the
reference's intermediates / from feature_matrix.npz to overlaps.tsv, byte-compared with the
reference's writer applied to the oracle's results (config 1 of BASELINE.json: plumbing)."""
import os
import struct

import numpy as np
import pytest

from fedrann_amd import __main__ as cli
from fedrann_amd import feature_extraction as fx
from fedrann_amd.synth import synth

pytestmark = pytest.mark.gpu


def _write_intermediates(tmp_path, s, names):
    """The two files the reference leaves in out/temp with --keep-intermediates."""
    L = s["n_features"] // 2
    fasta = tmp_path / "fwd_kmer_library.fasta"
    with open(fasta, "w") as f:
        for c in s["counts"]:
            f.write(">%d\nACGTACGTACGTACG\n" % int(c))
    out_bin = tmp_path / "output.bin"
    rng = np.random.default_rng(0)
    with open(out_bin, "wb") as f:
        f.write(struct.pack("<4sB3sQ", b"KMER", 1, b"\0\0\0", len(names)))
        for i, n in enumerate(names):
            idx = s["indices"][s["indptr"][i]:s["indptr"][i + 1]].astype(np.int64)
            idx = idx[rng.permutation(idx.size)]  # kmer_searcher emits a hash-set order
            nb = n.encode()
            f.write(struct.pack("<H", len(nb)) + nb + struct.pack("<I", idx.size))
            f.write(struct.pack("<%dQ" % idx.size, *idx.tolist()))
    return str(out_bin), str(fasta), L


@pytest.mark.parametrize("d,k", [(128, 20), (500, 50), (1000, 100)])  # (the last: beyond the MFMA kernels' shapes)
def test_cli_from_kmer_searcher_output(tmp_path, oracle, d, k):
    s = synth(900, seed=5, m=80)  # one row per record; the CLI doubles them (fwd + mirrored strand)
    names = ["read_%d/ccs" % i for i in range(900)]
    out_bin, fasta, L = _write_intermediates(tmp_path, s, names)
    out_dir = tmp_path / "out"
    cli.main(["-o", str(out_dir), "-n", str(d), "--nndescent-n-neighbors", str(k),
              "--kmer-searcher-output", out_bin, "--kmer-library", fasta, "--save-feature-matrix"])
    got = open(out_dir / "overlaps.tsv", newline="").read()
    # oracle pipeline: reference-style parse -> P -> E -> exact k-NN -> the reference's writer loop
    o_names, o_strands, o_rows = oracle.parse_output_bin(out_bin, L)
    P = oracle.precompute_matrix(s["counts"], d)
    indptr, indices = oracle.rows_to_csr(o_rows)
    E = oracle.embed(indptr, indices, P, 2 * L, d)
    idx, dist = oracle.knn(E, k)
    want = oracle.overlaps_tsv(idx, dist, o_names, o_strands)
    assert got == want
    assert os.path.exists(out_dir / "fedrann.log")
    # the saved feature matrix is the doubled binary CSR in scipy's npz layout
    ip, ix, F = fx.load_feature_matrix_npz(str(out_dir / "feature_matrix.npz"))
    assert F == 2 * L and ip.size - 1 == 1800
    for r in (0, 1, 777, 1799):
        assert ix[ip[r]:ip[r + 1]].tolist() == sorted(o_rows[r])


def test_cli_from_feature_matrix_npz(tmp_path, oracle):
    s = synth(1500, seed=9, m=60, doubling=True)
    npz = tmp_path / "feature_matrix.npz"
    fx.save_feature_matrix_npz(str(npz), s["indptr"], s["indices"], s["n_features"])
    counts = tmp_path / "counts.npy"
    np.save(counts, s["counts"])
    names = tmp_path / "names.txt"
    with open(names, "w") as f:
        for n, st in zip(s["names"], s["strands"]):
            f.write("%s\t%d\n" % (n, st))
    out_dir = tmp_path / "out"
    cli.main(["-o", str(out_dir), "-n", "128", "--nndescent-n-neighbors", "20", "--feature-matrix", str(npz),
              "--kmer-counts", str(counts), "--read-names", str(names)])
    got = open(out_dir / "overlaps.tsv", newline="").read()
    P = oracle.precompute_matrix(s["counts"], 128)
    E = oracle.embed(s["indptr"], s["indices"], P, s["n_features"], 128)
    idx, dist = oracle.knn(E, 20)
    assert got == oracle.overlaps_tsv(idx, dist, s["names"], s["strands"])


def test_cli_from_reads_and_kmer_library(tmp_path, oracle):
    """-i reads.fasta + --kmer-library: k-mer search, loader, embed, k-NN and writer all on this build;
    the oracle side restates kmer_searcher.cpp, the reference's parser and its writer."""
    from fedrann_amd.synth import synth_sequences
    k = 15
    s = synth_sequences(700, genome_len=120_000, mean_len=2500, k=k, sample=0.04, seed=31)
    rng = np.random.default_rng(2)
    counts = rng.integers(2, 40, size=len(s["fwd"]))
    lib = tmp_path / "fwd_kmer_library.fasta"
    lib.write_bytes(b"".join(b">%d\n%s\n" % (int(c), x) for c, x in zip(counts, s["fwd"])))
    reads = [bytes(s["seqs"][s["seq_off"][i]:s["seq_off"][i + 1]]) for i in range(700)]
    fa = tmp_path / "reads.fasta"
    fa.write_bytes(b"".join(b">%s len=%d\n%s\n" % (i, len(r), r) for i, r in zip(s["ids"], reads)))
    out_dir = tmp_path / "out"
    cli.main(["-i", str(fa), "-k", str(k), "-o", str(out_dir), "-n", "128", "--nndescent-n-neighbors", "20",
              "--kmer-library", str(lib), "--keep-intermediates"])
    got = open(out_dir / "overlaps.tsv", newline="").read()
    L = len(s["fwd"])
    codes = oracle.kmer_library(b"\n".join(s["fwd"] + s["rev"]), k)
    ip, ix = oracle.kmer_search(reads, codes, k)
    rows = []
    for r in range(700):
        idx = ix[ip[r]:ip[r + 1]].astype(np.int64)
        rows.append(idx.tolist())
        rows.append(np.where(idx < L, idx + L, idx - L).tolist())
    P = oracle.precompute_matrix(counts, 128)
    indptr, indices = oracle.rows_to_csr(rows)
    E = oracle.embed(indptr, indices, P, 2 * L, 128)
    idx, dist = oracle.knn(E, 20)
    names = [i.decode() for i in s["ids"] for _ in (0, 1)]
    assert got == oracle.overlaps_tsv(idx, dist, names, [0, 1] * 700)
    assert os.path.exists(out_dir / "temp" / "kmer_searcher" / "output.bin")


def test_cli_from_reads_only(tmp_path, oracle):
    """-i reads.fasta alone: k-mer counting, sampling, search, loader, embed, k-NN, writer -- the whole
    reference pipeline on this build.  The oracle side starts from the library the run sampled (the sample
    itself depends on the RNG, as it does on the awk implementation in the reference) and restates
    everything after it."""
    from fedrann_amd.precompute import read_kmer_counts
    from fedrann_amd.synth import synth_sequences
    k = 15
    s = synth_sequences(600, genome_len=100_000, mean_len=2500, k=k, seed=41)
    reads = [bytes(s["seqs"][s["seq_off"][i]:s["seq_off"][i + 1]]) for i in range(600)]
    fa = tmp_path / "reads.fa"
    fa.write_bytes(b"".join(b">%s\n%s\n" % (i, r) for i, r in zip(s["ids"], reads)))
    out_dir = tmp_path / "out"
    cli.main(["-i", str(fa), "-k", str(k), "--kmer-sample-fraction", "0.05", "-o", str(out_dir), "-n", "128",
              "--nndescent-n-neighbors", "20", "--keep-intermediates", "--seed", "77"])
    got = open(out_dir / "overlaps.tsv", newline="").read()
    lib_path = out_dir / "temp" / "fwd_kmer_library.fasta"
    fwd = [l for l in lib_path.read_bytes().split(b"\n") if l and not l.startswith(b">")]
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    rev = [x.translate(comp)[::-1] for x in fwd]
    # the library is what jellyfish count -C | dump -L 2 would hold, sub-sampled
    wc, wn = oracle.kmer_count(reads, k, 2)
    from fedrann_amd.count_kmers import codes_to_kmers
    universe = {r.tobytes(): int(c) for r, c in zip(codes_to_kmers(wc, k), wn)}
    counts = read_kmer_counts(str(lib_path))
    assert 0 < len(fwd) < len(universe) and all(universe[x] == c for x, c in zip(fwd, counts.tolist()))
    L = len(fwd)
    codes = oracle.kmer_library(b"\n".join(fwd + rev), k)
    ip, ix = oracle.kmer_search(reads, codes, k)
    rows = []
    for r in range(600):
        idx = ix[ip[r]:ip[r + 1]].astype(np.int64)
        rows.append(idx.tolist())
        rows.append(np.where(idx < L, idx + L, idx - L).tolist())
    P = oracle.precompute_matrix(counts, 128)
    indptr, indices = oracle.rows_to_csr(rows)
    E = oracle.embed(indptr, indices, P, 2 * L, 128)
    idx, dist = oracle.knn(E, 20)
    names = [i.decode() for i in s["ids"] for _ in (0, 1)]
    assert got == oracle.overlaps_tsv(idx, dist, names, [0, 1] * 600)


def test_cli_devices_sharded_run_is_byte_identical(tmp_path):
    """`--devices a,b,c` starts one process per entry BEFORE the parent touches a GPU; every rank embeds its
    row block, the blocks are all-gathered, every rank searches its rows and writes its part, the parent
    concatenates the parts.  Here the three ranks share GPU 0 over gloo (RCCL needs one GPU per rank): the
    result must be the single-GPU overlaps.tsv byte for byte, incl. a ragged last block (1800 rows / 3
    ranks in 32-row-aligned blocks)."""
    import subprocess
    import sys
    s = synth(900, seed=5, m=80)
    names = ["read_%d/ccs" % i for i in range(900)]
    out_bin, fasta, L = _write_intermediates(tmp_path, s, names)
    base = ["-n", "128", "--nndescent-n-neighbors", "20", "--kmer-searcher-output", out_bin, "--kmer-library", fasta]
    one = tmp_path / "one"
    cli.main(["-o", str(one)] + base)
    many = tmp_path / "many"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "fedrann_amd", "-o", str(many), "--devices", "0,0,0", "--dist-backend",
                        "gloo", "--keep-intermediates"] + base, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert (many / "overlaps.tsv").read_bytes() == (one / "overlaps.tsv").read_bytes()
    parts = sorted(p.name for p in (many / "temp").iterdir() if p.name.startswith("overlaps.rank"))
    assert parts == ["overlaps.rank0.tsv", "overlaps.rank1.tsv", "overlaps.rank2.tsv"]


def test_cli_devices_from_reads_runs_stage1_first_and_shards_the_host_side(tmp_path):
    """`-i reads --devices a,b`: stage 1 runs in a child of its own BEFORE the ranks exist (nobody waits in a
    collective meanwhile), every rank then loads only ITS rows of output.bin (the ranged native loader) and the
    result is the single-GPU file byte for byte."""
    import re
    import subprocess
    import sys
    from fedrann_amd.synth import synth_sequences
    k = 15
    s = synth_sequences(500, genome_len=80_000, mean_len=2500, k=k, sample=0.05, seed=71)
    rng = np.random.default_rng(5)
    counts = rng.integers(2, 40, size=len(s["fwd"]))
    lib = tmp_path / "fwd_kmer_library.fasta"
    lib.write_bytes(b"".join(b">%d\n%s\n" % (int(c), x) for c, x in zip(counts, s["fwd"])))
    fa = tmp_path / "reads.fasta"
    fa.write_bytes(b"".join(b">%s\n%s\n" % (i, bytes(s["seqs"][s["seq_off"][j]:s["seq_off"][j + 1]]))
                            for j, i in enumerate(s["ids"])))
    base = ["-i", str(fa), "-k", str(k), "-n", "128", "--nndescent-n-neighbors", "20", "--kmer-library", str(lib)]
    one = tmp_path / "one"
    cli.main(["-o", str(one)] + base)
    many = tmp_path / "many"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "fedrann_amd", "-o", str(many), "--devices", "0,0", "--dist-backend", "gloo"]
                       + base, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert (many / "overlaps.tsv").read_bytes() == (one / "overlaps.tsv").read_bytes()
    held = [(int(m.group(1)), int(m.group(2))) for m in re.finditer(r"holds rows \[(\d+), (\d+)\) of 1000: (?:\d+) column ids", r.stderr)]
    assert sorted(held) == [(0, 512), (512, 1000)]  # two ranks, 32-row-aligned blocks, each its own rows only


def test_cli_devices_parent_ends_the_other_ranks_when_one_fails(tmp_path):
    """A rank that dies (here: a device ordinal that does not exist) must not leave its siblings waiting in a
    collective until the process group's timeout: the parent polls all children and ends the rest."""
    import subprocess
    import sys
    import time
    s = synth(300, seed=5, m=60)
    out_bin, fasta, L = _write_intermediates(tmp_path, s, ["r%d" % i for i in range(300)])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "fedrann_amd", "-o", str(tmp_path / "o"), "--devices", "0,99", "--dist-backend",
                        "gloo", "-n", "128", "--nndescent-n-neighbors", "20", "--kmer-searcher-output", out_bin,
                        "--kmer-library", fasta], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "rank worker(s) failed" in (r.stderr + r.stdout)
    assert time.time() - t0 < 120


def test_cli_fastq_headers_follow_the_fasta_id_rule(tmp_path, oracle):
    """The reference converts FASTQ to FASTA (seqkit fq2fa, count_kmers.py:76-79) before kmer_searcher sees
    it, so a read's name is its header up to the first space OR TAB: ONT / PacBio style descriptions and
    samtools tags must not end up in overlaps.tsv (nor make the output.bin writer refuse the TAB)."""
    from fedrann_amd.synth import synth_sequences
    k = 15
    s = synth_sequences(300, genome_len=60_000, mean_len=2000, k=k, sample=0.05, seed=53)
    rng = np.random.default_rng(3)
    counts = rng.integers(2, 40, size=len(s["fwd"]))
    lib = tmp_path / "fwd_kmer_library.fasta"
    lib.write_bytes(b"".join(b">%d\n%s\n" % (int(c), x) for c, x in zip(counts, s["fwd"])))
    reads = [bytes(s["seqs"][s["seq_off"][i]:s["seq_off"][i + 1]]) for i in range(300)]
    fq = tmp_path / "reads.fastq"
    with open(fq, "wb") as f:
        for i, (rid, r) in enumerate(zip(s["ids"], reads)):
            desc = b" runid=abc ch=%d" % i if i % 2 else b"\tRG:Z:x\tch=%d" % i
            f.write(b"@" + rid + desc + b"\n" + r + b"\n+\n" + b"I" * len(r) + b"\n")
    fa = tmp_path / "reads.fasta"
    fa.write_bytes(b"".join(b">%s\n%s\n" % (i, r) for i, r in zip(s["ids"], reads)))
    outs = []
    for path, tag in ((fq, "q"), (fa, "a")):
        out_dir = tmp_path / ("out_" + tag)
        cli.main(["-i", str(path), "-k", str(k), "-o", str(out_dir), "-n", "128", "--nndescent-n-neighbors", "20",
                  "--kmer-library", str(lib)])
        outs.append((out_dir / "overlaps.tsv").read_bytes())
    assert outs[0] == outs[1] and b"runid" not in outs[0] and b"RG:Z" not in outs[0]
    assert outs[0].count(b"\n") > 300


def test_cli_refuses_unsupported_sizes_before_any_work(tmp_path):
    for extra in (["-n", "3000"], ["--nndescent-n-neighbors", "200"]):
        with pytest.raises(SystemExit) as e:
            cli.main(["-o", str(tmp_path / "o"), "--feature-matrix", "missing.npz", "--kmer-counts", "missing.npy"] + extra)
        assert "GPU k-NN kernels" in str(e.value)
