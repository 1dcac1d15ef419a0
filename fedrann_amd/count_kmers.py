"""Stage 1 of the reference pipeline on the GPU: canonical k-mer counting, sampling, k-mer search.

Mirror of fedrann/count_kmers.py:52-148 (`run_kmer_searcher`), whose work is done there by third-party
binaries:

    jellyfish count -m k -s 10G -t T -C reads.fasta      -> fdr_kmer_count   (canonical counts)
    jellyfish dump -L min_multiplicity                    -> the min_count argument
    awk 'BEGIN{srand(seed)} ... rand() > 1-p'              -> numpy Generator(PCG64(seed)).random() > 1-p
    seqkit seq -r -p -t DNA fwd > rev                      -> reverse complement, same order
    cat fwd rev | grep -v '^>' | kmer_searcher ...         -> fedrann_amd.kmer_search.kmer_searcher

What cannot be reproduced bit for bit, by construction: jellyfish dumps in its hash-table order and awk's
rand() stream depends on the awk implementation, so WHICH k-mers are sampled and how they are numbered
differs from a reference run (any run of the reference on another awk differs in the same way).  The
feature order only names the features; everything downstream of fwd_kmer_library.fasta is exact again.
Here the library is in ascending code order (= lexicographic k-mer order).
"""
import os
from os.path import join

import numpy as np

from . import _lib, global_variables
from .kmer_search import iter_sequence_blocks, kmer_searcher, unique_first

_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def codes_to_kmers(codes, k):
    """uint64 2-bit codes -> uint8 [n, k] of 'ACGT' characters (first base in the highest position)."""
    codes = np.asarray(codes, dtype=np.uint64)
    out = np.empty((codes.size, k), dtype=np.uint8)
    for j in range(k):
        out[:, j] = _BASES[((codes >> np.uint64(2 * (k - 1 - j))) & np.uint64(3)).astype(np.intp)]
    return out


def count_canonical_kmers(seqs, seq_off, k, min_multiplicity=1, context=None):
    """(codes ascending, counts) of the canonical k-mers occurring at least min_multiplicity times."""
    ctx = context or _lib.default_context()
    return ctx.kmer_count(seqs, seq_off, int(k), int(min_multiplicity))


def sample_kmers(n, sample_fraction, seed):
    """Bernoulli sample of n library candidates: keep where rand() > 1 - p (count_kmers.py:103-117)."""
    rng = np.random.Generator(np.random.PCG64(int(seed)))
    return np.flatnonzero(rng.random(int(n)) > 1.0 - float(sample_fraction))


def revcomp_codes(codes, k):
    """2-bit codes of the reverse complements (what `seqkit seq -r -p` makes of each library k-mer)."""
    c = np.asarray(codes, dtype=np.uint64) ^ np.uint64((1 << (2 * k)) - 1)  # complement: A<->T, C<->G = 3 - base
    out = np.zeros_like(c)
    for _ in range(k):
        out = (out << np.uint64(2)) | (c & np.uint64(3))
        c = c >> np.uint64(2)
    return out


def kmer_library_text(codes, counts, k):
    """jellyfish-dump style FASTA as one uint8 array: '>count' then the k-mer, a line each."""
    counts = np.asarray(counts, dtype=np.uint64)
    n = counts.size
    nd = np.ones(n, dtype=np.int64)  # decimal digits of each count
    p = np.uint64(10)
    for _ in range(19):
        more = counts >= p
        if not more.any():
            break
        nd += more
        p = p * np.uint64(10)
    rec = nd + (k + 3)  # '>' digits '\n' k-mer '\n'
    start = np.zeros(n, dtype=np.int64)
    np.cumsum(rec[:-1], out=start[1:])
    out = np.empty(int(rec.sum()), dtype=np.uint8)
    out[start] = ord(">")
    rest = counts.copy()
    for j in range(int(nd.max()) if n else 0):  # digit j from the right
        m = nd > j
        out[(start + nd - j)[m]] = (48 + rest[m] % np.uint64(10)).astype(np.uint8)
        rest //= np.uint64(10)
    out[start + nd + 1] = 10
    kmers = codes_to_kmers(codes, k)
    for j in range(k):
        out[start + nd + 2 + j] = kmers[:, j]
    out[start + nd + 2 + k] = 10
    return out


def write_kmer_library(path, codes, counts, k):
    """jellyfish-dump style FASTA: '>count' then the k-mer (count_kmers.py:119-121; read back by
    precompute.py:44-55)."""
    kmer_library_text(codes, counts, k).tofile(path)


def run_kmer_searcher(input_path, k, sample_fraction, min_multiplicity=2, context=None):
    """Same signature and return value as the reference (count_kmers.py:52-148):
    (path of output.bin, number of features = 2 x |forward library|, number of reads).  Writes
    temp/fwd_kmer_library.fasta, temp/rev_kmer_library.fasta and temp/kmer_searcher/{output.bin,
    kmer_frequency.bin} like the reference does."""
    tmp = global_variables.temp_dir
    if not tmp:
        raise RuntimeError("global_variables.temp_dir is not set")
    if input_path.endswith(".gz"):
        import gzip
        from shutil import copyfileobj
        plain = join(tmp, os.path.basename(input_path[:-3]))
        with gzip.open(input_path, "rb") as src, open(plain, "wb") as dst:
            copyfileobj(src, dst, 1 << 24)
        input_path = plain
    if not input_path.endswith((".fasta", ".fa", ".fastq", ".fq")):
        raise ValueError("Unsupported file format. Please provide a FASTA or FASTQ file.")  # count_kmers.py:72-75
    # the reads are streamed twice (counting, then the search): the host never holds the read set
    # (fastq_ids_as_fasta: the reference runs seqkit fq2fa first)
    ctx = context or _lib.default_context()
    ctx.kmer_count_begin(int(k))
    for _, seqs, off in iter_sequence_blocks(input_path, fastq_ids_as_fasta=True, reuse_buffers=True):
        ctx.kmer_count_add(seqs, off)
    codes, counts = ctx.kmer_count_finish(int(min_multiplicity))
    keep = sample_kmers(codes.size, sample_fraction, global_variables.seed)
    fwd = join(tmp, "fwd_kmer_library.fasta")
    write_kmer_library(fwd, codes[keep], counts[keep], k)
    kmer_count = int(keep.size)
    rev = join(tmp, "rev_kmer_library.fasta")
    rev_codes = revcomp_codes(codes[keep], k)  # (seqkit seq -r -p: same headers, same order)
    write_kmer_library(rev, rev_codes, counts[keep], k)
    out_dir = join(tmp, "kmer_searcher")
    # the library as `cat fwd rev | grep -v '^>'` reads: forward k-mers then their reverse complements, a k-mer seen
    # before (a palindrome) loses its second slot -- from the codes at hand instead of the two files just written
    lib_codes = unique_first(np.concatenate((codes[keep], rev_codes)))
    n_reads, _, _, _ = kmer_searcher([fwd, rev], input_path, out_dir, k, context=context, fastq_ids_as_fasta=True,
                                     collect=False, lib_codes=lib_codes)
    return join(out_dir, "output.bin"), kmer_count * 2, n_reads
