"""IDF-weighted very-sparse random projection matrix P (F x d).

Host-side mirror of the reference's fedrann/precompute.py (get_precompute_matrix :58-115,
kmer_count_generator :44-55).  P is tiny next to the data (nnz ~ d * sqrt(F)) and its values
depend on numpy's PCG64 stream, so it is built on the host with numpy and handed to the GPU as
CSR by feature (fdr_projection_load).  Bit-identical to the reference under its pinned numpy
1.26.4 (float32 arithmetic) -- see tests/golden/precompute_*.
"""
import logging
import math

import numpy as np
from scipy.sparse import csr_matrix

logger = logging.getLogger("fedrann_amd")

DEFAULT_SEED = 2094  # precompute.py:63 -- the CLI --seed is NOT forwarded (__main__.py:331-335)


def read_kmer_counts(counter_file):
    """Counts of the forward k-mer library in file order (int64 [L]).

    The file is jellyfish-dump FASTA (count_kmers.py:101,121): a '>count' line, then the k-mer.
    As in kmer_count_generator (precompute.py:44-55) every non-header line is one k-mer whose
    count is the most recent header's.
    """
    with open(counter_file, "rb") as f:
        data = f.read()
    if not data:
        return np.zeros(0, dtype=np.int64)
    buf = np.frombuffer(data, dtype=np.uint8)
    nl = np.flatnonzero(buf == 10)
    starts = np.concatenate(([0], nl + 1)).astype(np.int64)
    ends = np.concatenate((nl, [buf.size])).astype(np.int64)
    keep = starts < buf.size  # drop the empty "line" after a trailing newline
    starts, ends = starts[keep], ends[keep]
    nonempty = ends > starts
    is_hdr = np.zeros(starts.size, dtype=bool)
    is_hdr[nonempty] = buf[starts[nonempty]] == ord(">")
    hs, he = starts[is_hdr] + 1, ends[is_hdr]
    # vectorised decimal parse of the header integers (whitespace / CR tolerated like int(str.strip()))
    vals = np.zeros(hs.size, dtype=np.int64)
    done = np.zeros(hs.size, dtype=bool)
    seen = np.zeros(hs.size, dtype=bool)
    maxlen = int((he - hs).max()) if hs.size else 0
    for p in range(maxlen):
        pos = hs + p
        inside = (pos < he) & ~done
        ch = np.where(inside, buf[np.minimum(pos, buf.size - 1)], 32)
        digit = (ch >= 48) & (ch <= 57)
        space = (ch == 32) | (ch == 13) | (ch == 9)
        if np.any(inside & ~digit & ~space):
            raise ValueError("%s: non-numeric count in a '>' header line" % counter_file)
        vals = np.where(digit & ~done, vals * 10 + (ch.astype(np.int64) - 48), vals)
        seen |= digit
        done |= seen & space & inside  # trailing whitespace ends the number
    if hs.size and not np.all(seen):
        raise ValueError("%s: empty count in a '>' header line" % counter_file)
    kmer_lines = ~is_hdr
    hdr_before = np.cumsum(is_hdr) - 1
    which = hdr_before[kmer_lines]
    if which.size and which.min() < 0:
        raise ValueError("%s: k-mer line before the first '>' header" % counter_file)
    return vals[which].astype(np.int64)


def kmer_count_generator(filename, kmer_count):
    """API-compatible with the reference generator: yields (i, count) then (i + kmer_count, count)."""
    for i, c in enumerate(read_kmer_counts(filename)):
        yield i, int(c)
        yield i + kmer_count, int(c)


def build_precompute_matrix(counts, n_components, n_features=None, density="auto",
                            seed=DEFAULT_SEED):
    """P from the forward-library counts.  Returns scipy CSR (F x d, float32, sorted indices).

    Arithmetic (reference line numbers in precompute.py):
      :68-75  count[i] = count[i + L] = counts[i]   (uint64; features without a line stay 0)
      :77     idf = float32(ln(F / (count + 1e-12)))          (float64 log, one rounding)
      :80-84  density = 1 / sqrt(F)
      :86-101 rng = default_rng(seed); per component n_c = binomial(F, density),
              idx_c = choice(F, n_c, replace=False); then ONE draw sign = binomial(1, .5, nnz)*2-1
      :107    scale = float32(sqrt(1/density) / sqrt(d))     (value-based casting of numpy 1.x)
      :113    P[f, c] = (scale * sign) * idf[f]                (two float32 multiplies)
    """
    counts = np.asarray(counts, dtype=np.int64)
    L = int(counts.size)
    F = 2 * L if n_features is None else int(n_features)
    half = int(F / 2)  # the reference passes int(n_features / 2) as kmer_count
    if L > half:
        raise IndexError("counter file has %d k-mers but n_features/2 = %d" % (L, half))
    if F <= 0:
        raise ValueError("n_features must be positive")
    cnt = np.zeros(F, dtype=np.uint64)
    cnt[:L] = counts
    cnt[half:half + L] = counts
    with np.errstate(divide="ignore"):
        idf = np.log(F / (cnt + 1e-12)).astype(np.float32)
    logger.debug("idf.shape=%s", idf.shape)
    if density == "auto":
        _density = 1 / math.sqrt(F)
    else:
        assert isinstance(density, float) and 0 < density <= 1
        _density = density
    rng = np.random.default_rng(seed)
    cols, nnz_per_comp = [], []
    for _ in range(n_components):
        n_i = rng.binomial(F, _density)
        cols.append(rng.choice(F, n_i, replace=False))
        nnz_per_comp.append(n_i)
    feat = np.concatenate(cols) if cols else np.zeros(0, dtype=np.int64)
    sign = rng.binomial(1, 0.5, size=np.size(feat)) * 2 - 1
    comp = np.repeat(np.arange(n_components, dtype=np.int32), nnz_per_comp)
    scale = np.float32(np.sqrt(1 / _density) / np.sqrt(n_components))
    vals = ((scale * sign.astype(np.float32)).astype(np.float32) * idf[feat]).astype(np.float32)
    # CSR by feature, columns ascending inside a feature row (a (f, c) pair occurs at most once:
    # choice(replace=False) never repeats a feature inside one component)
    order = np.lexsort((comp, feat))
    feat, comp, vals = feat[order], comp[order], vals[order]
    indptr = np.zeros(F + 1, dtype=np.int64)
    np.cumsum(np.bincount(feat, minlength=F), out=indptr[1:])
    P = csr_matrix((vals, comp.astype(np.int32), indptr), shape=(F, n_components))
    P.has_sorted_indices = True
    return P


def get_precompute_matrix(n_components, counter_file, n_features, density="auto",
                          seed=DEFAULT_SEED):
    """Drop-in for the reference's get_precompute_matrix (same arguments, same return pair).

    Returns (P, n_features): P is scipy sparse F x d float32 (CSR here; the reference returns the
    COO result of .multiply() and converts it to CSR on first use, feature_extraction.py:227).
    """
    counts = read_kmer_counts(counter_file)
    P = build_precompute_matrix(counts, n_components, n_features=n_features, density=density,
                                seed=seed)
    logger.debug("precompute_matrix.shape=%s nnz=%d", P.shape, P.nnz)
    return P, n_features
