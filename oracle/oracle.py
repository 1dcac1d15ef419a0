"""CPU oracle for the FEDRANN hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package (fedrann_amd/) never does.  Each function restates one
reference function and cites it (paths relative to the reference checkout).

Pinned against the reference's own outputs (tests/golden/, made by tests/golden/make_golden.py):
    precompute_matrix  <- precompute.py:58-115              PINNED (precompute_{tiny,mid,big})
    parse_output_bin   <- feature_extraction.py:108-140     PINNED (embed_*.npz inputs, metadata_tiny.json)
    embed              <- feature_extraction.py:167-292     PINNED (embed_{tiny,mid}.npz)
    overlaps_tsv       <- __main__.py:261-300, :385         PINNED (overlaps_{edge,rand}.tsv)
    knn                <- nearest_neighbors.py:39-55        PARITY UNPINNED (pynndescent absent; see
                          the header of fedrann_oracle.c)
    nndescent          <- the same call, as the ALGORITHM pynndescent runs (nndescent.c: RP forest + NN-descent
                          restated from its published description): CPU baseline and recall figure only
    read_sequences / kmer_library / kmer_search
                       <- kmer_searcher/kmer_searcher.cpp   PARITY UNPINNED (needs the un-vendored
                          robin_hood.h; its own test data pin an obsolete format)
"""
import ctypes
import io
import math
import os
import struct
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile oracle/libfedrann_oracle.so (gcc, a second or two)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libfedrann_oracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        i64, i32, vp = ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p
        L.orc_embed.argtypes = [i64, vp, vp, i64, vp, vp, vp, i32, vp]
        L.orc_embed.restype = ctypes.c_int
        L.orc_normalize.argtypes = [vp, i64, i32, vp, vp, vp]
        L.orc_normalize.restype = None
        L.orc_knn.argtypes = [vp, vp, i64, vp, vp, i64, i64, i32, i32, vp, vp]
        L.orc_knn.restype = ctypes.c_int
        L.orc_pair_dist.argtypes = [vp, vp, i32, ctypes.c_int, ctypes.c_int]
        L.orc_pair_dist.restype = ctypes.c_float
        L.orc_nndescent.argtypes = [vp, vp, i64, i32, i32, i32, i32, ctypes.c_uint64, vp, vp, vp]
        L.orc_nndescent.restype = ctypes.c_int
        L.orc_num_threads.restype = ctypes.c_int
        L.orc_set_num_threads.argtypes = [ctypes.c_int]
        L.orc_set_num_threads.restype = None
        _LIB = L
    return _LIB


def set_num_threads(n):
    """Threads the OpenMP loops of the oracle use (bench.py sets the cgroup CPU budget)."""
    lib().orc_set_num_threads(int(n))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------------------------------
# precompute.py:44-55 kmer_count_generator + :58-115 get_precompute_matrix
# --------------------------------------------------------------------------------------------
def read_kmer_counts(counter_file):
    """Counts of the forward k-mer library, in file order (precompute.py:44-55: a '>count'
    header line followed by the k-mer; the count applies to index i and to i + L)."""
    counts = []
    with open(counter_file) as f:
        for line in f:
            if line.startswith(">"):
                counts.append(int(line.strip()[1:]))
    return np.asarray(counts, dtype=np.int64)


def precompute_matrix(counts, n_components, seed=2094):
    """Literal restatement of precompute.py:58-115 with the arithmetic types the reference gets
    under its pinned numpy 1.26.4 (value-based casting keeps everything float32).

    counts: int array [L] (forward library); features F = 2L, count[i+L] = count[i] (:52-54).
    Returns (indptr int64 [F+1], cols int32 [nnz], vals float32 [nnz]) = P (F x d) as CSR by
    feature with columns ascending inside a feature row.
    """
    import scipy.sparse as sp
    L = int(len(counts))
    F = 2 * L
    result_array = np.zeros(F, dtype=np.uint64)              # :68
    result_array[:L] = counts                                  # :71-75 (i, count)
    result_array[L:] = counts                                  #        (i + L, count)
    idf = np.log(F / (result_array + 1e-12)).astype(np.float32)  # :77  (f64 log -> f32)
    density = 1 / math.sqrt(F)                                 # :80
    rng = np.random.default_rng(seed)                          # :86
    indices, indptr, offset = [], [0], 0
    for _ in range(n_components):                              # :90-96
        n_i = rng.binomial(F, density)
        indices.append(rng.choice(F, n_i, replace=False))
        offset += n_i
        indptr.append(offset)
    indices = np.concatenate(indices)
    data = rng.binomial(1, 0.5, size=np.size(indices)) * 2 - 1  # :101
    comp = sp.csr_matrix((data, indices, indptr), shape=(n_components, F), dtype=np.float32)  # :104
    # :107 under numpy 1.26: python-float64 scalar * float32 matrix -> float32 data, i.e. the
    # scalar is rounded to float32 first and the product is one float32 multiply.
    scale = np.float32(np.sqrt(1 / density) / np.sqrt(n_components))
    comp.data = (scale * comp.data.astype(np.float32)).astype(np.float32)
    Pt = comp.T.tocsr()                                         # :111  F x d
    Pt.sort_indices()
    rows = np.repeat(np.arange(F), np.diff(Pt.indptr))
    vals = (Pt.data.astype(np.float32) * idf[rows]).astype(np.float32)  # :113 multiply(idf)
    return Pt.indptr.astype(np.int64), Pt.indices.astype(np.int32), vals


# --------------------------------------------------------------------------------------------
# feature_extraction.py:108-140 parse_kmer_searcher_output (+ :295-302 get_metadata)
# --------------------------------------------------------------------------------------------
def parse_output_bin(path, L):
    """Record-by-record struct.unpack loop exactly as the reference does it.  Returns
    (names[2R], strands[2R], index_lists[2R]); row 2i = forward set, row 2i+1 = mirrored set
    (i + L if i < L else i - L)."""
    names, strands, rows = [], [], []
    with open(path, "rb") as f:
        header = f.read(16)
        if len(header) < 16:
            raise ValueError("incomplete header")
        magic, version, _, total = struct.unpack("<4sB3sQ", header)
        if magic != b"KMER":
            raise ValueError("bad magic")
        if version != 1:
            raise ValueError("unsupported version %d" % version)
        for _ in range(total):
            (id_len,) = struct.unpack("<H", f.read(2))
            idb = f.read(id_len)
            try:
                name = idb.decode("utf-8")
            except UnicodeDecodeError:
                name = "".join(chr(b) if b < 128 else "_" for b in idb)
            (cnt,) = struct.unpack("<I", f.read(4))
            idx = struct.unpack("<%dQ" % cnt, f.read(8 * cnt))
            names += [name, name]
            strands += [0, 1]
            rows.append(list(idx))
            rows.append([i + L if i < L else i - L for i in idx])
    return names, strands, rows


def rows_to_csr(rows):
    indptr = np.zeros(len(rows) + 1, dtype=np.int64)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    indices = np.fromiter((i for r in rows for i in r), dtype=np.int64, count=int(indptr[-1]))
    return indptr, indices


# --------------------------------------------------------------------------------------------
# feature_extraction.py:167-292: E = A . P
# --------------------------------------------------------------------------------------------
def embed(a_indptr, a_indices, P, n_features, d):
    p_indptr, p_cols, p_vals = P
    a_indptr = np.ascontiguousarray(a_indptr, dtype=np.int64)
    a_indices = np.ascontiguousarray(a_indices, dtype=np.int64)
    p_indptr = np.ascontiguousarray(p_indptr, dtype=np.int64)
    p_cols = np.ascontiguousarray(p_cols, dtype=np.int32)
    p_vals = np.ascontiguousarray(p_vals, dtype=np.float32)
    n = len(a_indptr) - 1
    E = np.empty((n, d), dtype=np.float32)
    rc = lib().orc_embed(n, _p(a_indptr), _p(a_indices), int(n_features), _p(p_indptr),
                         _p(p_cols), _p(p_vals), int(d), _p(E))
    if rc != 0:
        raise ValueError("feature index out of range")
    return E


def normalize(E):
    E = np.ascontiguousarray(E, dtype=np.float32)
    n, d = E.shape
    Eh = np.empty_like(E)
    rinv = np.empty(n, dtype=np.float32)
    zero = np.empty(n, dtype=np.uint8)
    lib().orc_normalize(_p(E), n, d, _p(Eh), _p(rinv), _p(zero))
    return Eh, rinv, zero


def knn_normalized(Qh, q_zero, Th, t_zero, k, t_base=0):
    Qh = np.ascontiguousarray(Qh, dtype=np.float32)
    Th = np.ascontiguousarray(Th, dtype=np.float32)
    nq, d = Qh.shape
    nt = Th.shape[0]
    idx = np.empty((nq, k), dtype=np.int32)
    dist = np.empty((nq, k), dtype=np.float32)
    qz = np.ascontiguousarray(q_zero, dtype=np.uint8)
    tz = np.ascontiguousarray(t_zero, dtype=np.uint8)
    rc = lib().orc_knn(_p(Qh), _p(qz), nq, _p(Th), _p(tz), nt, int(t_base), d, int(k),
                       _p(idx), _p(dist))
    if rc != 0:
        raise ValueError("orc_knn failed (%d): need n_targets >= k" % rc)
    return idx, dist


def knn(E, k, q_lo=0, q_hi=None):
    """Canonical exact cosine k-NN of rows [q_lo, q_hi) of E against all rows of E."""
    Eh, _, zero = normalize(E)
    q_hi = Eh.shape[0] if q_hi is None else q_hi
    return knn_normalized(Eh[q_lo:q_hi], zero[q_lo:q_hi], Eh, zero, k)


def nndescent(Eh, zero, k, n_trees=300, leaf_size=200, seed=602):
    """The ALGORITHM of the reference's k-NN stage (nearest_neighbors.py:39-55 -> pynndescent.NNDescent with the
    arguments of __main__.py:184-197: angular RP forest of n_trees trees, leaf_size, NN-descent with
    max_candidates = min(60, k), delta = 0.001, max(5, log2 N) rounds) restated in oracle/nndescent.c from its
    published description -- the package itself is absent (PARITY UNPINNED; an approximate, randomised method).
    Eh / zero: normalised rows and zero flags (normalize()).  Returns (idx int32 [n,k], dist float32 [n,k]
    ascending, stats dict)."""
    Eh = np.ascontiguousarray(Eh, dtype=np.float32)
    zero = np.ascontiguousarray(zero, dtype=np.uint8)
    n, d = Eh.shape
    idx = np.empty((n, k), dtype=np.int32)
    dist = np.empty((n, k), dtype=np.float32)
    stats = np.zeros(3, dtype=np.int64)
    rc = lib().orc_nndescent(_p(Eh), _p(zero), n, d, int(k), int(n_trees), int(leaf_size), int(seed), _p(idx), _p(dist),
                             _p(stats))
    if rc != 0:
        raise ValueError("orc_nndescent failed (%d)" % rc)
    return idx, dist, {"rounds": int(stats[0]), "leaves": int(stats[1]), "distance_evaluations": int(stats[2])}


def pair_dist_normalized(a, a_zero, b, b_zero):
    """Canonical distance of two NORMALISED rows (fp32 fma chain, clamp, zero-row rules)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return float(lib().orc_pair_dist(_p(a), _p(b), len(a), int(a_zero), int(b_zero)))


def pair_dist(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return float(lib().orc_pair_dist(_p(a), _p(b), len(a), 0, 0))


# --------------------------------------------------------------------------------------------
# __main__.py:261-300 get_output_dataframe + :385 df.to_csv(sep="\t", index=False)
# --------------------------------------------------------------------------------------------
def overlaps_tsv(indices, distances, read_names, strands):
    """Pure-Python N x k loop exactly as the reference writes it; returns the TSV text.
    Only the row == query test skips a neighbour; the rank keeps its column number; a -1 index
    aliases the LAST read through Python negative indexing, as in the reference."""
    import pandas as pd
    qn, qo, tn, to, rk, ds = [], [], [], [], [], []
    for q in range(indices.shape[0]):
        for rank, t in enumerate(indices[q]):
            if t == q:
                continue
            qn.append(read_names[q])
            qo.append("+-"[strands[q]])
            tn.append(read_names[t])
            to.append("+-"[strands[t]])
            rk.append(rank)
            ds.append(distances[q][rank])
    df = pd.DataFrame({"query_name": qn, "query_orientation": qo, "target_name": tn,
                       "target_orientation": to, "neighbor_rank": rk, "distance": ds})
    buf = io.StringIO()
    df.to_csv(buf, sep="\t", index=False)
    return buf.getvalue()


# --------------------------------------------------------------------------------------------
# kmer_searcher (kmer_searcher/kmer_searcher.cpp) -- PARITY UNPINNED, see fedrann_oracle.c
# --------------------------------------------------------------------------------------------
def read_sequences(path):
    """kmer_searcher.cpp:153-200, line by line.  FASTA unless the first line starts with '@'.
    FASTA: id = header up to the first space/tab, sequence = the following lines concatenated (only the
    '\\n' is stripped: a '\\r' stays in the sequence and is an invalid character); a header with an empty
    id does not start a record.  FASTQ: id = the whole header line after '@', sequence = the next line,
    two lines skipped.  Returns (ids, sequences) as lists of bytes."""
    ids, seqs = [], []
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()  # getline does not yield a final empty piece
    is_fastq = bool(lines) and lines[0][:1] == b"@"
    cur_id, cur_seq, i = b"", b"", 0
    while i < len(lines):
        line = lines[i]
        i += 1
        if not line:
            continue
        if not is_fastq:
            if line[:1] == b">":
                if cur_id:
                    ids.append(cur_id)
                    seqs.append(cur_seq)
                head = line[1:]
                cut = min([p for p in (head.find(b" "), head.find(b"\t")) if p >= 0], default=-1)
                cur_id = head if cut < 0 else head[:cut]
                cur_seq = b""
            else:
                cur_seq += line
        elif line[:1] == b"@":
            cur_id = line[1:]
            cur_seq = lines[i] if i < len(lines) else b""
            i += 3
            ids.append(cur_id)
            seqs.append(cur_seq)
    if not is_fastq and cur_id:
        ids.append(cur_id)
        seqs.append(cur_seq)
    return ids, seqs


def kmer_library(text, k):
    """Library text (bytes) -> uint64 codes of the unique valid k-mers in index order (:262-279)."""
    buf = np.frombuffer(bytes(text), dtype=np.uint8)
    cap = max(16, len(text) // max(k, 1) + 16)
    codes = np.empty(cap, dtype=np.uint64)
    L = lib()
    L.orc_kmer_library.restype = ctypes.c_int64
    L.orc_kmer_library.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]
    n = L.orc_kmer_library(buf.ctypes.data if buf.size else None, buf.size, int(k), _p(codes), cap)
    if n < 0:
        raise ValueError("orc_kmer_library failed (%d)" % n)
    return codes[:n].copy()


def kmer_search(seqs, codes, k):
    """Per-read sorted unique library indices (:306-352).  seqs: list of bytes.  Returns (indptr, indices)."""
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    cat = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    codes = np.ascontiguousarray(codes, dtype=np.uint64)
    indptr = np.empty(len(seqs) + 1, dtype=np.int64)
    L = lib()
    L.orc_kmer_search.restype = ctypes.c_int64
    L.orc_kmer_search.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                  ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    cap = max(1024, int(off[-1]) + len(seqs))
    indices = np.empty(cap, dtype=np.int32)
    nnz = L.orc_kmer_search(cat.ctypes.data if cat.size else None, _p(off), len(seqs), _p(codes), codes.size, int(k),
                            _p(indptr), _p(indices), cap)
    if nnz < 0 or nnz > cap:
        raise ValueError("orc_kmer_search failed (%d)" % nnz)
    return indptr, indices[:nnz].copy()


def kmer_count(seqs, k, min_count=1):
    """`jellyfish count -C` + `dump -L` restated with numpy (third-party tool, PARITY UNPINNED): every
    window of k characters inside one read, all in ACGTacgt, counts under the smaller of its 2-bit code
    and its reverse complement's.  Returns (codes ascending, counts)."""
    lut = np.full(256, 255, dtype=np.uint8)
    for i, c in enumerate(b"ACGT"):
        lut[c] = i
        lut[c | 0x20] = i
    allc = []
    for s in seqs:
        a = lut[np.frombuffer(bytes(s), dtype=np.uint8)]
        if a.size < k:
            continue
        w = np.lib.stride_tricks.sliding_window_view(a, k)
        ok = (w != 255).all(axis=1)
        w = w[ok].astype(np.uint64)
        fw = np.zeros(w.shape[0], dtype=np.uint64)
        rc = np.zeros(w.shape[0], dtype=np.uint64)
        for j in range(k):
            fw = (fw << np.uint64(2)) | w[:, j]
            rc = (rc << np.uint64(2)) | (np.uint64(3) - w[:, k - 1 - j])
        allc.append(np.minimum(fw, rc))
    if not allc:
        return np.zeros(0, np.uint64), np.zeros(0, np.uint64)
    codes, counts = np.unique(np.concatenate(allc), return_counts=True)
    keep = counts >= min_count
    return codes[keep], counts[keep].astype(np.uint64)
