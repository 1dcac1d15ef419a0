"""Read x feature matrix handling and E = A . P on the GPU.

Host-side mirror of the live part of the reference's fedrann/feature_extraction.py:
    parse_kmer_searcher_output  :108-140   (output.bin reader, fwd/rev doubling)
    get_feature_matrix          :216-292   (E = A . P; here one fdr_embed call on the GPU)
    get_metadata                :295-302
plus the `feature_matrix.npz` (scipy.sparse.save_npz, binary int8 CSR) reader/writer that the
reference documents (README.md:66) but no longer implements (SURVEY.md section 5 note).
"""
import logging
import struct

import numpy as np
import scipy.sparse as sp

from . import _lib

logger = logging.getLogger("fedrann_amd")

_HEADER = struct.Struct("<4sB3sQ")


def read_kmer_searcher_output(ks_file):
    """Parse kmer_searcher's output.bin (writer: kmer_searcher.cpp:98-130).

    Layout: header '<4sB3sQ' = b"KMER", version 1, 3 pad bytes, record count; per record
    '<H' id length, id bytes, '<I' index count, that many '<Q' feature indices.
    Returns (names [R] list of str, indptr int64 [R+1], indices int64 [nnz]) in file order.
    """
    with open(ks_file, "rb") as f:
        data = f.read()
    if len(data) < 16:
        raise ValueError("incomplete file header")  # feature_extraction.py:111-112
    magic, version, _, total = _HEADER.unpack_from(data, 0)
    if magic != b"KMER":
        raise ValueError("invalid file format (bad magic)")  # :116-117
    if version != 1:
        raise ValueError("unsupported version: %d" % version)  # :118-119
    mv = memoryview(data)
    names = []
    offs = np.empty(total, dtype=np.int64)
    cnts = np.empty(total, dtype=np.int64)
    pos = 16
    n = len(data)
    for r in range(total):
        if pos + 2 > n:
            raise ValueError("truncated record %d" % r)
        id_len = data[pos] | (data[pos + 1] << 8)
        pos += 2
        idb = bytes(mv[pos:pos + id_len])
        pos += id_len
        try:
            names.append(idb.decode("utf-8"))
        except UnicodeDecodeError:  # :125-128
            names.append("".join(chr(b) if b < 128 else "_" for b in idb))
        if pos + 4 > n:
            raise ValueError("truncated record %d" % r)
        c = int.from_bytes(mv[pos:pos + 4], "little")
        pos += 4
        offs[r] = pos
        cnts[r] = c
        pos += 8 * c
        if pos > n:
            raise ValueError("truncated record %d" % r)
    indptr = np.zeros(total + 1, dtype=np.int64)
    np.cumsum(cnts, out=indptr[1:])
    nnz = int(indptr[-1])
    # gather the uint64 payloads (records are not 8-byte aligned in the file)
    raw = np.frombuffer(data, dtype=np.uint8)
    if nnz:
        byte_start = np.repeat(offs - 8 * indptr[:-1], cnts) + 8 * np.arange(nnz, dtype=np.int64)
        gather = (byte_start[:, None] + np.arange(8, dtype=np.int64)[None, :]).ravel()
        indices = raw[gather].view("<u8").astype(np.int64)
    else:
        indices = np.zeros(0, dtype=np.int64)
    return names, indptr, indices


def parse_kmer_searcher_output(ks_file, kmer_count):
    """Generator with the reference's signature and yield order (feature_extraction.py:108-140):
    (id, indices, 0) then (id, mirrored indices, 1) per record; `kmer_count` is L = F / 2."""
    names, indptr, indices = read_kmer_searcher_output(ks_file)
    L = int(kmer_count)
    for r, name in enumerate(names):
        idx = indices[indptr[r]:indptr[r + 1]]
        yield name, tuple(int(i) for i in idx), 0
        yield name, [int(i) + L if i < L else int(i) - L for i in idx], 1


def get_metadata(ks_file, kmer_count):
    """(read_names, strands) of the 2R embedding rows (feature_extraction.py:295-302)."""
    names, _, _ = read_kmer_searcher_output(ks_file)
    read_names = [n for n in names for _ in (0, 1)]
    strands = [0, 1] * len(names)
    return read_names, strands


def canonical_csr(indptr, indices, n_features, what="feature matrix"):
    """Validate and return (indptr int64, indices int32) with ascending columns per row --
    what scipy's COO->CSR conversion gives the reference (feature_extraction.py:204)."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.asarray(indices)
    if indptr.ndim != 1 or indptr.size < 1 or indptr[0] != 0 or indptr[-1] != indices.size:
        raise ValueError("%s: inconsistent indptr" % what)
    if np.any(np.diff(indptr) < 0):
        raise ValueError("%s: indptr not monotone" % what)
    if indices.size:
        lo, hi = int(indices.min()), int(indices.max())
        if lo < 0 or hi >= n_features:
            raise ValueError("%s: feature index out of range [0, %d): min %d max %d"
                             % (what, n_features, lo, hi))
    if n_features > np.iinfo(np.int32).max:
        raise ValueError("n_features exceeds int32")
    indices = indices.astype(np.int64, copy=False)
    nrows = indptr.size - 1
    rows = np.repeat(np.arange(nrows, dtype=np.int64), np.diff(indptr))
    if indices.size > 1:
        same_row = rows[1:] == rows[:-1]
        if np.any(same_row & (indices[1:] <= indices[:-1])):  # not strictly ascending: sort
            key = rows * np.int64(n_features) + indices
            key.sort()
            indices = key - rows * np.int64(n_features)
            if np.any(same_row & (indices[1:] == indices[:-1])):
                raise ValueError("%s: duplicate feature index inside a row (kmer_searcher emits "
                                 "sets; duplicates are not supported)" % what)
    return indptr, np.ascontiguousarray(indices, dtype=np.int32)


def _decode_names(name_off, name_buf):
    """Record ids as str: UTF-8, or -- for ids that are not valid UTF-8 -- ASCII with every byte >= 128
    replaced by '_' (feature_extraction.py:125-128)."""
    raw = name_buf.tobytes()
    off = name_off.tolist()
    try:
        if raw.isascii():
            txt = raw.decode("ascii")  # one decode; byte offsets are character offsets
            return [txt[off[r]:off[r + 1]] for r in range(len(off) - 1)]
    except AttributeError:
        pass
    names = []
    for r in range(len(off) - 1):
        b = raw[off[r]:off[r + 1]]
        try:
            names.append(b.decode("utf-8"))
        except UnicodeDecodeError:
            names.append("".join(chr(c) if c < 128 else "_" for c in b))
    return names


def build_feature_csr(ks_file, n_features, n_threads=0):
    """output.bin -> binary CSR of the 2R doubled rows (row 2i = record i, row 2i+1 = its strand
    mirror: i + L if i < L else i - L; feature_extraction.py:136-140), parsed, mirrored and sorted by
    the native loader (fdr_kmer_output_load; no Python fallback).
    Returns (indptr, indices int32 sorted, read_names, strands)."""
    try:
        indptr, indices, name_off, name_buf = _lib.kmer_output_load(ks_file, int(n_features), n_threads)
    except _lib.FedrannHipError as e:
        msg = str(e)
        if "output.bin:" in msg:  # format errors keep the reference's exception type
            raise ValueError(msg.split("output.bin:", 1)[1].strip()) from None
        raise
    names = _decode_names(name_off, name_buf)
    read_names = [n for n in names for _ in (0, 1)]
    strands = [0, 1] * len(names)
    return indptr, indices, read_names, strands


def save_feature_matrix_npz(path, indptr, indices, n_features):
    """Write the binary read x feature CSR exactly as scipy.sparse.save_npz does (int8 ones)."""
    nrows = len(indptr) - 1
    A = sp.csr_matrix((np.ones(len(indices), dtype=np.int8), np.asarray(indices, dtype=np.int32),
                       np.asarray(indptr, dtype=np.int32 if indptr[-1] < 2**31 else np.int64)),
                      shape=(nrows, int(n_features)))
    sp.save_npz(path, A)


def load_feature_matrix_npz(path):
    """feature_matrix.npz -> (indptr int64, indices int32 ascending per row, n_features)."""
    A = sp.load_npz(path).tocsr()
    if A.nnz and not np.all(A.data == 1):
        raise ValueError("%s: the feature matrix must be binary (all stored values 1)" % path)
    F = int(A.shape[1])
    indptr, indices = canonical_csr(A.indptr, A.indices, F, what=path)
    return indptr, indices, F


def _projection_csr(precompute_matrix):
    P = precompute_matrix.tocsr()  # feature_extraction.py:227
    if P.data.dtype != np.float32:
        # numpy >= 2 would give the reference a float64 P; the pinned behaviour is float32
        raise TypeError("precompute_matrix must be float32 (got %s)" % P.data.dtype)
    P.sort_indices()
    return P


def embed_csr(indptr, indices, precompute_matrix, context=None):
    """E = A . P for a canonical binary CSR (indptr, indices); float32 [rows, d]."""
    ctx = context or _lib.default_context()
    P = _projection_csr(precompute_matrix)
    F, d = P.shape
    ctx.projection_load(P.indptr, P.indices, P.data, F, d)
    return ctx.embed(indptr, indices)


def get_feature_matrix(ks_file, precompute_matrix, kmer_count, read_count, chunk_size=1000,
                       context=None):
    """Drop-in for the reference's get_feature_matrix (feature_extraction.py:216-292).

    ks_file: kmer_searcher output.bin; precompute_matrix: P (F x d scipy sparse, float32);
    kmer_count: F; read_count: R.  Returns float32 [2R, d].  `chunk_size` is accepted for
    signature compatibility; the GPU embeds all rows in one launch.
    A strand with no sampled k-mer yields an all-zero row (the reference leaves such rows
    uninitialised).
    """
    indptr, indices, _, _ = build_feature_csr(ks_file, int(kmer_count))
    if indptr.size - 1 != 2 * int(read_count):
        raise ValueError("output.bin holds %d records but read_count is %d"
                         % ((indptr.size - 1) // 2, read_count))
    logger.debug("embedding %d rows, %d nnz on the GPU", indptr.size - 1, indices.size)
    return embed_csr(indptr, indices, precompute_matrix, context=context)
