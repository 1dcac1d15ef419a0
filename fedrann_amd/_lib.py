"""ctypes binding of include/fedrann_hip.h (libfedrann_hip.so).

No fallback: if the library is missing or no GPU is visible, FedrannHipError is raised.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FEDRANN_HIP_LIB") or os.path.join(HERE, "libfedrann_hip.so")  # env: dev A/B builds

# symbols declared in include/fedrann_hip.h (tests check that the library exports every one)
SYMBOLS = (
    "fdr_create", "fdr_destroy", "fdr_last_error", "fdr_device_info", "fdr_padded_dim",
    "fdr_projection_load", "fdr_embed", "fdr_knn", "fdr_embed_knn", "fdr_embed_dev",
    "fdr_normalize_dev", "fdr_knn_workspace_bytes", "fdr_knn_dev", "fdr_timing", "fdr_timing_read",
    "fdr_last_uncertified", "fdr_set_knn_mode", "fdr_set_dedup_mode", "fdr_last_unique", "fdr_kmer_output_scan",
    "fdr_kmer_output_load", "fdr_kmer_search", "fdr_kmer_search_indices", "fdr_kmer_count",
    "fdr_kmer_count_fetch", "fdr_set_kmer_count_block", "fdr_last_kmer_count_blocks", "fdr_csr_compact", "fdr_host_register", "fdr_host_unregister",
    "fdr_overlaps_write", "fdr_last_prefilter_launches", "fdr_knn_classes_dev", "fdr_knn_unique_dev",
    "fdr_knn_expand_dev", "fdr_kmer_output_scan_range", "fdr_kmer_output_load_range",
    "fdr_kmer_count_begin", "fdr_kmer_count_add", "fdr_kmer_count_finish", "fdr_reads_scan", "fdr_reads_parse",
    "fdr_kmer_output_append", "fdr_last_query_paths",
)
FDR_MAX_K = 128
KERNELS = ("embed_csr", "normalize_rows", "knn_tile", "knn_merge", "knn_prefilter", "knn_rerank",
           "knn_dedup", "kmer_search", "kmer_compact")
FDR_MAX_DIM = 2048
# fdr_last_query_paths codes (include/fedrann_hip.h: FDR_PATH_*)
PATH_CERTIFIED, PATH_RANGE, PATH_EXACT, PATH_ZERO, PATH_RANGE_OVERFLOW, PATH_GENERIC, PATH_CLASS_MEMBER = 1, 2, 3, 4, 5, 6, 0x80


class FedrannHipError(RuntimeError):
    pass


_lib = None


def _share_torch_hip_runtime():
    """One HIP / HSA runtime per process.  A PyTorch-ROCm wheel bundles its own libamdhip64.so.7; if
    libfedrann_hip.so were loaded first it would pull in /opt/rocm's copy, and torch (imported later,
    e.g. by fedrann_amd.distributed) would start a second runtime that sees no GPU.  So map torch's copy
    first -- located without importing torch -- and let the dynamic linker resolve our DT_NEEDED
    libamdhip64.so.7 to it (same SONAME).  FEDRANN_HIP_SYSTEM_RUNTIME=1 keeps the system runtime."""
    if os.environ.get("FEDRANN_HIP_SYSTEM_RUNTIME") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    except (OSError, ImportError, ValueError):
        pass  # no torch, or an unloadable copy: the system runtime is used


def load_library():
    """dlopen libfedrann_hip.so and declare the prototypes.  Needs no GPU."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FedrannHipError(
            "%s is missing: build it with `python -m fedrann_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    _share_torch_hip_runtime()
    L = ctypes.CDLL(LIB_PATH)
    i32, i64, vp, sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t
    L.fdr_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.fdr_destroy.argtypes = [vp]
    L.fdr_last_error.argtypes = []
    L.fdr_last_error.restype = ctypes.c_char_p
    L.fdr_device_info.argtypes = [vp, ctypes.c_char_p, ctypes.c_int]
    L.fdr_padded_dim.argtypes = [ctypes.c_int]
    L.fdr_projection_load.argtypes = [vp, i64, i32, vp, vp, vp]
    L.fdr_embed.argtypes = [vp, i64, vp, vp, vp]
    L.fdr_knn.argtypes = [vp, vp, i64, i32, i32, vp, vp]
    L.fdr_embed_knn.argtypes = [vp, i64, vp, vp, i32, vp, vp, vp]
    L.fdr_embed_dev.argtypes = [vp, i64, vp, vp, vp, vp]
    L.fdr_normalize_dev.argtypes = [vp, vp, i64, i32, vp, vp, vp]
    L.fdr_knn_workspace_bytes.argtypes = [vp, i64, i64, i32, i32]
    L.fdr_knn_workspace_bytes.restype = sz
    L.fdr_knn_dev.argtypes = [vp, vp, vp, i64, vp, vp, i64, i64, i32, i32, vp, vp, vp, sz, vp]
    L.fdr_knn_classes_dev.argtypes = [vp, vp, vp, i64, i32, i32, i64, vp, sz, vp, ctypes.POINTER(i32)]
    L.fdr_knn_unique_dev.argtypes = [vp, i64, i64, vp, vp, vp]
    L.fdr_knn_expand_dev.argtypes = [vp, i64, i64, i64, vp, vp, i64, vp, vp, vp]
    L.fdr_last_uncertified.argtypes = [vp]
    L.fdr_last_query_paths.argtypes = [vp, vp, i64]
    L.fdr_set_knn_mode.argtypes = [vp, ctypes.c_int]
    L.fdr_set_dedup_mode.argtypes = [vp, ctypes.c_int]
    L.fdr_last_unique.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    L.fdr_last_prefilter_launches.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    p64 = ctypes.POINTER(ctypes.c_int64)
    L.fdr_kmer_output_scan.argtypes = [ctypes.c_char_p, p64, p64, p64]
    L.fdr_kmer_output_load.argtypes = [ctypes.c_char_p, i64, i32, i64, i64, i64, vp, vp, vp, vp]
    L.fdr_kmer_output_scan_range.argtypes = [ctypes.c_char_p, i64, i64, p64, p64, p64]
    L.fdr_kmer_output_load_range.argtypes = [ctypes.c_char_p, i64, i32, i64, i64, i64, i64, i64, vp, vp, vp, vp]
    L.fdr_kmer_output_append.argtypes = [ctypes.c_char_p, i64, vp, vp, vp, vp]
    L.fdr_reads_scan.argtypes = [vp, i64, i32, i32, i32, p64, p64, p64]
    L.fdr_reads_parse.argtypes = [vp, i64, i32, i32, i64, i64, vp, vp, vp]
    L.fdr_csr_compact.argtypes = [vp, i64, vp, vp, vp, vp, i64, i32]
    L.fdr_host_register.argtypes = [vp, vp, sz]
    L.fdr_host_unregister.argtypes = [vp, vp]
    L.fdr_overlaps_write.argtypes = [ctypes.c_char_p, i32, i32, i64, i64, i64, i32, vp, vp, vp, vp, vp, i32, p64]
    L.fdr_kmer_search.argtypes = [vp, vp, vp, i64, vp, i64, i32, vp, p64]
    L.fdr_kmer_search_indices.argtypes = [vp, vp]
    L.fdr_kmer_count.argtypes = [vp, vp, vp, i64, i32, i64, p64]
    L.fdr_kmer_count_fetch.argtypes = [vp, vp, vp]
    L.fdr_kmer_count_begin.argtypes = [vp, i32]
    L.fdr_kmer_count_add.argtypes = [vp, vp, vp, i64]
    L.fdr_kmer_count_finish.argtypes = [vp, i64, p64]
    L.fdr_set_kmer_count_block.argtypes = [vp, ctypes.c_int64]
    L.fdr_last_kmer_count_blocks.argtypes = [vp]
    L.fdr_timing.argtypes = [vp, ctypes.c_int]
    L.fdr_timing_read.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                  ctypes.POINTER(ctypes.c_float)]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name not in ("fdr_last_error", "fdr_knn_workspace_bytes"):
            fn.restype = ctypes.c_int
    _lib = L
    return L


def _ptr(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def _as(a, dtype, name):
    b = np.ascontiguousarray(a, dtype=dtype)
    if b.dtype != np.dtype(dtype):
        raise TypeError("%s must be %s" % (name, np.dtype(dtype)))
    return b


def kmer_output_load(path, n_features, n_threads=0):
    """output.bin -> (indptr int64 [2R+1], indices int32 ascending per row, name_off int64 [R+1],
    names uint8 buffer) through fdr_kmer_output_scan / fdr_kmer_output_load (host only, no GPU)."""
    L = load_library()
    bpath = os.fsencode(path)
    R, nnz, nb = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    rc = L.fdr_kmer_output_scan(bpath, ctypes.byref(R), ctypes.byref(nnz), ctypes.byref(nb))
    if rc != 0:
        raise FedrannHipError("fdr_kmer_output_scan failed (%d): %s" % (rc, L.fdr_last_error().decode()))
    indptr = np.empty(2 * R.value + 1, dtype=np.int64)
    indices = np.empty(2 * nnz.value, dtype=np.int32)
    name_off = np.empty(R.value + 1, dtype=np.int64)
    names = np.empty(nb.value, dtype=np.uint8)
    rc = L.fdr_kmer_output_load(bpath, int(n_features), int(n_threads), R.value, nnz.value, nb.value,
                                indptr.ctypes.data, indices.ctypes.data, name_off.ctypes.data, names.ctypes.data)
    if rc != 0:
        raise FedrannHipError("fdr_kmer_output_load failed (%d): %s" % (rc, L.fdr_last_error().decode()))
    return indptr, indices, name_off, names


def kmer_output_records(path):
    """Record count of an output.bin from its 16-byte header ('<4sB3sQ', feature_extraction.py:110-119); the
    reference's ValueError for a bad magic / version.  No record is read."""
    import struct
    with open(path, "rb") as f:
        head = f.read(16)
    if len(head) < 16:
        raise ValueError("incomplete file header")
    magic, version, _, total = struct.unpack("<4sB3sQ", head)
    if magic != b"KMER":
        raise ValueError("invalid file format (bad magic)")
    if version != 1:
        raise ValueError("unsupported version: %d" % version)
    return int(total)


def kmer_output_load_range(path, n_features, rec_lo, rec_hi, n_threads=0, with_names=True):
    """Rows of the records [rec_lo, rec_hi) of output.bin (rec_hi = None: to the end) -- a rank's block of a
    row-sharded run -- as (n_records of the file, indptr int64 [2 (hi - lo) + 1] rebased to 0, indices int32,
    name_off, names of ALL records or (None, None))."""
    L = load_library()
    bpath = os.fsencode(path)
    R, nnz, nb = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    total = kmer_output_records(path)  # (the header's count clamps the range; one walk over the records below, not two)
    lo = max(0, min(int(rec_lo), total))
    hi = total if rec_hi is None else max(lo, min(int(rec_hi), total))
    rc = L.fdr_kmer_output_scan_range(bpath, lo, hi, ctypes.byref(R), ctypes.byref(nnz), ctypes.byref(nb))
    if rc != 0:
        raise FedrannHipError("fdr_kmer_output_scan_range failed (%d): %s" % (rc, L.fdr_last_error().decode()))
    indptr = np.empty(2 * (hi - lo) + 1, dtype=np.int64)
    indices = np.empty(2 * nnz.value, dtype=np.int32)
    name_off = np.empty(R.value + 1, dtype=np.int64) if with_names else None
    names = np.empty(nb.value, dtype=np.uint8) if with_names else None
    rc = L.fdr_kmer_output_load_range(bpath, int(n_features), int(n_threads), R.value, lo, hi, nnz.value, nb.value,
                                      indptr.ctypes.data, indices.ctypes.data,
                                      name_off.ctypes.data if with_names else None,
                                      names.ctypes.data if with_names and names.size else None)
    if rc != 0:
        raise FedrannHipError("fdr_kmer_output_load_range failed (%d): %s" % (rc, L.fdr_last_error().decode()))
    return R.value, indptr, indices, name_off, names


def reads_parse(buf, n, is_fastq, fastq_ids_as_fasta, eof, seq_buf=None):
    """Whole records in buf[:n] (uint8 array: the tail of the previous piece of a FASTA / FASTQ file + new bytes)
    -> (consumed bytes, ids list of bytes, seqs uint8, seq_off int64 [R + 1]) through fdr_reads_scan /
    fdr_reads_parse (host only, no GPU).  seq_buf: a uint8 array of at least n bytes to hold the sequences (seqs is
    then a view of it) instead of a fresh one."""
    L = load_library()
    used, R, nb = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    base = buf.ctypes.data if n else None
    rc = L.fdr_reads_scan(base, int(n), int(bool(is_fastq)), int(bool(fastq_ids_as_fasta)), int(bool(eof)),
                          ctypes.byref(used), ctypes.byref(R), ctypes.byref(nb))
    if rc != 0:
        raise FedrannHipError("fdr_reads_scan failed (%d): %s" % (rc, L.fdr_last_error().decode()))
    seqs = np.empty(nb.value, dtype=np.uint8) if seq_buf is None else seq_buf[:nb.value]
    off = np.empty(R.value + 1, dtype=np.int64)
    span = np.empty(2 * R.value, dtype=np.int64)
    rc = L.fdr_reads_parse(base, used.value, int(bool(is_fastq)), int(bool(fastq_ids_as_fasta)), R.value, nb.value,
                           seqs.ctypes.data if nb.value else None, off.ctypes.data, span.ctypes.data if R.value else None)
    if rc != 0:
        raise FedrannHipError("fdr_reads_parse failed (%d): %s" % (rc, L.fdr_last_error().decode()))
    view = memoryview(buf)
    sp = span.tolist()
    ids = [bytes(view[sp[2 * r]:sp[2 * r + 1]]) for r in range(R.value)]
    return used.value, ids, seqs, off


def kmer_output_append(path, ids, indptr, indices):
    """Records of output.bin appended to `path` (fdr_kmer_output_append; host only).  ids: list of bytes."""
    L = load_library()
    name_off, names = pack_names(ids)
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    if indptr.size != len(ids) + 1:
        raise ValueError("indptr must have one entry per record + 1")
    rc = L.fdr_kmer_output_append(os.fsencode(path), len(ids), _ptr(name_off), _ptr(names) if names.size else None,
                                  _ptr(indptr), _ptr(indices) if indices.size else None)
    if rc != 0:
        msg = L.fdr_last_error().decode()
        if "non-ASCII" in msg or "bytes long" in msg:
            raise ValueError(msg)
        raise FedrannHipError("fdr_kmer_output_append failed (%d): %s" % (rc, msg))


def pack_names(read_names):
    """list of str / bytes -> (name_off int64 [n + 1], names uint8 buffer) for fdr_overlaps_write."""
    enc = [n if isinstance(n, bytes) else str(n).encode("utf-8") for n in read_names]
    off = np.zeros(len(enc) + 1, dtype=np.int64)
    if enc:
        np.cumsum([len(b) for b in enc], out=off[1:])
    return off, np.frombuffer(b"".join(enc), dtype=np.uint8)


def overlaps_write(path, idx, dist, name_off, names, strands, row0=0, append=False, header=True, n_threads=0):
    """overlaps.tsv rows of neighbour-graph rows row0 .. row0 + idx.shape[0] (host only, no GPU); returns the
    number of data lines.  strands=None: name_off / names describe the RECORDS of a fwd / rev doubled matrix (row t =
    record t >> 1, strand t & 1).  See fdr_overlaps_write."""
    L = load_library()
    idx = _as(idx, np.int32, "idx")
    dist = _as(dist, np.float32, "dist")
    if idx.ndim != 2 or dist.shape != idx.shape:
        raise ValueError("idx and dist must be [rows, k] arrays of the same shape")
    name_off = _as(name_off, np.int64, "name_off")
    names = np.ascontiguousarray(names, dtype=np.uint8)
    if strands is None:  # doubled rows: one name per RECORD, row t = record t >> 1 on strand t & 1
        n_total = 2 * (name_off.size - 1)
    else:
        strands = np.ascontiguousarray(strands, dtype=np.uint8)
        n_total = name_off.size - 1
        if strands.size != n_total:
            raise ValueError("strands must have one entry per row")
    lines = ctypes.c_int64()
    rc = L.fdr_overlaps_write(os.fsencode(path), 1 if append else 0, 1 if header else 0, n_total, int(row0),
                              idx.shape[0], idx.shape[1], _ptr(idx), _ptr(dist), _ptr(name_off),
                              names.ctypes.data if names.size else None, _ptr(strands) if strands is not None else None,
                              int(n_threads),
                              ctypes.byref(lines))
    if rc != 0:
        raise FedrannHipError("fdr_overlaps_write failed (%d): %s" % (rc, L.fdr_last_error().decode()))
    return int(lines.value)


class Context:
    """One GPU context (= fdr_ctx).  Methods raise FedrannHipError on any non-zero return code."""

    def __init__(self, device=0):
        self._L = load_library()
        h = ctypes.c_void_p()
        rc = self._L.fdr_create(int(device), ctypes.byref(h))
        if rc != 0:
            raise FedrannHipError("fdr_create(%d) failed (%d): %s" % (device, rc, self._err()))
        self._h = h
        self.device = int(device)
        self.n_features = 0
        self.d = 0

    def _err(self):
        return self._L.fdr_last_error().decode("utf-8", "replace")

    def _check(self, rc, what):
        if rc != 0:
            raise FedrannHipError("%s failed (%d): %s" % (what, rc, self._err()))

    def close(self):
        if getattr(self, "_h", None):
            self._L.fdr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- info ---------------------------------------------------------------------------------
    def device_info(self):
        buf = ctypes.create_string_buffer(256)
        self._check(self._L.fdr_device_info(self._h, buf, 256), "fdr_device_info")
        name, arch, cus, mem = buf.value.decode().split("|")
        return {"name": name, "arch": arch, "cus": int(cus), "hbm_bytes": int(mem)}

    def padded_dim(self, d):
        dp = self._L.fdr_padded_dim(int(d))
        if dp < 0:
            raise FedrannHipError("embedding dimension %d unsupported (1..%d)" % (d, FDR_MAX_DIM))
        return dp

    def set_knn_mode(self, mode):
        """mode: "auto" (default), "exact" or "prefilter" -- same results, see include/fedrann_hip.h."""
        code = {"auto": 0, "exact": 1, "prefilter": 2}[mode]
        self._check(self._L.fdr_set_knn_mode(self._h, code), "fdr_set_knn_mode")

    def set_dedup_mode(self, mode):
        """Duplicate-row classes: "auto" (default), "off", "on" (at every size) or "force" (always expand;
        tests) -- same results, see include/fedrann_hip.h."""
        code = {"auto": 0, "off": 1, "on": 2, "force": 3}[mode]
        self._check(self._L.fdr_set_dedup_mode(self._h, code), "fdr_set_dedup_mode")

    def kmer_search(self, seqs, seq_off, lib_codes, k):
        """Per-read ascending unique library indices: (indptr int64 [R+1], indices int32 [nnz])."""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        seq_off = np.ascontiguousarray(seq_off, dtype=np.int64)
        lib_codes = np.ascontiguousarray(lib_codes, dtype=np.uint64)
        R = seq_off.size - 1
        if R < 0 or (R >= 0 and seq_off.size and int(seq_off[-1]) != seqs.size):
            raise ValueError("seq_off does not describe seqs")
        indptr = np.empty(R + 1, dtype=np.int64)
        nnz = ctypes.c_int64()
        self._check(self._L.fdr_kmer_search(self._h, seqs.ctypes.data, seq_off.ctypes.data, R,
                                            lib_codes.ctypes.data, lib_codes.size, int(k), indptr.ctypes.data,
                                            ctypes.byref(nnz)), "fdr_kmer_search")
        indices = np.empty(nnz.value, dtype=np.int32)
        self._check(self._L.fdr_kmer_search_indices(self._h, indices.ctypes.data), "fdr_kmer_search_indices")
        return indptr, indices

    def kmer_count(self, seqs, seq_off, k, min_count=1):
        """Canonical k-mers with >= min_count occurrences: (codes uint64 ascending, counts uint64)."""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        seq_off = np.ascontiguousarray(seq_off, dtype=np.int64)
        n = ctypes.c_int64()
        self._check(self._L.fdr_kmer_count(self._h, seqs.ctypes.data, seq_off.ctypes.data, seq_off.size - 1, int(k),
                                           int(min_count), ctypes.byref(n)), "fdr_kmer_count")
        codes = np.empty(n.value, dtype=np.uint64)
        counts = np.empty(n.value, dtype=np.uint64)
        self._check(self._L.fdr_kmer_count_fetch(self._h, codes.ctypes.data, counts.ctypes.data), "fdr_kmer_count_fetch")
        return codes, counts

    def kmer_count_begin(self, k):
        """Incremental counting for a streaming reader: begin, add whole reads piece by piece, finish."""
        self._check(self._L.fdr_kmer_count_begin(self._h, int(k)), "fdr_kmer_count_begin")

    def kmer_count_add(self, seqs, seq_off):
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        seq_off = np.ascontiguousarray(seq_off, dtype=np.int64)
        self._check(self._L.fdr_kmer_count_add(self._h, seqs.ctypes.data, seq_off.ctypes.data, seq_off.size - 1),
                    "fdr_kmer_count_add")

    def kmer_count_finish(self, min_count=1):
        """(codes uint64 ascending, counts uint64) of the canonical k-mers with >= min_count occurrences in all the
        reads added since kmer_count_begin."""
        n = ctypes.c_int64()
        self._check(self._L.fdr_kmer_count_finish(self._h, int(min_count), ctypes.byref(n)), "fdr_kmer_count_finish")
        codes = np.empty(n.value, dtype=np.uint64)
        counts = np.empty(n.value, dtype=np.uint64)
        self._check(self._L.fdr_kmer_count_fetch(self._h, codes.ctypes.data, counts.ctypes.data), "fdr_kmer_count_fetch")
        return codes, counts

    def set_kmer_count_block(self, chars):
        """Characters per block of kmer_count (0 = default 2^31); small blocks exercise the merge in tests."""
        self._check(self._L.fdr_set_kmer_count_block(self._h, int(chars)), "fdr_set_kmer_count_block")

    def last_kmer_count_blocks(self):
        """Non-empty blocks the last kmer_count call counted."""
        return int(self._L.fdr_last_kmer_count_blocks(self._h))

    def last_unique(self):
        """(unique target rows, unique query rows) searched by the last k-NN call."""
        a, b = ctypes.c_int(), ctypes.c_int()
        self._check(self._L.fdr_last_unique(self._h, ctypes.byref(a), ctypes.byref(b)), "fdr_last_unique")
        return int(a.value), int(b.value)

    def last_prefilter_launches(self):
        """(launches, queues) of the fp16 candidate pass of the last k-NN call (0, 0 after an exact-mode call)."""
        a, b = ctypes.c_int(), ctypes.c_int()
        self._check(self._L.fdr_last_prefilter_launches(self._h, ctypes.byref(a), ctypes.byref(b)),
                    "fdr_last_prefilter_launches")
        return int(a.value), int(b.value)

    def last_uncertified(self):
        """Prefilter mode: query rows of the last k-NN call that were searched by the exact kernel."""
        return int(self._L.fdr_last_uncertified(self._h))

    def last_query_paths(self, n_queries):
        """uint8 [n_queries]: which way every query row of the last k-NN call took to its result (PATH_* codes, bit 7 =
        member of a duplicate-row class of several rows): fdr_last_query_paths.  Diagnostics; the tests' strata."""
        out = np.empty(int(n_queries), dtype=np.uint8)
        self._check(self._L.fdr_last_query_paths(self._h, _ptr(out), int(n_queries)), "fdr_last_query_paths")
        return out

    def timing(self, enable):
        self._check(self._L.fdr_timing(self._h, 1 if enable else 0), "fdr_timing")

    def timing_read(self, which):
        """(launch count, total ms) of kernel kind `which` since the last read; KERNELS names them."""
        n, ms = ctypes.c_int(), ctypes.c_float()
        self._check(self._L.fdr_timing_read(self._h, int(which), ctypes.byref(n), ctypes.byref(ms)),
                    "fdr_timing_read")
        return int(n.value), float(ms.value)

    # -- host-pointer API -----------------------------------------------------------------------
    def projection_load(self, p_indptr, p_cols, p_vals, n_features, d):
        p_indptr = _as(p_indptr, np.int64, "p_indptr")
        p_cols = _as(p_cols, np.int32, "p_cols")
        p_vals = _as(p_vals, np.float32, "p_vals")
        if p_indptr.shape != (int(n_features) + 1,):
            raise ValueError("p_indptr must have n_features + 1 entries")
        self._check(self._L.fdr_projection_load(self._h, int(n_features), int(d), _ptr(p_indptr),
                                                _ptr(p_cols), _ptr(p_vals)), "fdr_projection_load")
        self.n_features, self.d = int(n_features), int(d)

    def csr_compact(self, a_indptr, a_indices, n_threads=0):
        """The CSR without the column ids whose projection row is empty (same E, ~10x fewer ids)."""
        a_indptr = _as(a_indptr, np.int64, "a_indptr")
        a_indices = _as(a_indices, np.int32, "a_indices")
        n = a_indptr.shape[0] - 1
        out_ip = np.empty(n + 1, dtype=np.int64)
        out_ix = np.empty(max(int(a_indices.size), 1), dtype=np.int32)
        self._check(self._L.fdr_csr_compact(self._h, n, _ptr(a_indptr), _ptr(a_indices), _ptr(out_ip),
                                            _ptr(out_ix), int(out_ix.size), int(n_threads)), "fdr_csr_compact")
        return out_ip, np.ascontiguousarray(out_ix[:int(out_ip[-1])])

    def host_register(self, *arrays):
        """Pin numpy arrays the caller keeps passing to embed / knn / embed_knn (PCIe-rate copies)."""
        for a in arrays:
            if a is not None and a.nbytes:
                self._check(self._L.fdr_host_register(self._h, a.ctypes.data, a.nbytes), "fdr_host_register")

    def host_unregister(self, *arrays):
        for a in arrays:
            if a is not None and a.nbytes:
                self._check(self._L.fdr_host_unregister(self._h, a.ctypes.data), "fdr_host_unregister")

    def embed(self, a_indptr, a_indices):
        a_indptr = _as(a_indptr, np.int64, "a_indptr")
        a_indices = _as(a_indices, np.int32, "a_indices")
        n = a_indptr.shape[0] - 1
        E = np.empty((n, self.d), dtype=np.float32)
        self._check(self._L.fdr_embed(self._h, n, _ptr(a_indptr), _ptr(a_indices), _ptr(E)),
                    "fdr_embed")
        return E

    def knn(self, E, k):
        E = _as(E, np.float32, "E")
        if E.ndim != 2:
            raise ValueError("E must be 2-D")
        n, d = E.shape
        idx = np.empty((n, k), dtype=np.int32)
        dist = np.empty((n, k), dtype=np.float32)
        self._check(self._L.fdr_knn(self._h, _ptr(E), n, d, int(k), _ptr(idx), _ptr(dist)),
                    "fdr_knn")
        return idx, dist

    def embed_knn(self, a_indptr, a_indices, k, return_embedding=False, out=None):
        """out=(idx int32 [n,k], dist float32 [n,k]): caller-owned (e.g. pinned, reused) result arrays."""
        a_indptr = _as(a_indptr, np.int64, "a_indptr")
        a_indices = _as(a_indices, np.int32, "a_indices")
        n = a_indptr.shape[0] - 1
        if out is None:
            idx = np.empty((n, k), dtype=np.int32)
            dist = np.empty((n, k), dtype=np.float32)
        else:
            idx, dist = out
            if (idx.shape, dist.shape) != ((n, k), (n, k)) or idx.dtype != np.int32 or dist.dtype != np.float32 \
                    or not (idx.flags.c_contiguous and dist.flags.c_contiguous):
                raise ValueError("out must be C-contiguous (int32 [n,k], float32 [n,k])")
        E = np.empty((n, self.d), dtype=np.float32) if return_embedding else None
        self._check(self._L.fdr_embed_knn(self._h, n, _ptr(a_indptr), _ptr(a_indices), int(k),
                                          _ptr(idx), _ptr(dist), _ptr(E)), "fdr_embed_knn")
        return (idx, dist, E) if return_embedding else (idx, dist)

    # -- device-pointer API (integers are raw device addresses, e.g. torch.Tensor.data_ptr()) -----
    def embed_dev(self, n_rows, d_indptr, d_indices, d_E, stream=0):
        self._check(self._L.fdr_embed_dev(self._h, int(n_rows), d_indptr, d_indices, d_E,
                                          stream or None), "fdr_embed_dev")

    def normalize_dev(self, d_E, n_rows, d, d_Ehat, d_zero, stream=0):
        self._check(self._L.fdr_normalize_dev(self._h, d_E, int(n_rows), int(d), d_Ehat, d_zero,
                                              stream or None), "fdr_normalize_dev")

    def knn_workspace_bytes(self, nq, nt, d, k):
        return int(self._L.fdr_knn_workspace_bytes(self._h, int(nq), int(nt), int(d), int(k)))

    def knn_dev(self, d_Qhat, d_qzero, nq, d_That, d_tzero, nt, t_base, d, k, d_idx, d_dist,
                d_ws, ws_bytes, stream=0):
        self._check(self._L.fdr_knn_dev(self._h, d_Qhat, d_qzero, int(nq), d_That, d_tzero,
                                        int(nt), int(t_base), int(d), int(k), d_idx, d_dist, d_ws,
                                        int(ws_bytes), stream or None), "fdr_knn_dev")


    def knn_classes_dev(self, d_That, d_tzero, nt, d, k, nq_max, d_ws, ws_bytes, stream=0):
        """Duplicate-row classes of the target set for knn_unique_dev / knn_expand_dev; returns the number of
        unique rows (0: not worth it, use knn_dev)."""
        nu = ctypes.c_int32()
        self._check(self._L.fdr_knn_classes_dev(self._h, d_That, d_tzero, int(nt), int(d), int(k), int(nq_max), d_ws,
                                                int(ws_bytes), stream or None, ctypes.byref(nu)), "fdr_knn_classes_dev")
        return int(nu.value)

    def knn_unique_dev(self, u_lo, u_hi, d_idx_u, d_dist_u, stream=0):
        self._check(self._L.fdr_knn_unique_dev(self._h, int(u_lo), int(u_hi), d_idx_u, d_dist_u, stream or None),
                    "fdr_knn_unique_dev")

    def knn_expand_dev(self, q0, nq, t_base, d_idx_u_all, d_dist_u_all, d_idx, d_dist, stream=0, u_row_stride=0):
        self._check(self._L.fdr_knn_expand_dev(self._h, int(q0), int(nq), int(t_base), d_idx_u_all, d_dist_u_all,
                                               int(u_row_stride), d_idx, d_dist, stream or None), "fdr_knn_expand_dev")


_default_ctx = None


def default_context():
    """Process-wide context on device $FEDRANN_DEVICE (default: LOCAL_RANK, else 0)."""
    global _default_ctx
    if _default_ctx is None:
        dev = int(os.environ.get("FEDRANN_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _default_ctx = Context(dev)
    return _default_ctx
