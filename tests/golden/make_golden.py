#!/opt/conda/bin/python3.9
"""Generate the golden vectors under tests/golden/ by RUNNING the reference.

Run in the build container only (the reference lives at /root/reference there
and never travels to the GPU box):

    /opt/conda/bin/python3.9 -B tests/golden/make_golden.py

Interpreter: /opt/conda python3.9 with numpy 1.26.4 -- the reference's pinned
numpy (requirements.txt:11); under numpy >= 2 the same reference source yields
a float64 projection (NEP-50 promotion) that differs by 1 ulp after the final
float32 cast, so the pinned interpreter is the one that defines "reference
output" (SURVEY.md section 8a-2).

What is produced (data only: inputs + expected outputs, no reference source):

  precompute_tiny.npz      P for L=6 counts [2,5,3,7,2,11], d=4
  precompute_mid.npz       P for L=50_000 random counts (seed 7), d=128
  precompute_big.json      sha256 of P (CSR by feature) for L=1_000_000, d=256
  embed_tiny.npz           E for 5 hand-written reads through get_feature_matrix
  embed_mid.npz            E for 1500 synthetic reads (3000 rows), d=128
  metadata_tiny.json       get_metadata() names / strands for the tiny reads
  overlaps_edge.{npz,tsv}  get_output_dataframe + to_csv on hand-written edge cases
  overlaps_rand.{npz,tsv}  the same on a random 400-row x 12 neighbour table

Modules that the reference imports but never uses on this path (Bio, numba,
ahocorasick, sharedmem, pysam, isal, xxhash, memory_profiler, pynndescent,
hnswlib) are absent from this container; empty in-memory stand-ins are
registered so that `import` succeeds.  No arithmetic on the path goes through
a stand-in.  pynndescent (the k-NN arithmetic itself) is NOT available, so no
k-NN golden vector can be made from the reference: see DESIGN.md ("parity
unpinned" for the k-NN stage).
"""
import hashlib
import io
import json
import os
import struct
import sys
import tempfile
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_stubs():
    _stub("Bio", SeqIO=types.ModuleType("Bio.SeqIO"))
    sys.modules["Bio.SeqIO"] = sys.modules["Bio"].SeqIO
    _stub("numba", njit=lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f)))
    for n in ("ahocorasick", "sharedmem", "pysam", "xxhash", "memory_profiler",
              "pynndescent", "hnswlib"):
        if n not in sys.modules:
            try:
                __import__(n)
            except Exception:
                _stub(n, memory_usage=None)
    # scipy 1.7 (py3.9 env) keeps csr_matrix in scipy.sparse.csr; the reference imports
    # the scipy >= 1.8 private path scipy.sparse._csr (type annotations only).
    import scipy.sparse
    if "scipy.sparse._csr" not in sys.modules:
        try:
            __import__("scipy.sparse._csr")
        except Exception:
            _stub("scipy.sparse._csr", csr_matrix=scipy.sparse.csr_matrix)
    try:
        import isal  # noqa: F401
    except Exception:
        isal = _stub("isal")
        isal.igzip = _stub("isal.igzip")
    try:
        import colorama  # noqa: F401
    except Exception:
        class _E:
            def __getattr__(self, k):
                return ""
        _stub("colorama", Fore=_E(), Style=_E(), init=lambda **k: None)


def _import_reference():
    sys.path.insert(0, REF)
    _install_stubs()
    import fedrann  # package __init__ only
    # precompute.py:62 uses a PEP-604 annotation; python3.9 needs postponed evaluation.
    path = os.path.join(REF, "fedrann", "precompute.py")
    with open(path) as f:
        src = "from __future__ import annotations\n" + f.read()
    mod = types.ModuleType("fedrann.precompute")
    mod.__package__ = "fedrann"
    mod.__file__ = path
    sys.modules["fedrann.precompute"] = mod
    exec(compile(src, path, "exec"), mod.__dict__)
    fedrann.precompute = mod
    # same annotation issue in nearest_neighbors.py / __main__.py
    for name in ("nearest_neighbors", "__main__"):
        p = os.path.join(REF, "fedrann", name + ".py")
        with open(p) as f:
            s = "from __future__ import annotations\n" + f.read()
        m = types.ModuleType("fedrann." + name)
        m.__package__ = "fedrann"
        m.__file__ = p
        m.__dict__["__name__"] = "fedrann." + name
        sys.modules["fedrann." + name] = m
        exec(compile(s, p, "exec"), m.__dict__)
    import fedrann.feature_extraction as fe
    return mod, fe, sys.modules["fedrann.__main__"]


def write_counts_fasta(path, counts):
    with open(path, "w") as f:
        for i, c in enumerate(counts):
            f.write(">%d\n%s\n" % (int(c), "ACGT"))


def write_output_bin(path, names, index_lists):
    with open(path, "wb") as f:
        f.write(struct.pack("<4sB3sQ", b"KMER", 1, b"\0\0\0", len(names)))
        for n, idx in zip(names, index_lists):
            nb = n.encode()
            f.write(struct.pack("<H", len(nb)))
            f.write(nb)
            f.write(struct.pack("<I", len(idx)))
            f.write(struct.pack("<%dQ" % len(idx), *[int(i) for i in idx]))


def p_to_csr_arrays(P):
    import numpy as np
    c = P.tocsr()
    c.sort_indices()
    assert c.data.dtype == np.float32, c.data.dtype
    return (c.indptr.astype(np.int64), c.indices.astype(np.int32),
            c.data.view(np.uint32).copy())


def main():
    import numpy as np
    import scipy
    import pandas as pd
    pre, fe, mainmod = _import_reference()
    import fedrann.global_variables as gv
    gv.threads = 2
    versions = {"numpy": np.__version__, "scipy": scipy.__version__,
                "pandas": pd.__version__, "python": sys.version.split()[0]}
    print("versions", versions)
    assert np.__version__.startswith("1.26"), "run under the pinned numpy 1.26.x"
    tmp = tempfile.mkdtemp(prefix="golden_")

    # ---- precompute -------------------------------------------------------
    def run_pre(counts, d):
        fa = os.path.join(tmp, "lib_%d_%d.fasta" % (len(counts), d))
        write_counts_fasta(fa, counts)
        P, F = pre.get_precompute_matrix(n_components=d, counter_file=fa,
                                         n_features=2 * len(counts))
        assert F == 2 * len(counts)
        return P

    counts_tiny = np.array([2, 5, 3, 7, 2, 11], dtype=np.int64)
    P_tiny = run_pre(counts_tiny, 4)
    ip, ix, bits = p_to_csr_arrays(P_tiny)
    np.savez_compressed(os.path.join(HERE, "precompute_tiny.npz"), counts=counts_tiny,
                        d=4, indptr=ip, indices=ix, data_bits=bits)

    counts_mid = np.random.default_rng(7).integers(2, 61, size=50_000).astype(np.int64)
    P_mid = run_pre(counts_mid, 128)
    ip, ix, bits = p_to_csr_arrays(P_mid)
    np.savez_compressed(os.path.join(HERE, "precompute_mid.npz"), counts=counts_mid,
                        d=128, indptr=ip, indices=ix, data_bits=bits)

    counts_big = np.random.default_rng(11).integers(2, 61, size=1_000_000).astype(np.int64)
    P_big = run_pre(counts_big, 256)
    ip, ix, bits = p_to_csr_arrays(P_big)
    h = hashlib.sha256()
    for a in (ip, ix, bits):
        h.update(np.ascontiguousarray(a).tobytes())
    with open(os.path.join(HERE, "precompute_big.json"), "w") as f:
        json.dump({"L": 1_000_000, "d": 256, "counts_rng": "default_rng(11).integers(2,61,L)",
                   "nnz": int(ix.size), "sha256_indptr_i64_indices_i32_data_f32": h.hexdigest(),
                   "versions": versions}, f, indent=1)

    # ---- embed (get_feature_matrix) ----------------------------------------
    names_tiny = ["r0", "r1", "r2", "r3", "r4"]
    reads_tiny = [[0, 3, 5], [1], [7, 2], [11, 0, 1, 2], [4, 10, 9]]
    ob = os.path.join(tmp, "tiny.bin")
    write_output_bin(ob, names_tiny, reads_tiny)
    E = fe.get_feature_matrix(ks_file=ob, precompute_matrix=P_tiny, kmer_count=12,
                              read_count=5, chunk_size=4)
    assert E.dtype == np.float32 and E.shape == (10, 4)
    np.savez_compressed(os.path.join(HERE, "embed_tiny.npz"),
                        read_lens=np.array([len(r) for r in reads_tiny], np.int64),
                        read_idx=np.concatenate([np.array(r, np.int64) for r in reads_tiny]),
                        L=6, d=4, E_bits=E.view(np.uint32))
    rn, st = fe.get_metadata(ks_file=ob, kmer_count=12)
    with open(os.path.join(HERE, "metadata_tiny.json"), "w") as f:
        json.dump({"names": names_tiny, "read_names": list(rn), "strands": [int(s) for s in st]}, f)

    # mid: 1500 reads over the L=50_000 library; every read has >= 1 index (the
    # reference is undefined for empty reads), order as kmer_searcher
    # emits it (a set in arbitrary order, no duplicates).
    rng = np.random.default_rng(1234)
    L = 50_000
    reads_mid, names_mid = [], []
    for i in range(1500):
        n = int(rng.integers(1, 400))
        idx = rng.choice(2 * L, size=n, replace=False)
        reads_mid.append(idx.astype(np.int64))
        names_mid.append("read_%d/%d" % (i, n))
    ob = os.path.join(tmp, "mid.bin")
    write_output_bin(ob, names_mid, reads_mid)
    E = fe.get_feature_matrix(ks_file=ob, precompute_matrix=P_mid, kmer_count=2 * L,
                              read_count=1500, chunk_size=1000)
    assert E.dtype == np.float32 and E.shape == (3000, 128)
    np.savez_compressed(os.path.join(HERE, "embed_mid.npz"),
                        read_lens=np.array([len(r) for r in reads_mid], np.int64),
                        read_idx=np.concatenate(reads_mid), L=L, d=128,
                        E_bits=E.view(np.uint32))

    # ---- overlaps.tsv writer ------------------------------------------------
    def run_tsv(tag, idx, dist, names, strands):
        df = mainmod.get_output_dataframe(neighbor_matrix=idx, neighbor_distances=dist,
                                          read_names=names, strands=strands)
        buf = io.StringIO()
        df.to_csv(buf, sep="\t", index=False)
        with open(os.path.join(HERE, tag + ".tsv"), "w", newline="") as f:
            f.write(buf.getvalue())
        np.savez_compressed(os.path.join(HERE, tag + ".npz"), indices=idx,
                            dist_bits=dist.view(np.uint32), names=np.array(names),
                            strands=np.array(strands, np.int64))

    idx = np.array([[0, 2, 3], [1, 0, -1], [3, 2, 0], [3, 1, 2]], dtype=np.int32)
    dist = np.array([[0, .660664, .7207873], [1.1920929e-07, .5, np.inf],
                     [.25, .25, 1], [0, .33333334, .9]], dtype=np.float32)
    run_tsv("overlaps_edge", idx, dist, ["rA", "rA", "rB", "rB"], [0, 1, 0, 1])

    rng = np.random.default_rng(99)
    n, k = 400, 12
    idx = np.empty((n, k), np.int32)
    for i in range(n):
        idx[i] = rng.choice(n, size=k, replace=False)
        if rng.random() < 0.8:
            idx[i, 0] = i
    dist = np.sort(rng.random((n, k), dtype=np.float32) ** 3, axis=1).astype(np.float32)
    dist[rng.random((n, k)) < 0.05] = np.float32(1.0)
    dist[:, 0][idx[:, 0] == np.arange(n)] = 0.0
    dist[3, 4] = np.float32(5.9604645e-08)
    dist[7, 2] = np.float32(1.0000001e-05)
    dist[9, 1] = np.float32(0.0001)
    dist[11, 1] = np.float32(9.999999e-05)
    names = []
    for i in range(n // 2):
        names += ["read/%d_x" % i] * 2
    run_tsv("overlaps_rand", idx, dist, names, [0, 1] * (n // 2))

    with open(os.path.join(HERE, "VERSIONS.json"), "w") as f:
        json.dump(versions, f, indent=1)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
