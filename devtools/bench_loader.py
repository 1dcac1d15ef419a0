"""output.bin -> CSR: the native loader (fdr_kmer_output_load) against the reference-style
struct.unpack loop (oracle.parse_output_bin + sort, timed on a sample of the records).
usage: python devtools/bench_loader.py [reads]"""
import os
import struct
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fedrann_amd import feature_extraction as fx  # noqa: E402
from fedrann_amd.synth import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
s = synth(R, seed=602)
F = int(s["n_features"])
L = F // 2
ip, ix = s["indptr"], s["indices"].astype(np.uint64)
rng = np.random.default_rng(1)
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "output.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("<4sB3sQ", b"KMER", 1, b"\0\0\0", R))
        for r in range(R):
            row = ix[ip[r]:ip[r + 1]].copy()
            rng.shuffle(row)  # kmer_searcher emits a hash set: arbitrary order
            name = b"read_%09d" % r
            f.write(struct.pack("<H", len(name)) + name + struct.pack("<I", row.size) + row.tobytes())
    size = os.path.getsize(path)
    fx.build_feature_csr(path, F)  # page cache + thread pool warm
    for thr in (1, 0):
        t0 = time.perf_counter()
        indptr, indices, names, strands = fx.build_feature_csr(path, F, n_threads=thr)
        dt = time.perf_counter() - t0
        print("native loader, %s threads: %.3f s  %.2f GB/s of file  %.1f M index/s (%d records, %d MB)"
              % (thr or "all", dt, size / dt / 1e9, ix.size / dt / 1e6, R, size >> 20))
    # reference-style loop on the first records only
    sample = min(R, 20_000)
    spath = os.path.join(tmp, "sample.bin")
    with open(path, "rb") as f, open(spath, "wb") as g:
        g.write(struct.pack("<4sB3sQ", b"KMER", 1, b"\0\0\0", sample))
        f.seek(16)
        nbytes = sample * (2 + 14 + 4) + 8 * int(ip[sample])
        g.write(f.read(nbytes))
    t0 = time.perf_counter()
    n, st, rows = O.parse_output_bin(spath, L)
    ptr, idx = O.rows_to_csr([sorted(r) for r in rows])
    dt = time.perf_counter() - t0
    print("struct.unpack loop + sort (reference style, 1 core): %d records in %.3f s -> %.2f M index/s"
          % (sample, dt, int(ip[sample]) / dt / 1e6))
    assert np.array_equal(idx, indices[:idx.size])
