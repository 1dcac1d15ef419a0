"""overlaps.tsv: the native writer (fdr_overlaps_write) against the reference's route (N x k loop replaced by the
round-1 numpy columns -> DataFrame.to_csv), lines per second.  Host only.
usage: python devtools/bench_writer.py [rows=1000000] [k=20]"""
import io
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fedrann_amd import _lib  # noqa: E402
from fedrann_amd.__main__ import get_output_dataframe, write_overlaps  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(0)
idx = rng.integers(0, n, size=(n, k), dtype=np.int32)
idx[:, 0] = np.arange(n)
dist = np.sort(rng.random((n, k), dtype=np.float32), axis=1)
dist[:, 0] = 0
names = ["read_%07d" % (i // 2) for i in range(n)]
strands = [i % 2 for i in range(n)]
tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    off, buf = _lib.pack_names(names)
    st8 = np.array(strands, np.uint8)
    for threads in (1, 4, 16, 0):
        t0 = time.perf_counter()
        lines = _lib.overlaps_write(os.path.join(tmp, "o.tsv"), idx, dist, off, buf, st8, n_threads=threads)
        dt = time.perf_counter() - t0
        size = os.path.getsize(os.path.join(tmp, "o.tsv"))
        print("native writer, %s threads: %d lines, %.1f MB in %.2f s = %.1f M lines/s"
              % (threads or "all", lines, size / 1e6, dt, lines / dt / 1e6))
    t0 = time.perf_counter()
    write_overlaps(os.path.join(tmp, "o.tsv"), idx, dist, names, strands)
    print("write_overlaps (packs %d Python names first, all threads): %.2f s" % (n, time.perf_counter() - t0))
    m = min(n, 100_000)  # pandas on a tenth (it is linear)
    t0 = time.perf_counter()
    df = get_output_dataframe(idx[:m] % m, dist[:m], names[:m], strands[:m])
    df.to_csv(os.path.join(tmp, "p.tsv"), sep="\t", index=False)
    dt = time.perf_counter() - t0
    print("numpy columns + DataFrame.to_csv: %d lines in %.2f s = %.2f M lines/s" % (df.shape[0], dt, df.shape[0] / dt / 1e6))
finally:
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)
