"""Wall time of `python -m fedrann_amd -i reads.fasta -o out` by stage, on reads cut from a random genome:
    PYTHONPATH=. python devtools/bench_from_reads.py [n_reads] [mean_len] [genome_len]
Prints the time between the pipeline's stage banners (stage 1 = counting + library + search, all streamed)."""
import logging
import os
import sys
import tempfile
import time

import numpy as np

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000
mean_len = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
genome_len = int(sys.argv[3]) if len(sys.argv) > 3 else 30_000_000

rng = np.random.default_rng(5)
alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
comp = np.zeros(256, dtype=np.uint8)
comp[list(b"ACGT")] = list(b"TGCA")
genome = alpha[rng.integers(0, 4, size=genome_len)]
lens = np.clip(rng.lognormal(np.log(mean_len) - 0.125, 0.5, size=n_reads), 500, genome_len).astype(np.int64)
starts = rng.integers(0, genome_len - lens + 1)
tmp = tempfile.mkdtemp(prefix="fdr_reads_")
fa = os.path.join(tmp, "reads.fasta")
t0 = time.time()
with open(fa, "wb", buffering=1 << 24) as f:
    for i, (s, n) in enumerate(zip(starts.tolist(), lens.tolist())):
        r = genome[s:s + n].copy()
        if i & 1:
            r = comp[r[::-1]]
        e = np.flatnonzero(rng.random(n) < 0.06)
        r[e] = alpha[rng.integers(0, 4, size=e.size)]
        f.write(b">read_%07d\n" % i)
        f.write(r.tobytes())
        f.write(b"\n")
print("reads: %d, %.1f MB FASTA written in %.1f s" % (n_reads, os.path.getsize(fa) / 1e6, time.time() - t0), flush=True)

from fedrann_amd import __main__ as cli  # noqa: E402

marks = []


class Marks(logging.Handler):
    def emit(self, record):
        msg = record.getMessage()
        if msg.startswith("---") or msg.startswith("Pipeline completed"):
            marks.append((time.time(), msg))


cli.logger.addHandler(Marks())
t0 = time.time()
cli.main(["-i", fa, "-o", os.path.join(tmp, "out")] + sys.argv[4:])
t1 = time.time()
for (ta, ma), (tb, _) in zip(marks, marks[1:]):
    print("%8.2f s  %s" % (tb - ta, ma))
print("%8.2f s  total (%.1f MB/s of FASTA)" % (t1 - t0, os.path.getsize(fa) / 1e6 / (t1 - t0)))
