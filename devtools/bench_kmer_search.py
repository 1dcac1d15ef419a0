"""k-mer search: fdr_kmer_search (GPU, host buffers in / CSR out) against the oracle's single-thread
restatement of kmer_searcher.cpp on a sample.  usage: python devtools/bench_kmer_search.py [reads] [mean_len]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fedrann_amd import _lib, kmer_search as ks  # noqa: E402
from fedrann_amd.synth import synth_sequences  # noqa: E402
from oracle import oracle as O  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
mean_len = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
k = 15
s = synth_sequences(R, genome_len=5_000_000, mean_len=mean_len, k=k, sample=0.05, seed=602)
codes = ks.load_kmer_library([b"\n".join(s["fwd"]) + b"\n", b"\n".join(s["rev"]) + b"\n"], k)
seqs, off = s["seqs"], s["seq_off"]
ctx = _lib.Context(0)
ctx.kmer_search(seqs, off, codes, k)  # warm: allocations
ctx.timing(True)
t0 = time.perf_counter()
ip, ix = ctx.kmer_search(seqs, off, codes, k)
dt = time.perf_counter() - t0
ms = {name: ctx.timing_read(i)[1] for i, name in enumerate(_lib.KERNELS) if name.startswith("kmer")}
print("GPU: %d reads, %.1f Mbases, library %d k-mers, %d index entries" % (R, seqs.size / 1e6, codes.size, ix.size))
print("  end to end (host in, host out): %.1f ms = %.2f Gbases/s" % (dt * 1e3, seqs.size / dt / 1e9))
print("  device spans: search (table build + passes) %.2f ms = %.1f GB/s of sequence; sort+compact %.2f ms"
      % (ms["kmer_search"], seqs.size / ms["kmer_search"] / 1e6, ms["kmer_compact"]))
ctx.kmer_count(seqs, off, k, 2)  # warm
ctx.timing_read(7), ctx.timing_read(8)
t0 = time.perf_counter()
cc, cn = ctx.kmer_count(seqs, off, k, 2)
dt = time.perf_counter() - t0
ms = {name: ctx.timing_read(i)[1] for i, name in enumerate(_lib.KERNELS) if name.startswith("kmer")}
print("canonical k-mer counting (jellyfish count -C | dump -L 2): %d k-mers kept; end to end %.1f ms; device: "
      "codes %.2f ms, sort + run lengths + compaction %.2f ms" % (cc.size, dt * 1e3, ms["kmer_search"], ms["kmer_compact"]))
n = min(R, 300)
reads = [bytes(seqs[off[i]:off[i + 1]]) for i in range(n)]
t0 = time.perf_counter()
wp, wx = O.kmer_search(reads, codes, k)
dt = time.perf_counter() - t0
nb = int(off[n])
print("oracle (1 thread, restated kmer_searcher.cpp): %d reads, %.1f Mbases in %.2f s = %.4f Gbases/s"
      % (n, nb / 1e6, dt, nb / dt / 1e9))
assert np.array_equal(ix[:wx.size], wx)
