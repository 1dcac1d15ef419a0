"""PCIe probe: host -> device rate for a CSR-sized buffer (pinned by hipHostMalloc vs hipHostRegister, 1 / 2 / 4
concurrent copies), device -> host for a result-sized buffer, and the host-side fdr_csr_compact rate.
usage: python devtools/h2d_probe.py [reads=1000000]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fedrann_amd import _lib  # noqa: E402
from fedrann_amd.precompute import build_precompute_matrix  # noqa: E402
from fedrann_amd.synth import synth  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", 0)
s = synth(R, seed=602)
ix = s["indices"]
nbytes = ix.nbytes
print("indices: %.1f MB" % (nbytes / 1e6))
dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)


def rate(fn, reps=5):
    fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize(dev)
    return nbytes * reps / (time.perf_counter() - t0) / 1e9


src_pin = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
src_pin.numpy()[:] = ix.view(np.uint8)
for nstream in (1, 2, 4, 8):
    streams = [torch.cuda.Stream(dev) for _ in range(nstream)]
    cut = [nbytes * i // nstream // 4096 * 4096 for i in range(nstream)] + [nbytes]

    def go():
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                dst[cut[i]:cut[i + 1]].copy_(src_pin[cut[i]:cut[i + 1]], non_blocking=True)
    print("hipHostMalloc pinned, %d stream(s): %.1f GB/s" % (nstream, rate(go)))
ctx = _lib.Context(0)
ctx.host_register(ix)
src_reg = torch.from_numpy(ix.view(np.uint8))
for nstream in (1, 4):
    streams = [torch.cuda.Stream(dev) for _ in range(nstream)]
    cut = [nbytes * i // nstream // 4096 * 4096 for i in range(nstream)] + [nbytes]

    def go():
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                dst[cut[i]:cut[i + 1]].copy_(src_reg[cut[i]:cut[i + 1]], non_blocking=True)
    print("hipHostRegister'd numpy, %d stream(s): %.1f GB/s" % (nstream, rate(go)))
ctx.host_unregister(ix)
print("pageable numpy: %.1f GB/s" % rate(lambda: dst.copy_(src_reg), reps=2))
# device -> host, result sized (R x 20 x 8 bytes)
res = torch.empty(R * 160, dtype=torch.uint8, device=dev)
hres = torch.empty(R * 160, dtype=torch.uint8, pin_memory=True)
nb0, nbytes = nbytes, res.numel()
print("D2H %.0f MB pinned: %.1f GB/s" % (nbytes / 1e6, rate(lambda: hres.copy_(res, non_blocking=True))))
nbytes = nb0
# host-side compaction
P = build_precompute_matrix(s["counts"], 128)
ctx.projection_load(P.indptr, P.indices, P.data, s["n_features"], 128)
for thr in (1, 4, 8, 16, 32, 0):
    t0 = time.perf_counter()
    for _ in range(3):
        cip, cix = ctx.csr_compact(s["indptr"], ix, n_threads=thr)
    dt = (time.perf_counter() - t0) / 3
    print("fdr_csr_compact threads=%d: %.1f ms (%.1f GB/s of ids), %d of %d ids survive" % (thr, dt * 1e3, nbytes / dt / 1e9, cix.size, ix.size))
print(open("/proc/cpuinfo").read().split("flags")[1].split("\n")[0][:1500].count("avx512"), "avx512 flag groups; model:",
      [l for l in open("/proc/cpuinfo") if "model name" in l][0].strip())
