"""Development: parity of the d = 256 / k = 50 and d = 500 / k = 50 prefilter shapes against the CPU oracle with the library
FEDRANN_HIP_LIB names (a development build may hold only some shapes, so the whole pytest suite does not apply):
the checks of tests/test_gpu_configs.py, called directly.  usage: FEDRANN_HIP_LIB=... python devtools/check_shapes.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fedrann_amd import _lib  # noqa: E402
from oracle import oracle  # noqa: E402
import test_gpu_configs as T  # noqa: E402

oracle.lib()
ctx = _lib.Context(0)
ctx.set_knn_mode("auto")
ctx.set_dedup_mode("auto")
if os.environ.get("CHECK_D128") == "only":
    os.environ["CHECK_D128"] = "1"
    E = T._device_embeddings(400_000, 128, nnz=6, loci=150_000, seed=7)
    print("d128 k20 400k all pairs:", T._check_rank_share(ctx, oracle, E, 400_000, 20, sample=192), ctx.last_prefilter_launches(), flush=True)
    print("d128 k20 130k x 400k:", T._check_rank_share(ctx, oracle, E, 130_017, 20, sample=128), ctx.last_prefilter_launches(), flush=True)
    print("CHECK D128 OK")
    sys.exit(0)
E = T._device_embeddings(1_000_000, 256, nnz=8, loci=300_000, seed=5, doubling=True)
print("d256 k50 125k x 1M:", T._check_rank_share(ctx, oracle, E, 125_000, 50), ctx.last_prefilter_launches(), flush=True)
print("d256 k50 200k all pairs:", T._check_rank_share(ctx, oracle, E[:200_000].contiguous(), 200_000, 50, sample=128),
      ctx.last_prefilter_launches(), flush=True)
del E
E = T._device_embeddings(600_000, 500, nnz=8, loci=250_000, seed=6)
print("d500 k50 150k x 600k:", T._check_rank_share(ctx, oracle, E, 150_000, 50, sample=128), ctx.last_prefilter_launches(), flush=True)
print("CHECK OK")
if os.environ.get("CHECK_D128") == "1":  # (the d <= 128 shapes: a development library that holds them)
    E = T._device_embeddings(400_000, 128, nnz=6, loci=150_000, seed=7)
    print("d128 k20 400k all pairs:", T._check_rank_share(ctx, oracle, E, 400_000, 20, sample=192), ctx.last_prefilter_launches(), flush=True)
    print("d128 k20 130k x 400k:", T._check_rank_share(ctx, oracle, E, 130_017, 20, sample=128), ctx.last_prefilter_launches(), flush=True)
    print("CHECK D128 OK")
