"""One rank's share of BASELINE config 4 or 5 at full size from the real generator, on one GPU (the measured
line committed under profiles/rN_config{4,5}_rank_share.json; tests/test_gpu_configs.py runs the same code).
usage: python devtools/rank_share_real.py 4|5 [reads=10000000] [ranks=8] [reps=2]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fedrann_amd import _lib  # noqa: E402
from oracle import oracle  # noqa: E402
from _rank_share import run_rank_share  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
R = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
G = int(sys.argv[3]) if len(sys.argv) > 3 else 8
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
oracle.lib()
ctx = _lib.Context(0)
kw = dict(d=128, k=20, doubling=False) if cfg == 4 else dict(d=256, k=50, doubling=True, sample=64)
info = run_rank_share(ctx, oracle, R=R, ranks=G, reps=reps, unique_split=True,
                      log=lambda *a: print("[config%d]" % cfg, *a, file=sys.stderr, flush=True), **kw)
info["device"] = ctx.device_info()
print(json.dumps(info))
