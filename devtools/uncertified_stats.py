"""Development: why does the prefilter certificate fail?  Emulates it on the host (numpy fp16 rows,
fp32 accumulation) for a sample of queries of a synthetic read set and classifies the failures."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fedrann_amd.synth import synth
from fedrann_amd.precompute import build_precompute_matrix
from oracle import oracle as O
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d, k, extra, M = 128, 20, 8, 2 * 0.00105 + 4e-7
s = synth(R, seed=602)
P = build_precompute_matrix(s["counts"], d)
E = O.embed(s["indptr"], s["indices"], (P.indptr, P.indices, P.data), s["n_features"], d)
Eh, _, zero = O.normalize(E)
H = Eh.astype(np.float16).astype(np.float32)
rng = np.random.default_rng(0)
qs = rng.choice(R, size=1500, replace=False)
kp = k + extra
cls = {"ok": 0, "zero_row": 0, "too_few_positive": 0, "plateau": 0}
widths, nnzs = [], []
t0 = time.time()
for q0 in range(0, len(qs), 100):
    qq = qs[q0:q0 + 100]
    S = H[qq] @ H.T
    Dt = np.clip(1 - S, 0, 1).astype(np.float32)
    part = np.partition(Dt, kp + 200, axis=1)[:, :kp + 201]
    part.sort(axis=1)
    for i, q in enumerate(qq):
        dK, dKP = part[i, k - 1], part[i, kp - 1]
        if zero[q]:
            cls["zero_row"] += 1
        elif not (dK + M < 1):
            cls["too_few_positive"] += 1
        elif not (dK + M < dKP):
            cls["plateau"] += 1
            widths.append(int((part[i] <= dK + M).sum()))
            nnzs.append(int((Eh[q] != 0).sum()))
        else:
            cls["ok"] += 1
print("reads", R, "sampled", len(qs), cls, "time %.0fs" % (time.time() - t0))
if widths:
    w = np.array(widths)
    print("plateau: candidates within margin of d~(K): median %d, p90 %d, >200: %d of %d" %
          (np.median(w), np.percentile(w, 90), (w > 200).sum(), len(w)))
    print("non-zero components of plateau queries: ", np.bincount(np.array(nnzs))[:12])
print("non-zero components of all rows:       ", np.bincount((Eh != 0).sum(1))[:12] / R)
