"""Randomised cross-check of the k-NN code paths (a short, seeded version of devtools/soak.py): for random
shapes and data -- sparse rows, duplicate classes, near-ties, exact ties, all-zero rows -- the prefilter mode
(with and without the duplicate-row layer), the exact mode and, for the smaller cases, the oracle agree bit
for bit.  Sizes reach the synchronised rounds on two queues (> 131 k rows at d <= 128)."""
import numpy as np
import pytest

from fedrann_amd import _lib

pytestmark = pytest.mark.gpu


def _case(rng):
    n = int(rng.choice([9000, 17000, 33000, 70000, 150000]))
    d = int(rng.choice([64, 128, 128, 200, 256, 500]))
    k = int(rng.choice([5, 20, 20, 33, 50, 56, 64]))
    kind = int(rng.integers(0, 4))
    if kind == 0:    # sparse rows like real embeddings
        E = rng.standard_normal((n, d)).astype(np.float32)
        E[rng.random(E.shape) < rng.choice([0.8, 0.93, 0.97])] = 0
    elif kind == 1:  # duplicate classes
        u = rng.standard_normal((int(rng.integers(40, 3000)), d)).astype(np.float32)
        u[rng.random(u.shape) < 0.9] = 0
        E = u[rng.integers(0, u.shape[0], size=n)]
    elif kind == 2:  # tight clusters: near-ties everywhere
        base = rng.standard_normal((int(rng.integers(3, 60)), d)).astype(np.float32)
        E = base[rng.integers(0, base.shape[0], size=n)] + 1e-4 * rng.standard_normal((n, d)).astype(np.float32)
    else:            # few distinct magnitudes: exact distance ties
        E = rng.integers(-1, 2, size=(n, d)).astype(np.float32)
        E[rng.random(E.shape) < 0.9] = 0
    return E, k, kind


@pytest.mark.parametrize("seed", [7, 11])
def test_random_cases_all_modes_agree(oracle, seed):
    rng = np.random.default_rng(seed)
    ctx = _lib.Context(0)
    try:
        for case in range(6):
            E, k, kind = _case(rng)
            ctx.set_dedup_mode(str(rng.choice(["auto", "off", "force"])))
            ctx.set_knn_mode("prefilter")
            pi, pd = ctx.knn(E, k)
            ctx.set_knn_mode("exact")
            xi, xd = ctx.knn(E, k)
            what = "seed %d case %d: n=%d d=%d k=%d kind=%d" % (seed, case, E.shape[0], E.shape[1], k, kind)
            assert np.array_equal(pi, xi), what
            assert np.array_equal(pd.view(np.uint32), xd.view(np.uint32)), what
            if E.shape[0] <= 17000:
                wi, wd = oracle.knn(E, k)
                assert np.array_equal(xi, wi) and np.array_equal(xd.view(np.uint32), wd.view(np.uint32)), what
    finally:
        ctx.close()


@pytest.mark.parametrize("dedup", ["auto", "off"])
def test_ordered_scan_with_exact_ties(dedup):
    """The candidate pass scans > 131 k rows in chunk-mask order and keeps, among candidates at EQUAL approximate
    distance, the ones it met first -- not the ones with the smallest row numbers.  Rows of a few distinct
    magnitudes (plateaus of hundreds of exact ties around the k-th place) must still come out in (distance, row)
    order: the certificate, the merge of the segment lists (whose bound once assumed row order inside a list) and
    the range pass see to that."""
    rng = np.random.default_rng(1234)
    n, d, k = 300_000, 64, 20
    E = rng.integers(-1, 2, size=(n, d)).astype(np.float32)
    E[rng.random(E.shape) < 0.9] = 0
    ctx = _lib.Context(0)
    try:
        ctx.set_dedup_mode(dedup)
        ctx.set_knn_mode("prefilter")
        pi, pd = ctx.knn(E, k)
        launches, queues = ctx.last_prefilter_launches()
        assert launches > 1  # synchronised rounds: the sizes at which the scan is ordered
        ctx.set_knn_mode("exact")
        xi, xd = ctx.knn(E, k)
    finally:
        ctx.close()
    assert np.array_equal(pi, xi)
    assert np.array_equal(pd.view(np.uint32), xd.view(np.uint32))
