import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """libfedrann_hip.so is a build product (git-ignored): compile it if this checkout has none or an
    outdated one (hipcc cross-compiles gfx950 without a GPU; ~2 minutes the first time)."""
    from fedrann_amd import build
    build.build_library()


def golden(name):
    return os.path.join(GOLDEN, name)


def golden_embed_case(tag):
    """Inputs + expected E of tests/golden/embed_<tag>.npz as (indptr, indices, P(csr arrays), F, d, E_bits)."""
    g = np.load(golden("embed_%s.npz" % tag))
    pg = np.load(golden("precompute_%s.npz" % tag))
    L, d = int(g["L"]), int(g["d"])
    lens, idx = g["read_lens"], g["read_idx"]
    rows, o = [], 0
    for n in lens:
        r = idx[o:o + n]
        o += n
        rows.append(r)
        rows.append(np.where(r < L, r + L, r - L))
    indptr = np.zeros(len(rows) + 1, dtype=np.int64)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    indices = np.concatenate(rows).astype(np.int64)
    P = (pg["indptr"], pg["indices"], pg["data_bits"].view(np.float32))
    return indptr, indices, P, 2 * L, d, g["E_bits"]


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def ctx():
    """One GPU context for the whole session (GPU tests only)."""
    from fedrann_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()
