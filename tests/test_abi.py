"""The C-ABI library: builds for gfx950, loads, exports every symbol of include/fedrann_hip.h, and
fails loudly without a GPU (no compute calls here)."""
import os
import re

import pytest

from conftest import ROOT
from fedrann_amd import _lib, build


@pytest.fixture(scope="module")
def lib():
    build.build_library()
    return _lib.load_library()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "fedrann_hip.h")).read()
    declared = set(re.findall(r"\b(fdr_[a-z_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_constants_match_header():
    hdr = open(os.path.join(ROOT, "include", "fedrann_hip.h")).read()
    assert int(re.search(r"#define FDR_MAX_K (\d+)", hdr).group(1)) == _lib.FDR_MAX_K
    assert int(re.search(r"#define FDR_MAX_DIM (\d+)", hdr).group(1)) == _lib.FDR_MAX_DIM


def test_padded_dim_needs_no_gpu(lib):
    assert lib.fdr_padded_dim(1) == 128 and lib.fdr_padded_dim(128) == 128
    assert lib.fdr_padded_dim(129) == 256 and lib.fdr_padded_dim(256) == 256
    assert lib.fdr_padded_dim(257) == 512 and lib.fdr_padded_dim(500) == 512
    assert lib.fdr_padded_dim(513) == 1024 and lib.fdr_padded_dim(1025) == 2048  # (the generic kernel's sizes)
    assert lib.fdr_padded_dim(2049) < 0 and lib.fdr_padded_dim(0) < 0


def test_no_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.FedrannHipError) as e:
        _lib.Context(0)
    assert "fdr_create" in str(e.value)


def test_release_library_has_no_development_knobs():
    """The shipped .so reads one environment variable (FDR_KNN_MODE, once, in fdr_create).  The timing /
    wrong-result knobs of the development build (FDR_KNN_DEBUG, _EXTRA, _SLOTS, _OV, _NSEG, _SHAPE, _PAIR,
    _RANGE, _DEDUP) must not exist in it: neither the names nor, hence, the getenv calls."""
    build.build_library()
    blob = open(_lib.LIB_PATH, "rb").read()
    names = set(re.findall(rb"FDR_[A-Z0-9_]*KNN[A-Z0-9_]*", blob))
    assert names == {b"FDR_KNN_MODE"}, names


def test_product_never_imports_oracle():
    """No file of the product may import, load or link the CPU oracle (comments may mention it)."""
    pat = re.compile(r"^\s*(from\s+oracle|import\s+oracle)|libfedrann_oracle|orc_[a-z_]+\s*\(|oracle[/.]oracle",
                     re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "fedrann_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".inc", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not pat.search(src), os.path.join(dirpath, f)
