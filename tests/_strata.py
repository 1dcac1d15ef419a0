"""Oracle samples stratified by execution path.

The library reports, per query row of its last k-NN call, which way the row took to its result
(fdr_last_query_paths: certified fp16 candidates, range pass over a plateau, exact kernel, all-zero closed form,
expanded from a duplicate-row class).  All ways return the same canonical bits -- which is exactly what a sample drawn
uniformly over the rows barely tests for the rare ways (17 expected range-pass rows of 211 at config 4, 0.6 at config 5).
So the sampled oracle rows are drawn PER PATH: up to `per` rows from every non-empty stratum, plus the block's edges."""
import numpy as np

from fedrann_amd import _lib

STRATA = ("certified", "range", "exact", "zero", "class_member", "range_in_class", "edges")


def strata_masks(paths):
    code = paths & 0x7F
    member = (paths & _lib.PATH_CLASS_MEMBER) != 0
    edges = np.zeros(paths.size, dtype=bool)
    edges[:2] = True
    edges[-2:] = True
    return {
        "certified": code == _lib.PATH_CERTIFIED,
        "range": code == _lib.PATH_RANGE,
        "exact": (code == _lib.PATH_EXACT) | (code == _lib.PATH_RANGE_OVERFLOW) | (code == _lib.PATH_GENERIC),
        "zero": code == _lib.PATH_ZERO,
        "class_member": member,
        "range_in_class": member & (code == _lib.PATH_RANGE),  # (a plateau query whose result was expanded from its class)
        "edges": edges,
    }


def stratified_rows(paths, per=64, seed=1):
    """(sorted unique row numbers, {stratum: rows in it}, {stratum: rows sampled from it})."""
    rng = np.random.default_rng(seed)
    masks = strata_masks(paths)
    code = paths & 0x7F
    assert np.all((code >= 1) & (code <= 6)), "a finished query row without a path code"
    rows, counts, taken = [], {}, {}
    for name in STRATA:
        ids = np.flatnonzero(masks[name])
        counts[name] = int(ids.size)
        pick = ids if ids.size <= per else rng.choice(ids, size=per, replace=False)
        taken[name] = int(pick.size)
        rows.append(pick)
    # the exclusive codes partition the rows
    assert counts["certified"] + counts["range"] + counts["exact"] + counts["zero"] == paths.size
    return np.unique(np.concatenate(rows)).astype(np.int64), counts, taken
