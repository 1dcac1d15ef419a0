"""All-pairs cosine k-NN on the GPU.

Mirror of the reference's fedrann/nearest_neighbors.py (NNDescent_ava.get_neighbors :22-55), which
wraps pynndescent.NNDescent(...).neighbor_graph.  NN-descent approximates the exact k-NN graph;
this class returns the exact graph (tiled MFMA distance kernel + top-k, see
csrc/fedrann_hip.hip), so the forest / descent hyper-parameters are accepted and ignored.
Rows come back ascending by (distance, index); a row's self match is a neighbour like any other,
as in `index.neighbor_graph`.
"""
import logging

import numpy as np
import scipy.sparse as sp

from . import _lib

logger = logging.getLogger("fedrann_amd")


class _NearestNeighbors:
    def get_neighbors(self, ref, que, n_neighbors):
        raise NotImplementedError()


class NNDescent_ava(_NearestNeighbors):
    def get_neighbors(self, data, metric="cosine", *, index_n_neighbors=50, n_trees=300,
                      leaf_size=200, n_iters=None, diversify_prob=1, pruning_degree_multiplier=1.5,
                      low_memory=True, n_jobs=64, seed=683985, verbose=True, context=None):
        if metric != "cosine":
            raise ValueError("only metric='cosine' is implemented (the reference's only call, "
                             "__main__.py:186)")
        if sp.issparse(data):
            data = data.toarray()
        data = np.ascontiguousarray(data, dtype=np.float32)  # pynndescent also casts to float32
        if data.ndim != 2:
            raise ValueError("data must be 2-D")
        n, d = data.shape
        k = int(index_n_neighbors)
        if n < k:
            raise ValueError("n_neighbors (%d) must not exceed the number of rows (%d)" % (k, n))
        ctx = context or _lib.default_context()
        if verbose:
            logger.info("exact cosine k-NN on %s: %d rows x %d dims, k = %d (n_trees / leaf_size / "
                        "n_iters are NN-descent parameters and do not apply)",
                        ctx.device_info()["name"], n, d, k)
        nbr_indices, distances = ctx.knn(data, k)
        return nbr_indices, distances
