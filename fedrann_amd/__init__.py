"""fedrann_amd -- MI355X (gfx950) implementation of FEDRANN's dimensionality-reduction + k-NN hot path.

Host side (Python) mirrors the three calls the reference makes from
fedrann/__main__.py:run_fedrann_pipeline:

    get_precompute_matrix   (reference precompute.py:58-115)
    get_feature_matrix      (reference feature_extraction.py:216-292)
    NNDescent_ava.get_neighbors / get_neighbors_ava  (reference nearest_neighbors.py:22-55)

and drives hand-written HIP kernels through the C-ABI in include/fedrann_hip.h
(libfedrann_hip.so, loaded with ctypes by fedrann_amd._lib).  There is no CPU fallback: without
the built library and a GPU every compute entry point raises.
"""
__version__ = "0.1.0"
__description__ = ("fedrann-amd: FEDRANN's sparse random projection + all-pairs cosine k-NN on "
                   "AMD MI355X (gfx950)")
