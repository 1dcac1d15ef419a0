"""Development: the pipelined host upload (host_upload.inc) against the oracle with the library FEDRANN_HIP_LIB names --
tests/test_gpu_parity.py::test_embed_host_upload_in_chunks_raw_and_compacted called directly (no conftest rebuild)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fedrann_amd import _lib  # noqa: E402
from oracle import oracle  # noqa: E402
import test_gpu_parity as T  # noqa: E402

oracle.lib()
ctx = _lib.Context(0)
T.test_embed_host_upload_in_chunks_raw_and_compacted.__wrapped__(ctx, oracle) if hasattr(
    T.test_embed_host_upload_in_chunks_raw_and_compacted, "__wrapped__") else T.test_embed_host_upload_in_chunks_raw_and_compacted(ctx, oracle)
print("UPLOAD OK")
