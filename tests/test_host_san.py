"""The library's plain-C++ parts (output.bin loader, projection table builder, dead-feature filter, launch
planner) built HOST-ONLY with AddressSanitizer + UBSan and driven through tests/host_san/host_san.cpp.
CPU only: sanitizers never run on the GPU build.  Each harness result is compared with what the regular
libfedrann_hip.so returns for the same input (same code, built by hipcc)."""
import ctypes
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from fedrann_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_san", "host_san.cpp")
BIN = os.path.join(HERE, "host_san", "host_san")
FLAGS = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
         "-fno-sanitize-recover=undefined", "-Wall", "-Wno-unused-function", "-pthread"]


@pytest.fixture(scope="module")
def san():
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    csrc = os.path.join(os.path.dirname(HERE), "fedrann_amd", "csrc")
    deps = [SRC] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".inc")]
    if not os.path.exists(BIN) or any(os.path.getmtime(d) > os.path.getmtime(BIN) for d in deps):
        subprocess.run([gxx] + FLAGS + [SRC, "-o", BIN], check=True)

    def run(*args, ok_codes=(0,)):
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
        r = subprocess.run([BIN] + [str(a) for a in args], capture_output=True, text=True, timeout=900, env=env)
        assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
        assert r.returncode in ok_codes, (r.returncode, r.stdout, r.stderr[-2000:])
        return r.stdout.strip()
    return run


def _fields(line):
    return dict(kv.split("=", 1) for kv in line.split() if "=" in kv)


def _wsum(a):
    a = np.ascontiguousarray(a)
    u = a.view({1: np.uint8, 4: np.uint32, 8: np.uint64}[a.dtype.itemsize]).astype(np.uint64)
    with np.errstate(over="ignore"):
        return int((np.arange(1, u.size + 1, dtype=np.uint64) * u).sum(dtype=np.uint64))


def _hdr(magic, ver, n):
    return struct.pack("<4sB3sQ", magic, ver, b"\0\0\0", n)


def _rec(name, idx):
    return struct.pack("<H", len(name)) + name + struct.pack("<I", len(idx)) + struct.pack("<%dQ" % len(idx), *idx)


def test_planner_sweep_under_sanitizers(san):
    """nt in {k, 64, 8191..8193, 2^19-1..2^19+1, 1 M, 10 M, 20 M, FDR_MAX_SEG << 19} x d x k x query shares,
    exact / prefilter / range shapes, 3 CU counts: segment tables well-formed, sizes consistent; one rank's
    plan of BASELINE configs 4 and 5 fits FDR_MAX_SEG segments."""
    f = _fields(san("plan"))
    assert f["rc"] == "0" and int(f["plans"]) > 4000
    assert 1 <= int(f["config4_nseg"]) <= 48 and 1 <= int(f["config5_nseg"]) <= 48


@pytest.mark.parametrize("seed,F,d", [(3, 200_000, 128), (4, 5000, 16), (5, 1_000_003, 500), (6, 64, 4)])
def test_projection_tables_and_compaction_under_sanitizers(san, seed, F, d):
    f = _fields(san("tables", seed, F, d))
    assert f["rc"] == "0"


@pytest.mark.parametrize("threads", [1, 5])
def test_loader_under_sanitizers_matches_library(san, tmp_path, threads):
    rng = np.random.default_rng(5)
    L, R = 40_000, 20_000
    lens = rng.integers(0, 60, size=R)
    lens[::97] = 0
    blob = [_hdr(b"KMER", 1, R)]
    for i, n in enumerate(lens):
        name = b"read_%d_" % i + b"x" * int(rng.integers(0, 30))
        blob.append(_rec(name, rng.choice(2 * L, size=int(n), replace=False).tolist()))
    p = tmp_path / "output.bin"
    p.write_bytes(b"".join(blob))
    f = _fields(san("loader", p, 2 * L, threads))
    indptr, indices, name_off, names = _lib.kmer_output_load(str(p), 2 * L, threads)
    assert f["rc"] == "0" and int(f["R"]) == R and int(f["nnz"]) == int(lens.sum())
    assert [int(x) for x in f["sums"].split(",")] == [_wsum(indptr), _wsum(indices), _wsum(name_off), _wsum(names)]


def test_ranged_loader_under_sanitizers_matches_library(san, tmp_path):
    """fdr_kmer_output_scan_range / _load_range (a rank's row block) with exactly sized arrays: the sanitized build's
    sums equal the library's for a middle block, the first record, an empty block and the whole file; with and
    without names; a block past the end is refused."""
    rng = np.random.default_rng(8)
    L, R = 9000, 5000
    blob = [_hdr(b"KMER", 1, R)]
    for i in range(R):
        n = 0 if i % 89 == 0 else int(rng.integers(1, 80))
        blob.append(_rec(b"r%d" % i, rng.choice(2 * L, size=n, replace=False).tolist()))
    p = tmp_path / "output.bin"
    p.write_bytes(b"".join(blob))
    for lo, hi, names in ((1200, 3456, 1), (0, 1, 1), (77, 77, 1), (0, R, 1), (4000, R, 0)):
        f = _fields(san("loader-range", p, 2 * L, 3, lo, hi, names))
        total, ip, ix, noff, nbuf = _lib.kmer_output_load_range(str(p), 2 * L, lo, hi, n_threads=3, with_names=bool(names))
        assert f["rc"] == "0" and int(f["R"]) == R == total
        want = [_wsum(ip), _wsum(ix), _wsum(noff) if names else 0, _wsum(nbuf) if names else 0]
        assert [int(x) for x in f["sums"].split(",")] == want
    assert int(_fields(san("loader-range", p, 2 * L, 1, 10, R + 1, 1))["rc"]) == -1


def _fnv(chunks):
    h = 0
    for c in chunks:
        for b in c:
            h = (h * 1099511628211 + b) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.parametrize("kind", ["fasta", "fastq"])
def test_reads_parser_and_appender_under_sanitizers(san, tmp_path, kind):
    """fdr_reads_scan / fdr_reads_parse / fdr_kmer_output_append on a messy FASTA / FASTQ file streamed in pieces
    of 1 byte to 1 MB with exactly sized arrays: no sanitizer report, and the records are those of the numpy
    statement of the reader (names, sequences) whatever the piece size."""
    from fedrann_amd import kmer_search as ks
    from test_kmer_host import _mixed_fasta, _mixed_fastq
    rng = np.random.default_rng(12)
    raw = (_mixed_fasta if kind == "fasta" else _mixed_fastq)(rng, 200)
    p = tmp_path / ("reads." + kind)
    p.write_bytes(raw)
    for flag in (0, 1):
        ids, seqs, off = ks._parse_records(raw, kind == "fastq", bool(flag))
        want_ids = _fnv(i + b"\xff" for i in ids)
        want_seq = _fnv([seqs.tobytes()])
        for chunk in (1, 7, 300, 1 << 20):
            out = tmp_path / "out.bin"
            f = _fields(san("reads", p, chunk, flag, out))
            assert f["rc"] == "0" and int(f["R"]) == len(ids) and int(f["bases"]) == seqs.size
            assert int(f["seq"]) == want_seq and int(f["ids"]) == want_ids
            # the appended records: one index per record = its sequence length
            blob = out.read_bytes()
            pos, lens = 16, []
            while pos < len(blob):
                nb = struct.unpack_from("<H", blob, pos)[0]
                cnt = struct.unpack_from("<I", blob, pos + 2 + nb)[0]
                lens.append(struct.unpack_from("<Q", blob, pos + 6 + nb)[0])
                assert cnt == 1
                pos += 6 + nb + 8
            assert lens == np.diff(off).tolist()


def test_loader_errors_under_sanitizers_match_library(san, tmp_path):
    """Every malformed file of tests/test_host.py::test_native_output_bin_loader_errors: same return code
    from the sanitized build as from the library, and no out-of-bounds access on the way."""
    L = _lib.load_library()
    cases = [b"KME", _hdr(b"XXXX", 1, 0), _hdr(b"KMER", 2, 0),
             _hdr(b"KMER", 1, 1) + struct.pack("<H", 2) + b"ab" + struct.pack("<I", 3) + struct.pack("<2Q", 1, 2),
             _hdr(b"KMER", 1, 2) + _rec(b"a", [1, 2]), _hdr(b"KMER", 1, 1) + _rec(b"a", [1, 10]),
             _hdr(b"KMER", 1, 1) + _rec(b"a", [3, 4, 3]), _hdr(b"KMER", 1, 1 << 40), _hdr(b"KMER", 1, 0), b""]
    p = tmp_path / "x.bin"
    for blob in cases:
        p.write_bytes(blob)
        got = int(_fields(san("loader", p, 10, 2))["rc"])
        R, nnz, nb = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        rc = L.fdr_kmer_output_scan(os.fsencode(str(p)), ctypes.byref(R), ctypes.byref(nnz), ctypes.byref(nb))
        if rc == 0:
            ip = np.zeros(2 * R.value + 1, np.int64)
            ix = np.zeros(max(2 * nnz.value, 1), np.int32)
            no = np.zeros(R.value + 1, np.int64)
            nm = np.zeros(max(nb.value, 1), np.uint8)
            rc = L.fdr_kmer_output_load(os.fsencode(str(p)), 10, 2, R.value, nnz.value, nb.value, ip.ctypes.data,
                                        ix.ctypes.data, no.ctypes.data, nm.ctypes.data)
        assert got == rc, blob[:24]
    assert int(_fields(san("loader", tmp_path / "missing.bin", 10, 1))["rc"]) == -6  # FDR_E_IO
    p.write_bytes(_hdr(b"KMER", 1, 1) + _rec(b"a", [1]))
    assert int(_fields(san("loader", p, 9, 1))["rc"]) == -1  # odd n_features
    # the file changed between scan and load (capacities no longer match): refused before any write
    assert int(_fields(san("loader-stale", p, 10))["rc"]) == -5  # FDR_E_STATE


def test_overlaps_writer_under_sanitizers(san, tmp_path):
    """A 3000-row graph with -1 fillers, self hits and inf, written whole and as two appended row blocks (1 and
    5 threads): same bytes; an out-of-range neighbour index is refused."""
    for threads in (1, 5):
        out = tmp_path / ("o%d.tsv" % threads)
        f = _fields(san("overlaps", out, threads))
        assert f["rc"] == "0" and f["lines"] == f["parts"] and f["bad_rc"] == "-1"
        assert out.read_bytes() == (tmp_path / ("o%d.tsv.parts" % threads)).read_bytes()
        assert out.read_bytes().count(b"\n") == int(f["lines"]) + 1
    assert (tmp_path / "o1.tsv").read_bytes() == (tmp_path / "o5.tsv").read_bytes()
