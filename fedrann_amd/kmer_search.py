"""reads x k-mer library -> per-read sets of library indices, on the GPU.

Host-side mirror of the reference's native tool `kmer_searcher` (kmer_searcher/kmer_searcher.cpp) as
fedrann/count_kmers.py:131-139 drives it:

    cat fwd_kmer_library.fasta rev_kmer_library.fasta | grep -v '^>' | kmer_searcher /dev/stdin reads.fasta OUT k threads

    load_kmer_library   kmer_searcher.cpp:262-279  (tokens of length k, first occurrence wins)
    read_sequences      kmer_searcher.cpp:153-200  (FASTA / FASTQ reader)
    kmer_searcher       kmer_searcher.cpp:232-375  (the search: fdr_kmer_search on the GPU)
    write_output_bin    kmer_searcher.cpp:98-130   (output.bin) and :203-230 (kmer_frequency.bin)

The search itself has no CPU path: without libfedrann_hip.so and a GPU it raises.
"""
import os
import struct

import numpy as np

from . import _lib

_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate("ACGT"):
    _CODE[ord(_c)] = _i
    _CODE[ord(_c.lower())] = _i
_WS = np.zeros(256, dtype=bool)
_WS[[9, 10, 11, 12, 13, 32]] = True  # what `istream >> std::string` skips


def load_kmer_library(texts, k):
    """Library text(s) -> uint64 codes of the unique valid k-mers, in index order.

    `texts`: bytes or a list of bytes (concatenated as `cat` would).  Tokens are separated by white
    space; a token whose length is not k (e.g. a '>count' header), a token with a character outside
    ACGTacgt and a k-mer seen before are skipped (kmer_searcher.cpp:268-277).  Note that the reverse
    library repeats every palindromic k-mer of the forward one; those lose their slot, as in the
    reference."""
    if not 1 <= int(k) <= 31:
        raise ValueError("Invalid k value: %r" % (k,))  # kmer_searcher.cpp:246-249
    if isinstance(texts, (bytes, bytearray, memoryview)):
        texts = [texts]
    data = np.frombuffer(b"".join(bytes(t) for t in texts), dtype=np.uint8)
    if data.size == 0:
        return np.zeros(0, dtype=np.uint64)
    ws = _WS[data]
    start = np.flatnonzero(~ws & np.concatenate(([True], ws[:-1])))
    end = np.flatnonzero(~ws & np.concatenate((ws[1:], [True]))) + 1
    start = start[end - start == k]
    codes = np.zeros(start.size, dtype=np.uint64)
    valid = np.ones(start.size, dtype=bool)
    for j in range(k):  # k <= 31 passes over the token starts
        c = _CODE[data[start + j]]
        valid &= c != 255
        codes = (codes << np.uint64(2)) | (c & 3).astype(np.uint64)
    return unique_first(codes[valid])


def unique_first(codes):
    """The distinct codes in order of first occurrence (a k-mer seen before loses: kmer_searcher.cpp:274-277)."""
    codes = np.asarray(codes, dtype=np.uint64)
    _, first = np.unique(codes, return_index=True)
    first.sort()
    return np.ascontiguousarray(codes[first])


def fasta_id(header):
    """The part of a header line kmer_searcher keeps for a FASTA record: up to the first space or tab."""
    cut = len(header)
    for sep in (b" ", b"\t"):
        p = header.find(sep)
        if 0 <= p < cut:
            cut = p
    return header[:cut]


def _parse_records(raw, is_fastq, fastq_ids_as_fasta):
    """Whole records in `raw` (bytes) -> (ids, seqs uint8, seq_off int64 [R+1]); see read_sequences.  The numpy
    statement of the reader's rules: the tests hold the native reader (fdr_reads_parse) against it."""
    data = np.frombuffer(raw, dtype=np.uint8)
    if data.size == 0:
        return [], np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.int64)
    nl = np.flatnonzero(data == 10)
    starts = np.concatenate(([0], nl + 1))
    ends = np.concatenate((nl, [data.size]))  # (exclusive, without the '\n')
    if starts[-1] == data.size:               # the piece ends with '\n': no further line
        starts, ends = starts[:-1], ends[:-1]
    lens = ends - starts
    first = data[np.minimum(starts, data.size - 1)]
    if is_fastq:
        ids, pieces, i, n = [], [], 0, starts.size
        while i < n:
            if lens[i] > 0 and first[i] == ord("@"):
                name = raw[starts[i] + 1:ends[i]]
                if fastq_ids_as_fasta:
                    name = fasta_id(name)
                if not (fastq_ids_as_fasta and len(name) == 0):
                    ids.append(name)
                    pieces.append(raw[starts[i + 1]:ends[i + 1]] if i + 1 < n else b"")
                i += 4
            else:
                i += 1
        off = np.zeros(len(ids) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(p) for p in pieces])
        return ids, np.frombuffer(b"".join(pieces), dtype=np.uint8), off
    # ---- FASTA ----
    nonempty = lens > 0
    is_head = nonempty & (first == ord(">"))
    rec_of_line = np.cumsum(is_head) - 1  # -1: before the first header
    heads = np.flatnonzero(is_head)
    ids = []
    for h in heads.tolist():
        ids.append(fasta_id(raw[starts[h] + 1:ends[h]]))
    has_id = np.array([len(x) > 0 for x in ids], dtype=bool)
    keep_line = nonempty & ~is_head & (rec_of_line >= 0)
    if has_id.size:
        keep_line &= has_id[np.maximum(rec_of_line, 0)]
    # bases = the bytes of the kept lines
    span = np.diff(np.concatenate((starts, [data.size])))  # line length incl. its '\n'
    mask = np.repeat(keep_line, span)
    mask[nl] = False
    seqs = np.ascontiguousarray(data[mask])
    seq_len = np.bincount(rec_of_line[keep_line], weights=lens[keep_line], minlength=len(ids)).astype(np.int64)
    seq_len = seq_len[has_id] if has_id.size else seq_len
    off = np.zeros(seq_len.size + 1, dtype=np.int64)
    np.cumsum(seq_len, out=off[1:])
    return [x for x, ok in zip(ids, has_id) if ok], seqs, off


def _complete_prefix(raw, is_fastq, eof):
    """Bytes of `raw` that hold whole records only (what follows waits for the next piece of the file).  FASTA: up
    to the last line that starts with '>'.  FASTQ: the reader's own walk over the lines -- a non-empty line starting
    with '@' opens a record of four lines, any other line is skipped alone -- stopped at the first record whose four
    lines are not all there yet."""
    if eof:
        return len(raw)
    if not is_fastq:
        p = raw.rfind(b"\n>")
        return p + 1 if p >= 0 else 0
    data = np.frombuffer(raw, dtype=np.uint8)
    nl = np.flatnonzero(data == 10)
    if nl.size == 0:
        return 0
    starts = np.concatenate(([0], nl[:-1] + 1))  # the complete (terminated) lines
    ends = nl
    n, i = starts.size, 0
    while i < n:
        if ends[i] > starts[i] and data[starts[i]] == ord("@"):
            if i + 3 >= n:
                return int(starts[i])
            i += 4
        else:
            i += 1
    return int(nl[-1]) + 1 if i == n else int(starts[min(i, n - 1)])


def iter_sequence_blocks(path, fastq_ids_as_fasta=False, chunk_bytes=1 << 28, reuse_buffers=False):
    """The records of a FASTA / FASTQ file in file order, a piece of the file at a time: yields (ids, seqs uint8,
    seq_off int64) for consecutive groups of whole records.  The reference streams its reads through pipes
    (count_kmers.py:131-139, kmer_searcher.cpp:284-292); this keeps the host at about chunk_bytes + one record
    whatever the size of the read set.  Same records, names and sequences as read_sequences.  The bytes are read
    into one reused buffer and cut into records by the library's native reader (fdr_reads_scan / fdr_reads_parse);
    the unconsumed tail of a piece moves to the front of the buffer, which grows only for a record longer than it.
    reuse_buffers: the yielded seqs are views of ONE buffer, valid until the next piece is asked for (what the
    pipeline wants: no fresh pages per piece); default: every piece owns its arrays."""
    chunk_bytes = max(int(chunk_bytes), 1)
    with open(path, "rb", buffering=0) as f:
        buf = np.empty(chunk_bytes, dtype=np.uint8)
        have = f.readinto(memoryview(buf)) or 0
        if have == 0:
            return
        is_fastq = buf[0] == ord("@")  # (a first line that is just '\n' is empty: FASTA)
        eof = False
        seq_buf = None
        while True:
            if not eof and have == buf.size:  # (a record longer than the buffer is waiting for its end)
                buf = np.concatenate((buf, np.empty(max(chunk_bytes, buf.size), dtype=np.uint8)))
            while not eof and have < buf.size:
                got = f.readinto(memoryview(buf)[have:]) or 0
                if got == 0:
                    eof = True
                have += got
            if reuse_buffers and (seq_buf is None or seq_buf.size < have):
                seq_buf = np.empty(buf.size, dtype=np.uint8)
            used, ids, seqs, off = _lib.reads_parse(buf, have, is_fastq, fastq_ids_as_fasta, eof, seq_buf)
            if ids:
                yield ids, seqs, off
            if eof:
                return
            buf[:have - used] = buf[used:have]  # (numpy copies overlapping ranges safely)
            have -= used


def read_sequences(path, fastq_ids_as_fasta=False):
    """FASTA / FASTQ -> (ids list of bytes, seqs uint8 [total], seq_off int64 [R+1]) exactly as
    kmer_searcher.cpp:153-200 reads them: FASTQ iff the first line starts with '@'.
    fastq_ids_as_fasta: the PIPELINE never shows kmer_searcher a FASTQ file -- count_kmers.py:76-79 converts
    it with `seqkit fq2fa` first, so a read's name is its header up to the first space or tab (the FASTA
    rule) there, and a record with an empty name is dropped; the stand-alone tool keeps the whole line.  FASTA: id = header up
    to the first space or tab; the sequence is every following line with only the '\\n' removed (a '\\r'
    stays and is an invalid character); empty lines are skipped; a record whose id is empty, and anything
    before the first header, is dropped.  FASTQ: id = the whole header line after '@', sequence = the
    next line, then two lines are skipped.  (Whole file in memory: the pipeline uses iter_sequence_blocks.)"""
    raw = np.fromfile(path, dtype=np.uint8)
    if raw.size == 0:
        return [], np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.int64)
    _, ids, seqs, off = _lib.reads_parse(raw, raw.size, raw[0] == ord("@"), fastq_ids_as_fasta, True)
    return ids, seqs, off


def search(seqs, seq_off, lib_codes, k, context=None, block_chars=1 << 31):
    """(indptr int64 [R+1], indices int32) -- ascending unique library indices per read (GPU only).
    Reads are independent, so read sets of more than block_chars characters go to the device in blocks of
    whole reads (one call holds its reads, their hits and the sort buffers in HBM: about 28 B per base)."""
    ctx = context or _lib.default_context()
    seq_off = np.ascontiguousarray(seq_off, dtype=np.int64)
    R = seq_off.size - 1
    if R <= 0 or int(seq_off[-1]) <= block_chars:
        return ctx.kmer_search(seqs, seq_off, lib_codes, int(k))
    ptr_parts, idx_parts, base, r0 = [np.zeros(1, dtype=np.int64)], [], 0, 0
    while r0 < R:
        # the reads r0 .. r1 - 1: at most block_chars characters, at least one read
        r1 = int(np.searchsorted(seq_off, seq_off[r0] + block_chars, side="right")) - 1
        r1 = min(max(r1, r0 + 1), R)
        ip, ix = ctx.kmer_search(seqs[seq_off[r0]:seq_off[r1]], seq_off[r0:r1 + 1] - seq_off[r0], lib_codes, int(k))
        ptr_parts.append(ip[1:] + base)
        idx_parts.append(ix)
        base += int(ip[-1])
        r0 = r1
    return np.concatenate(ptr_parts), np.concatenate(idx_parts) if idx_parts else np.empty(0, dtype=np.int32)


def write_output_bin(path, ids, indptr, indices):
    """output.bin as kmer_searcher.cpp:98-130 writes it (header '<4sB3sQ', per record '<H' id length, id,
    '<I' count, count x '<Q').  Ids must be printable ASCII, as there (:113-117)."""
    with open(path, "wb", buffering=1 << 24) as f:
        f.write(struct.pack("<4sB3sQ", b"KMER", 1, b"\0\0\0", len(ids)))
        idx64 = np.asarray(indices).astype("<u8")
        ptr = np.asarray(indptr).tolist()
        for r, name in enumerate(ids):
            if any(c < 32 or c > 126 for c in name):
                raise ValueError("ID contains non-ASCII characters")
            a, b = ptr[r], ptr[r + 1]
            f.write(struct.pack("<H", len(name)))
            f.write(name)
            f.write(struct.pack("<I", b - a))
            f.write(idx64[a:b].tobytes())


def write_kmer_frequency_bin(path, indices, n_lib):
    """kmer_frequency.bin (kmer_searcher.cpp:203-230): (index, number of reads containing it) as '<QQ' for
    every library k-mer found in at least one read, ascending index."""
    counts = np.bincount(np.asarray(indices, dtype=np.int64), minlength=int(n_lib))
    nz = np.flatnonzero(counts)
    out = np.empty((nz.size, 2), dtype="<u8")
    out[:, 0] = nz
    out[:, 1] = counts[nz]
    out.tofile(path)


def kmer_searcher(kmer_lib, input_reads, output_dir, k, threads=None, context=None, fastq_ids_as_fasta=False,
                  collect=True, chunk_bytes=1 << 28, lib_codes=None):
    """Drop-in for the command line `kmer_searcher <kmer_lib> <input> <output_dir> <k> <threads>`
    (kmer_searcher.cpp:232-375).  `kmer_lib`: a path, a list of paths (read in order, like
    `cat fwd rev | grep -v '^>'`; '>' header tokens are not k long and drop out by themselves unless a
    count happens to have k digits -- so, as in the reference's pipeline, header lines are removed first).
    Writes output_dir/output.bin and output_dir/kmer_frequency.bin.  The reads are STREAMED (iter_sequence_blocks):
    a piece of the file is parsed (natively), searched on the GPU and its records appended (natively:
    fdr_kmer_output_append); the record count in the header is patched at the end.  Returns (ids, indptr, indices, n_lib) -- with collect=False (the pipeline: nothing of the
    read set is kept on the host) ids is the number of reads, indptr None and indices the number of hits.
    fastq_ids_as_fasta: see read_sequences (the pipeline's callers set it).  lib_codes: the library as
    load_kmer_library would return it for kmer_lib, for a caller that has just written those files and still holds
    their k-mers (run_kmer_searcher): the files are then not read back."""
    if lib_codes is not None:
        if not 1 <= int(k) <= 31:
            raise ValueError("Invalid k value: %r" % (k,))
        codes = np.ascontiguousarray(lib_codes, dtype=np.uint64)
    else:
        paths = [kmer_lib] if isinstance(kmer_lib, (str, bytes, os.PathLike)) else list(kmer_lib)
        texts = []
        for p in paths:
            with open(p, "rb") as f:
                t = f.read()
            texts.append(b"\n".join(l for l in t.split(b"\n") if not l.startswith(b">")) + b"\n")
        codes = load_kmer_library(texts, k)
    os.makedirs(output_dir, exist_ok=True)
    freq = np.zeros(int(codes.size), dtype=np.int64)
    all_ids, ptr_parts, idx_parts, n_reads, nnz = [], [np.zeros(1, dtype=np.int64)], [], 0, 0
    # The records go to output.bin.tmp, which becomes output.bin once the header holds the record count: a piece that
    # fails (an id fdr_kmer_output_append refuses, a GPU error) leaves no output.bin with a VALID header saying "0
    # records" behind for a later --kmer-searcher-output run to take for an empty read set.
    final_bin = os.path.join(output_dir, "output.bin")
    out_bin = final_bin + ".tmp"
    if os.path.exists(final_bin):
        os.remove(final_bin)
    try:
        with open(out_bin, "wb") as f:
            f.write(struct.pack("<4sB3sQ", b"KMER", 1, b"\0\0\0", 0))
        for ids, seqs, off in iter_sequence_blocks(input_reads, fastq_ids_as_fasta=fastq_ids_as_fasta,
                                                   chunk_bytes=chunk_bytes, reuse_buffers=True):
            indptr, indices = search(seqs, off, codes, k, context=context)
            _lib.kmer_output_append(out_bin, ids, indptr, indices)  # (native: no per-record Python)
            if indices.size:
                freq += np.bincount(indices, minlength=freq.size)
            if collect:
                all_ids += ids
                ptr_parts.append(indptr[1:] + nnz)
                idx_parts.append(indices)
            n_reads += len(ids)
            nnz += int(indices.size)
        with open(out_bin, "r+b") as f:  # the record count, known now
            f.seek(8)
            f.write(struct.pack("<Q", n_reads))
        os.replace(out_bin, final_bin)
    except BaseException:
        if os.path.exists(out_bin):
            os.remove(out_bin)
        raise
    nz = np.flatnonzero(freq)
    out = np.empty((nz.size, 2), dtype="<u8")
    out[:, 0] = nz
    out[:, 1] = freq[nz]
    out.tofile(os.path.join(output_dir, "kmer_frequency.bin"))
    if not collect:
        return n_reads, None, nnz, int(codes.size)
    indices = np.concatenate(idx_parts) if idx_parts else np.empty(0, dtype=np.int32)
    return all_ids, np.concatenate(ptr_parts), indices, int(codes.size)
