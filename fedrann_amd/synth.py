"""Synthetic ONT-like read x k-mer feature matrices.

A circular "genome" of L sampled-k-mer slots; the feature id of slot s is perm[s] (jellyfish
order is uncorrelated with position).  Read i: start U[0, L), length max(4, lognormal(ln m -
sigma^2/2, sigma)) slots, each slot kept with probability q (sequencing error), at least one kept;
strand flip with probability 0.5 (flipped => feature + L).  F = 2L features.  Counts for the IDF
are U{2..2c}, shared by f and f + L.  With doubling=True row 2i is the read and row 2i+1 its
strand mirror (the reference's behaviour, feature_extraction.py:136-140); with doubling=False one
row per read.  No row is empty.
"""
import os

import numpy as np


def _synth_chunk(args):
    """Rows of reads [r0, r0 + n): every chunk has its own generator, seeded by (seed, chunk number), so any
    row range can be generated without the rows before it (rank-local generation, threads)."""
    seed, cid, r0, n, L, m, q, sigma, doubling, perm = args
    F = 2 * L
    rng = np.random.default_rng([seed, 1 + cid])
    start = rng.integers(0, L, size=n)
    length = np.maximum(4, rng.lognormal(np.log(m) - sigma * sigma / 2, sigma, size=n)).astype(np.int64)
    length = np.minimum(length, L)
    flip = rng.random(n) < 0.5
    T = int(length.sum())
    row = np.repeat(np.arange(n, dtype=np.int64), length)
    first = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(length, out=first[1:])
    off = np.arange(T, dtype=np.int64) - np.repeat(first[:-1], length)
    keep = rng.random(T) < q
    kept_per_row = np.bincount(row[keep], minlength=n)
    keep[first[:-1][kept_per_row == 0]] = True  # at least one slot survives
    row, off = row[keep], off[keep]
    slot = (start[row] + off) % L
    feat = perm[slot] + np.where(flip[row], L, 0)
    if doubling:
        mirror = np.where(feat < L, feat + L, feat - L)
        row = np.concatenate((2 * row, 2 * row + 1))
        feat = np.concatenate((feat, mirror))
        nrows = 2 * n
    else:
        nrows = n
    key = row * np.int64(F) + feat
    key.sort()
    row_s = key // np.int64(F)
    return (key - row_s * np.int64(F)).astype(np.int32), np.bincount(row_s, minlength=nrows)


def synth(R, seed=602, m=200, q=0.85, c=30, Lcap=12_500_000, sigma=0.5, doubling=False,
          chunk=100_000, reads=None, threads=None):
    """Returns dict(indptr int64, indices int32 (ascending per row), n_features, counts int64 [L],
    names list[str], strands list[int]).  reads=(lo, hi): only the rows of reads [lo, hi) (lo a multiple
    of `chunk`; indptr rebased to 0) -- what one rank of a row-sharded run needs; the genome (perm),
    the counts and every read are the same whichever range is asked for."""
    rng = np.random.default_rng([seed, 0])
    L = int(min(max(R * m // c, 64), Lcap))
    F = 2 * L
    perm = rng.permutation(L).astype(np.int64)
    counts = rng.integers(2, 2 * c + 1, size=L).astype(np.int64)
    lo, hi = (0, R) if reads is None else (int(reads[0]), int(reads[1]))
    if lo % chunk or not (0 <= lo <= hi <= R):
        raise ValueError("reads=(lo, hi): lo must be a multiple of chunk=%d and 0 <= lo <= hi <= R" % chunk)
    jobs = []
    for r0 in range(lo, hi, chunk):
        full = min(chunk, R - r0)  # a chunk is always generated whole (its generator is consumed in order)
        jobs.append((seed, r0 // chunk, r0, full, L, m, q, sigma, doubling, perm))
    if threads is None:
        threads = min(16, len(os.sched_getaffinity(0)))
    if len(jobs) > 1 and threads > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=threads) as ex:  # numpy's sort and RNG release the GIL
            parts = list(ex.map(_synth_chunk, jobs))
    else:
        parts = [_synth_chunk(j) for j in jobs]
    per = 2 if doubling else 1
    ip_parts, ix_parts, total = [np.zeros(1, dtype=np.int64)], [], 0
    for job, (ix, cnt) in zip(jobs, parts):
        want = (min(hi, job[2] + job[3]) - job[2]) * per  # the last chunk may be cut by hi
        if want < cnt.size:
            cnt = cnt[:want]
            ix = ix[:int(cnt.sum())]
        ix_parts.append(ix)
        ip_parts.append(total + np.cumsum(cnt))
        total += int(cnt.sum())
    indptr = np.concatenate(ip_parts).astype(np.int64)
    indices = np.concatenate(ix_parts) if ix_parts else np.zeros(0, np.int32)
    if doubling:
        names = ["read_%d" % i for i in range(lo, hi) for _ in (0, 1)]
        strands = [0, 1] * (hi - lo)
    else:
        names = ["read_%d" % i for i in range(lo, hi)]
        strands = [0] * (hi - lo)
    return {"indptr": indptr, "indices": indices, "n_features": F, "counts": counts,
            "names": names, "strands": strands}


def synth_sequences(n_reads, genome_len=200_000, mean_len=3000, k=15, sample=0.05, error=0.06, n_rate=2e-4,
                    seed=602):
    """Sequence-level synthetic input for the k-mer search: a random genome,
    reads cut from either strand with substitution errors and a few N, and a k-mer library = a Bernoulli
    sample of the genome's canonical k-mers followed by their reverse complements (what count_kmers.py
    builds with jellyfish -C + awk + seqkit).  Returns dict(ids, seqs uint8, seq_off int64, fwd list of
    bytes, rev list of bytes)."""
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = np.zeros(256, dtype=np.uint8)
    comp[list(b"ACGTN")] = list(b"TGCAN")
    genome = alpha[rng.integers(0, 4, size=genome_len)]
    lens = np.maximum(1, rng.lognormal(np.log(mean_len) - 0.125, 0.5, size=n_reads)).astype(np.int64)
    lens = np.minimum(lens, genome_len)
    starts = rng.integers(0, genome_len - lens + 1)
    pieces = []
    for s, n in zip(starts.tolist(), lens.tolist()):
        r = genome[s:s + n].copy()
        if rng.random() < 0.5:
            r = comp[r[::-1]]
        e = rng.random(n) < error
        r[e] = alpha[rng.integers(0, 4, size=int(e.sum()))]
        r[rng.random(n) < n_rate] = ord("N")
        pieces.append(r)
    seq_off = np.zeros(n_reads + 1, dtype=np.int64)
    np.cumsum(lens, out=seq_off[1:])
    seqs = np.concatenate(pieces) if pieces else np.zeros(0, dtype=np.uint8)
    # canonical k-mers of the genome, sampled
    win = np.lib.stride_tricks.sliding_window_view(genome, k)
    pick = np.flatnonzero(rng.random(win.shape[0]) < sample)
    fwd, seen = [], set()
    for p in pick.tolist():
        a = win[p].tobytes()
        b = comp[win[p][::-1]].tobytes()
        c = min(a, b)
        if c not in seen:
            seen.add(c)
            fwd.append(c)
    rev = [comp[np.frombuffer(x, dtype=np.uint8)[::-1]].tobytes() for x in fwd]
    ids = [b"read_%07d" % i for i in range(n_reads)]
    return {"ids": ids, "seqs": seqs, "seq_off": seq_off, "fwd": fwd, "rev": rev, "k": k}
