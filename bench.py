#!/usr/bin/env python
"""bench.py -- read-pairs/s of the FEDRANN hot path (embed -> normalise -> all-pairs cosine k-NN).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--dim D] [--knn k]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over the whole synthetic read set: every rank embeds and
normalises its row block, the normalised embeddings are all-gathered (N > 1), and every rank
searches its rows against all targets.  Default workload = BASELINE.json configs[2], the largest
single-GPU configuration: 1 M synthetic ONT reads, 128-dim projection, k-NN = 20 (configs[1] =
--reads 100000; configs[3] = --reads 10000000 on 8 GPUs; configs[4] = --reads 10000000 --doubling
--dim 256 --knn 50).  The total work is the same for every N (strong scaling).

`value` = R * k * steps / time with the inputs (the rank's CSR rows, the projection tables) resident
in HBM when the timed region starts and the results left in HBM.  `host_to_host` (N = 1) is the span
SURVEY.md section 8(d) words: CSR in (pinned) host memory -> (indices, distances) in host memory through
fdr_embed_knn, i.e. with the PCIe copies inside the timed region -- once with the CSR as the loader
delivers it and once with the dead features dropped on the host first (fdr_csr_compact; not timed,
it belongs to the loader).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel of the timed mode (the fp16 MFMA
prefilter pass by default, the fp32 MFMA tile kernel in exact mode), priced on the unique rows it
actually searched; `cpu_baseline` times this repo's CPU oracle (a port of the path, not the reference's
pynndescent, which is not installed) on a bounded sample of the same workload; `recall` compares the
timed run's rows with the oracle's (tie-aware recall@k) and, when it imports, with pynndescent called
as the reference calls it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak (same guide; never the 2:1-sparse figure)
HBM_PEAK_GBS = 8000.0


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--reads", type=int, default=1_000_000,
                   help="reads (= rows of the feature matrix, x2 with --doubling)")
    p.add_argument("--dim", type=int, default=128)
    p.add_argument("--knn", type=int, default=20)
    p.add_argument("--seed", type=int, default=602)
    p.add_argument("--cpu-baseline-seconds", type=float, default=15.0,
                   help="target CPU time of the cpu_baseline sample (0 = skip)")
    p.add_argument("--doubling", action="store_true", help="fwd/rev row doubling (2 rows per read)")
    p.add_argument("--mode", choices=["auto", "exact", "prefilter"], default="auto",
                   help="k-NN mode of the timed run (same results in every mode); auto = the library's "
                        "default: fp16 prefilter + certificate + exact re-rank when it applies")
    p.add_argument("--no-compare", action="store_true",
                   help="skip the extra (separately timed) pass in the other k-NN mode")
    p.add_argument("--compare-steps", type=int, default=2, help="steps of the other-mode pass")
    p.add_argument("--no-host-span", action="store_true", help="skip the host-to-host (PCIe-inclusive) passes")
    p.add_argument("--nndescent-full", action="store_true",
                   help="also run the restated reference algorithm (oracle/nndescent.c: RP forest + NN-descent with the "
                        "reference's arguments) on ALL rows of the workload -- minutes of host time, outside every timed region; "
                        "its recall is scored against the GPU run's (exact) result")
    p.add_argument("--host-steps", type=int, default=0,
                   help="steps of the host-to-host pass over the full CSR (0 = --steps, like the device-resident run); the "
                        "compacted-CSR variant runs min(3, that)")
    return p.parse_args()


def cpu_budget():
    """Host threads this process may really use: the affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def tie_aware_recall(got_idx, want_dist, Eh, zero, k):
    """recall@k of `got_idx` against the exact k-NN whose ascending distances are `want_dist`: a returned
    neighbour counts when its canonical distance to the query is <= the exact k-th distance (so any member
    of a tie across the rank-k boundary is as good as another).  Query i is row i of Eh."""
    import numpy as np
    from oracle import oracle as O
    m = got_idx.shape[0]
    if m == 0:
        return None
    step = max(1, m // 2048)  # pair distances one at a time: a sample of the sample
    rows = np.arange(0, m, step)
    hits = 0
    for q in rows:
        kth = want_dist[q, k - 1]
        for t in got_idx[q]:
            hits += O.pair_dist_normalized(Eh[q], zero[q], Eh[t], zero[t]) <= kth
    return hits / float(rows.size * k)


def pynndescent_recall(ctx, E, k, cores, seed, rows=20000):
    """The reference's own k-NN call (nearest_neighbors.py:39-55 with the arguments of __main__.py:184-197)
    on the first `rows` embeddings, if the third-party package imports on this box.  Its graph and the
    GPU's result for the same sub-sample are both scored against the exact oracle (tie-aware recall@k).
    Returns (dict | None, reason)."""
    import numpy as np
    try:
        import pynndescent
    except Exception as e:  # not installed in this image (no network): reported, never faked
        return None, "import pynndescent failed (%s: %s)" % (type(e).__name__, e)
    from oracle import oracle as O
    sub = np.ascontiguousarray(E[:rows])
    Eh, _, zero = O.normalize(sub)
    _, want_dist = O.knn_normalized(Eh, zero, Eh, zero, k)
    t0 = time.perf_counter()
    index = pynndescent.NNDescent(sub, metric="cosine", n_neighbors=k, n_trees=300, leaf_size=200, n_iters=None,
                                  diversify_prob=1.0, pruning_degree_multiplier=1.5, low_memory=True,
                                  n_jobs=cores, random_state=seed, verbose=False)
    p_idx, _ = index.neighbor_graph
    dt = time.perf_counter() - t0
    g_idx, _ = ctx.knn(sub, k)
    return {"rows": int(sub.shape[0]), "pynndescent_vs_exact": tie_aware_recall(np.maximum(p_idx, 0), want_dist, Eh, zero, k),
            "gpu_vs_exact": tie_aware_recall(g_idx, want_dist, Eh, zero, k),
            "pynndescent_read_pairs_per_s": sub.shape[0] * k / dt, "pynndescent_seconds": dt, "cores": cores}, None


def nndescent_baseline(O, E, k, cores, target_seconds):
    """The ALGORITHM the reference's k-NN stage runs -- pynndescent's RP forest + NN-descent with the arguments of
    __main__.py:184-197 (n_trees = 300, leaf_size = 200, max_candidates = min(60, k), delta = 0.001) -- restated in
    oracle/nndescent.c (the package does not import here), timed on the first m rows (m sized for ~target_seconds:
    the forest's leaf pairs, n_trees * m * leaf_size / 2 distances, dominate) and scored against the exact k-NN of the
    same m rows (tie-aware recall@k).  An approximate, randomised method: a figure beside the exact search, not parity."""
    import numpy as np
    n, d = E.shape
    # size the sample from a measured probe, not a constant: the same call on the first 8192 rows gives this box's cost per
    # distance evaluation; the forest's leaf pairs (n_trees * m * leaf_size / 2 distances) dominate and grow linearly in m
    m0 = int(min(n, 8192))
    Eh0, _, zero0 = O.normalize(np.ascontiguousarray(E[:m0]))
    t0 = time.perf_counter()
    _, _, st0 = O.nndescent(Eh0, zero0, k, n_trees=300, leaf_size=200, seed=602)
    t_probe = time.perf_counter() - t0
    per_row = t_probe / m0  # seconds per row of the sample at this d, k and core count
    m = int(min(n, max(20000, target_seconds / per_row)))
    m = min(m, 100_000)  # (+ the exact k-NN of the same rows for the recall: m^2 pairs)
    sub = np.ascontiguousarray(E[:m])
    Eh, _, zero = O.normalize(sub)
    t0 = time.perf_counter()
    a_idx, a_dist, stats = O.nndescent(Eh, zero, k, n_trees=300, leaf_size=200, seed=602)
    dt = time.perf_counter() - t0
    _, want_dist = O.knn_normalized(Eh, zero, Eh, zero, k)
    recall = float(((a_dist <= want_dist[:, k - 1:k]) & (a_idx >= 0)).mean())
    return {"rows": m, "seconds": dt, "read_pairs_per_s": m * k / dt, "recall_at_k_tie_aware_vs_exact": recall,
            "cores": cores, "descent_rounds": stats["rounds"], "distance_evaluations": stats["distance_evaluations"],
            "sample": "the first %d of %d rows, SEARCHED AMONG THEMSELVES (targets = the same %d rows, not the workload's %d): "
                      "an m-row problem, not a share of the n-row one; no extrapolation to n is made" % (m, n, m, n),
            "sizing_basis": {"probe_rows": m0, "probe_seconds": t_probe, "probe_distance_evaluations": st0["distance_evaluations"],
                             "seconds_per_distance": t_probe / max(1, st0["distance_evaluations"]),
                             "rule": "rows = target_seconds / (probe_seconds / probe_rows), clamped to [20000, 100000]"},
            "note": "oracle/nndescent.c: restatement of pynndescent's published algorithm with the reference's arguments"}


def cpu_baseline(s, P, d, k, target_seconds, gpu_result=None):
    """Time the CPU oracle on a bounded sample: embed + normalise ALL rows (they are the targets),
    then k-NN for as many query rows as fit the time budget; extrapolate linearly in queries.
    `gpu_result` = (idx, dist) numpy arrays of the GPU run's first rows: the oracle's rows are compared
    with them bit for bit (reported as `parity_sample`; the measurement itself is unaffected)."""
    import numpy as np
    from oracle import oracle as O
    O.set_num_threads(cpu_budget())  # (libgomp is already loaded by torch: the env var would be too late)
    cores = O.lib().orc_num_threads()
    n = len(s["indptr"]) - 1
    t0 = time.perf_counter()
    E = O.embed(s["indptr"], s["indices"], (P.indptr, P.indices, P.data), s["n_features"], d)
    t_embed = time.perf_counter() - t0
    t0 = time.perf_counter()
    Eh, _, zero = O.normalize(E)
    t_norm = time.perf_counter() - t0
    probe = min(n, max(256, 64 * cores))
    O.knn_normalized(Eh[:64], zero[:64], Eh, zero, k)  # warm the thread pool / page in the targets
    t0 = time.perf_counter()
    O.knn_normalized(Eh[:probe], zero[:probe], Eh, zero, k)
    t_probe = time.perf_counter() - t0
    nq = int(min(n, max(probe, target_seconds / max(t_probe / probe, 1e-9))))
    nq = max(64, nq // 64 * 64) if n >= 64 else n
    t0 = time.perf_counter()
    o_idx, o_dist = O.knn_normalized(Eh[:nq], zero[:nq], Eh, zero, k)
    t_knn = time.perf_counter() - t0
    est_total = t_embed + t_norm + t_knn * n / nq
    parity = None
    if gpu_result is not None:
        m = min(nq, gpu_result[0].shape[0])
        same = bool(np.array_equal(gpu_result[0][:m], o_idx[:m])) and bool(
            np.array_equal(gpu_result[1][:m].view(np.uint32), o_dist[:m].view(np.uint32)))
        parity = {"rows": int(m), "identical_indices_and_distance_bits": same,
                  "recall_at_k_tie_aware": tie_aware_recall(gpu_result[0][:m], o_dist[:m], Eh, zero, k),
                  "max_abs_distance_error": float(np.abs(gpu_result[1][:m] - o_dist[:m]).max()) if m else 0.0}
    nnd = nndescent_baseline(O, E, k, cores, target_seconds)
    return {
        "parity_sample": parity,
        "nndescent": nnd,
        "value": n * k / est_total, "unit": "read-pairs/s", "cores": cores, "kind": "port",
        "sample": ("oracle/fedrann_oracle.c (exact fp32 cosine k-NN, AVX2+OpenMP): embed+normalise all "
                   "%d rows (%.2fs+%.2fs), k-NN of %d of %d query rows vs all targets in %.2fs, "
                   "extrapolated x%.1f in queries" % (n, t_embed, t_norm, nq, n, t_knn, n / nq)),
    }


def main():
    args = parse_args()
    # (multi-process GPU work on this pool needs dmabuf IPC: RCCL fails with hipIpcGetMemHandle otherwise; exported by the
    # image already, kept here for environments that drop it)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # FEDRANN_BENCH_REHEARSE=1: every rank on GPU 0 over gloo -- a one-GPU rehearsal of the N > 1 code path
    # (sharding, exchange, per-rank k-NN, reductions); never a measurement
    rehearse = os.environ.get("FEDRANN_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from fedrann_amd import _lib
    from fedrann_amd.distributed import HipEngine, ShardedPipeline, local_csr
    from fedrann_amd.precompute import build_precompute_matrix
    from fedrann_amd.synth import synth

    R, d, k = args.reads, args.dim, args.knn
    per = 2 if args.doubling else 1
    n = R * per
    # every rank generates only ITS rows (chunk-seeded generator: the same reads whichever rank makes them)
    from fedrann_amd.distributed import shard_rows
    _, blocks = shard_rows(n, world)
    lo, hi = blocks[rank]
    chunk = 100_000
    r_lo, r_hi = lo // per, -(-hi // per)
    g_lo = r_lo // chunk * chunk
    # The rank's rows are generated in pieces of <= 1 M reads and compacted piece by piece (what the product uploads --
    # python -m fedrann_amd -- is the CSR without the ids P has no entry for: >= 90 % of them, density 1 / sqrt(F); same
    # E bit for bit, tests); the full CSR is kept too while it is small enough to be the host-to-host pass's and the CPU
    # baseline's input (10 M reads: 6.8 GB of ids per copy -- the pieces are dropped and those two legs are skipped).
    keep_full = (r_hi - g_lo) <= 2_000_000
    piece = 1_000_000
    gen_threads = cpu_budget()  # (a 100 k-read chunk of the generator holds ~2 GB of temporaries: as many at once as fit)
    try:
        import psutil
        gen_threads = max(2, min(gen_threads, 16, int(psutil.virtual_memory().available / 2**30 / 3.0)))
    except Exception:
        gen_threads = max(1, min(gen_threads, 8))
    ctx = _lib.Context(local_rank)
    ctx.set_knn_mode(args.mode)
    P = None
    s = None
    c_ip, c_ix, f_ip, f_ix = [np.zeros(1, dtype=np.int64)], [], [np.zeros(1, dtype=np.int64)], []
    nnz_total = 0
    for p0 in range(g_lo, r_hi, piece):
        p1 = min(r_hi, p0 + piece)
        sp = synth(R, seed=args.seed, doubling=args.doubling, chunk=chunk, reads=(p0, p1), threads=gen_threads)
        if P is None:
            P = build_precompute_matrix(sp["counts"], d)
            ctx.projection_load(P.indptr, P.indices, P.data, sp["n_features"], d)
            s = {"n_features": sp["n_features"], "counts": sp["counts"]}
        a = max(lo, p0 * per) - p0 * per      # this rank's rows of the piece
        b = min(hi, p1 * per) - p0 * per
        pip = np.ascontiguousarray(sp["indptr"][a:b + 1] - sp["indptr"][a])
        pix = np.ascontiguousarray(sp["indices"][sp["indptr"][a]:sp["indptr"][b]])
        del sp
        nnz_total += int(pix.size)
        cip, cix = ctx.csr_compact(pip, pix)
        c_ip.append(cip[1:] + c_ip[-1][-1])
        c_ix.append(cix)
        if keep_full:
            f_ip.append(pip[1:] + f_ip[-1][-1])
            f_ix.append(pix)
        del pip, pix
    ip = np.concatenate(c_ip)
    ix = np.concatenate(c_ix) if c_ix else np.zeros(0, dtype=np.int32)
    del c_ip, c_ix
    if keep_full:
        ip_full, ix_full = np.concatenate(f_ip), (np.concatenate(f_ix) if f_ix else np.zeros(0, dtype=np.int32))
        s.update(indptr=ip_full, indices=ix_full)
    else:
        ip_full = ix_full = None
    del f_ip, f_ix
    engine = HipEngine(ctx, device)
    pipe = ShardedPipeline(engine, n, d, k, rank=rank, world_size=world, device=device)
    assert (pipe.lo, pipe.hi) == (lo, hi) and ip.size == hi - lo + 1
    d_ip = torch.from_numpy(ip).to(device)
    d_ix = torch.from_numpy(ix).to(device)
    nloc = pipe.hi - pipe.lo
    nnz_live = int(ix.size)
    if world > 1:
        t = torch.tensor([nnz_live], dtype=torch.int64, device=device)
        dist.all_reduce(t)
        nnz_live = int(t.item())
    if world > 1:
        t = torch.tensor([nnz_total], dtype=torch.int64, device=device)
        dist.all_reduce(t)
        nnz_total = int(t.item())

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    def timed_run(steps, warmup):
        """W untimed + K timed steps; returns (seconds, per-kernel average ms, last result)."""
        res = None
        for _ in range(warmup):
            res = pipe.step(d_ip, d_ix)
        barrier()
        ctx.timing(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            res = pipe.step(d_ip, d_ix)
        barrier()
        dt = time.perf_counter() - t0
        kms, kcnt = {}, {}
        for i, name in enumerate(_lib.KERNELS):
            cnt, ms = ctx.timing_read(i)  # timed spans of this kind since the last read, their total
            kms[name] = ms / max(steps, 1)  # per step (a kind may have several spans per step)
            kcnt[name] = cnt / max(steps, 1)
        ctx.timing(False)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, kms, kcnt, res

    elapsed, kernel_ms, kernel_cnt, out = timed_run(args.steps, args.warmup)
    uncertified = ctx.last_uncertified()
    uniq_t, uniq_q = ctx.last_unique()  # rows actually searched (duplicate-row classes)
    pass_launches, pass_queues = ctx.last_prefilter_launches()
    used_prefilter = kernel_cnt["knn_prefilter"] > 0

    # the same workload in the other mode, separately timed (never part of `value`)
    other = None
    if not args.no_compare and (used_prefilter or args.mode != "exact"):
        other_mode = "exact" if used_prefilter else "prefilter"
        ctx.set_knn_mode(other_mode)
        o_steps = max(1, min(args.steps, args.compare_steps))
        o_elapsed, o_ms, o_cnt, o_out = timed_run(o_steps, min(args.warmup, 1))
        same = bool(torch.equal(out[0], o_out[0])) and bool(torch.equal(out[1], o_out[1]))
        if o_cnt["knn_prefilter"] > 0 or other_mode == "exact":
            other = {"mode": other_mode, "value": n * k * o_steps / o_elapsed, "unit": "read-pairs/s",
                     "steps": o_steps, "ms_per_step": o_elapsed / o_steps * 1e3, "kernels_ms": o_ms,
                     "kernel_launches": o_cnt,
                     "identical_to_timed_run": same}
        ctx.set_knn_mode(args.mode)

    # SURVEY.md 8(d)'s span, N = 1: CSR in pinned host memory -> (idx, dist) in pinned host memory through the
    # fused host-pointer call (H2D, embed, normalise, k-NN, D2H).  Never `value` (the contract times the
    # device-resident path); same results, checked.
    host_span = None
    if world == 1 and not args.no_host_span and not keep_full:
        host_span = {"skipped": "the full CSR of %d reads is not held on the host (see keep_full)" % R}
    elif world == 1 and not args.no_host_span:
        h_idx = np.empty((n, k), dtype=np.int32)
        h_dst = np.empty((n, k), dtype=np.float32)
        hsteps = args.host_steps if args.host_steps > 0 else args.steps
        host_span = {"steps": hsteps, "unit": "read-pairs/s"}
        # full_csr: the CSR as the loader / feature_matrix.npz holds it; compacted_csr: after fdr_csr_compact (what
        # the CLI uploads; the compaction itself is host work of the loader stage and not timed here)
        for label, (a_ip, a_ix) in (("full_csr", (ip_full, ix_full)), ("compacted_csr", (ip, ix))):
            ctx.host_register(a_ip, a_ix, h_idx, h_dst)
            ctx.embed_knn(a_ip, a_ix, k, out=(h_idx, h_dst))  # warm-up: scratch buffers sized, tables hot
            nst = hsteps if label == "full_csr" else min(3, hsteps)
            t0 = time.perf_counter()
            for _ in range(nst):
                ctx.embed_knn(a_ip, a_ix, k, out=(h_idx, h_dst))  # (synchronises before it returns)
            dt = time.perf_counter() - t0
            ctx.host_unregister(a_ip, a_ix, h_idx, h_dst)
            same = bool(np.array_equal(h_idx, out[0].cpu().numpy())) and bool(
                np.array_equal(h_dst.view(np.uint32), out[1].cpu().numpy().view(np.uint32)))
            host_span[label] = {"value": n * k * nst / dt, "ms_per_step": dt / nst * 1e3, "steps": nst,
                                "h2d_bytes": int(a_ip.nbytes + a_ix.nbytes), "d2h_bytes": int(h_idx.nbytes + h_dst.nbytes),
                                "identical_to_timed_run": same}
        del h_idx, h_dst

    # sanity on the last step's result (not timed): self is its own nearest neighbour
    idx = out[0][: min(nloc, 4096)].cpu().numpy()
    dst = out[1][: min(nloc, 4096)].cpu().numpy()
    E = out[2][: min(nloc, 4096)].cpu().numpy()
    nz = np.abs(E).sum(1) > 0
    rows = np.arange(pipe.lo, pipe.lo + idx.shape[0])
    # a non-zero row finds itself unless k rows with identical embeddings precede it (ties go by index)
    found = (idx == rows[:, None]).any(1) | (dst[:, -1] <= 1e-6)
    ok = bool(np.all(found[nz])) and bool(np.all(np.diff(dst, axis=1) >= 0))
    zero_frac = float(1.0 - nz.mean()) if nz.size else 0.0
    # rows whose list ends inside a run of equal distances (the k-th neighbour is decided by the index)
    tie_frac = float((dst[:, -1] == dst[:, -2]).mean()) if dst.shape[0] and dst.shape[1] > 1 else 0.0

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n * k * args.steps / elapsed
        # roofline of the dominant kernel: the ordered (unique query, unique target) pairs this rank
        # actually evaluated, 2*d flop each (= all nloc * n pairs when the class layer did not engage)
        flops = 2.0 * uniq_q * uniq_t * d

        def mfma_roofline(kms, kcnt, prefilter, timed_mode=True):
            name, peak = ("knn_prefilter", MFMA_F16_PEAK_TFLOPS) if prefilter else ("knn_tile", MFMA_F32_PEAK_TFLOPS)
            ms = kms[name]  # HIP events on the launch stream, per step: the whole pass
            ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            # The pass is one launch, or (synchronised rounds) `launches` equal launches dealt to `queues` queues.
            # With two queues two launches are in flight at any time, so a launch lasts pass_ms * queues / launches
            # -- the per-dispatch duration a kernel trace shows -- and runs at 1 / queues of `achieved`.
            launches, queues = (max(1, pass_launches), max(1, pass_queues)) if (prefilter and timed_mode) else \
                (max(1.0, kcnt[name]), 1)
            nominal = 2.0 * nloc * n * d  # every ordered (query row, target row) pair, duplicates included
            pp_kernel = prefilter and ctx.padded_dim(d) > 128 and launches >= 2 and queues == 2 and timed_mode
            return {"kernel": (("knn_prefilter_pp_kernel (fp16 MFMA 32x32x16, ping-pong turns)" if pp_kernel else
                                "knn_prefilter_kernel (fp16 MFMA 32x32x16)") if prefilter
                               else "knn_tile_kernel<%d> (fp32 MFMA 32x32x2)" % ctx.padded_dim(d)),
                    "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                    "priced_on": "ordered (unique query, unique target) pairs actually evaluated, 2 d flop each",
                    "nominal_flops": nominal, "nominal_frac": (nominal / (ms * 1e-3) / 1e12 / peak) if ms > 0 else 0.0,
                    "skipped_pair_fraction": 1.0 - flops / nominal if nominal > 0 else 0.0,
                    "pass_ms": ms, "launches_per_step": launches, "queues": queues,
                    "flops_per_launch": flops / launches, "avg_launch_ms": ms * queues / launches}

        roof = mfma_roofline(kernel_ms, kernel_cnt, used_prefilter)
        nnz_loc = int(ix.size)
        embed_bytes = 4.0 * nnz_loc + 8.0 * nloc + 4.0 * nloc * d
        # HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE doubled as the
        # gfx950 guide prescribes + WRITE_SIZE); only valid for the workload the profile was collected on
        traffic, traffic_src = None, None
        try:
            import glob
            import re
            summaries = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")),
                               key=lambda f: int(re.search(r"r(\d+)", os.path.basename(f)).group(1)), reverse=True)
            for path in summaries:  # the newest round's profile of THIS workload
                with open(path) as f:
                    prof = json.load(f)
                if prof.get("bench_args") == {"reads": R, "dim": d, "knn": k, "gpus": world, "doubling": args.doubling}:
                    pmc = prof["pmc_per_launch_avg"]
                    kname = ("knn_prefilter_pp_kernel" if "knn_prefilter_pp_kernel" in pmc else "knn_prefilter_kernel") \
                        if used_prefilter else "knn_tile_kernel"
                    traffic = pmc[kname]["hbm_bytes_per_launch"]
                    traffic_src = ("profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, bytes "
                                   "per launch, same workload)" % os.path.basename(path))
                    break
        except Exception:
            traffic = None
        result = {
            "metric": "read-pairs/sec (overlap candidates)", "value": value, "unit": "read-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 (canonical fp32 results; fp16 MFMA candidate prefilter + exact fp32 re-rank)"
                     if used_prefilter else "f32",
            "data": "synthetic",
            "config": {"workload": "%d synthetic ONT reads (%d rows%s), %d-dim projection, k-NN=%d, "
                                   "row-sharded over %d GPU(s)" % (R, n, ", fwd/rev doubled" if args.doubling else "",
                                                                  d, k, world),
                       "reads": R, "rows": n, "dim": d, "knn": k, "doubling": bool(args.doubling),
                       "n_features": int(s["n_features"]),
                       "nnz": nnz_total, "nnz_with_projection_entries": nnz_live,
                       "parallelism": "rows/%d + all-gather" % world,
                       "zero_row_fraction_sample": zero_frac, "boundary_tie_fraction_sample": tie_frac,
                       "self_check": ok},
            "roofline": dict(roof, traffic=traffic, traffic_source=traffic_src),
            "value_span": "inputs resident in HBM -> results in HBM (the bench contract's definition of `value`); "
                          "value_host_to_host = SURVEY 8(d)'s span: CSR in host memory -> (indices, distances) in host memory "
                          "through fdr_embed_knn with the full CSR, PCIe copies inside the timed region, same --steps",
            "value_device": value,
            "value_host_to_host": ((host_span or {}).get("full_csr") or {}).get("value"),
            "host_to_host": host_span,
            "knn_mode": "prefilter" if used_prefilter else "exact",
            "uncertified_queries_last_step": uncertified if used_prefilter else None,
            "unique_rows_searched": {"targets": uniq_t, "queries": uniq_q, "of_targets": n, "of_queries": nloc},
            "kernels_ms": kernel_ms,
            "embed_roofline": {"bound": "hbm", "achieved": embed_bytes / (kernel_ms["embed_csr"] * 1e-3) / 1e9
                               if kernel_ms["embed_csr"] > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "bytes_per_launch": embed_bytes,
                               "note": "4 B per column id of the compacted CSR + 8 B per row pointer + 4 d B per row of E"},
        }
        if other is not None:
            other["roofline"] = mfma_roofline(other["kernels_ms"], other["kernel_launches"], other["mode"] == "prefilter",
                                              timed_mode=False)
            result["other_mode"] = other
        if world == 1 and args.cpu_baseline_seconds > 0 and keep_full:
            m = min(nloc, 1 << 17)
            base = cpu_baseline(s, P, d, k, args.cpu_baseline_seconds,
                                gpu_result=(out[0][:m].cpu().numpy(), out[1][:m].cpu().numpy()))
            par = base.get("parity_sample") or {}
            pyn, why = pynndescent_recall(ctx, out[2].cpu().numpy(), k, base["cores"], args.seed)
            result["recall"] = {"recall_vs_oracle": par.get("recall_at_k_tie_aware"),
                                "oracle_rows_compared": par.get("rows"),
                                "ranks_and_distance_bits_identical": par.get("identical_indices_and_distance_bits"),
                                "recall_vs_pynndescent": pyn, "recall_vs_pynndescent_reason": why,
                                "nndescent_restatement_recall_vs_exact":
                                    (base.get("nndescent") or {}).get("recall_at_k_tie_aware_vs_exact")}
            result["cpu_baseline"] = base
            if args.nndescent_full:
                # The reference's k-NN ALGORITHM on the whole workload ("missing" 3): every row searched among
                # all rows, 300 trees, leaves of 200, on the box's host cores.  Recall against the GPU's result, which the
                # parity tests pin to the exact oracle bit for bit: a returned neighbour counts when its (restated,
                # canonical) distance is <= the exact k-th distance of its query.
                from oracle import oracle as O
                E_all = out[2].cpu().numpy()
                Eh_all, _, zero_all = O.normalize(E_all)
                t0 = time.perf_counter()
                a_idx, a_dist, st = O.nndescent(Eh_all, zero_all, k, n_trees=300, leaf_size=200, seed=602)
                dt = time.perf_counter() - t0
                kth = out[1][:, k - 1:k].cpu().numpy()
                result["cpu_baseline"]["nndescent_full"] = {
                    "rows": int(n), "seconds": dt, "read_pairs_per_s": n * k / dt, "cores": base["cores"],
                    "recall_at_k_tie_aware_vs_exact": float(((a_dist <= kth) & (a_idx >= 0)).mean()),
                    "descent_rounds": st["rounds"], "distance_evaluations": st["distance_evaluations"],
                    "gpu_over_this": value / (n * k / dt),
                    "note": "oracle/nndescent.c on every row of the workload (targets = all rows), n_trees = 300, leaf_size = 200, "
                            "max_candidates = min(60, k), delta = 0.001: the reference's call (__main__.py:184-197) restated; "
                            "pynndescent itself is not installed"}
        else:
            result["recall"] = None
            result["cpu_baseline"] = None
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
