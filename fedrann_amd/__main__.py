"""`python -m fedrann_amd` -- the fedrann command line on the GPU hot path.

Keeps the reference's flags and defaults (fedrann/__main__.py:69-171) and its stage sequence
(run_fedrann_pipeline :302-391): k-mer counting / sampling / search (stage 1, on the GPU too) ->
projection matrix -> embeddings -> k-NN -> overlaps.tsv.  Entry points:

    -i reads.fa[.gz]                              the whole pipeline, like the reference
    -i reads.fa --kmer-library fwd_kmer_library.fasta          stage 1b on (search only)
    --kmer-searcher-output out/temp/kmer_searcher/output.bin --kmer-library out/temp/fwd_kmer_library.fasta
        (what the reference leaves behind with --keep-intermediates), or
    --feature-matrix feature_matrix.npz --kmer-counts counts.npy [--read-names names.txt]
        (scipy.sparse.save_npz binary CSR of the rows to search; see feature_extraction.py).

--devices 0,1,...: the rows are sharded over several GPUs of the node.  The parent process starts one
child per GPU BEFORE it touches a GPU itself; every child embeds its row block, the blocks are
all-gathered (RCCL), every child searches its rows against all rows and writes its part of overlaps.tsv;
the parent concatenates the parts in rank order (byte-identical to the single-GPU file).
"""
import argparse
import logging
import os
from os.path import abspath, join
import subprocess
import sys
from shutil import copyfileobj, rmtree
from typing import List

import numpy as np
import pandas as pd

from . import __description__, __version__
from . import global_variables
from .feature_extraction import (build_feature_csr, embed_csr, get_feature_matrix, get_metadata,
                                 load_feature_matrix_npz, save_feature_matrix_npz)
from .nearest_neighbors import NNDescent_ava
from .precompute import build_precompute_matrix, get_precompute_matrix

LOG_FORMAT = "%(asctime)s - %(levelname)s - %(message)s"  # custom_logging.py:8
logger = logging.getLogger("fedrann_amd")
logger.setLevel(logging.DEBUG)


def _setup_logging(logfile=None):
    if not logger.handlers:
        h = logging.StreamHandler()
        h.setFormatter(logging.Formatter(LOG_FORMAT))
        logger.addHandler(h)
    if logfile:
        fh = logging.FileHandler(logfile)
        fh.setFormatter(logging.Formatter(LOG_FORMAT))
        logger.addHandler(fh)


def build_parser():
    p = argparse.ArgumentParser(prog="fedrann", description=__description__,
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("-i", "--input", type=str, required=False, default=None,
                   help="Path to the input FASTQ/FASTA file (needs the reference's stage-1 tools; "
                        "use --kmer-searcher-output or --feature-matrix instead).")
    p.add_argument("-o", "--output-dir", type=str, required=True, help="Directory to save output files.")
    p.add_argument("-k", "--kmer-size", type=int, default=16, help="K-mer size for feature extraction.")
    p.add_argument("--kmer-sample-fraction", type=float, default=0.005,
                   help="Percentage of k-mer used to build feature matrix.")
    p.add_argument("--kmer-min-multiplicity", type=int, default=2,
                   help="Minimum allowed frequency of a k-mer in all reads.")
    p.add_argument("--threads", type=int, default=1)
    p.add_argument("--chunk-size", type=int, default=1000)
    p.add_argument("-n", "--embedding-dimension", type=int, default=500)
    p.add_argument("--nndescent-n-trees", type=int, default=300, help="Number of trees to use in NNDescent.")
    p.add_argument("--nndescent-n-neighbors", type=int, default=50,
                   help="Number of neighbors to use in building NNDescent index.")
    p.add_argument("--seed", type=int, default=356115, help="Random seed for reproducibility.")
    p.add_argument("--save-feature-matrix", action="store_true", default=False,
                   help="Save the feature matrix to a file.")
    p.add_argument("--keep-intermediates", action="store_true", default=False,
                   help="Do not remove intermediate files")
    p.add_argument("--mprof", action="store_true", default=False, help="Record memory usage.")
    g = p.add_argument_group("hot-path entry points (GPU build)")
    g.add_argument("--kmer-searcher-output", type=str, default=None,
                   help="kmer_searcher output.bin (reference intermediate temp/kmer_searcher/output.bin).")
    g.add_argument("--kmer-library", type=str, default=None,
                   help="fwd_kmer_library.fasta (jellyfish dump with counts) for the IDF weights.")
    g.add_argument("--feature-matrix", type=str, default=None,
                   help="feature_matrix.npz: binary read x feature CSR (scipy.sparse.save_npz).")
    g.add_argument("--kmer-counts", type=str, default=None,
                   help="With --feature-matrix: .npy of forward-library counts (int, length F/2) or a "
                        "fwd_kmer_library.fasta.")
    g.add_argument("--read-names", type=str, default=None,
                   help="With --feature-matrix: text file, one 'name<TAB>strand' (or just name) per row.")
    g.add_argument("--device", type=int, default=None, help="GPU ordinal (default $LOCAL_RANK or 0).")
    g.add_argument("--devices", type=str, default=None,
                   help="Comma-separated GPU ordinals: shard the rows over these GPUs (one process each).")
    g.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                   help="With --devices: nccl (= RCCL over xGMI; one GPU per process) or gloo (processes may "
                        "share a GPU: one-GPU rehearsal of the sharded path).")
    g.add_argument("--rank-worker", action="store_true", default=False, help=argparse.SUPPRESS)
    g.add_argument("--stage1-worker", action="store_true", default=False, help=argparse.SUPPRESS)
    return p


def parse_command_line_arguments(argv=None):
    return build_parser().parse_args(argv)


def get_neighbors_ava(embedding_matrix, nndescent_n_trees, nndescent_n_neighbors, leaf_size=200):
    """Same call shape as the reference (__main__.py:174-199)."""
    logger.info("Using exact GPU k-NN in place of NNDescent (n_trees = %s, leaf_size = %s are inert)",
                nndescent_n_trees, leaf_size)
    return NNDescent_ava().get_neighbors(
        embedding_matrix, metric="cosine", index_n_neighbors=nndescent_n_neighbors,
        n_trees=nndescent_n_trees, leaf_size=leaf_size, n_iters=None, diversify_prob=1.0,
        pruning_degree_multiplier=1.5, low_memory=True, n_jobs=global_variables.threads,
        seed=global_variables.seed, verbose=True)


def get_output_dataframe(neighbor_matrix, neighbor_distances, read_names: List[str],
                         strands: List[int]) -> pd.DataFrame:
    """Vectorised equivalent of the reference's N x k Python loop (__main__.py:261-300).

    Same rows in the same order with the same dtypes: a neighbour is skipped only when it is the
    query row itself; neighbor_rank keeps the column number; the distance column is float32 (pandas
    prints its shortest repr); a negative index aliases from the end like Python indexing does.
    """
    idx = np.asarray(neighbor_matrix)
    dist = np.asarray(neighbor_distances)
    n = idx.shape[0]
    names = np.asarray(read_names, dtype=object)
    orient = np.where(np.asarray(strands, dtype=np.int64) == 0, "+", "-").astype(object)
    keep = idx != np.arange(n, dtype=idx.dtype)[:, None]
    q, r = np.nonzero(keep)  # row-major = the loop's order
    t = idx[q, r]
    columns = {
        "query_name": names[q],
        "query_orientation": orient[q],
        "target_name": names[t],
        "target_orientation": orient[t],
        "neighbor_rank": r.astype(np.int64),
        "distance": dist[q, r],
    }
    df = pd.DataFrame(columns)
    logger.debug("Output DataFrame shape: %s", df.shape)
    return df


def write_overlaps(path, neighbor_matrix, distances, read_names, strands, row0=0, header=True):
    """overlaps.tsv through the native writer (fdr_overlaps_write): the bytes get_output_dataframe(...)
    .to_csv(path, sep="\\t", index=False) produces (__main__.py:261-300, :385), without the DataFrame.
    neighbor_matrix / distances may be the rows row0.. of the graph (a rank's block); read_names / strands
    describe all rows.  Returns the number of data lines."""
    from . import _lib
    name_off, names = _lib.pack_names(read_names)
    return _lib.overlaps_write(path, np.asarray(neighbor_matrix), np.asarray(distances), name_off, names,
                               np.asarray(strands, dtype=np.uint8), row0=row0, header=header,
                               n_threads=global_variables.threads if global_variables.threads > 1 else 0)


def _load_counts(path):
    if path.endswith(".npy"):
        return np.load(path).astype(np.int64)
    from .precompute import read_kmer_counts
    return read_kmer_counts(path)


def _load_names(path, nrows):
    if path is None:
        return ["row_%d" % i for i in range(nrows)], [0] * nrows
    names, strands = [], []
    with open(path) as f:
        for line in f:
            parts = line.rstrip("\n").split("\t")
            names.append(parts[0])
            strands.append(int(parts[1]) if len(parts) > 1 and parts[1] != "" else 0)
    if len(names) != nrows:
        raise ValueError("%s has %d lines, the feature matrix has %d rows" % (path, len(names), nrows))
    return names, strands


def check_limits(embedding_dimension, nndescent_n_neighbors):
    """The k-NN kernels' limits, checked before any work is done (the reference accepts any value)."""
    from . import _lib
    if not 1 <= embedding_dimension <= _lib.FDR_MAX_DIM:
        raise SystemExit("-n/--embedding-dimension must be in 1..%d for the GPU k-NN kernels (got %d)"
                         % (_lib.FDR_MAX_DIM, embedding_dimension))
    if not 1 <= nndescent_n_neighbors <= _lib.FDR_MAX_K:
        raise SystemExit("--nndescent-n-neighbors must be in 1..%d for the GPU k-NN kernels (got %d)"
                         % (_lib.FDR_MAX_K, nndescent_n_neighbors))


def load_inputs(*, output_dir, embedding_dimension, save_feature_matrix, kmer_searcher_output=None,
                kmer_library=None, feature_matrix=None, kmer_counts=None, read_names_path=None, save=True):
    """Stages 2-3a of the reference pipeline on the host (__main__.py:329-345): the projection matrix and
    the read x feature CSR.  Returns (indptr, indices, n_features, P, read_names, strands)."""
    if kmer_searcher_output:
        from .precompute import read_kmer_counts
        n_features = 2 * int(read_kmer_counts(kmer_library).size)  # (count_kmers.py:148)
        logger.info("--- 2. Generate dimension reduction and IDF matrix ---")
        P, n_features = get_precompute_matrix(n_components=embedding_dimension, counter_file=kmer_library,
                                              n_features=n_features)
        logger.info("--- 3. Generate feature matrix ---")
        indptr, indices, read_names, strands = build_feature_csr(kmer_searcher_output, n_features)
    else:
        indptr, indices, n_features = load_feature_matrix_npz(feature_matrix)
        counts = _load_counts(kmer_counts)
        logger.info("--- 2. Generate dimension reduction and IDF matrix ---")
        P = build_precompute_matrix(counts, embedding_dimension, n_features=n_features)
        logger.info("--- 3. Generate feature matrix ---")
        read_names, strands = _load_names(read_names_path, indptr.size - 1)
    if save_feature_matrix and save:
        save_feature_matrix_npz(join(output_dir, "feature_matrix.npz"), indptr, indices, n_features)
    return indptr, indices, n_features, P, read_names, strands


def run_fedrann_pipeline(*, output_dir, embedding_dimension, nndescent_n_trees,
                         nndescent_n_neighbors, save_feature_matrix, keep_intermediates, chunk_size,
                         kmer_searcher_output=None, kmer_library=None, feature_matrix=None,
                         kmer_counts=None, read_names_path=None):
    """Stages 2-4 of the reference pipeline (__main__.py:329-391) on one GPU.  The embeddings never leave
    HBM between the projection and the search (fdr_embed_knn), and only the features P has entries for
    cross PCIe (fdr_csr_compact; the saved feature_matrix.npz is the full matrix)."""
    from . import _lib
    from .feature_extraction import _projection_csr
    if kmer_searcher_output:
        logger.info("--- 1. (skipped) using kmer_searcher output %s ---", kmer_searcher_output)
    else:
        logger.info("--- 1. (skipped) using feature matrix %s ---", feature_matrix)
    indptr, indices, n_features, P, read_names, strands = load_inputs(
        output_dir=output_dir, embedding_dimension=embedding_dimension, save_feature_matrix=save_feature_matrix,
        kmer_searcher_output=kmer_searcher_output, kmer_library=kmer_library, feature_matrix=feature_matrix,
        kmer_counts=kmer_counts, read_names_path=read_names_path)
    ctx = _lib.default_context()
    Pc = _projection_csr(P)
    ctx.projection_load(Pc.indptr, Pc.indices, Pc.data, n_features, embedding_dimension)
    cip, cix = ctx.csr_compact(indptr, indices)
    logger.debug("embedding %d rows: %d of %d feature ids have an entry in P", indptr.size - 1, cix.size, indices.size)
    del indptr, indices
    logger.info("--- 4. Nearest Neighbors Search ---")
    logger.info("Using exact GPU k-NN in place of NNDescent (n_trees = %s, leaf_size = %s are inert)",
                nndescent_n_trees, 200)
    neighbor_matrix, distances = ctx.embed_knn(cip, cix, nndescent_n_neighbors)
    nbr_output_file = join(output_dir, "overlaps.tsv")
    logger.debug("Saving overlap table to %s", nbr_output_file)
    rows = write_overlaps(nbr_output_file, neighbor_matrix, distances, read_names, strands)
    logger.debug("wrote %d overlap rows", rows)
    if not keep_intermediates and global_variables.temp_dir and os.path.isdir(global_variables.temp_dir):
        logger.debug("Removing intermediate files")
        rmtree(global_variables.temp_dir)
    logger.info("Pipeline completed.")


def record_names(name_off, name_buf):
    """(name_off, names) of the R records as fdr_overlaps_write takes them in its doubled-rows mode (strands=None: row
    t carries record t >> 1's id -- no id is materialised twice, let alone gathered byte by byte).  Valid UTF-8 ids
    pass through untouched; ids that are not follow the reference's rule (feature_extraction.py:125-128) through the
    str path."""
    from . import _lib
    from .feature_extraction import _decode_names
    raw = name_buf.tobytes()
    if not raw.isascii():
        try:
            raw.decode("utf-8")
        except UnicodeDecodeError:
            return _lib.pack_names(_decode_names(name_off, name_buf))
    return name_off, name_buf


def load_rank_inputs(args, output_dir, rank, world):
    """What ONE rank of `--devices` needs on the host: the projection matrix, ITS row block of the read x feature
    CSR (1 / world of the matrix: the ranged native loader for output.bin; a feature_matrix.npz is inflated whole
    and sliced, scipy's format has no random access) and the names / strands of all rows for the writer.
    Returns (n_rows, lo, hi, indptr, indices, n_features, P, name_off, names, strands); strands is None when
    name_off / names describe the R records of a doubled matrix (fdr_overlaps_write's doubled-rows mode)."""
    from . import _lib
    from .distributed import local_csr, shard_rows
    if args.kmer_searcher_output:
        from .precompute import read_kmer_counts
        n_features = 2 * int(read_kmer_counts(args.kmer_library).size)  # (count_kmers.py:148)
        P, n_features = get_precompute_matrix(n_components=args.embedding_dimension, counter_file=args.kmer_library,
                                              n_features=n_features)
        try:
            R = _lib.kmer_output_records(args.kmer_searcher_output)  # (the header's count: no record is walked for it)
            n = 2 * R
            _, blocks = shard_rows(n, world)
            lo, hi = blocks[rank]
            _, ip, ix, name_off, name_buf = _lib.kmer_output_load_range(
                args.kmer_searcher_output, n_features, lo // 2, hi // 2,
                n_threads=global_variables.threads if global_variables.threads > 1 else 0)
        except _lib.FedrannHipError as e:
            if "output.bin:" in str(e):  # format errors keep the reference's exception type
                raise ValueError(str(e).split("output.bin:", 1)[1].strip()) from None
            raise
        name_off, names = record_names(name_off, name_buf)
        strands = None  # (fdr_overlaps_write's doubled-rows mode: row t = record t >> 1 on strand t & 1)
        if args.save_feature_matrix and rank == 0:  # (the whole matrix, once)
            from .feature_extraction import build_feature_csr
            fip, fix, _, _ = build_feature_csr(args.kmer_searcher_output, n_features)
            save_feature_matrix_npz(join(output_dir, "feature_matrix.npz"), fip, fix, n_features)
            del fip, fix
    else:
        indptr, indices, n_features = load_feature_matrix_npz(args.feature_matrix)
        P = build_precompute_matrix(_load_counts(args.kmer_counts), args.embedding_dimension, n_features=n_features)
        n = indptr.size - 1
        _, blocks = shard_rows(n, world)
        lo, hi = blocks[rank]
        if args.save_feature_matrix and rank == 0:
            save_feature_matrix_npz(join(output_dir, "feature_matrix.npz"), indptr, indices, n_features)
        ip, ix = local_csr(indptr, indices, lo, hi)
        del indptr, indices
        read_names, strand_list = _load_names(args.read_names, n)
        name_off, names = _lib.pack_names(read_names)
        strands = np.asarray(strand_list, dtype=np.uint8)
    return n, lo, hi, ip, ix, n_features, P, name_off, names, strands


def run_rank_worker(args, output_dir, temp_dir):
    """One rank of `--devices`: RANK / WORLD_SIZE / FEDRANN_DEVICE / FEDRANN_RENDEZVOUS come from the parent, and
    so does stage 1's output (the parent runs it in a child of its own before it starts the ranks).  A rank
    loads ITS rows of the feature matrix, embeds them, the normalised blocks are all-gathered, it searches its
    rows against all rows (distributed.ShardedPipeline) and writes temp/overlaps.rank<r>.tsv."""
    import datetime
    import torch
    import torch.distributed as dist
    from . import _lib
    from .distributed import HipEngine, ShardedPipeline
    from .feature_extraction import _projection_csr
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    device = torch.device("cuda", int(os.environ["FEDRANN_DEVICE"]))
    torch.cuda.set_device(device)
    # rendezvous through a file in temp/ (no port to lose to another process); the host stages between two
    # collectives (loading a rank's rows of a 10 M-read matrix, writing its part of overlaps.tsv) may take long
    # (FEDRANN_COLLECTIVE_TIMEOUT_S: how long a rank waits inside one collective, default 2 h -- the only long host stage
    # between two collectives is the load below, and a barrier right after it takes that wait; a rank that hangs without
    # dying no longer holds its siblings for half a day)
    kw = dict(init_method="file://" + os.environ["FEDRANN_RENDEZVOUS"], rank=rank, world_size=world,
              timeout=datetime.timedelta(seconds=float(os.environ.get("FEDRANN_COLLECTIVE_TIMEOUT_S", "7200"))))
    if args.dist_backend == "nccl":
        dist.init_process_group("nccl", device_id=device, **kw)
    else:
        dist.init_process_group("gloo", **kw)
    n, lo, hi, ip, ix, n_features, P, name_off, names, strands = load_rank_inputs(args, output_dir, rank, world)
    dist.barrier()  # every rank has its rows: what follows is GPU work and short collectives
    k = args.nndescent_n_neighbors
    ctx = _lib.Context(device.index)
    Pc = _projection_csr(P)
    ctx.projection_load(Pc.indptr, Pc.indices, Pc.data, n_features, args.embedding_dimension)
    pipe = ShardedPipeline(HipEngine(ctx, device), n, args.embedding_dimension, k, rank=rank, world_size=world,
                           device=device)
    assert (pipe.lo, pipe.hi) == (lo, hi)
    logger.debug("rank %d of %d holds rows [%d, %d) of %d: %d column ids", rank, world, lo, hi, n, ix.size)
    ip, ix = ctx.csr_compact(ip, ix)
    if rank == 0:
        logger.info("--- 4. Nearest Neighbors Search (%d rows over %d GPUs) ---", n, world)
    idx, dst, _ = pipe.step(torch.from_numpy(ip).to(device), torch.from_numpy(ix).to(device))
    torch.cuda.synchronize(device)
    part = join(temp_dir, "overlaps.rank%d.tsv" % rank)
    rows = _lib.overlaps_write(part, idx.cpu().numpy(), dst.cpu().numpy(), name_off, names, strands, row0=lo,
                               header=rank == 0,
                               n_threads=global_variables.threads if global_variables.threads > 1 else 0)
    logger.debug("rank %d: rows [%d, %d), %d overlap rows", rank, lo, hi, rows)
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


def run_stage1_worker(args, temp_dir):
    """Stage 1 of a `--devices` run (k-mer counting / sampling / search on the first GPU), as a child of its own
    that has exited before the ranks start: no rank waits inside a collective while it runs."""
    if args.kmer_library:
        logger.info("--- 1b. k-mer search on the GPU ---")
        _gpu_kmer_search(args.input, args.kmer_library, args.kmer_size, temp_dir)
    else:
        from .count_kmers import run_kmer_searcher
        logger.info("--- 1. Counter kmers (GPU) ---")
        run_kmer_searcher(input_path=args.input, k=args.kmer_size, sample_fraction=args.kmer_sample_fraction,
                          min_multiplicity=args.kmer_min_multiplicity)


def launch_rank_workers(argv, args, devices, output_dir, temp_dir, keep_intermediates):
    """The parent of `--devices`: no GPU call here (a process that has initialised the GPU must not start
    others on this pool, and the children own the devices).  Children = this module with --stage1-worker
    (once, when reads were given) and --rank-worker (one per device)."""
    import time
    argv = list(argv)
    if args.input and not args.kmer_searcher_output and not args.feature_matrix:
        env = dict(os.environ, FEDRANN_DEVICE=str(devices[0]))
        rc = subprocess.call([sys.executable, "-m", "fedrann_amd"] + argv + ["--stage1-worker"], env=env)
        if rc:
            raise SystemExit("stage 1 failed: exit code %d" % rc)
        argv += ["--kmer-searcher-output", join(temp_dir, "kmer_searcher", "output.bin")]
        if not args.kmer_library:
            argv += ["--kmer-library", join(temp_dir, "fwd_kmer_library.fasta")]
    rendezvous = join(temp_dir, "rendezvous.%d" % os.getpid())
    if os.path.exists(rendezvous):
        os.remove(rendezvous)
    procs = []
    for rank, dev in enumerate(devices):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(len(devices)), LOCAL_RANK=str(rank),
                   FEDRANN_DEVICE=str(dev), FEDRANN_RENDEZVOUS=rendezvous)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (RCCL between processes needs dmabuf IPC on hosts like this pool's)
        procs.append(subprocess.Popen([sys.executable, "-m", "fedrann_amd"] + argv + ["--rank-worker"], env=env))
    # poll all ranks: the first failure ends the others (they would sit in a collective until its timeout); whatever ends
    # the parent -- KeyboardInterrupt, SIGTERM turned into SystemExit, an error here -- ends the ranks and removes the
    # rendezvous file too
    codes = [None] * len(procs)

    def end_ranks():
        for i, p in enumerate(procs):
            if codes[i] is None and p.poll() is None:
                p.terminate()
        for i, p in enumerate(procs):
            if codes[i] is None:
                try:
                    codes[i] = p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
                    codes[i] = p.wait()

    import signal
    old_term = signal.signal(signal.SIGTERM, lambda *_: sys.exit(143))
    try:
        while any(c is None for c in codes):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    codes[i] = p.poll()
            if any(c not in (None, 0) for c in codes):
                end_ranks()
                break
            time.sleep(0.05)
    finally:
        end_ranks()
        signal.signal(signal.SIGTERM, old_term)
        if os.path.exists(rendezvous):
            os.remove(rendezvous)
    if any(codes):
        raise SystemExit("rank worker(s) failed: exit codes %s" % codes)
    out = join(output_dir, "overlaps.tsv")
    with open(out, "wb") as dst:
        for rank in range(len(devices)):
            with open(join(temp_dir, "overlaps.rank%d.tsv" % rank), "rb") as src:
                copyfileobj(src, dst, 1 << 24)
    if not keep_intermediates and os.path.isdir(temp_dir):
        rmtree(temp_dir)
    logger.info("Pipeline completed.")


def _gpu_kmer_search(reads_path, fwd_library, k, temp_dir):
    """reads + forward library -> temp/kmer_searcher/output.bin (and kmer_frequency.bin).  The reverse
    library is the reverse complement of every forward k-mer, in the same order (`seqkit seq -r -p`,
    count_kmers.py:127); both are passed in the reference's order: forward, then reverse."""
    from .kmer_search import kmer_searcher
    comp = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")
    with open(fwd_library, "rb") as f:
        lines = f.read().split(b"\n")
    rev_path = join(temp_dir, "rev_kmer_library.fasta")
    with open(rev_path, "wb") as f:
        f.write(b"\n".join(l if l.startswith(b">") else l.translate(comp)[::-1] for l in lines))
    if reads_path.endswith(".gz"):
        import gzip
        plain = join(temp_dir, os.path.basename(reads_path[:-3]))
        with gzip.open(reads_path, "rb") as src, open(plain, "wb") as dst:
            copyfileobj(src, dst, 1 << 24)
        reads_path = plain
    out_dir = join(temp_dir, "kmer_searcher")
    n_reads, _, nnz, n_lib = kmer_searcher([fwd_library, rev_path], reads_path, out_dir, k, fastq_ids_as_fasta=True,
                                           collect=False)  # (the reference runs seqkit fq2fa first; reads streamed)
    logger.debug("k-mer search: %d reads, %d library k-mers, %d hits", n_reads, n_lib, nnz)
    return join(out_dir, "output.bin")


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_command_line_arguments(argv)
    global_variables.threads = args.threads
    global_variables.seed = args.seed
    check_limits(args.embedding_dimension, args.nndescent_n_neighbors)  # before any work (the library would
    # only refuse them after stages 1-3)
    if args.device is not None and not (args.rank_worker or args.stage1_worker):
        os.environ["FEDRANN_DEVICE"] = str(args.device)
    output_dir = abspath(args.output_dir)
    os.makedirs(output_dir, exist_ok=True)
    global_variables.output_dir = output_dir
    _setup_logging(join(output_dir, "fedrann.log"))
    temp_dir = join(output_dir, "temp")
    os.makedirs(temp_dir, exist_ok=True)
    global_variables.temp_dir = temp_dir
    have_ks = bool(args.kmer_searcher_output)
    have_fm = bool(args.feature_matrix)
    if args.rank_worker and have_ks:
        args.input = None  # (stage 1 has run: the parent passes its output next to the user's arguments)
    if sum((bool(args.input), have_ks, have_fm)) != 1:
        raise SystemExit(
            "give exactly one of -i reads, -i reads + --kmer-library, --kmer-searcher-output + --kmer-library, "
            "or --feature-matrix + --kmer-counts")
    if have_ks and not args.kmer_library:
        raise SystemExit("--kmer-searcher-output needs --kmer-library")
    if have_fm and not args.kmer_counts:
        raise SystemExit("--feature-matrix needs --kmer-counts")
    if args.rank_worker:
        return run_rank_worker(args, output_dir, temp_dir)
    if args.stage1_worker:
        return run_stage1_worker(args, temp_dir)
    logger.info("FEDRANN (MI355X hot path) version: %s", __version__)
    logger.debug("Parameters: %s", args)
    if args.devices:
        devices = [int(x) for x in args.devices.split(",") if x.strip() != ""]
        if len(devices) > 1:
            return launch_rank_workers(argv, args, devices, output_dir, temp_dir, args.keep_intermediates)
        if devices:
            os.environ["FEDRANN_DEVICE"] = str(devices[0])
    if args.input and args.kmer_library and not have_ks and not have_fm:
        # stage 1b on the GPU: reads x sampled k-mer library -> output.bin (count_kmers.py:119-139 with the
        # reverse library made here instead of by seqkit, the search by fdr_kmer_search instead of kmer_searcher)
        logger.info("--- 1b. k-mer search on the GPU ---")
        args.kmer_searcher_output = _gpu_kmer_search(args.input, args.kmer_library, args.kmer_size, temp_dir)
        have_ks = True
    if args.input and not args.kmer_library and not have_ks and not have_fm:
        # the whole stage 1 on the GPU (count_kmers.py:52-148): canonical k-mer counts, threshold, Bernoulli
        # sample, reverse library, search
        from .count_kmers import run_kmer_searcher
        logger.info("--- 1. Counter kmers (GPU) ---")
        args.kmer_searcher_output, n_features, read_count = run_kmer_searcher(
            input_path=args.input, k=args.kmer_size, sample_fraction=args.kmer_sample_fraction,
            min_multiplicity=args.kmer_min_multiplicity)
        logger.debug("kmer_searcher n_features: %d, reads: %d", n_features, read_count)
        args.kmer_library = join(temp_dir, "fwd_kmer_library.fasta")
        have_ks = True
    run_fedrann_pipeline(
        output_dir=output_dir, embedding_dimension=args.embedding_dimension,
        nndescent_n_trees=args.nndescent_n_trees, nndescent_n_neighbors=args.nndescent_n_neighbors,
        save_feature_matrix=args.save_feature_matrix, keep_intermediates=args.keep_intermediates,
        chunk_size=args.chunk_size, kmer_searcher_output=args.kmer_searcher_output,
        kmer_library=args.kmer_library, feature_matrix=args.feature_matrix,
        kmer_counts=args.kmer_counts, read_names_path=args.read_names)


if __name__ == "__main__":
    main()
