/*
 * fedrann_hip.h -- C-ABI of libfedrann_hip.so, the MI355X (gfx950) implementation of FEDRANN's
 * dimensionality-reduction + k-NN hot path.
 *
 * The reference (jzhang-dev/FEDRANN v0.5.4) has no FFI for this path: the seam is three in-process
 * Python calls in fedrann/__main__.py (run_fedrann_pipeline):
 *     get_precompute_matrix(...)            __main__.py:331-335  -> precompute.py:58-115
 *     get_feature_matrix(...)               __main__.py:339-345  -> feature_extraction.py:216-292
 *     get_neighbors_ava(...)                __main__.py:361-365  -> nearest_neighbors.py:22-55
 * Each entry point below names the reference call it replaces.  The Python host
 * (fedrann_amd/) keeps those three call shapes and binds this header through ctypes; the stub a
 * reference maintainer would add is in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no C++ or torch types; every function returns 0 on success or a negative FDR_E_*
 *     code, and fdr_last_error() returns the message of the calling thread's last failure.
 *   - the caller owns every buffer it passes; the library borrows pointers for the duration of the
 *     call only.  "host" functions take host pointers and synchronise before returning; "_dev"
 *     functions take device pointers (hipMalloc'd or a torch tensor's data_ptr()) plus a
 *     hipStream_t passed as void* and enqueue work on that stream.  fdr_embed_dev and fdr_normalize_dev
 *     return without synchronising; fdr_knn_dev synchronises the stream twice on the common path (it sizes its
 *     follow-up passes from counters it reads back: unique-row counts; uncertified / all-zero / plateau queries; below
 *     2^18 targets a duplicate-row probe, and once more when a plateau's range overflows), so work the caller wants
 *     to overlap with it belongs on another stream.
 *   - one context = one GPU; one context per process is the intended use (one process per GPU).
 *     A context is not re-entrant: one call in flight at a time.
 *   - all arrays are C-contiguous with exactly the element types written here.
 */
#ifndef FEDRANN_HIP_H
#define FEDRANN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDR_OK 0
#define FDR_E_ARG (-1)     /* bad argument (null pointer, size, unsupported d or k, ...) */
#define FDR_E_HIP (-2)     /* a HIP runtime call or kernel launch failed */
#define FDR_E_NOMEM (-4)   /* device or host allocation failed */
#define FDR_E_STATE (-5)   /* call order (e.g. embed before a projection was loaded) */
#define FDR_E_IO (-6)      /* a file could not be opened / mapped */

#define FDR_MAX_K 128      /* neighbours per row (self included); up to 64 on the MFMA kernels, beyond on a generic one */
#define FDR_MAX_DIM 2048   /* embedding dimension; up to 512 (reference default: 500) on the MFMA kernels, beyond on a
                              generic vector-ALU kernel (same results, far slower: DESIGN.md) */

typedef struct fdr_ctx fdr_ctx;

/* ---- lifetime -------------------------------------------------------------------------- */
int fdr_create(int device_id, fdr_ctx **out);
int fdr_destroy(fdr_ctx *ctx);
const char *fdr_last_error(void);
/* "name|gcnArch|CUs|HBM bytes" of the context's device, NUL-terminated into buf. */
int fdr_device_info(fdr_ctx *ctx, char *buf, int buflen);
/* Padded row length (floats) of the internal normalised-embedding layout for dimension d:
 * 128 for d <= 128, 256 for d <= 256, 512 for d <= 512, 1024 for d <= 1024, 2048 for d <= 2048; negative if d is
 * unsupported. */
int fdr_padded_dim(int d);

/* ---- projection (replaces handing `precompute_matrix` to get_feature_matrix,
 *      feature_extraction.py:226-241: P as CSR by feature, float32 data) ---------------------
 * P is n_features x d; p_indptr int64[n_features+1], p_cols int32[nnz] in [0,d), p_vals
 * float32[nnz].  Builds the device-side lookup tables used by the embed kernel. */
int fdr_projection_load(fdr_ctx *ctx, int64_t n_features, int32_t d, const int64_t *p_indptr,
                        const int32_t *p_cols, const float *p_vals);

/* Drop the column ids of a read x feature CSR whose projection row is empty (host only; needs a loaded
 * projection).  P's density is 1/sqrt(F) (precompute.py:80-84): >= 90 % of the features have no entry in P
 * and add nothing to A.dot(P) (feature_extraction.py:204-213), so a caller that compacts its CSR before
 * fdr_embed / fdr_embed_knn moves ~10x fewer bytes over PCIe; E is bit for bit the same (the surviving
 * ids keep their order).  out_indptr int64 [n_rows + 1]; out_indices int32 with room for out_capacity ids
 * (a_indptr[n_rows] always suffices); the outputs must not alias the inputs.  n_threads <= 0: all
 * hardware threads. */
int fdr_csr_compact(fdr_ctx *ctx, int64_t n_rows, const int64_t *a_indptr, const int32_t *a_indices,
                    int64_t *out_indptr, int32_t *out_indices, int64_t out_capacity, int32_t n_threads);

/* Pin / unpin caller-owned host memory (hipHostRegister) so that the host-pointer calls below copy at PCIe
 * rate instead of through a staging buffer.  The caller still owns the memory and must unpin it before
 * freeing it. */
int fdr_host_register(fdr_ctx *ctx, void *ptr, size_t bytes);
int fdr_host_unregister(fdr_ctx *ctx, void *ptr);

/* ---- E = A . P  (replaces process_read_chunk_optimized + the scatter loop,
 *      feature_extraction.py:167-213, :280-290) --------------------------------------------
 * A is the binary read x feature CSR: a_indptr int64[n_rows+1], a_indices int32[nnz], column ids
 * ASCENDING inside each row (scipy canonical form; the host wrapper sorts).  E_out float32
 * [n_rows, d] row-major.  E[r,c] is the sequential fp32 sum, in ascending feature order, of
 * P[f,c] over the features f of row r -- bit-identical to the reference (golden vectors). */
int fdr_embed(fdr_ctx *ctx, int64_t n_rows, const int64_t *a_indptr, const int32_t *a_indices,
              float *E_out);

/* ---- exact cosine k-NN  (replaces NNDescent_ava().get_neighbors(E, metric="cosine",
 *      index_n_neighbors=k, ...).neighbor_graph, nearest_neighbors.py:39-55) ---------------
 * E float32 [n, d] (not normalised).  idx_out int32 [n,k], dist_out float32 [n,k], each row
 * ascending by (distance, index); self is a candidate like any other row.  Canonical arithmetic
 * (DESIGN.md section 4): rows are scaled by (float)(1/sqrt((double)chain(x,x))), the
 * similarity is the fp32 fma chain over components 0..d-1, dist = clamp(1 - c, 0, 1), two
 * all-zero rows are at distance 0.  Requires n >= k, 1 <= k <= FDR_MAX_K, d <= FDR_MAX_DIM (k > 64 or d > 512: the
 * generic kernel, every pair on the vector ALU). */
int fdr_knn(fdr_ctx *ctx, const float *E, int64_t n, int32_t d, int32_t k, int32_t *idx_out,
            float *dist_out);

/* embed + k-NN with E kept in HBM between the two (what run_fedrann_pipeline does in steps 3-4,
 * __main__.py:338-367).  E_out may be NULL. */
int fdr_embed_knn(fdr_ctx *ctx, int64_t n_rows, const int64_t *a_indptr, const int32_t *a_indices,
                  int32_t k, int32_t *idx_out, float *dist_out, float *E_out);

/* ---- device-resident API (multi-GPU host, bench.py) ----------------------------------------
 * All pointers are device pointers; work is enqueued on `stream` (a hipStream_t). */
int fdr_embed_dev(fdr_ctx *ctx, int64_t n_rows, const int64_t *d_indptr, const int32_t *d_indices,
                  float *d_E, void *stream);
/* E [n_rows,d] -> Ehat [n_rows, fdr_padded_dim(d)] in the kernel's internal layout (normalised,
 * zero padded, components permuted inside groups of 8) + zero-row flags uint8[n_rows]. */
int fdr_normalize_dev(fdr_ctx *ctx, const float *d_E, int64_t n_rows, int32_t d, float *d_Ehat,
                      uint8_t *d_zero, void *stream);
/* bytes of scratch fdr_knn_dev needs for (nq queries, nt targets, k). */
size_t fdr_knn_workspace_bytes(fdr_ctx *ctx, int64_t nq, int64_t nt, int32_t d, int32_t k);
/* k-NN of nq query rows against nt target rows, both in the Ehat layout.  Neighbour indices are
 * target row numbers + t_base.  Rows of a row-sharded run: queries = the rank's shard, targets =
 * the all-gathered Ehat of every rank.  The result arrays are complete when the call returns AND the
 * stream has finished (see "Conventions": the call itself waits for the stream a few times). */
int fdr_knn_dev(fdr_ctx *ctx, const float *d_Qhat, const uint8_t *d_qzero, int64_t nq,
                const float *d_That, const uint8_t *d_tzero, int64_t nt, int64_t t_base, int32_t d,
                int32_t k, int32_t *d_idx, float *d_dist, void *d_workspace, size_t workspace_bytes,
                void *stream);
/* ---- duplicate-row classes across the ranks of a row-sharded run -----------------------------------
 * fdr_knn_dev searches a duplicate QUERY row once per rank that holds a member of its class.  With these
 * three calls the ranks split the UNIQUE rows instead (same results):
 *   fdr_knn_classes_dev  builds the classes of the target set (the all-gathered Ehat: the same tables on
 *                        every rank) in the workspace -- fdr_knn_workspace_bytes(ctx, nq_max, nt, d, k) bytes,
 *                        nq_max = the most unique rows one later call will search -- and returns their number
 *                        in *n_unique_out; 0 = not worth it (small set, or unique^2 > 0.9 rows^2): use fdr_knn_dev.
 *                        The decision is a function of the exact unique count: the same on every rank.
 *   fdr_knn_unique_dev   k-NN of the unique rows [u_lo, u_hi) (ascending representative order) against all
 *                        unique rows: d_idx_u int32 [u_hi - u_lo, k] (unique-row numbers), d_dist_u float32.
 *   fdr_knn_expand_dev   given the results of ALL unique rows (the ranks' shares concatenated: [n_unique, k]),
 *                        the neighbours of the original rows [q0, q0 + nq): d_idx (+ t_base), d_dist [nq, k].
 *                        u_row_stride = elements between two unique rows' results in d_idx_u_all / d_dist_u_all
 *                        (0 = k; 2 k when a row's indices and distance bits travel side by side in ONE exchange:
 *                        d_dist_u_all = (float *)(d_idx_u_all + k)).
 * The workspace must stay untouched between the three calls; fdr_knn_classes_dev synchronises the stream. */
int fdr_knn_classes_dev(fdr_ctx *ctx, const float *d_That, const uint8_t *d_tzero, int64_t nt, int32_t d, int32_t k,
                        int64_t nq_max, void *d_workspace, size_t workspace_bytes, void *stream,
                        int32_t *n_unique_out);
int fdr_knn_unique_dev(fdr_ctx *ctx, int64_t u_lo, int64_t u_hi, int32_t *d_idx_u, float *d_dist_u, void *stream);
int fdr_knn_expand_dev(fdr_ctx *ctx, int64_t q0, int64_t nq, int64_t t_base, const int32_t *d_idx_u_all,
                       const float *d_dist_u_all, int64_t u_row_stride, int32_t *d_idx, float *d_dist, void *stream);

/* ---- per-kernel timing (bench.py's roofline figures) -----------------------------------------
 * With timing enabled every kernel launch is bracketed by its own hipEvent pair recorded on the
 * stream the kernel is launched on.  fdr_timing_read() waits for the recorded launches of one kernel
 * kind, returns how many there were and their summed duration, and clears the tally. */
#define FDR_KERNEL_EMBED 0
#define FDR_KERNEL_NORMALIZE 1
#define FDR_KERNEL_KNN_TILE 2
#define FDR_KERNEL_KNN_MERGE 3
#define FDR_KERNEL_KNN_PREFILTER 4 /* fp16 MFMA candidate pass of the prefilter mode: one span per launch, or one span
                                      over the whole pass when its launches overlap (fdr_last_prefilter_launches) */
#define FDR_KERNEL_KNN_RERANK 5    /* rest of the prefilter mode: fp16 conversion, key merge, certificate +
                                      exact fp32 re-rank (two timed spans per call) */
#define FDR_KERNEL_KNN_DEDUP 6     /* duplicate-row classes: hash, sort, class tables, gathers, expansion */
#define FDR_KERNEL_KMER_SEARCH 7   /* k-mer search: library table build + the search passes */
#define FDR_KERNEL_KMER_COMPACT 8  /* k-mer search: sort, de-duplication, row pointers */
#define FDR_NUM_KERNELS 9
int fdr_timing(fdr_ctx *ctx, int enable);
int fdr_timing_read(fdr_ctx *ctx, int which, int *count_out, float *total_ms_out);
/* ---- k-NN mode ---------------------------------------------------------------------------------
 * Both modes return the SAME canonical result.  EXACT: every pair through the
 * fp32 MFMA kernel.  PREFILTER (k <= 56): an fp16 MFMA pass proposes k + 12 (or k + 8) candidates per
 * query; a certificate proves they contain the exact top-k and their distances are recomputed with
 * the canonical fp32 chain; queries that cannot be certified are searched by the exact kernel.  The
 * prefilter mode reads one 4-byte counter back per call (a stream synchronisation).  AUTO (default):
 * PREFILTER when it applies and there are >= 8192 targets.  A new context starts in the mode named by
 * the environment variable FDR_KNN_MODE=exact|prefilter|auto (read once, in fdr_create; default auto) --
 * the only environment variable the library reads. */
#define FDR_MODE_AUTO 0
#define FDR_MODE_EXACT 1
#define FDR_MODE_PREFILTER 2
int fdr_set_knn_mode(fdr_ctx *ctx, int mode);
/* Duplicate-row classes: bitwise-identical rows are searched once and the result
 * expanded -- the same canonical result either way.  AUTO (default): from 8192 targets, when at least 5 % of
 * the rows repeat.  OFF: never.  ON: at every size (same 5 % test).  FORCE: always expand, even without
 * duplicates (the parity tests' "+classes" variants). */
#define FDR_DEDUP_AUTO 0
#define FDR_DEDUP_OFF 1
#define FDR_DEDUP_ON 2
#define FDR_DEDUP_FORCE 3
int fdr_set_dedup_mode(fdr_ctx *ctx, int mode);
/* Unique target / query rows the most recent k-NN call
 * actually searched (= the row counts when the call found too few duplicates to bother). */
int fdr_last_unique(fdr_ctx *ctx, int *unique_targets, int *unique_queries);
/* Prefilter mode only: how the fp16 candidate pass of the most recent k-NN call was launched -- the number
 * of kernel launches and of queues they were dealt to (with two queues two launches are in flight at any
 * time and FDR_KERNEL_KNN_PREFILTER is ONE timed span over the whole pass; 0 / 0 after an exact-mode call). */
int fdr_last_prefilter_launches(fdr_ctx *ctx, int *launches, int *queues);
/* Prefilter mode only: number of query rows of the most recent k-NN call whose candidate set could
 * not be certified and that were therefore searched by the exact kernel. */
int fdr_last_uncertified(fdr_ctx *ctx);
/* Diagnostics: which way each query row of the most recent fdr_knn_dev / fdr_knn / fdr_embed_knn call took to its
 * (canonical, identical on every way) result -- one byte per query row, copied to host memory `paths` [n_queries]:
 * the low seven bits one of FDR_PATH_*, bit 7 (FDR_PATH_CLASS_MEMBER) set when the row belongs to a duplicate-row
 * class of several rows and its result was expanded from the class representative's.  The codes live in the
 * workspace of that call: ask before the workspace is reused or freed, with the n_queries of that call
 * (FDR_E_STATE otherwise, and after fdr_knn_unique_dev / fdr_knn_expand_dev, which record none).  The parity
 * tests stratify their oracle samples by these codes; nothing in the product reads them. */
#define FDR_PATH_NONE 0            /* (never reported for a finished row) */
#define FDR_PATH_CERTIFIED 1       /* fp16 candidates certified, the K' candidates re-ranked with the canonical chain */
#define FDR_PATH_RANGE 2           /* plateau: every target within the bound collected by the range pass, ranked exactly */
#define FDR_PATH_EXACT 3           /* the exact fp32 kernel (exact mode; uncertifiable queries of the prefilter mode) */
#define FDR_PATH_ZERO 4            /* all-zero query row: closed-form answer */
#define FDR_PATH_RANGE_OVERFLOW 5  /* the range pass collected more than its capacity: the exact kernel */
#define FDR_PATH_GENERIC 6         /* d > 512 or k > 64: the generic kernel */
#define FDR_PATH_CLASS_MEMBER 0x80
int fdr_last_query_paths(fdr_ctx *ctx, uint8_t *paths, int64_t n_queries);

/* ---- k-mer search on the GPU: reads x k-mer library -> per-read set of library indices -----------
 * Replaces the reference's native tool kmer_searcher (kmer_searcher/kmer_searcher.cpp:232-375; called
 * from fedrann/count_kmers.py:131-139).  lib_codes: the unique valid library k-mers in index order as
 * 2-bit codes (A C G T = 0 1 2 3, first base in the most significant position; kmer_to_int :138-151).
 * seqs: the reads' characters concatenated, read r = seqs[seq_off[r] .. seq_off[r+1]).  A character
 * outside ACGTacgt makes every window that contains it invalid in the reference's particular way (see
 * kmer_search.inc).  Output: CSR rows of ascending unique library indices per read (the reference
 * writes them in hash-set order): indptr_out int64 [n_reads + 1] and *nnz_out from fdr_kmer_search,
 * then the indices (int32 [nnz], kept on the device until then) from fdr_kmer_search_indices, which also
 * releases the search's device scratch (~28 B per base: not to be held through the embed / k-NN stages).
 * Limits: the concatenated reads must be shorter than 2^32 characters and, with that scratch, fit in HBM. */
int fdr_kmer_search(fdr_ctx *ctx, const uint8_t *seqs, const int64_t *seq_off, int64_t n_reads,
                    const uint64_t *lib_codes, int64_t n_lib, int32_t k, int64_t *indptr_out,
                    int64_t *nnz_out);
int fdr_kmer_search_indices(fdr_ctx *ctx, int32_t *indices_out);

/* ---- canonical k-mer counting on the GPU ---------------------------------------------------------
 * Replaces `jellyfish count -m k -C` + `jellyfish dump -L min_count` (fedrann/count_kmers.py:80-121):
 * every window of k characters inside one read whose characters are all in ACGTacgt counts once under
 * the smaller of its 2-bit code and its reverse complement's.  fdr_kmer_count keeps the k-mers with at
 * least min_count occurrences on the device and returns their number; fdr_kmer_count_fetch copies
 * them out in ascending code order (jellyfish dumps in its hash order; the order only names the
 * features): codes_out, counts_out uint64 [n], and releases the device scratch.
 * Any number of characters: read sets of 2^31 characters and more are counted in blocks of whole reads whose
 * sorted (code, count) runs are merged into one table on the device; min_count applies to the totals.  Limits:
 * n_reads < 2^31, one read < the block size, < 2^31 distinct k-mers.  fdr_set_kmer_count_block sets the block
 * size in characters (0 = the default 2^31; for tests), fdr_last_kmer_count_blocks returns the number of
 * non-empty blocks the last fdr_kmer_count call counted. */
int fdr_kmer_count(fdr_ctx *ctx, const uint8_t *seqs, const int64_t *seq_off, int64_t n_reads, int32_t k,
                   int64_t min_count, int64_t *n_out);
int fdr_kmer_count_fetch(fdr_ctx *ctx, uint64_t *codes_out, uint64_t *counts_out);
/* The same in pieces, for a reader that streams the reads (the reference pipes them through jellyfish,
 * count_kmers.py:80-99) and never holds the whole read set: begin, add whole reads any number of times (seq_off[0] = 0
 * in every piece), finish = fdr_kmer_count's threshold and result; then fdr_kmer_count_fetch. */
int fdr_kmer_count_begin(fdr_ctx *ctx, int32_t k);
int fdr_kmer_count_add(fdr_ctx *ctx, const uint8_t *seqs, const int64_t *seq_off, int64_t n_reads);
int fdr_kmer_count_finish(fdr_ctx *ctx, int64_t min_count, int64_t *n_out);
int fdr_set_kmer_count_block(fdr_ctx *ctx, int64_t chars);
int fdr_last_kmer_count_blocks(fdr_ctx *ctx);

/* ---- FASTA / FASTQ reader of the k-mer stage (host only: no context, no GPU) -------------------------
 * Replaces read_sequences of kmer_searcher/kmer_searcher.cpp:153-200 (FASTQ iff the first line starts with '@';
 * FASTA id = header up to the first space or tab, sequence = the following non-empty lines with only the '\n'
 * removed, records with an empty id and anything before the first header dropped; FASTQ id = the header line
 * after '@', sequence = the next line, two lines skipped) for a PIECE of the file: raw[0, n) = the unconsumed
 * tail of the previous piece followed by new bytes, eof = nothing follows.  fastq_ids_as_fasta: the pipeline's
 * `seqkit fq2fa` step (fedrann/count_kmers.py:76-79) -- FASTQ names cut like FASTA ids, empty names dropped.
 *   fdr_reads_scan:  consumed = bytes holding whole records only (the rest waits for the next piece), their
 *                    record count and total sequence length;
 *   fdr_reads_parse: seqs [n_bases], seq_off int64 [n_records + 1], id_span int64 [2 n_records] = (begin, end)
 *                    of each id inside raw; FDR_E_STATE if raw[0, consumed) no longer matches the scan. */
int fdr_reads_scan(const uint8_t *raw, int64_t n, int32_t is_fastq, int32_t fastq_ids_as_fasta, int32_t eof,
                   int64_t *consumed, int64_t *n_records, int64_t *n_bases);
int fdr_reads_parse(const uint8_t *raw, int64_t consumed, int32_t is_fastq, int32_t fastq_ids_as_fasta,
                    int64_t n_records, int64_t n_bases, uint8_t *seqs, int64_t *seq_off, int64_t *id_span);

/* ---- kmer_searcher output.bin -> doubled binary CSR (host only: no context, no GPU) ---------------
 * Replaces fedrann/feature_extraction.py:108-140 (parse_kmer_searcher_output: header '<4sB3sQ' =
 * "KMER", version 1, record count; per record '<H' id length, id bytes, '<I' index count, that many
 * '<Q' feature indices; a record yields its index set and the strand mirror i + L if i < L else i - L,
 * L = n_features / 2) together with the COO -> CSR conversion of :191-204 (ascending columns per row).
 * Row 2r = record r, row 2r + 1 = its mirror.  Two calls, caller-allocated outputs:
 *   fdr_kmer_output_scan: record count R, sum of index counts nnz, sum of id lengths;
 *   fdr_kmer_output_load: indptr int64 [2R + 1], indices int32 [2 nnz], name_off int64 [R + 1],
 *                         names [name_bytes] (raw id bytes, record r at name_off[r] .. name_off[r+1]);
 *                         n_records / nnz / name_bytes = the scan's results the arrays were sized from
 *                         (FDR_E_STATE if the file no longer matches them: nothing is written then).
 * n_threads <= 0: all hardware threads.  Errors as in the reference (bad magic / version, short
 * header -> FDR_E_ARG), plus truncated records, indices outside [0, n_features) and repeated indices
 * inside a record (the reference would sum them; kmer_searcher emits sets). */
int fdr_kmer_output_scan(const char *path, int64_t *n_records, int64_t *nnz, int64_t *name_bytes);
int fdr_kmer_output_load(const char *path, int64_t n_features, int32_t n_threads, int64_t n_records, int64_t nnz,
                         int64_t name_bytes, int64_t *indptr, int32_t *indices, int64_t *name_off, char *names);
/* The same for the records [rec_lo, rec_hi) only -- one rank's row block of a row-sharded run (rows 2 rec_lo ..
 * 2 rec_hi of the matrix), so that G ranks hold 1/G of the CSR each instead of G copies of it:
 *   fdr_kmer_output_scan_range: record count of the FILE, sum of index counts of the RANGE, id bytes of the file;
 *   fdr_kmer_output_load_range: indptr int64 [2 (rec_hi - rec_lo) + 1] (rebased to 0), indices int32 [2 nnz_range];
 *                               name_off int64 [R + 1] + names of ALL records (the writer names any target row),
 *                               or name_off = NULL for no names. */
int fdr_kmer_output_scan_range(const char *path, int64_t rec_lo, int64_t rec_hi, int64_t *n_records,
                               int64_t *nnz_range, int64_t *name_bytes);
int fdr_kmer_output_load_range(const char *path, int64_t n_features, int32_t n_threads, int64_t n_records,
                               int64_t rec_lo, int64_t rec_hi, int64_t nnz_range, int64_t name_bytes, int64_t *indptr,
                               int32_t *indices, int64_t *name_off, char *names);

/* The writer of the same file, a group of records at a time (kmer_searcher/kmer_searcher.cpp:106-128: '<H' id
 * length, id, '<I' count, the indices as '<Q'): appends n_records records to `path`, record r with the id
 * names[name_off[r] .. name_off[r+1]) and the indices indices[indptr[r] .. indptr[r+1]).  Ids must be printable
 * ASCII as there (:113-117) and at most 65535 bytes (FDR_E_ARG before anything is written).  The caller writes
 * the 16-byte header and patches its record count. */
int fdr_kmer_output_append(const char *path, int64_t n_records, const int64_t *name_off, const char *names,
                           const int64_t *indptr, const int32_t *indices);

/* ---- overlaps.tsv writer (host only: no context, no GPU) ---------------------------------------------
 * Replaces get_output_dataframe + DataFrame.to_csv(sep="\t", index=False) (fedrann/__main__.py:261-300,
 * :385): for every row q = row0 + r (r < n_rows) and column c with target t = idx[r][c], in that order,
 * unless t == q:   name[q] \t "+-"[strand[q]] \t name[t] \t "+-"[strand[t]] \t c \t dist[r][c] \n
 * byte for byte what pandas writes for the reference's DataFrame (float32 distances in their shortest
 * round-trip form, numpy layout; a negative t aliases from the end like Python indexing).  idx / dist
 * [n_rows, k] are the rows row0 .. row0 + n_rows of the neighbour graph (a rank's block of a row-sharded
 * run); names / name_off / strands describe all n_total rows -- or, with strands == NULL, names / name_off describe
 * the n_total / 2 RECORDS of a fwd / rev doubled matrix (feature_extraction.py:136-140): row t carries record t >> 1's
 * id and strand t & 1, and no id is held twice.  append: open the file for appending;
 * write_header: the column-name line first.  n_threads <= 0: all hardware threads.  *lines_out (may be
 * NULL) = data lines written. */
int fdr_overlaps_write(const char *path, int32_t append, int32_t write_header, int64_t n_total, int64_t row0,
                       int64_t n_rows, int32_t k, const int32_t *idx, const float *dist, const int64_t *name_off,
                       const char *names, const uint8_t *strands, int32_t n_threads, int64_t *lines_out);

#ifdef __cplusplus
}
#endif
#endif /* FEDRANN_HIP_H */
