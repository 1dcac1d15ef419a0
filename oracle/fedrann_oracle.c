/*
 * oracle/fedrann_oracle.c -- CPU restatement of the FEDRANN hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in fedrann_amd/ (the product) may import,
 * link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and there only as the checker / the timed CPU baseline.
 *
 * Parity status
 *   orc_embed      PINNED   by tests/golden/embed_{tiny,mid}.npz, produced by running the
 *                           reference's get_feature_matrix (feature_extraction.py:216-292).
 *   orc_normalize  own spec (the reference never normalises E; pynndescent does it internally).
 *   orc_knn        PARITY UNPINNED.  The reference's k-NN arithmetic lives in the third-party
 *                  package pynndescent==0.5.12 (requirements.txt:14; call site
 *                  nearest_neighbors.py:39-55, __main__.py:184-197), which is neither vendored
 *                  under /root/reference nor installed, and the reference has no test or golden
 *                  vector at this boundary.  orc_knn restates the EXACT cosine k-NN that
 *                  NN-descent approximates, with a canonical
 *                  (distance, index) order; it is cross-checked against scikit-learn's brute-force
 *                  cosine neighbours in tests/test_oracle.py.
 *
 * Build: see oracle/Makefile (gcc -O3 -march=x86-64-v3 -ffp-contract=off -fopenmp).
 * -ffp-contract=off matters: every rounding below is spelled out, fmaf() is the only fused op.
 */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * E = A . P     (feature_extraction.py:204-213: A_chunk.dot(precompute_matrix).toarray())
 *
 * A: binary CSR, n_rows x F, column ids per row in ANY order (kmer_searcher emits a hash-set
 *    order); scipy's COO->CSR conversion (csr_matrix((data,(rows,cols))) at :204) sorts them, and
 *    its Gustavson SpGEMM then walks a row's columns in ascending order, adding P[f, c] * 1 into a
 *    per-column fp32 accumulator that starts at +0.  So
 *        E[r, c] = (((0 + P[f0,c]) + P[f1,c]) + ...)   f0 < f1 < ... the features of row r,
 *    one fp32 rounding per add.  Verified bit-for-bit against the reference by the golden vectors.
 * P: CSR by feature (F x d), fp32 values.
 * Rows with no feature come out as zeros (the reference leaves them undefined).
 * ---------------------------------------------------------------------------------------- */
static int cmp_i64(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}

ORC_API int orc_embed(int64_t n_rows, const int64_t *a_indptr, const int64_t *a_indices,
                      int64_t n_features, const int64_t *p_indptr, const int32_t *p_cols,
                      const float *p_vals, int32_t d, float *E) {
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t r = 0; r < n_rows; ++r) {
        float *e = E + r * (int64_t)d;
        for (int32_t c = 0; c < d; ++c) e[c] = 0.0f;
        int64_t b = a_indptr[r], n = a_indptr[r + 1] - b;
        if (n <= 0) continue;
        int64_t *cols = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
        memcpy(cols, a_indices + b, sizeof(int64_t) * (size_t)n);
        qsort(cols, (size_t)n, sizeof(int64_t), cmp_i64);
        for (int64_t t = 0; t < n; ++t) {
            int64_t f = cols[t];
            if (f < 0 || f >= n_features) { bad = 1; continue; }
            for (int64_t q = p_indptr[f]; q < p_indptr[f + 1]; ++q) {
                e[p_cols[q]] = e[p_cols[q]] + p_vals[q]; /* one fp32 add (SSE, FLT_EVAL_METHOD 0) */
            }
        }
        free(cols);
    }
    return bad ? -1 : 0;
}

/* ------------------------------------------------------------------------------------------
 * Canonical fp32 dot product: a fused-multiply-add chain in ascending component order starting
 * from +0.  This is also exactly what gfx950's v_mfma_f32_32x32x2_f32 computes along K, which is
 * why the GPU path can be bit-identical.
 * ---------------------------------------------------------------------------------------- */
static inline float chain_dot(const float *a, const float *b, int32_t d) {
    float acc = 0.0f;
    for (int32_t k = 0; k < d; ++k) acc = __builtin_fmaf(a[k], b[k], acc);
    return acc;
}

/* Row normalisation.  n = chain_dot(x, x); rinv = (float)(1.0 / sqrt((double)n)) (both double ops
 * correctly rounded, then one rounding to fp32); xhat[k] = x[k] * rinv (one fp32 rounding).
 * A row with n == 0 is flagged zero and stays all-zero. */
ORC_API void orc_normalize(const float *E, int64_t n_rows, int32_t d, float *Ehat, float *rinv,
                           uint8_t *is_zero) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n_rows; ++r) {
        const float *x = E + r * (int64_t)d;
        float *y = Ehat + r * (int64_t)d;
        float n = chain_dot(x, x, d);
        float ri = 0.0f;
        if (n > 0.0f) ri = (float)(1.0 / sqrt((double)n));
        for (int32_t k = 0; k < d; ++k) {
            y[k] = x[k] * ri;
        }
        if (rinv) rinv[r] = ri;
        if (is_zero) is_zero[r] = (uint8_t)(n > 0.0f ? 0 : 1);
    }
}

/* Canonical cosine distance from the chain dot of two normalised rows:
 * dist = clamp(1 - c, 0, 1); "x.y <= 0 -> 1.0" is the upper clamp; two zero rows -> 0. */
static inline float cos_dist(float c, int qz, int tz) {
    if (qz && tz) return 0.0f;
    float dv = 1.0f - c;
    if (dv < 0.0f) dv = 0.0f;
    if (dv > 1.0f) dv = 1.0f;
    return dv;
}

/* (dist, idx) packed so that unsigned comparison == lexicographic (dist asc, idx asc);
 * dist >= 0 so its IEEE bits are monotone. */
static inline uint64_t pack_key(float dist, int32_t idx) {
    uint32_t b;
    memcpy(&b, &dist, 4);
    return ((uint64_t)b << 32) | (uint32_t)idx;
}

#define TB 16  /* targets per transposed block (two AVX2 vectors) */
#define QG 4   /* queries per register group */
#define QB 64  /* queries per thread task: the target stream is re-read once per QB queries */
#define TCH 64 /* target blocks per cache chunk (64 * 16 * d * 4 B = 512 KB at d = 128) */

/* Exact k-NN of nq query rows against nt target rows (both already normalised), canonical order
 * (dist asc, target index asc), self included when it is among the targets.
 * q_zero / t_zero: zero-row flags from orc_normalize (may be NULL = no zero rows).
 * idx_out int32 [nq,k] (target row number + t_base), dist_out float [nq,k].  Needs nt >= k.
 * Every query sees its targets in ascending index order; the blocking only reorders work
 * BETWEEN queries, so it cannot change any result. */
ORC_API int orc_knn(const float *Q, const uint8_t *q_zero, int64_t nq, const float *T,
                    const uint8_t *t_zero, int64_t nt, int64_t t_base, int32_t d, int32_t k,
                    int32_t *idx_out, float *dist_out) {
    if (k <= 0 || nt < k || d <= 0) return -1;
    int64_t nblk = (nt + TB - 1) / TB;
    /* transpose targets into [block][component][TB] so the chain runs 16 targets per vector op */
    float *Tt = (float *)aligned_alloc(64, sizeof(float) * (size_t)nblk * d * TB);
    if (!Tt) return -2;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < nblk; ++b)
        for (int32_t c = 0; c < d; ++c)
            for (int j = 0; j < TB; ++j) {
                int64_t t = b * TB + j;
                Tt[((size_t)b * d + c) * TB + j] = t < nt ? T[t * (int64_t)d + c] : 0.0f;
            }
    int64_t ntask = (nq + QB - 1) / QB;
#pragma omp parallel
    {
        uint64_t *best = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)k * QB);
        int *filled = (int *)malloc(sizeof(int) * QB);
#pragma omp for schedule(dynamic, 1)
        for (int64_t task = 0; task < ntask; ++task) {
            int64_t q0 = task * QB, q1 = q0 + QB < nq ? q0 + QB : nq;
            for (int i = 0; i < QB; ++i) filled[i] = 0;
            for (int64_t c0 = 0; c0 < nblk; c0 += TCH) {
                int64_t c1 = c0 + TCH < nblk ? c0 + TCH : nblk;
                for (int64_t qg = q0; qg < q1; qg += QG) {
                    int ng = (int)(q1 - qg < QG ? q1 - qg : QG);
                    const float *qp[QG];
                    for (int g = 0; g < QG; ++g) qp[g] = Q + (qg + (g < ng ? g : 0)) * (int64_t)d;
                    for (int64_t b = c0; b < c1; ++b) {
                        /* 4 queries x 16 targets: eight AVX2 accumulators, each lane one
                         * (query, target) pair running its own fmaf chain over c = 0..d-1 */
                        const float *tb = Tt + (size_t)b * d * TB;
                        __m256 a00 = _mm256_setzero_ps(), a01 = a00, a10 = a00, a11 = a00;
                        __m256 a20 = a00, a21 = a00, a30 = a00, a31 = a00;
                        for (int32_t c = 0; c < d; ++c) {
                            __m256 t0 = _mm256_load_ps(tb + c * TB), t1 = _mm256_load_ps(tb + c * TB + 8);
                            __m256 q0v = _mm256_broadcast_ss(qp[0] + c), q1v = _mm256_broadcast_ss(qp[1] + c);
                            __m256 q2v = _mm256_broadcast_ss(qp[2] + c), q3v = _mm256_broadcast_ss(qp[3] + c);
                            a00 = _mm256_fmadd_ps(q0v, t0, a00); a01 = _mm256_fmadd_ps(q0v, t1, a01);
                            a10 = _mm256_fmadd_ps(q1v, t0, a10); a11 = _mm256_fmadd_ps(q1v, t1, a11);
                            a20 = _mm256_fmadd_ps(q2v, t0, a20); a21 = _mm256_fmadd_ps(q2v, t1, a21);
                            a30 = _mm256_fmadd_ps(q3v, t0, a30); a31 = _mm256_fmadd_ps(q3v, t1, a31);
                        }
                        float acc[QG][TB] __attribute__((aligned(32)));
                        _mm256_store_ps(acc[0], a00); _mm256_store_ps(acc[0] + 8, a01);
                        _mm256_store_ps(acc[1], a10); _mm256_store_ps(acc[1] + 8, a11);
                        _mm256_store_ps(acc[2], a20); _mm256_store_ps(acc[2] + 8, a21);
                        _mm256_store_ps(acc[3], a30); _mm256_store_ps(acc[3] + 8, a31);
                        for (int g = 0; g < ng; ++g) {
                            int li = (int)(qg + g - q0);
                            uint64_t *bl = best + (size_t)li * k;
                            int qz = q_zero ? q_zero[qg + g] : 0;
                            int fl = filled[li];
                            for (int j = 0; j < TB; ++j) {
                                int64_t t = b * TB + j;
                                if (t >= nt) break;
                                float dist = cos_dist(acc[g][j], qz, t_zero ? t_zero[t] : 0);
                                uint64_t key = pack_key(dist, (int32_t)(t + t_base));
                                if (fl == k && key >= bl[k - 1]) continue;
                                int pos = fl < k ? fl++ : k - 1;
                                while (pos > 0 && bl[pos - 1] > key) { bl[pos] = bl[pos - 1]; --pos; }
                                bl[pos] = key;
                            }
                            filled[li] = fl;
                        }
                    }
                }
            }
            for (int64_t qi = q0; qi < q1; ++qi) {
                const uint64_t *bl = best + (size_t)(qi - q0) * k;
                for (int r = 0; r < k; ++r) {
                    uint32_t b32 = (uint32_t)(bl[r] >> 32);
                    float dv;
                    memcpy(&dv, &b32, 4);
                    idx_out[qi * (int64_t)k + r] = (int32_t)(uint32_t)(bl[r] & 0xffffffffu);
                    dist_out[qi * (int64_t)k + r] = dv;
                }
            }
        }
        free(best);
        free(filled);
    }
    free(Tt);
    return 0;
}

/* Plain (untransposed, scalar) distance of one pair -- used by tests to check the blocked loop. */
ORC_API float orc_pair_dist(const float *a, const float *b, int32_t d, int az, int bz) {
    return cos_dist(chain_dot(a, b, d), az, bz);
}

ORC_API void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * kmer_searcher (the step upstream of the hot path).
 *
 * PARITY UNPINNED: kmer_searcher/kmer_searcher.cpp is the reference's only native file; it needs the
 * un-vendored robin_hood.h submodule (unbuildable here, see oracle/Makefile) and its own test data pin
 * an obsolete text format.  What follows restates its algorithm line by line, quirks included.
 *
 *  - kmer_to_int (kmer_searcher.cpp:138-151): 2 bits per base, A=0 C=1 G=2 T=3, either case; any other
 *    character makes the k-mer invalid.
 *  - library load (:262-279): whitespace-separated tokens; a token whose length is not k, an invalid
 *    token, and a code seen before are skipped; the others get indices 0, 1, 2, ... in file order.
 *  - scan of one read (:306-352): rolling code `cur = ((cur << 2) & mask) | base`; an invalid character
 *    sets cur = UINT64_MAX (all ones) and the NEXT characters keep shifting into that, so until k valid
 *    characters have passed the window reads as T...T + the characters since the invalid one.  No
 *    look-up at an invalid character itself.  The first look-up happens after min(k, len) characters
 *    (a read shorter than k is looked up once, left-padded with A = 0; an empty read looks up code 0).
 *    The hits of a read form a SET (output order = hash order, unspecified; sorted ascending here).
 * ------------------------------------------------------------------------------------------ */
static uint64_t orc_mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

static int orc_base_code(unsigned char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
    }
}

typedef struct {
    uint64_t *keys;  /* code + 1 (0 = empty) */
    int64_t *vals;
    uint64_t mask;
} orc_kmer_table;

static int orc_table_init(orc_kmer_table *t, int64_t n) {
    uint64_t size = 16;
    while (size < (uint64_t)(2 * n + 1)) size <<= 1;
    t->keys = (uint64_t *)calloc(size, sizeof(uint64_t));
    t->vals = (int64_t *)malloc(size * sizeof(int64_t));
    t->mask = size - 1;
    return t->keys && t->vals ? 0 : -1;
}

static void orc_table_free(orc_kmer_table *t) {
    free(t->keys);
    free(t->vals);
}

/* returns the stored index, or -1 after inserting (code -> idx) when insert != 0 */
static int64_t orc_table_find(orc_kmer_table *t, uint64_t code, int insert, int64_t idx) {
    uint64_t slot = orc_mix64(code) & t->mask;
    for (;;) {
        if (t->keys[slot] == 0) {
            if (insert) {
                t->keys[slot] = code + 1;
                t->vals[slot] = idx;
            }
            return -1;
        }
        if (t->keys[slot] == code + 1) return t->vals[slot];
        slot = (slot + 1) & t->mask;
    }
}

/* Library text -> unique codes in index order.  Returns their number (or -1: out of memory, -2: cap). */
ORC_API int64_t orc_kmer_library(const char *text, int64_t len, int k, uint64_t *codes_out, int64_t cap) {
    if (k <= 0 || k > 31) return -3;
    int64_t ntok = 0;
    for (int64_t i = 0; i < len;) {  /* upper bound on tokens, for the table size */
        while (i < len && (text[i] == ' ' || (text[i] >= '\t' && text[i] <= '\r'))) ++i;
        if (i >= len) break;
        ++ntok;
        while (i < len && !(text[i] == ' ' || (text[i] >= '\t' && text[i] <= '\r'))) ++i;
    }
    orc_kmer_table t;
    if (orc_table_init(&t, ntok)) return -1;
    int64_t n = 0;
    for (int64_t i = 0; i < len;) {
        while (i < len && (text[i] == ' ' || (text[i] >= '\t' && text[i] <= '\r'))) ++i;
        if (i >= len) break;
        const int64_t b = i;
        while (i < len && !(text[i] == ' ' || (text[i] >= '\t' && text[i] <= '\r'))) ++i;
        if (i - b != k) continue;  /* :271 */
        uint64_t code = 0;
        int ok = 1;
        for (int64_t j = b; j < i; ++j) {
            const int c = orc_base_code((unsigned char)text[j]);
            if (c < 0) {
                ok = 0;
                break;
            }
            code = (code << 2) | (uint64_t)c;
        }
        if (!ok) continue;  /* :273 code != UINT64_MAX */
        if (orc_table_find(&t, code, 1, n) >= 0) continue;  /* :273 seen before */
        if (n >= cap) {
            orc_table_free(&t);
            return -2;
        }
        codes_out[n++] = code;
    }
    orc_table_free(&t);
    return n;
}

static int orc_cmp_i32(const void *a, const void *b) {
    const int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

/* Reads (concatenated bytes + offsets) x library codes -> per-read sorted unique library indices.
 * Returns nnz; if nnz > cap nothing beyond indptr is valid (call again with a larger buffer). */
ORC_API int64_t orc_kmer_search(const char *seqs, const int64_t *off, int64_t n_reads, const uint64_t *codes,
                                int64_t n_lib, int k, int64_t *indptr, int32_t *indices, int64_t cap) {
    if (k <= 0 || k > 31) return -3;
    orc_kmer_table t;
    if (orc_table_init(&t, n_lib)) return -1;
    for (int64_t i = 0; i < n_lib; ++i) (void)orc_table_find(&t, codes[i], 1, i);
    const uint64_t mask = (1ull << (2 * k)) - 1;
    int64_t nnz = 0, hcap = 1024;
    int32_t *hits = (int32_t *)malloc((size_t)hcap * sizeof(int32_t));
    indptr[0] = 0;
    for (int64_t r = 0; r < n_reads; ++r) {
        const unsigned char *s = (const unsigned char *)seqs + off[r];
        const int64_t len = off[r + 1] - off[r];
        int64_t nh = 0;
        uint64_t cur = 0;
        for (int64_t i = 0; i < len || i == 0; ++i) {
            if (i < len) {  /* (an empty read still does the first look-up, with cur = 0) */
                const int c = orc_base_code(s[i]);
                cur = (cur << 2) & mask;
                if (c < 0) cur = UINT64_MAX; else cur |= (uint64_t)c;
            }
            const int lookup = (i >= k - 1) || (i == len - 1) || len == 0;  /* :323 after the init loop, :339 */
            if (lookup && cur != UINT64_MAX) {
                const int64_t idx = orc_table_find(&t, cur, 0, 0);
                if (idx >= 0) {
                    if (nh == hcap) {
                        hcap *= 2;
                        hits = (int32_t *)realloc(hits, (size_t)hcap * sizeof(int32_t));
                    }
                    hits[nh++] = (int32_t)idx;
                }
            }
            if (len == 0) break;
        }
        qsort(hits, (size_t)nh, sizeof(int32_t), orc_cmp_i32);
        int64_t u = 0;
        for (int64_t i = 0; i < nh; ++i)
            if (i == 0 || hits[i] != hits[i - 1]) {
                if (nnz + u < cap) indices[nnz + u] = hits[i];
                ++u;
            }
        nnz += u;
        indptr[r + 1] = nnz;
    }
    free(hits);
    orc_table_free(&t);
    return nnz;
}
