/*
 * nndescent.c -- CPU restatement of the ALGORITHM the reference's k-NN stage runs -- TEST INFRASTRUCTURE ONLY
 * (part of oracle/libfedrann_oracle.so; used by bench.py's cpu_baseline leg and tests/, never by the product).
 *
 * Reference call: fedrann/nearest_neighbors.py:39-55 -> pynndescent.NNDescent(data, metric="cosine", n_neighbors=k,
 * n_trees=300, leaf_size=200, n_iters=None, low_memory=True, ...).neighbor_graph with the arguments fixed in
 * fedrann/__main__.py:184-197.  The arithmetic lives in the third-party package pynndescent == 0.5.12
 * (requirements.txt:14), which is neither vendored under /root/reference nor installed nor installable here, so this
 * file restates its PUBLISHED algorithm (Dong, Charikar, Li: "Efficient k-nearest neighbor graph construction for
 * generic similarity measures", WWW 2011; pynndescent's documentation "How PyNNDescent works"):
 *
 *   1. a forest of n_trees ANGULAR random-projection trees (cosine metric): a node is split by the hyperplane
 *      normal to the difference of two randomly chosen, normalised member points; recursion stops at leaf_size;
 *   2. initialisation: every pair of points that share a leaf (a point with itself included: self enters the graph
 *      at distance 0), then random neighbours for rows still short of k;
 *   3. NN-descent: n_iters = max(5, round(log2 N)) rounds; per round every vertex samples up to
 *      max_candidates = min(60, k) "new" and "old" neighbours (by random priority, both directions of an edge),
 *      all new-new and new-old pairs of a vertex are joined (distance computed, pushed into both heaps when it
 *      improves them), joined new neighbours become old; stop when fewer than delta * k * N = 0.001 k N pushes
 *      succeeded;
 *   4. heaps sorted ascending.
 *
 * PARITY UNPINNED, by construction twice over: the package is absent, and NN-descent is a randomised approximation
 * whose result depends on numba's thread schedule (n_jobs) even for a fixed seed.  What this file is for: a CPU
 * time and a recall figure for the reference's ALGORITHM beside the exact search (bench.py: cpu_baseline.nndescent),
 * with the same canonical distance as the exact oracle (fedrann_oracle.c: normalised rows, fma chain, clamp).
 * Its random stream is its own (xorshift128+ seeded per tree / per vertex from `seed`).
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

typedef struct { uint64_t a, b; } rng_t;
static inline uint64_t rng_next(rng_t *r) {
    uint64_t s1 = r->a;
    const uint64_t s0 = r->b;
    r->a = s0;
    s1 ^= s1 << 23;
    r->b = s1 ^ s0 ^ (s1 >> 18) ^ (s0 >> 5);
    return r->b + s0;
}
static inline rng_t rng_seed(uint64_t seed, uint64_t stream) {
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + stream * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull;
    rng_t r;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    r.a = z | 1;
    z += 0x9E3779B97F4A7C15ull; z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27;
    r.b = z | 2;
    for (int i = 0; i < 4; ++i) (void)rng_next(&r);
    return r;
}
static inline float rng_unit(rng_t *r) { return (float)(rng_next(r) >> 40) * (1.0f / 16777216.0f); }

static inline float nd_dot(const float *a, const float *b, int d) {  /* the oracle's canonical chain */
    float acc = 0.0f;
    for (int k = 0; k < d; ++k) acc = __builtin_fmaf(a[k], b[k], acc);
    return acc;
}
static inline float nd_dist(const float *Eh, const uint8_t *zero, int d, int p, int q) {
    if (zero[p] && zero[q]) return 0.0f;
    float dv = 1.0f - nd_dot(Eh + (size_t)p * d, Eh + (size_t)q * d, d);
    return dv < 0.0f ? 0.0f : (dv > 1.0f ? 1.0f : dv);
}

/* ---- max-heaps of (distance, index, flag), one per vertex, with a lock --------------------------------------- */
typedef struct {
    int n, k;
    int *idx;       /* [n][k], -1 = empty */
    float *dist;    /* [n][k], +inf = empty */
    uint8_t *flag;  /* [n][k], 1 = new */
    omp_lock_t *lock;
} heaps_t;

/* pynndescent's checked_flagged_heap_push: rejects d >= root and indices already present */
static int heap_push(heaps_t *h, int row, float d, int j, uint8_t f) {
    int *ix = h->idx + (size_t)row * h->k;
    float *ds = h->dist + (size_t)row * h->k;
    uint8_t *fl = h->flag + (size_t)row * h->k;
    const int k = h->k;
    if (d >= ds[0]) return 0;
    for (int i = 0; i < k; ++i)
        if (ix[i] == j) return 0;
    int i = 0;  /* replace the root, sift down */
    for (;;) {
        const int l = 2 * i + 1, r = l + 1;
        int s;
        if (l >= k) break;
        if (r >= k) s = ds[l] > d ? l : i;
        else if (ds[l] >= ds[r]) s = ds[l] > d ? l : i;
        else s = ds[r] > d ? r : i;
        if (s == i) break;
        ds[i] = ds[s]; ix[i] = ix[s]; fl[i] = fl[s];
        i = s;
    }
    ds[i] = d; ix[i] = j; fl[i] = f;
    return 1;
}
static int heap_push_locked(heaps_t *h, int row, float d, int j, uint8_t f) {
    if (d >= h->dist[(size_t)row * h->k]) return 0;  /* (racy pre-test: the bound only falls) */
    omp_set_lock(&h->lock[row]);
    const int c = heap_push(h, row, d, j, f);
    omp_unset_lock(&h->lock[row]);
    return c;
}

/* ---- angular random-projection forest ----------------------------------------------------------------------- */
typedef struct { int *leaves; int64_t n_leaves; int leaf_size; } leafset_t;

static void split_rec(const float *Eh, int d, int *pts, int n, int leaf_size, rng_t *rng, int depth, int *out,
                      int64_t *n_out, int64_t cap, float *hyper) {
    if (n <= leaf_size || depth > 200) {
        if (*n_out >= cap) return;
        int *leaf = out + (*n_out) * (int64_t)leaf_size;
        const int m = n < leaf_size ? n : leaf_size;
        for (int i = 0; i < m; ++i) leaf[i] = pts[i];
        for (int i = m; i < leaf_size; ++i) leaf[i] = -1;
        ++*n_out;
        return;
    }
    /* two distinct random members; rows are already normalised, so the hyperplane is their difference */
    const int li = (int)(rng_next(rng) % (uint64_t)n);
    int ri = (int)(rng_next(rng) % (uint64_t)(n - 1));
    if (ri >= li) ++ri;
    const float *L = Eh + (size_t)pts[li] * d, *R = Eh + (size_t)pts[ri] * d;
    for (int c = 0; c < d; ++c) hyper[c] = L[c] - R[c];
    int nl = 0;
    int i = 0, j = n - 1;  /* partition in place: left side first */
    while (i <= j) {
        const float m = nd_dot(hyper, Eh + (size_t)pts[i] * d, d);
        int left;
        if (fabsf(m) < 1e-8f) left = (int)(rng_next(rng) & 1);
        else left = m > 0.0f;
        if (left) { ++i; ++nl; }
        else { const int t = pts[i]; pts[i] = pts[j]; pts[j] = t; --j; }
    }
    if (nl == 0 || nl == n) {  /* degenerate: random halves (pynndescent does the same) */
        for (int a = n - 1; a > 0; --a) { const int b = (int)(rng_next(rng) % (uint64_t)(a + 1)); const int t = pts[a]; pts[a] = pts[b]; pts[b] = t; }
        nl = n / 2;
    }
    split_rec(Eh, d, pts, nl, leaf_size, rng, depth + 1, out, n_out, cap, hyper);
    split_rec(Eh, d, pts + nl, n - nl, leaf_size, rng, depth + 1, out, n_out, cap, hyper);
}

/* ---- the whole thing ---------------------------------------------------------------------------------------------
 * Eh [n,d] normalised rows + zero flags (orc_normalize); idx_out int32 [n,k], dist_out float [n,k] ascending.
 * stats_out[0] = descent rounds run, [1] = leaves, [2] = distance evaluations.  Returns 0, or -1 on bad arguments /
 * allocation failure. */
ORC_API int orc_nndescent(const float *Eh, const uint8_t *zero, int64_t n64, int32_t d, int32_t k, int32_t n_trees,
                          int32_t leaf_size, uint64_t seed, int32_t *idx_out, float *dist_out, int64_t *stats_out) {
    if (!Eh || !zero || !idx_out || !dist_out || n64 < k || k < 1 || n64 > 0x7fffffff || leaf_size < 2 || n_trees < 1) return -1;
    const int n = (int)n64;
    heaps_t H;
    H.n = n; H.k = k;
    H.idx = (int *)malloc((size_t)n * k * sizeof(int));
    H.dist = (float *)malloc((size_t)n * k * sizeof(float));
    H.flag = (uint8_t *)malloc((size_t)n * k);
    H.lock = (omp_lock_t *)malloc((size_t)n * sizeof(omp_lock_t));
    if (!H.idx || !H.dist || !H.flag || !H.lock) return -1;
    for (size_t i = 0; i < (size_t)n * k; ++i) { H.idx[i] = -1; H.dist[i] = INFINITY; H.flag[i] = 0; }
    for (int i = 0; i < n; ++i) omp_init_lock(&H.lock[i]);
    int64_t n_dist = 0, n_leaves_total = 0;

    /* 1 + 2: forest, leaf by leaf initialisation (a tree at a time per thread) */
#pragma omp parallel reduction(+ : n_dist, n_leaves_total)
    {
        int *pts = (int *)malloc((size_t)n * sizeof(int));
        const int64_t cap = 4 * ((int64_t)n / leaf_size + 1) + 16;  /* leaves of one tree (splits are near the middle) */
        int *leaves = (int *)malloc((size_t)cap * leaf_size * sizeof(int));
        float *hyper = (float *)malloc((size_t)d * sizeof(float));
#pragma omp for schedule(dynamic, 1)
        for (int t = 0; t < n_trees; ++t) {
            if (!pts || !leaves || !hyper) continue;
            rng_t rng = rng_seed(seed, (uint64_t)t + 1);
            for (int i = 0; i < n; ++i) pts[i] = i;
            int64_t nl = 0;
            split_rec(Eh, d, pts, n, leaf_size, &rng, 0, leaves, &nl, cap, hyper);
            n_leaves_total += nl;
            for (int64_t l = 0; l < nl; ++l) {
                const int *leaf = leaves + l * leaf_size;
                for (int a = 0; a < leaf_size && leaf[a] >= 0; ++a)
                    for (int b = a; b < leaf_size && leaf[b] >= 0; ++b) {  /* (b = a: a point with itself) */
                        const int p = leaf[a], q = leaf[b];
                        const float dv = nd_dist(Eh, zero, d, p, q);
                        ++n_dist;
                        heap_push_locked(&H, p, dv, q, 1);
                        if (p != q) heap_push_locked(&H, q, dv, p, 1);
                    }
            }
        }
        free(pts); free(leaves); free(hyper);
    }
    /* rows still short of k neighbours: random ones */
#pragma omp parallel for schedule(static) reduction(+ : n_dist)
    for (int i = 0; i < n; ++i) {
        rng_t rng = rng_seed(seed ^ 0xabcdefull, (uint64_t)i + 1);
        int tries = 0;
        while (H.idx[(size_t)i * k] < 0 && tries++ < 8 * k) {  /* (the root is empty while any slot is) */
            const int j = (int)(rng_next(&rng) % (uint64_t)n);
            const float dv = nd_dist(Eh, zero, d, i, j);
            ++n_dist;
            heap_push_locked(&H, i, dv, j, 1);
        }
    }

    /* 3: NN-descent */
    const int max_cand = k < 60 ? k : 60;
    int n_iters = (int)lround(log2((double)n));
    if (n_iters < 5) n_iters = 5;
    int *newc = (int *)malloc((size_t)n * max_cand * sizeof(int)), *oldc = (int *)malloc((size_t)n * max_cand * sizeof(int));
    float *newp = (float *)malloc((size_t)n * max_cand * sizeof(float)), *oldp = (float *)malloc((size_t)n * max_cand * sizeof(float));
    if (!newc || !oldc || !newp || !oldp) return -1;
    int rounds = 0;
    for (int it = 0; it < n_iters; ++it) {
        ++rounds;
        for (size_t i = 0; i < (size_t)n * max_cand; ++i) { newc[i] = oldc[i] = -1; newp[i] = oldp[i] = INFINITY; }
        /* candidate sampling: an edge (i, j) offers j to i's list and i to j's list, kept by smallest random priority
           (a bounded "heap" as a small array: max_cand <= 60) */
#pragma omp parallel for schedule(dynamic, 256)
        for (int i = 0; i < n; ++i) {
            rng_t rng = rng_seed(seed + 77ull * (uint64_t)(it + 1), (uint64_t)i + 1);
            for (int s = 0; s < k; ++s) {
                const int j = H.idx[(size_t)i * k + s];
                if (j < 0) continue;
                const float pr = rng_unit(&rng);
                const int is_new = H.flag[(size_t)i * k + s];
                int *cl = is_new ? newc : oldc;
                float *pl = is_new ? newp : oldp;
                for (int side = 0; side < 2; ++side) {
                    const int v = side ? j : i, w = side ? i : j;
                    omp_set_lock(&H.lock[v]);
                    int *c = cl + (size_t)v * max_cand;
                    float *p = pl + (size_t)v * max_cand;
                    int worst = 0, dup = 0;
                    for (int m = 0; m < max_cand; ++m) {
                        if (c[m] == w) dup = 1;
                        if (p[m] > p[worst]) worst = m;
                    }
                    if (!dup && pr < p[worst]) { p[worst] = pr; c[worst] = w; }
                    omp_unset_lock(&H.lock[v]);
                }
            }
        }
        /* a new neighbour that made it into its vertex's new-candidate list is old from now on */
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; ++i)
            for (int s = 0; s < k; ++s) {
                if (!H.flag[(size_t)i * k + s]) continue;
                const int j = H.idx[(size_t)i * k + s];
                for (int m = 0; m < max_cand; ++m)
                    if (newc[(size_t)i * max_cand + m] == j) { H.flag[(size_t)i * k + s] = 0; break; }
            }
        /* local joins */
        int64_t c_updates = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : c_updates, n_dist)
        for (int i = 0; i < n; ++i) {
            const int *nc = newc + (size_t)i * max_cand, *oc = oldc + (size_t)i * max_cand;
            for (int a = 0; a < max_cand; ++a) {
                const int p = nc[a];
                if (p < 0) continue;
                for (int b = a + 1; b < max_cand; ++b) {
                    const int q = nc[b];
                    if (q < 0) continue;
                    const float dv = nd_dist(Eh, zero, d, p, q);
                    ++n_dist;
                    c_updates += heap_push_locked(&H, p, dv, q, 1);
                    c_updates += heap_push_locked(&H, q, dv, p, 1);
                }
                for (int b = 0; b < max_cand; ++b) {
                    const int q = oc[b];
                    if (q < 0 || q == p) continue;
                    const float dv = nd_dist(Eh, zero, d, p, q);
                    ++n_dist;
                    c_updates += heap_push_locked(&H, p, dv, q, 1);
                    c_updates += heap_push_locked(&H, q, dv, p, 1);
                }
            }
        }
        if ((double)c_updates <= 0.001 * (double)k * (double)n) break;
    }
    /* 4: ascending (distance, index) per row */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        int *ix = H.idx + (size_t)i * k;
        float *ds = H.dist + (size_t)i * k;
        for (int a = 1; a < k; ++a) {  /* insertion sort: k <= a few dozen */
            const float dv = ds[a];
            const int jv = ix[a];
            int b = a - 1;
            while (b >= 0 && (ds[b] > dv || (ds[b] == dv && ix[b] > jv))) { ds[b + 1] = ds[b]; ix[b + 1] = ix[b]; --b; }
            ds[b + 1] = dv; ix[b + 1] = jv;
        }
        memcpy(idx_out + (size_t)i * k, ix, (size_t)k * sizeof(int));
        memcpy(dist_out + (size_t)i * k, ds, (size_t)k * sizeof(float));
    }
    if (stats_out) { stats_out[0] = rounds; stats_out[1] = n_leaves_total; stats_out[2] = n_dist; }
    for (int i = 0; i < n; ++i) omp_destroy_lock(&H.lock[i]);
    free(H.idx); free(H.dist); free(H.flag); free(H.lock); free(newc); free(oldc); free(newp); free(oldp);
    return 0;
}
