// fedrann_hip.hip -- gfx950 (MI355X / CDNA4) kernels + C-ABI for FEDRANN's hot path.
//
//   K1 embed_csr_kernel      E = A . P          (feature_extraction.py:167-213 in the reference)
//   K2 normalize_rows_kernel E -> Ehat          (done inside pynndescent in the reference)
//   K3 knn_tile_kernel       exact cosine top-k (nearest_neighbors.py:39-55 -> pynndescent)
//   K4 knn_merge_kernel      merge of per-segment top-k lists
//
// Written for wave64 / MFMA / 160 KB LDS directly; there is no other backend.
// ABI: include/fedrann_hip.h.  Design notes and rooflines: DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstring>
#include <type_traits>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_run_length_encode.hpp>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/fedrann_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

#define FDR_EXPORT extern "C" __attribute__((visibility("default")))

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? FDR_E_NOMEM : FDR_E_HIP, "%s failed: %s",  \
                        #expr, hipGetErrorString(e_));                                         \
    } while (0)

// ------------------------------------------------------------------------------------------
// K1  E = A . P   -- CSR-row-parallel, one wave per read row.
//
// P (F x d) is "very sparse": >= 90 % of its feature rows are empty (density 1/sqrt(F)), so the
// projection is stored as
//   ftab[w]    = { bits: which of features 32w..32w+31 have a non-empty P row,
//                  prefix: number of non-empty rows among features < 32w }          (8 B / 32 features)
//   rowinfo[r] = { start, count, first column, first value bits } of the r-th non-empty row  (16 B)
//   ent[q]     = { column, fp32 bits }   (entries beyond a row's first)
// A wave streams its row's column ids 256 at a time (four coalesced loads in flight), tests the
// bitmap words (L2 resident) and fetches rowinfo for the hits, so a row pays the dependent
// id -> bitmap -> rowinfo latency chain once per 256 non-zeros.  Hits are then applied in ascending
// feature order by the whole wave (lane l owns columns l, l+64, ...), which reproduces scipy's
// sequential fp32 sums bit for bit.
// ------------------------------------------------------------------------------------------
template <int DP>
__global__ __launch_bounds__(256) void embed_csr_kernel(
    long long n_rows, const long long *__restrict__ a_indptr, const int *__restrict__ a_indices,
    long long n_features, const uint2 *__restrict__ ftab, const uint4 *__restrict__ rowinfo,
    const uint2 *__restrict__ ent, int d, float *__restrict__ E) {
    constexpr int NACC = DP / 64;
    constexpr int NB = 4;  // 64-id chunks in flight per wave
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
    for (long long row = wave0; row < n_rows; row += nwaves) {
        float acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = 0.0f;
        const long long beg = a_indptr[row], end = a_indptr[row + 1];
        for (long long base = beg; base < end; base += 64 * NB) {
            int f[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const long long pos = base + 64 * u + lane;
                f[u] = pos < end ? a_indices[pos] : -1;
            }
            uint2 w[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                w[u] = make_uint2(0u, 0u);
                if (f[u] >= 0 && (long long)f[u] < n_features) w[u] = ftab[f[u] >> 5];
            }
            uint4 info[NB];
            bool hit[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const unsigned bit = 1u << (f[u] & 31);
                hit[u] = (w[u].x & bit) != 0u;
                info[u] = make_uint4(0u, 0u, 0u, 0u);
                if (hit[u]) info[u] = rowinfo[w[u].y + __popc(w[u].x & (bit - 1u))];
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                u64 m = __ballot(hit[u]);
                while (m) {  // wave-uniform: hits in ascending lane = ascending feature order
                    const int src = __builtin_ctzll(m);
                    m &= m - 1;
                    int q = __builtin_amdgcn_readlane((int)info[u].x, src);
                    const int ee = q + __builtin_amdgcn_readlane((int)info[u].y, src);
                    unsigned c = (unsigned)__builtin_amdgcn_readlane((int)info[u].z, src);
                    float v = __int_as_float(__builtin_amdgcn_readlane((int)info[u].w, src));
                    while (true) {
#pragma unroll
                        for (int i = 0; i < NACC; ++i)
                            if (c == (unsigned)(lane + 64 * i)) acc[i] += v;
                        if (++q >= ee) break;
                        const uint2 en = ent[q];  // same address in every lane
                        c = en.x;
                        v = __uint_as_float(en.y);
                    }
                }
            }
        }
        float *out = E + row * (long long)d;
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            if (lane + 64 * i < d) out[lane + 64 * i] = acc[i];
    }
}

// ------------------------------------------------------------------------------------------
// K2  row normalisation into the k-NN kernel's layout.
//
// One lane per row runs the canonical chain n = fma(x_k, x_k, n), k ascending (the same order
// the MFMA uses along K), rinv = (float)(1/sqrt((double)n)), xhat_k = x_k * rinv.  A 64-row tile is
// staged through LDS so that both the global read and the global write are coalesced.
// Output row: DP floats, zero padded; inside each group of 8 components the order is
// [k0 k2 k4 k6 | k1 k3 k5 k7] so that lane-half h of the MFMA reads its four K-steps
// (components 8g + 2s + h, s = 0..3) as ONE 16-byte access.
// ------------------------------------------------------------------------------------------
template <int DP, int RB>
__global__ __launch_bounds__(64) void normalize_rows_kernel(const float *__restrict__ E,
                                                            long long n_rows, int d,
                                                            float *__restrict__ Ehat,
                                                            unsigned char *__restrict__ zero) {
    __shared__ float tile[RB][DP + 1];  // +1: the per-lane row walk below is bank-conflict free
    const int tid = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * RB;
    const int nr = (int)min((long long)RB, n_rows - row0);
    const float *src = E + row0 * (long long)d;
    for (int i = tid; i < RB * DP; i += 64) tile[i / DP][i % DP] = 0.0f;
    __syncthreads();
    const int total = nr * d;
    for (int i = tid; i < total; i += 64) tile[i / d][i % d] = src[i];
    __syncthreads();
    if (tid < RB) {
        float n = 0.0f;
#pragma unroll 8
        for (int k = 0; k < DP; ++k) {
            const float x = tile[tid][k];
            n = __builtin_fmaf(x, x, n);
        }
        float ri = 0.0f;
        if (n > 0.0f) ri = (float)(1.0 / sqrt((double)n));
#pragma unroll 8
        for (int k = 0; k < DP; ++k) tile[tid][k] = tile[tid][k] * ri;
        if (tid < nr) zero[row0 + tid] = n > 0.0f ? 0 : 1;
    }
    __syncthreads();
    float *dst = Ehat + row0 * (long long)DP;
    for (int i = tid; i < nr * DP; i += 64) {
        const int r = i / DP, p = i % DP;
        const int g = p >> 3, hh = (p >> 2) & 1, s = p & 3;
        dst[i] = tile[r][8 * g + 2 * s + hh];
    }
}

// ------------------------------------------------------------------------------------------
// K3  tiled exact cosine k-NN.
//
// Workgroup = 8 waves = 256 query rows; grid = (query blocks, target segments).  Each wave keeps
// its 32 queries as the B operand of v_mfma_f32_32x32x2_f32 in DP/2 VGPRs for the whole kernel
// and streams 32-target tiles (A operand) from a double-buffered, XOR-swizzled LDS image shared by
// the 8 waves.  The MFMA accumulates along K exactly like an fp32 fmaf chain in ascending
// component order, so every similarity is bit-identical to the CPU oracle's chain_dot().
//
// Top-k: one list of K (dist,idx) keys per query in LDS (key = dist bits << 32 | idx, unsigned
// order == (dist asc, idx asc)); the list is an unsorted set whose maximum (tau, taupos) is kept
// in registers.  Fast path per tile: max of the 16 accumulators -> one distance -> compare with
// tau.  Slow path (rare after warm-up): candidates are visited in ascending target order --
// lane-half 0 takes tile rows 0..15, then lane-half 1 rows 16..31 -- so a strict "dist < tau" is
// exact under the (dist, idx) order; an accepted candidate replaces the maximum and the K keys
// are rescanned for the new maximum.
// ------------------------------------------------------------------------------------------
#define KEY_INF 0x7F800000FFFFFFFFull

__device__ __forceinline__ float dist_from_sim(float c) {
    return __builtin_amdgcn_fmed3f(1.0f - c, 0.0f, 1.0f);  // clamp(1 - c, 0, 1): v_sub + v_med3
}

// Conservative similarity-space form of "dist_from_sim(c) < tau": every c that passes the exact
// test satisfies c > sim_floor(tau) (both roundings involved are below 6e-8 in [0,1]); candidates
// above the floor are re-tested exactly before they are queued.
__device__ __forceinline__ float sim_floor(float tau) {
    return tau <= 1.0f ? (1.0f - tau) - 3.0e-7f : -__builtin_inff();
}

// zero-row byte flags -> one bit per target row (bit r of word w = row 32w + r)
__global__ __launch_bounds__(256) void pack_zero_bits_kernel(const unsigned char *__restrict__ zero,
                                                             int n, unsigned *__restrict__ bits,
                                                             unsigned *__restrict__ tau_shared,
                                                             int n_shared) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_shared; i += gridDim.x * 256)
        tau_shared[i] = 0x7F800000u;  // +inf: no segment has a full list yet
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool z = i < n && zero[i] != 0;
    const u64 m = __ballot(z);
    const int lane = threadIdx.x & 63;
    if (lane == 0 && i < n) bits[i >> 5] = (unsigned)m;
    if (lane == 32 && i < n) bits[i >> 5] = (unsigned)(m >> 32);
}

// ---- top-k state of one 32-query set of a wave ---------------------------------------------------
// A query (lane pair j / j+32) owns one unsorted list of K keys in LDS (column `ql` of lists[K][256]);
// taukey / taupos = its current maximum, identical in both lanes.  Each LANE additionally owns a
// 4-entry append queue in LDS.  Per tile, candidates that beat tau (which may be stale, i.e. too
// large, between flushes -- that only admits extra candidates) are appended to the lane's queue with
// predicated stores; when a queue overflows, and at the end of the segment, the whole wave flushes:
// every queued key that is still smaller than the list maximum replaces it and the two lanes of the
// query rescan the list together.  Exactness: keys order by (dist, idx); a rejected candidate has
// dist >= tau.dist and a larger index than every listed key, so it can never belong to the top-k.
#define QCAP 4  // default entries per lane append queue (the kernels take the actual value, 2 or 4)
#define FDR_MAX_SEG 48

// Target segment boundaries (in rows, multiples of 32 except the last): segment s = [b[s], b[s+1]).
struct SegBounds {
    int b[FDR_MAX_SEG + 1];
};

#ifdef FDR_DEBUG_COUNTERS  // development build only: event counters read back with FDR_KNN_DEBUG=2
__device__ unsigned long long g_dbg_counters[8];
#define DBG_COUNT(i) do { if (dbgc && (threadIdx.x & 63) == 0) atomicAdd(&g_dbg_counters[i], 1ull); } while (0)
#define DBG_ADD(i, n) do { if (dbgc && (threadIdx.x & 63) == 0) atomicAdd(&g_dbg_counters[i], (unsigned long long)(n)); } while (0)
#else
#define DBG_COUNT(i) do { (void)dbgc; } while (0)
#define DBG_ADD(i, n) do { (void)dbgc; } while (0)
#endif

struct TopkState {
    u64 taukey;   // maximum key of the query's list (both lanes of the query hold the same value)
    float tau;    // admission bound: min(distance part of taukey, cross-segment bound), see topk_share
    float cfloor; // sim_floor(tau)
    int taupos;   // position of taukey in the list
    int qcnt;     // entries in this LANE's append queue
    float foreign;  // last cross-segment bound seen (strict form, +inf if none): see topk_share
};

// Cross-segment bound.  Workgroups that search different target segments for the same queries
// publish the k-th best distance of their (full) list with a relaxed device-scope atomicMin on one
// word per query and adopt the minimum any segment has published.  Every published value is an upper
// bound of the query's final k-th best distance D, so a candidate with dist > bound can never be in
// the final top-k; candidates with dist == bound are kept (ties are decided by index in the merge):
// the admission test is dist < nextup(bound).  A stale or missing value only admits more candidates,
// so the result does not depend on scheduling, timing or placement.
__device__ __forceinline__ float topk_share(unsigned *__restrict__ slot, const u64 taukey, const int h,
                                            float &foreign_out) {
    const unsigned mine = (unsigned)(taukey >> 32);  // 0x7F800000 while the list still has empty slots
    unsigned seen = mine;
    if (h == 0) {
        const unsigned old = mine < 0x7F800000u
                                 ? __hip_atomic_fetch_min(slot, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                 : __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seen = old < mine ? old : mine;
    }
    seen = __shfl(seen, (threadIdx.x & 31), 64) ;  // lane j (h == 0) of this query holds the value
    // strict bound: the list's own maximum admits dist < tau; a foreign bound admits dist <= bound
    const float own = __uint_as_float(mine);
    const float foreign = seen < 0x7F800000u ? __uint_as_float(seen + 1u) : __builtin_inff();
    foreign_out = foreign;
    return fminf(own, foreign);
}

template <int NT, int QW>
__device__ __noinline__ TopkState topk_flush(TopkState st, u64 *__restrict__ lists,
                                             u64 *__restrict__ queue, unsigned *__restrict__ shared,
                                             const int ql, const int K, const int tid, const int h,
                                             const int qcap, const bool dbgc) {
    DBG_COUNT(2);
    const int cnt_me = st.qcnt;
    const int cnt_other = __shfl_xor(cnt_me, 32);
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
        const int owner_cnt = (h == ph) ? cnt_me : cnt_other;
        const int owner_tid = (tid & ~32) | (ph << 5);
#pragma unroll 1
        for (int i = 0; i < qcap; ++i) {
            const bool active = i < owner_cnt;
            if (!__any(active)) break;
            u64 key = KEY_INF;
            if (active) key = queue[i * NT + owner_tid];
            const bool ins = active && key < st.taukey;
            DBG_COUNT(3);
            if (__any(ins)) {
                DBG_COUNT(4);
                if (ins) {
                    if (h == 0) lists[st.taupos * QW + ql] = key;
                    // both lanes of the query rescan the list: lane-half h takes entries h, h+2, ...
                    u64 best = 0;
                    int bp = 0;
#pragma unroll 2
                    for (int e = h; e < K; e += 2) {
                        const u64 kv = lists[e * QW + ql];
                        if (kv > best) {
                            best = kv;
                            bp = e;
                        }
                    }
                    const u64 ob = __shfl_xor(best, 32);
                    const int op = __shfl_xor(bp, 32);
                    if (ob > best) {
                        best = ob;
                        bp = op;
                    }
                    st.taukey = best;
                    st.taupos = bp;
                }
            }
        }
    }
    // (no atomic here: a flush must not wait for a global round trip; the bound is exchanged by the
    // periodic topk_share calls of the tile loop)
    st.tau = fminf(__uint_as_float((unsigned)(st.taukey >> 32)), st.foreign);
    st.cfloor = sim_floor(st.tau);
    st.qcnt = 0;
    return st;
}

// Queue every candidate of this tile that beats tau.  acc[r] = similarity of query j with tile row
// (r&3) + 8*(r>>2) + 4*h; rows >= nvalid do not exist (last tile of a segment only).
template <int NT, int QW>
__device__ __forceinline__ void topk_append(const f32x16 acc, TopkState &st, u64 *__restrict__ lists,
                                            u64 *__restrict__ queue, unsigned *__restrict__ shared,
                                            const int ql, const int K, const int tid, const int h,
                                            int idx0, int nvalid, const int qcap, const bool dbgc) {
    // this block is cold: keep its address / index arithmetic from being hoisted into the hot loop
    asm volatile("" : "+s"(idx0), "+s"(nvalid));
    unsigned todo = 0xffffu;
    if (nvalid < 32) {
        todo = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            todo |= ((r & 3) + 8 * (r >> 2) + 4 * h < nvalid) ? (1u << r) : 0u;
    }
    const int idxh = idx0 + 4 * h;
    // First pass over a tile: "dist < tau" is exact, because every listed or queued key comes from an
    // earlier tile (smaller index).  After a mid-tile flush the list may hold rows of THIS tile from
    // the partner lane, whose indices interleave with mine, so a retried candidate with dist == tau
    // can still win on the index: retries admit dist <= tau and the flush's full-key test decides.
    bool retry = false;
#pragma unroll 1
    while (true) {
        DBG_COUNT(1);
        unsigned ovf = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (acc[r] > st.cfloor) {
                // everything below is kept inside this (rarely taken) block: the asm makes the
                // candidate opaque so that hipcc cannot evaluate the block's arithmetic eagerly
                float a = acc[r];
                asm volatile("" : "+v"(a));
                const float dist = dist_from_sim(a);
                if (((todo >> r) & 1u) && (dist < st.tau || (retry && dist == st.tau))) {
                    if (st.qcnt < qcap) {
                        queue[st.qcnt * NT + tid] =
                            ((u64)__float_as_uint(dist) << 32) | (unsigned)(idxh + (r & 3) + 8 * (r >> 2));
                        ++st.qcnt;
                    } else {
                        ovf |= 1u << r;
                    }
                }
            }
        }
        if (!__any(ovf != 0u)) break;
        st = topk_flush<NT, QW>(st, lists, queue, shared, ql, K, tid, h, qcap, dbgc);
        todo = ovf;
        retry = true;
    }
}

// DP: padded embedding length.  NQ: 32-query sets per wave (independent MFMA accumulator chains
// that share every A fragment).  NW: waves per workgroup; a workgroup owns QW = 32*NQ*NW queries.
// WPS: waves per SIMD the register budget is sized for.
// LDS: 2-stage ring of 32 target rows x 64 components (16 KB) | lists K x QW keys | queues.
template <int DP, int NQ, int NW, int WPS>
__global__ __launch_bounds__(64 * NW, WPS) void knn_tile_kernel(
    const float *__restrict__ Qh, const unsigned char *__restrict__ qzero, int nq,
    const float *__restrict__ Th, const unsigned *__restrict__ tzbits, int nt, int t_base,
    SegBounds segs, int K, int nq_pad, u64 *__restrict__ partial, unsigned *__restrict__ tau_shared,
    int qcap, int dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = 64 * NW;              // threads per workgroup
    constexpr int QW = 32 * NQ * NW;         // queries per workgroup
    constexpr int NCH = DP / 64;             // 64-component K-chunks per tile
    constexpr int STAGE_BYTES = 32 * 64 * 4; // one stage = 32 target rows x 64 components (8 KB)
    constexpr int SLOTS = 16;                // 16-byte slots per staged row
    constexpr int NSTAGE = 2;                // LDS ring: stage it lives in buffer it % 2
    u64 *lists = reinterpret_cast<u64 *>(smem + NSTAGE * STAGE_BYTES);                         // K * QW keys
    u64 *queues = reinterpret_cast<u64 *>(smem + NSTAGE * STAGE_BYTES + (size_t)K * QW * 8);  // NQ*QCAP*NT

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;

    // queries: this lane's B fragments for all DP/2 K-steps of each of its NQ query sets
    float b[NQ][DP / 2];
    bool qz[NQ];
    int ql[NQ];
    int qslot[NQ];  // index of this query's cross-segment bound word
    bool any_qz = false;
#pragma unroll
    for (int s = 0; s < NQ; ++s) {
        ql[s] = (wave * NQ + s) * 32 + j;
        const int qg = blockIdx.x * QW + ql[s];
        qslot[s] = qg;
        const int qrow = qg < nq ? qg : nq - 1;
        qz[s] = qzero[qrow] != 0;
        any_qz = any_qz || __any(qz[s]);
        const f32x4 *qp = reinterpret_cast<const f32x4 *>(Qh + (size_t)qrow * DP);
#pragma unroll
        for (int g = 0; g < DP / 8; ++g) {
            const f32x4 v = qp[2 * g + h];
            b[s][4 * g + 0] = v.x;
            b[s][4 * g + 1] = v.y;
            b[s][4 * g + 2] = v.z;
            b[s][4 * g + 3] = v.w;
        }
    }
    for (int i = tid; i < K * QW; i += NT) lists[i] = KEY_INF;
    TopkState st[NQ];
#pragma unroll
    for (int s = 0; s < NQ; ++s) {
        st[s].taukey = KEY_INF;
        st[s].taupos = 0;
        st[s].qcnt = 0;
        st[s].tau = topk_share((tau_shared + qslot[s]), KEY_INF, h, st[s].foreign);  // other segments may already have a bound
        st[s].cfloor = sim_floor(st[s].tau);
    }

    const int t_begin = segs.b[blockIdx.y];  // multiple of 32
    const int t_end = min(nt, segs.b[blockIdx.y + 1]);
    const int ntiles = (t_end - t_begin + 31) >> 5;
    const int nstages = ntiles * NCH;

    // staging by LDS-DMA (global_load_lds_dwordx4): each wave-instruction fills 1 KiB = four staged
    // rows of 256 B, linearly; the XOR swizzle that makes the MFMA loop's ds_read_b128 (32 rows x one
    // slot) bank-conflict free is applied to the SOURCE slot instead.  Rows past the segment end
    // re-read the last valid row (finite garbage; such candidates are masked by nvalid).
    constexpr int NPIECE = 8;  // 1 KiB pieces per stage, dealt round-robin to the waves
    auto issue_stage = [&](int it, int buf) {
        const int t = it / NCH, ch = it % NCH;
        const int trow0 = t_begin + t * 32;
#pragma unroll
        for (int u = 0; u < (NPIECE + NW - 1) / NW; ++u) {
            const int piece = wave + NW * u;
            if (NPIECE % NW != 0 && piece >= NPIECE) break;
            const int row = 4 * piece + (lane >> 4), pslot = lane & 15;
            const int trow = min(trow0 + row, t_end - 1);
            const float *src = Th + (size_t)trow * DP + (size_t)(ch * SLOTS + (pslot ^ (row & 15))) * 4;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)src,
                (__attribute__((address_space(3))) void *)(smem + buf * STAGE_BYTES + piece * 1024), 16, 0, 0);
        }
    };

    // Two-stage ring: stage it+1 is issued at the top of iteration it and must have landed by the
    // barrier at its bottom (hipcc drains vmcnt before __syncthreads()).  A three-stage ring with a
    // counted vmcnt and a raw s_barrier was measured and bought nothing here (three waves per SIMD
    // already hide the DMA latency), so the simpler form stays.
    if (nstages > 0) issue_stage(0, 0);
    __syncthreads();  // also publishes the list initialisation

    for (int t = 0; t < ntiles; ++t) {
        f32x16 acc[NQ];
#pragma unroll
        for (int s = 0; s < NQ; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int it = t * NCH + ch;
            const int buf = it % NSTAGE;
            if (it + 1 < nstages) issue_stage(it + 1, (it + 1) % NSTAGE);

            // ---- 32 targets x (NQ x 32) queries x 64 components ----
            {
                const f32x4 *sb = reinterpret_cast<const f32x4 *>(smem + buf * STAGE_BYTES) + j * SLOTS;
                const int sw = j & 15;
                f32x4 a[3];  // fragment ring: two groups prefetched ahead of the MFMAs
                a[0] = sb[(0 + h) ^ sw];
                a[1] = sb[(2 + h) ^ sw];
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    if (g + 2 < 8) a[(g + 2) % 3] = sb[(2 * (g + 2) + h) ^ sw];
                    const f32x4 av = a[g % 3];
                    const int bb = 32 * ch + 4 * g;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int s = 0; s < NQ; ++s)
                            acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b[s][bb + e], acc[s], 0, 0, 0);
                }
            }

            if (ch == NCH - 1) {
                const int tile_row0 = t_begin + t * 32;
                if (any_qz) {  // rare: an all-zero query is at distance 0 from all-zero targets, 1 from the rest
                    const unsigned zm = tzbits[(t_begin >> 5) + t] >> (4 * h);  // wave-uniform load
#pragma unroll
                    for (int s = 0; s < NQ; ++s)
                        if (qz[s]) {
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                acc[s][r] = (float)((zm >> ((r & 3) + 8 * (r >> 2))) & 1u);
                        }
                }
#pragma unroll
                for (int s = 0; s < NQ; ++s) {
                    // fast path: can any of my 16 candidates beat the current k-th best?
                    float mx = acc[s][0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[s][r]);
                    if (dbg & 1) {  // timing experiment: MFMA + fast path only
                        if (mx > 3.0e38f) st[s].tau = mx;
                        continue;
                    }
                    if (__any(mx > st[s].cfloor))
                        topk_append<NT, QW>(acc[s], st[s], lists, queues + s * qcap * NT, (tau_shared + qslot[s]), ql[s],
                                            K, tid, h, t_base + tile_row0, t_end - tile_row0, qcap, (dbg & 2) != 0);
                    if ((t & 31) == 31 && !(dbg & 4)) {  // refresh the cross-segment bound now and then
                        st[s].tau = topk_share((tau_shared + qslot[s]), st[s].taukey, h, st[s].foreign);
                        st[s].cfloor = sim_floor(st[s].tau);
                    }
                }
            }
            __syncthreads();  // stage it+1 is complete (every wave's pieces) before anyone reads it
        }
    }

    // ---- drain the queues, then write this segment's lists: partial[seg][query][K] ----
#pragma unroll
    for (int s = 0; s < NQ; ++s)
        if (__any(st[s].qcnt > 0))
            st[s] = topk_flush<NT, QW>(st[s], lists, queues + s * qcap * NT, (tau_shared + qslot[s]), ql[s], K, tid, h,
                                       qcap, (dbg & 2) != 0);
    __syncthreads();
    {
        u64 *out = partial + ((size_t)blockIdx.y * nq_pad + (size_t)blockIdx.x * QW) * K;
        const int total = QW * K;
        for (int i = tid; i < total; i += NT) {
            const int q = i / K, e = i % K;
            out[i] = lists[e * QW + q];
        }
    }
}

// ------------------------------------------------------------------------------------------
// K4  merge: one wave per query selects the K smallest keys out of nseg * K, ascending.
// Keys are unique (each target lives in exactly one segment), so "smallest key greater than the
// previous pick" enumerates them in order.
// ------------------------------------------------------------------------------------------
#define MERGE_CAP 512  // keys per query staged in LDS; longer candidate sets are re-read from global

__global__ __launch_bounds__(256) void knn_merge_kernel(const u64 *__restrict__ partial, int nseg,
                                                        int nq, int nq_pad, int K,
                                                        int *__restrict__ idx_out,
                                                        float *__restrict__ dist_out) {
    __shared__ u64 stage[4][MERGE_CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    if (q >= nq) return;
    const int M = nseg * K;
    const bool staged = M <= MERGE_CAP;
    u64 *mine_lds = stage[wave];
    if (staged) {  // one pass over global memory; the k selection rounds then run out of LDS
        for (int m = lane; m < M; m += 64) {
            const int seg = m / K, e = m - seg * K;
            mine_lds[m] = partial[((size_t)seg * nq_pad + q) * K + e];
        }
    }
    u64 prev1 = 0;  // previous pick + 1 (0 = none yet)
    u64 mine = 0;
    for (int r = 0; r < K; ++r) {
        u64 best = ~0ull;
        for (int m = lane; m < M; m += 64) {
            u64 kv;
            if (staged) {
                kv = mine_lds[m];
            } else {
                const int seg = m / K, e = m - seg * K;
                kv = partial[((size_t)seg * nq_pad + q) * K + e];
            }
            if (kv + 1 > prev1 && kv < best) best = kv;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const u64 o = __shfl_xor(best, off);
            best = o < best ? o : best;
        }
        prev1 = best + 1;
        if (lane == r) mine = best;
    }
    if (lane < K) {
        idx_out[(size_t)q * K + lane] = (int)(unsigned)(mine & 0xffffffffull);
        dist_out[(size_t)q * K + lane] = __uint_as_float((unsigned)(mine >> 32));
    }
}

// ------------------------------------------------------------------------------------------
// Half-precision prefilter (the default mode whenever k + 8 <= 64 and the lists fit in LDS).
//
// P1 knn_prefilter_kernel: the same tiling and top-k machinery as K3, but the similarities come
//    from v_mfma_f32_32x32x16_f16 on fp16 copies of the normalised rows (16x the fp32 MFMA rate)
//    and the lists keep K' = K + 8 candidates per query, ordered by the APPROXIMATE distance.
// P2 knn_rerank_kernel: per query, certifies that the K' candidates contain the exact top-K and, if
//    so, recomputes their distances with the canonical fp32 fma chain and selects the K best by
//    (dist, idx); otherwise the query is queued for the exact kernel (K3).
//
// Certificate.  Let eps bound |s~ - c| over all pairs (fp16 rounding of unit rows: 2*2^-11 relative
// on sum |x||y| <= 1, plus subnormal and fp32 accumulation terms; FDR_PREFILTER_EPS), d~ the
// approximate distances, d~(K) and d~(K') the K-th and K'-th smallest.  If
//        d~(K) + M < 1   and   d~(K) + M < d~(K'),      M = 2*eps + 4e-7,
// then every target outside the list has d~ >= d~(K') > d~(K) + M, hence an exact distance larger
// than d~(K) + eps + 4e-7, while the K list members with the smallest d~ have exact distances
// <= d~(K) + eps: nothing outside the list can reach the exact top-K, not even through an fp32
// rounding tie (the 4e-7), and "< 1" keeps the argument inside the region where the distance is
// strictly monotone in the similarity (no clamp plateau).  All-zero queries (d~ = 1 everywhere) and
// heavy near-tie plateaus fail the test and take the exact path, so the final result is always the
// canonical one.
// ------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define FDR_PREFILTER_EPS 0.00105f
#define FDR_PREFILTER_EXTRA 8

// Key layout of the prefilter pass (see RegList): ib index bits for the longest segment, the rest (at
// most 20) for the quantised distance.  At least 13 distance bits, i.e. segments of at most 2^19 rows.
#define FDR_PREFILTER_MAX_IB 19
static int prefilter_index_bits(int max_segment_rows) {
    int ib = 8;
    while ((1ll << ib) < max_segment_rows) ++ib;
    return ib;
}
// |approximate distance - canonical distance| of a prefilter candidate: fp16 operands + the key grid
static float prefilter_eps(int ib) {
    const int qbits = std::min(20, 32 - ib);
    return FDR_PREFILTER_EPS + 0.5f / (float)((1u << qbits) - 2u) + 1.0e-6f;
}

static int prefilter_extra() {  // candidates kept beyond k (development knob FDR_KNN_EXTRA)
    if (const char *e = getenv("FDR_KNN_EXTRA")) {
        const int v = atoi(e);
        if (v >= 2 && v <= 44) return v;
    }
    return FDR_PREFILTER_EXTRA;
}

// Ehat fp32 [n, DP] (k0 k2 k4 k6 k1 k3 k5 k7 inside each group of 8) -> fp16 [n, DP], natural order
__global__ __launch_bounds__(256) void to_half_kernel(const float *__restrict__ Ehat, long long n_groups,
                                                      _Float16 *__restrict__ out) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;  // one thread per 8 components
    if (t >= n_groups) return;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(Ehat) + t * 2;
    const f32x4 e = src[0], o = src[1];  // even components k0 k2 k4 k6 | odd components k1 k3 k5 k7
    f16x8 h;
    h[0] = (_Float16)e.x; h[1] = (_Float16)o.x; h[2] = (_Float16)e.y; h[3] = (_Float16)o.y;
    h[4] = (_Float16)e.z; h[5] = (_Float16)o.z; h[6] = (_Float16)e.w; h[7] = (_Float16)o.w;
    reinterpret_cast<f16x8 *>(out)[t] = h;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {  // s_waitcnt vmcnt(N) with a literal count
    static_assert(N >= 0 && N <= 16, "count out of range");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else static_assert(N == 0, "add the literal form for this count");
}

// ---- P1 top-k state: 32-bit keys in sorted register lists ------------------------------------
// key = qd << ib | row - segment start.  qd = QM1 - rint(clamp(sim, 0, 1) * QM1) is the approximate
// distance on a grid of QM1 = 2^qbits - 2 steps (qbits = min(20, 32 - ib); ib = bits of the longest
// segment, chosen per launch), so the order of keys is (quantised distance, row).  The grid adds
// 0.5 / QM1 to the prefilter's error bound (see prefilter_eps()).  EMPTY = all ones is larger than any key.
// The K' keys of a query live in REGISTERS: lane j holds LH of them and lane j + 32 the other LH, each
// half sorted ascending (positions beyond K' are pinned to key 0 and skipped at write-out).  The
// list's maximum is therefore max(v[LH-1], partner's v[LH-1]); inserting c means: the half that
// holds the maximum drops it and takes c, one v_med3_u32 per element
//     v'[e] = med3(v[e-1], c, v[e]),  v'[0] = min(v[0], c)
// (inserting EMPTY changes nothing, which is how the other half and idle queries sit the round out).
// A round costs ~LH + 12 VALU instructions for all 32 queries of the wave at once, no LDS and no scan.
#define PK_EMPTY 0xffffffffu

__device__ __forceinline__ unsigned partner32(unsigned x, int h) {  // value held by lane ^ 32
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return h ? r[0] : r[1];
}

template <int LH>
struct RegList {
    unsigned v[LH];
    unsigned pmax;  // partner half's maximum
};

// one insertion round: every query takes the smaller of its two lanes' candidates (PK_EMPTY = none);
// returns the candidate that was considered (the other lane's, if any, must be offered again)
template <int LH>
__device__ __forceinline__ unsigned reglist_round(RegList<LH> &L, unsigned cand, int h) {
    const unsigned c = min(cand, partner32(cand, h));
    const unsigned mymax = L.v[LH - 1];
    const bool hold = mymax > L.pmax || (mymax == L.pmax && h == 0);
    const unsigned ce = (hold && c < mymax) ? c : PK_EMPTY;  // (hold => mymax is the list maximum)
#pragma unroll
    for (int e = LH - 1; e >= 1; --e) L.v[e] = max(min(L.v[e - 1], ce), min(max(L.v[e - 1], ce), L.v[e]));
    L.v[0] = min(L.v[0], ce);
    L.pmax = partner32(L.v[LH - 1], h);
    return c;
}

// NW waves with NQ 32-query sets each (QW = 32*NQ*NW queries per workgroup; the NQ accumulator chains
// of a wave share every target fragment read from LDS).  The targets stream through a two-stage LDS
// ring; a stage holds U "units" of 32 rows x 128 fp16 components (8 KB each).  LDS holds nothing else
// (the top-k lists are in registers), so the workgroups per CU are set by the register budget (WPS
// waves per SIMD).
// The loop is bound by VALU ISSUE, not by the MFMA pipe: a tile's 8 MFMAs occupy the pipe for 256
// cycles, and every vector instruction of the SIMD's waves costs 4 issue cycles beside them.  Hence:
// integer max3 tree on the accumulator bits (10 instructions per tile, group maxima as by-products),
// stage parity unrolled so that every ds_read address is a register + immediate, LDS-DMA sources as
// 32-bit offsets advanced by a constant, the wave number in an SGPR.
template <int DP, int NQ, int NW, int WPS, int U, int LH, bool PAIRED = false>
__global__ __launch_bounds__(64 * NW, WPS) void knn_prefilter_kernel(
    const _Float16 *__restrict__ Qh, int nq, const _Float16 *__restrict__ Th, int nt, int t_base,
    SegBounds segs, int K, int nq_pad, u64 *__restrict__ partial, unsigned *__restrict__ tau_shared,
    int ib, int dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int QW = 32 * NQ * NW;
    constexpr int NCH = DP / 128;            // 128-component chunks (= units) per tile
    constexpr int UNIT_BYTES = 32 * 256;
    constexpr int STAGE_BYTES = U * UNIT_BYTES;  // U units per stage
    constexpr int ROW_BYTES = DP * 2;
    static_assert(NCH % U == 0 || U % NCH == 0, "a stage holds whole tiles or a tile spans whole stages");
    static_assert(NCH <= 2 * U, "a tile spans at most two stages (the loop is unrolled by stage parity)");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: LDS-DMA destinations stay in SGPRs
    const int j = lane & 31, h = lane >> 5;
    const int ql0 = wave * 32 * NQ + j;               // query set n: ql0 + 32 n
    const int qg0 = blockIdx.x * QW + ql0;

    f16x8 b[NQ][NCH * 8];  // B fragments: chunk c, k-step s covers components 128c + 16s + 8h .. + 7
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        const int qg = qg0 + 32 * n;
        const f16x8 *qp = reinterpret_cast<const f16x8 *>(Qh + (size_t)(qg < nq ? qg : nq - 1) * DP);
#pragma unroll
        for (int i = 0; i < NCH * 8; ++i) b[n][i] = qp[2 * i + h];
    }
    // quantisation grid and the lists
    const int qbits = min(20, 32 - ib);
    const unsigned QM1 = (1u << qbits) - 2u;
    const float qscale = (float)QM1, qinv = 1.0f / qscale;
    const int nlive = (K - h + 1) >> 1, dead = LH - nlive;  // live entries of this half
    RegList<LH> L[NQ];
    unsigned flim[NQ];  // cross-segment bound on qd (admits qd <= flim); QM1 + 1 = none
    int cthr[NQ];       // a similarity can enter only if its bit pattern, as a signed int, is >= cthr
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
#pragma unroll
        for (int e = 0; e < LH; ++e) L[n].v[e] = e < dead ? 0u : PK_EMPTY;
        L[n].pmax = PK_EMPTY;
    }
    // bound exchange (see topk_share): publish the list maximum's qd once the list is full
    auto share = [&](RegList<LH> &Ln, unsigned &fl, int n) {
        unsigned *slot = tau_shared + qg0 + 32 * n;
        const unsigned tk = max(Ln.v[LH - 1], Ln.pmax);
        const unsigned mine = tk == PK_EMPTY ? 0x7F800000u : (tk >> ib);
        unsigned seen = mine;
        if (h == 0) {
            const unsigned old = mine <= QM1
                                     ? __hip_atomic_fetch_min(slot, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                     : __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            seen = min(old, mine);
        }
        const unsigned ps = partner32(seen, h);
        fl = min(h ? ps : seen, QM1 + 1);
    };
    // Threshold below which no row of a LATER tile can enter.  Rows arrive in ascending order, so a
    // later row with the list maximum's qd has a larger key than the maximum: the own bound is strict
    // (qd < maximum's qd) -- which is what keeps plateaus (e.g. an all-zero query, every similarity 0)
    // on the fast path.  The cross-segment bound admits ties.  As a similarity the threshold is
    // ((QM1 - lim) - 0.75) / QM1 > 0 (a quarter step below the rounding boundary of the last admitted
    // grid value), or -inf (everything enters), or +inf (nothing does).  Positive floats order like
    // their bit patterns and a negative similarity has a negative pattern, so the test is one signed
    // integer compare: INT_MIN = everything, INT_MAX = nothing.
    auto rethreshold = [&](const RegList<LH> &Ln, unsigned fl) -> int {
        const unsigned tk = max(Ln.v[LH - 1], Ln.pmax);
        const unsigned oq = tk >> ib;
        const unsigned lim = tk == PK_EMPTY ? fl : min(oq - 1u, fl);  // (oq == 0: wraps, handled below)
        int th = lim >= QM1 ? (int)0x80000000u : __float_as_int(((float)(QM1 - lim) - 0.75f) * qinv) + 1;
        if (tk != PK_EMPTY && oq == 0u) th = 0x7fffffff;
        return th;
    };
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        share(L[n], flim[n], n);
        cthr[n] = rethreshold(L[n], flim[n]);
    }

    const int t_begin = segs.b[blockIdx.y];
    const int t_end = min(nt, segs.b[blockIdx.y + 1]);
    const int ntiles = (t_end - t_begin + 31) >> 5;
    const int nunits = ntiles * NCH;
    const int nstages = (nunits + U - 1) / U;

    // LDS-DMA: piece p = wave + NW*u (1 KiB: unit p>>3 of the stage, rows 4*(p&7)..+3 of that unit,
    // 16 lanes per row; the 16-byte slot a lane fetches is XOR-swizzled with the row).  Sources are
    // 32-bit byte offsets from the segment's first row (segments are at most 2^19 rows of <= 1 KiB);
    // rows past the segment end are clamped to its last row (and masked when the tile is scored).
    constexpr int PPW = 8 * U / NW;  // pieces per wave per stage
    static_assert((8 * U) % NW == 0, "unsupported wave count");
    const char *seg_base = reinterpret_cast<const char *>(Th + (size_t)t_begin * DP);
    const unsigned last_row_off = (unsigned)(t_end - 1 - t_begin) * ROW_BYTES;
    unsigned soff[PPW];  // this lane's source row (as a byte offset) for the stage issued next
    unsigned scol[PPW];  // its byte offset inside the row
#pragma unroll
    for (int u = 0; u < PPW; ++u) {
        const int piece = wave + NW * u;
        const int row = 4 * (piece & 7) + (lane >> 4), pslot = lane & 15;
        // unit piece>>3 of a stage: tile (piece>>3) / NCH of the stage (NCH <= U), chunk (piece>>3) % NCH
        // (+ U per odd stage when a tile spans two stages)
        scol[u] = (unsigned)((pslot ^ (row & 15)) * 16 + ((piece >> 3) % NCH) * 256);
        soff[u] = (unsigned)(32 * ((piece >> 3) / NCH) + row) * ROW_BYTES;
    }
    auto issue_stage = [&](auto par_c) {  // par = parity of the stage being issued
        constexpr int par = decltype(par_c)::value;
        unsigned char *dst = smem + par * STAGE_BYTES;
#pragma unroll
        for (int u = 0; u < PPW; ++u) {
            const int piece = wave + NW * u;
            const unsigned off = min(soff[u], last_row_off) + scol[u] + (NCH > U ? par * U * 256 : 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(seg_base + off),
                                             (__attribute__((address_space(3))) void *)(dst + piece * 1024),
                                             16, 0, 0);
            if constexpr (NCH <= U) soff[u] += (unsigned)(32 * (U / NCH)) * ROW_BYTES;  // U / NCH tiles further
            else if constexpr (par == 1) soff[u] += (unsigned)32 * ROW_BYTES;  // second half done: next tile
        }
    };
    if (nstages > 0) issue_stage(std::integral_constant<int, 0>{});
    __syncthreads();  // (hipcc drains the DMA before the barrier)

    // Candidates wait in a two-entry queue per lane (registers) and enter the lists in batches: late in a
    // scan a tile offers ~1 candidate to ONE of the wave's 32 queries, and a round costs the same for 1
    // query as for 32.  A flush drains the queues with at most four rounds.  Queued candidates are not
    // reflected in the threshold: it is only looser for that, never wrong.
    unsigned q0[NQ], q1[NQ];
#pragma unroll
    for (int n = 0; n < NQ; ++n) q0[n] = q1[n] = PK_EMPTY;
    auto flush = [&](RegList<LH> &Ln, unsigned &a0, unsigned &a1) {
        const bool dbgc = (dbg & 2) != 0;
        while (__any(a0 != PK_EMPTY)) {
            DBG_COUNT(4);
            const unsigned took = reglist_round<LH>(Ln, a0, h);  // the smaller head of the query's two lanes
            if (a0 == took) {
                a0 = a1;
                a1 = PK_EMPTY;
            }
        }
    };
    // the rows of one finished tile against query set n's list (cold: most tiles have no candidate)
    auto offer = [&](const f32x16 &a, const int (&g)[4], RegList<LH> &Ln, unsigned &a0, unsigned &a1,
                     const unsigned fl, const int th, int lrow, int nvalid) {
        const bool dbgc = (dbg & 2) != 0;
        DBG_COUNT(1);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            if (!__any(g[q4] >= th)) continue;
            DBG_COUNT(2);
#pragma unroll
            for (int r = 4 * q4; r < 4 * q4 + 4; ++r) {
                const int roff = (r & 3) + 8 * (r >> 2);
                const bool pass = __float_as_int(a[r]) >= th && roff + 4 * h < nvalid;
                if (!__any(pass)) continue;
                unsigned cand = PK_EMPTY;
                if (pass) {
                    const float sc = fminf(fmaxf(a[r], 0.0f), 1.0f);
                    const unsigned qd = QM1 - (unsigned)__builtin_rintf(sc * qscale);
                    if (qd <= fl) cand = (qd << ib) | (unsigned)(lrow + roff);
                }
                DBG_COUNT(3);
                if (__any(cand != PK_EMPTY && a1 != PK_EMPTY)) flush(Ln, a0, a1);  // some lane's queue is full
                a1 = (a0 != PK_EMPTY && a1 == PK_EMPTY) ? cand : a1;
                a0 = a0 == PK_EMPTY ? cand : a0;
            }
        }
    };

    // this lane's eight fragment addresses inside a unit (k-step s reads slot (2s + h) ^ (j & 15) of row j)
    // (as LDS addresses, ring base included, so that a read is `ds_read_b128 v, fa offset:stage/unit`)
    typedef const f16x8 __attribute__((address_space(3))) lds_f16x8;
    const unsigned ring = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    unsigned fa[8];
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
        fa[s8] = ring + (unsigned)(j * 256 + (((2 * s8 + h) ^ (j & 15)) * 16));
        asm volatile("" : "+v"(fa[s8]));  // (opaque: otherwise hipcc re-derives it with a v_add per read)
    }

    f32x16 acc[NQ];
#pragma unroll
    for (int n = 0; n < NQ; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

    // score one finished tile of query set n (hot: a max tree and one compare; cold: offer())
    auto score = [&](const f32x16 &av, int n, int t) __attribute__((always_inline)) {
        int g[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
            g[q4] = max(max(max(__float_as_int(av[4 * q4]), __float_as_int(av[4 * q4 + 1])),
                            __float_as_int(av[4 * q4 + 2])),
                        __float_as_int(av[4 * q4 + 3]));
        const int mx = max(max(max(g[0], g[1]), g[2]), g[3]);
        const bool dbgc = (dbg & 2) != 0;
        DBG_COUNT(0);
        if (dbg & 1) {  // timing experiment: MFMA + fast path only
            if (mx == 0x7fffffff) cthr[n] = mx;
        } else if (__any(mx >= cthr[n])) {
            int lrow = t * 32 + 4 * h;  // row - segment start of this lane's first row
            int nvalid = t_end - (t_begin + t * 32);
            asm volatile("" : "+v"(lrow), "+s"(nvalid));  // keep the cold block's set-up cold
            offer(av, g, L[n], q0[n], q1[n], flim[n], cthr[n], lrow, nvalid);
            cthr[n] = rethreshold(L[n], flim[n]);
        }
    };
    constexpr bool PAIR = PAIRED && NCH == 1 && U == 2 && NQ == 1;  // the stage's two tiles as two MFMA chains
    auto stage_body = [&](auto par_c, int it) {
        constexpr int par = decltype(par_c)::value;
        if (it + 1 < nstages) issue_stage(std::integral_constant<int, par ^ 1>{});  // lands before the barrier below
        if constexpr (PAIR) {
            // two independent accumulator chains (a dependent 32x32x16 MFMA waits for its predecessor's
            // result; the other tile's MFMA fills that slot); both share every query fragment
            const int t0 = 2 * it;
            if (t0 < nunits) {
                f32x16 accB;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[0][r] = 0.f;
                    accB[r] = 0.f;
                }
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) {
                    const f16x8 a0 = *(lds_f16x8 *)(size_t)(fa[s8] + (unsigned)(par * STAGE_BYTES));
                    const f16x8 a1 = *(lds_f16x8 *)(size_t)(fa[s8] + (unsigned)(par * STAGE_BYTES + UNIT_BYTES));
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[0][s8], acc[0], 0, 0, 0);
                    accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[0][s8], accB, 0, 0, 0);
                }
                score(acc[0], 0, t0);
                if (t0 + 1 < nunits) score(accB, 0, t0 + 1);  // (the second unit of a last, odd stage is padding)
            }
        } else {
#pragma unroll
        for (int uu = 0; uu < U; ++uu) {
            constexpr int dummy = 0;
            (void)dummy;
            const int c = NCH <= U ? uu % NCH : (U * par + uu) % NCH;  // static chunk number
            const int unit = U * it + uu;
            if (unit < nunits) {  // wave-uniform
                const int t = unit / NCH;
                if (c == 0) {
#pragma unroll
                    for (int n = 0; n < NQ; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
                }
                if constexpr (WPS <= 2) {  // 256 registers: all eight fragments in flight (+5 % at d = 500)
                    f16x8 a[8];
#pragma unroll
                    for (int s8 = 0; s8 < 8; ++s8)
                        a[s8] = *(lds_f16x8 *)(size_t)(fa[s8] + (unsigned)(par * STAGE_BYTES + uu * UNIT_BYTES));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
                        for (int n = 0; n < NQ; ++n)
                            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s8], b[n][c * 8 + s8], acc[n], 0, 0, 0);
                } else {
#pragma unroll
                    for (int s8 = 0; s8 < 8; ++s8) {
                        const f16x8 a = *(lds_f16x8 *)(size_t)(fa[s8] + (unsigned)(par * STAGE_BYTES + uu * UNIT_BYTES));
#pragma unroll
                        for (int n = 0; n < NQ; ++n)
                            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[n][c * 8 + s8], acc[n], 0, 0, 0);
                    }
                }
                if (c == NCH - 1) {
                    // acc[n][r] = similarity of query j of set n with tile row (r&3) + 8*(r>>2) + 4*h
#pragma unroll
                    for (int n = 0; n < NQ; ++n) score(acc[n], n, t);
                }
            }
        }
        }
        if ((it & 15) == 15) {
#pragma unroll
            for (int n = 0; n < NQ; ++n) {
                flush(L[n], q0[n], q1[n]);
                share(L[n], flim[n], n);
                cthr[n] = rethreshold(L[n], flim[n]);
            }
        }
        __syncthreads();  // stage it+1 is complete (all waves' pieces) before anyone reads it
    };
    for (int it0 = 0; it0 < nstages; it0 += 2) {
        stage_body(std::integral_constant<int, 0>{}, it0);
        if (it0 + 1 < nstages) stage_body(std::integral_constant<int, 1>{}, it0 + 1);
    }
    const unsigned imask = (1u << ib) - 1u;
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        flush(L[n], q0[n], q1[n]);
        u64 *out = partial + ((size_t)blockIdx.y * nq_pad + (size_t)blockIdx.x * QW + ql0 + 32 * n) * K +
                   (h ? (K + 1) >> 1 : 0);
#pragma unroll
        for (int e = 0; e < LH; ++e) {
            if (e >= dead) {
                const unsigned kv = L[n].v[e];
                u64 o = KEY_INF;
                if (kv != PK_EMPTY)
                    o = ((u64)__float_as_uint((float)(kv >> ib) / qscale) << 32) |
                        (unsigned)(t_base + t_begin + (int)(kv & imask));
                out[e - dead] = o;
            }
        }
    }
}

// Ascending bitonic sort of one u64 key per lane across the wave (21 compare-exchange stages).
__device__ __forceinline__ u64 wave_sort64(u64 key, const int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const u64 other = __shfl_xor(key, j);
            const bool up = k == 64 || (lane & k) == 0;
            const bool take_min = ((lane & j) == 0) == up;
            key = (take_min == (other < key)) ? other : key;
        }
    }
    return key;
}

// Ascending bitonic sort of 128 keys, two per lane: a = element `lane`, b = element `lane + 64`.
__device__ __forceinline__ void wave_sort128(u64 &a, u64 &b, const int lane) {
#pragma unroll
    for (int k = 2; k <= 128; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j == 64) {  // (k == 128: ascending) partner = the lane's other element
                const u64 lo = a < b ? a : b, hi = a < b ? b : a;
                a = lo;
                b = hi;
            } else {
                const u64 oa = __shfl_xor(a, j), ob = __shfl_xor(b, j);
                const bool up_a = k >= 64 || (lane & k) == 0;               // direction of element lane
                const bool up_b = k == 128 || (k < 64 && (lane & k) == 0);  // ... of element lane + 64
                const bool lower = (lane & j) == 0;
                a = ((lower == up_a) == (oa < a)) ? oa : a;
                b = ((lower == up_b) == (ob < b)) ? ob : b;
            }
        }
    }
}

// P1's merge: the KP smallest keys of a query's nseg segment lists, sorted.  One wave per query.
// A segment list arrives as two ascending halves ([0, ceil(KP/2)) and the rest; KEY_INF = empty slot),
// so a full list's maximum is the larger of the two last entries, and the smallest such maximum over
// the segments bounds the query's KP-th key.  About KP * (rows / rows of the longest segment) keys
// survive that bound (74 of 140 for the usual five segments); they are compacted into LDS, the first 64
// are sorted with a 64-lane bitonic network, whose KP-th key is a tighter bound for the rest, and a
// second sort of the KP best + the few remaining survivors finishes.  More than 128 survivors (wide
// plateaus), more than 64 in the second sort, or no full list: KP rounds of a wave-wide minimum.
__global__ __launch_bounds__(256) void knn_merge_keys_kernel(const u64 *__restrict__ partial, int nseg,
                                                             int nq, int nq_pad, int KP,
                                                             u64 *__restrict__ cand) {
    __shared__ u64 stage[4][MERGE_CAP];
    static_assert(MERGE_CAP >= 256, "a wave's stage row doubles as its survivor buffer (256 keys)");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u64 *sv = stage[wave];
    const int q = blockIdx.x * 4 + wave;
    if (q >= nq) return;
    const int M = nseg * KP;
    {
        // (the first eight segments' keys and the bound's operands are all in flight together)
        u64 kvs[8];
        auto load_batch = [&](int sg0) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                kvs[i] = (lane < KP && sg0 + i < nseg) ? partial[((size_t)(sg0 + i) * nq_pad + q) * KP + lane]
                                                       : KEY_INF;
        };
        load_batch(0);
        u64 bound = KEY_INF;
        if (lane < nseg) {
            const u64 *l = partial + ((size_t)lane * nq_pad + q) * KP;
            const u64 a = l[((KP + 1) >> 1) - 1], b2 = l[KP - 1];
            if (a != KEY_INF && b2 != KEY_INF) bound = a > b2 ? a : b2;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const u64 o = __shfl_xor(bound, off);
            bound = o < bound ? o : bound;
        }
        int total = 0;
        if (bound != KEY_INF) {
            for (int sg0 = 0;;) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const u64 kv = kvs[i];
                    const bool keep = kv <= bound;  // (bound < KEY_INF)
                    const u64 mask = __ballot(keep);
                    const int pos = total + __popcll(mask & ((1ull << lane) - 1ull));
                    if (keep && pos < 256) sv[pos] = kv;
                    total += __popcll(mask);
                }
                sg0 += 8;
                if (sg0 >= nseg) break;
                load_batch(sg0);
            }
        }
        if (bound != KEY_INF && total <= 256) {  // (total >= KP: the bounding list alone has KP such keys)
            if (total <= 64) {
                const u64 kv = wave_sort64(lane < total ? sv[lane] : KEY_INF, lane);
                if (lane < KP) cand[(size_t)q * KP + lane] = kv;
                return;
            }
            u64 ka = sv[lane];  // (total > 64)
            u64 kb = 64 + lane < total ? sv[64 + lane] : KEY_INF;
            wave_sort128(ka, kb, lane);
            bool done = total <= 128;
            if (!done) {
                // the KP-th of the first 128 bounds what is still needed from the other (at most 128) survivors
                const u64 b2 = __shfl(ka, KP - 1);
                const u64 e0 = 128 + lane < total ? sv[128 + lane] : KEY_INF;
                const u64 e1 = 192 + lane < total ? sv[192 + lane] : KEY_INF;
                const bool k0 = e0 <= b2, k1 = e1 <= b2;
                const u64 m0 = __ballot(k0), m1 = __ballot(k1);
                const int n0 = __popcll(m0), nx = n0 + __popcll(m1);
                if (KP + nx <= 128) {
                    const u64 below = (1ull << lane) - 1ull;
                    if (lane < KP) sv[lane] = ka;
                    if (k0) sv[KP + __popcll(m0 & below)] = e0;
                    if (k1) sv[KP + n0 + __popcll(m1 & below)] = e1;
                    ka = lane < KP + nx ? sv[lane] : KEY_INF;
                    kb = 64 + lane < KP + nx ? sv[64 + lane] : KEY_INF;
                    wave_sort128(ka, kb, lane);
                    done = true;
                }
            }
            if (done) {
                if (lane < KP) cand[(size_t)q * KP + lane] = ka;  // (KP <= 64: the first element of each lane)
                return;
            }
        }
    }
    const bool staged = M <= MERGE_CAP;
    u64 *mine_lds = stage[wave];
    if (staged)
        for (int m = lane; m < M; m += 64) {
            const int seg = m / KP, e = m - seg * KP;
            mine_lds[m] = partial[((size_t)seg * nq_pad + q) * KP + e];
        }
    u64 prev1 = 0, mine = KEY_INF;
    for (int r = 0; r < KP; ++r) {
        u64 best = ~0ull;
        for (int m = lane; m < M; m += 64) {
            u64 kv;
            if (staged) {
                kv = mine_lds[m];
            } else {
                const int seg = m / KP, e = m - seg * KP;
                kv = partial[((size_t)seg * nq_pad + q) * KP + e];
            }
            if (kv + 1 > prev1 && kv < best) best = kv;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const u64 o = __shfl_xor(best, off);
            best = o < best ? o : best;
        }
        // several KEY_INF entries (lists that never filled) are not unique: stop advancing there
        if (best >= KEY_INF) best = KEY_INF; else prev1 = best + 1;
        if (lane == r) mine = best;
    }
    if (lane < KP) cand[(size_t)q * KP + lane] = mine;
}

// P2: certificate + exact re-rank.  One wave per query, one lane per candidate.
__global__ __launch_bounds__(256) void knn_rerank_kernel(
    const u64 *__restrict__ cand, int KP, int K, const float *__restrict__ Qhat,
    const unsigned char *__restrict__ qzero, const float *__restrict__ That, int nq, int DP, int t_base,
    float margin, int *__restrict__ idx_out, float *__restrict__ dist_out, int *__restrict__ counter,
    int *__restrict__ flagged, int *__restrict__ range_list, float *__restrict__ theta) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    if (qzero[q]) {  // all-zero query: closed-form answer (zero_answer_kernel); listed from the back
        if (lane == 0) flagged[nq - 1 - atomicAdd(counter + 1, 1)] = q;
        return;
    }
    u64 key = KEY_INF;
    if (lane < KP) key = cand[(size_t)q * KP + lane];
    const float dt = __uint_as_float((unsigned)(key >> 32));
    const float dK = __shfl(dt, K - 1), dKP = __shfl(dt, KP - 1);
    const bool valid = (dK + margin < 1.0f) && (dK + margin < dKP);
    if (!valid) {
        if (lane == 0) {
            if (range_list && dK + margin < 1.0f) {  // plateau: collect {d~ <= d~(K) + M} in a range pass
                range_list[atomicAdd(counter + 2, 1)] = q;
                theta[q] = dK + margin;
            } else {
                flagged[atomicAdd(counter, 1)] = q;
            }
        }
        return;
    }
    // A candidate with d~ > d~(K) + M has an exact distance above d~(K) + eps + 4e-7, i.e. above the exact
    // distances of the K candidates with the smallest d~: it cannot be in the top K, so its row is not read.
    u64 exact = ~0ull;
    if (lane < KP && key < KEY_INF && dt <= dK + margin) {
        const int tidx = (int)(unsigned)(key & 0xffffffffull);
        const f32x4 *qp = reinterpret_cast<const f32x4 *>(Qhat + (size_t)q * DP);
        const f32x4 *tp = reinterpret_cast<const f32x4 *>(That + (size_t)(tidx - t_base) * DP);
        float c = 0.0f;
        for (int g0 = 0; g0 < DP / 8; g0 += 8) {  // (DP is a multiple of 128: 8 groups = 16 loads in flight)
            f32x4 te[8], to[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                te[i] = tp[2 * (g0 + i)];
                to[i] = tp[2 * (g0 + i) + 1];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {  // canonical chain: components 8g .. 8g+7 in ascending order
                const f32x4 qe = qp[2 * (g0 + i)], qo = qp[2 * (g0 + i) + 1];
                c = __builtin_fmaf(qe.x, te[i].x, c);
                c = __builtin_fmaf(qo.x, to[i].x, c);
                c = __builtin_fmaf(qe.y, te[i].y, c);
                c = __builtin_fmaf(qo.y, to[i].y, c);
                c = __builtin_fmaf(qe.z, te[i].z, c);
                c = __builtin_fmaf(qo.z, to[i].z, c);
                c = __builtin_fmaf(qe.w, te[i].w, c);
                c = __builtin_fmaf(qo.w, to[i].w, c);
            }
        }
        exact = ((u64)__float_as_uint(dist_from_sim(c)) << 32) | (unsigned)tidx;
    }
    const u64 mine = wave_sort64(exact, lane);  // (distinct targets: no equal keys; unused lanes sort last)
    if (lane < K) {
        idx_out[(size_t)q * K + lane] = (int)(unsigned)(mine & 0xffffffffull);
        dist_out[(size_t)q * K + lane] = __uint_as_float((unsigned)(mine >> 32));
    }
}

// ---- range pass for uncertified queries --------------------------------------------------------
// A query whose K' candidates could not be certified usually sits on a plateau of (near-)ties wider
// than K' (rows with one or two non-zero components have hundreds of exact duplicates).  Its exact
// top-K is still contained in { targets with d~ <= theta }, theta = d~(K) + M (same lemma as the
// certificate, which does not need the K' list to be complete), as long as theta < 1.  This pass
// re-streams the fp16 targets for those queries only and collects that set (no top-k state at all, so
// it runs at the MFMA / staging rate); knn_rerank_long_kernel then ranks it with the canonical fp32
// chain.  Queries whose set exceeds RANGE_CAP fall back to the exact kernel.
#define RANGE_CAP 1024

template <int DP, int NW, int WPS>
__global__ __launch_bounds__(64 * NW, WPS) void knn_range_kernel(
    const _Float16 *__restrict__ Qh, const float *__restrict__ theta, int nq,
    const _Float16 *__restrict__ Th, int nt, int t_base, SegBounds segs, int *__restrict__ cnt,
    int *__restrict__ cand) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int QW = 32 * NW;
    constexpr int NCH = DP / 128;
    constexpr int UNIT_BYTES = 32 * 256;
    constexpr int SLOTS = 16;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int qg = blockIdx.x * QW + wave * 32 + j;
    const bool live = qg < nq;
    const int qrow = live ? qg : nq - 1;
    const float th = live ? theta[qrow] : -1.0f;          // admit d~ <= th
    const float sfloor = live ? (1.0f - th) - 3.0e-7f : __builtin_inff();  // conservative similarity form

    f16x8 b[NCH * 8];
    {
        const f16x8 *qp = reinterpret_cast<const f16x8 *>(Qh + (size_t)qrow * DP);
#pragma unroll
        for (int i = 0; i < NCH * 8; ++i) b[i] = qp[2 * i + h];
    }
    const int t_begin = segs.b[blockIdx.y];
    const int t_end = min(nt, segs.b[blockIdx.y + 1]);
    const int ntiles = (t_end - t_begin + 31) >> 5;
    const int nunits = ntiles * NCH;  // one unit per stage, two-stage ring

    constexpr int PPW = 8 / NW;
    static_assert(8 % NW == 0, "unsupported wave count");
    auto issue_stage = [&](int it) {
        unsigned char *dst = smem + (it & 1) * UNIT_BYTES;
        const int t = it / NCH, c = it % NCH;
#pragma unroll
        for (int u = 0; u < PPW; ++u) {
            const int piece = wave + NW * u;
            const int row = 4 * piece + (lane >> 4), pslot = lane & 15;
            const int trow = min(t_begin + 32 * t + row, t_end - 1);
            const _Float16 *src =
                Th + (size_t)trow * DP + (size_t)(c * 128) + (size_t)((pslot ^ (row & 15)) * 8);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dst + piece * 1024),
                                             16, 0, 0);
        }
    };
    if (nunits > 0) issue_stage(0);
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int it0 = 0; it0 < nunits; it0 += NCH) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int it = it0 + c;
            if (it + 1 < nunits) issue_stage(it + 1);
            if (c == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            }
            const f16x8 *sb = reinterpret_cast<const f16x8 *>(smem + (it & 1) * UNIT_BYTES) + j * SLOTS;
            const int sw = j & 15;
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(sb[(2 * s2 + h) ^ sw], b[c * 8 + s2], acc, 0, 0, 0);
            if (c == NCH - 1) {
                float mx = acc[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[r]);
                if (__any(mx > sfloor)) {
                    const int t = it / NCH;
                    const int row0 = t_begin + 32 * t + 4 * h;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (acc[r] > sfloor) {
                            float a = acc[r];
                            asm volatile("" : "+v"(a));
                            const int row = row0 + (r & 3) + 8 * (r >> 2);
                            if (dist_from_sim(a) <= th && row < t_end) {
                                const int pos = atomicAdd(cnt + qg, 1);
                                if (pos < RANGE_CAP) cand[(size_t)qg * RANGE_CAP + pos] = t_base + row;
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
}

// exact ranking of a collected range: one wave per query, keys staged in LDS
__global__ __launch_bounds__(256) void knn_rerank_long_kernel(
    const int *__restrict__ list, int count, const int *__restrict__ cnt, const int *__restrict__ cand,
    int K, const float *__restrict__ Qhat, const float *__restrict__ That, int DP, int t_base,
    int *__restrict__ idx_out, float *__restrict__ dist_out, int *__restrict__ counter,
    int *__restrict__ flagged) {
    __shared__ u64 keys[4][RANGE_CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    if (i >= count) return;
    const int q = list[i];
    const int n = cnt[i];
    if (n > RANGE_CAP || n < K) {  // set too large (or, defensively, too small): exact kernel
        if (lane == 0) flagged[atomicAdd(counter, 1)] = q;
        return;
    }
    const f32x4 *qp = reinterpret_cast<const f32x4 *>(Qhat + (size_t)q * DP);
    for (int m = lane; m < n; m += 64) {
        const int tidx = cand[(size_t)i * RANGE_CAP + m];
        const f32x4 *tp = reinterpret_cast<const f32x4 *>(That + (size_t)(tidx - t_base) * DP);
        float c = 0.0f;
        for (int g = 0; g < DP / 8; ++g) {  // canonical chain, ascending components
            const f32x4 qe = qp[2 * g], qo = qp[2 * g + 1], te = tp[2 * g], to = tp[2 * g + 1];
            c = __builtin_fmaf(qe.x, te.x, c);
            c = __builtin_fmaf(qo.x, to.x, c);
            c = __builtin_fmaf(qe.y, te.y, c);
            c = __builtin_fmaf(qo.y, to.y, c);
            c = __builtin_fmaf(qe.z, te.z, c);
            c = __builtin_fmaf(qo.z, to.z, c);
            c = __builtin_fmaf(qe.w, te.w, c);
            c = __builtin_fmaf(qo.w, to.w, c);
        }
        keys[wave][m] = ((u64)__float_as_uint(dist_from_sim(c)) << 32) | (unsigned)tidx;
    }
    u64 prev1 = 0, mine = 0;
    for (int r = 0; r < K; ++r) {
        u64 best = ~0ull;
        for (int m = lane; m < n; m += 64) {
            const u64 kv = keys[wave][m];
            if (kv + 1 > prev1 && kv < best) best = kv;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const u64 o = __shfl_xor(best, off);
            best = o < best ? o : best;
        }
        prev1 = best + 1;
        if (lane == r) mine = best;
    }
    if (lane < K) {
        idx_out[(size_t)q * K + lane] = (int)(unsigned)(mine & 0xffffffffull);
        dist_out[(size_t)q * K + lane] = __uint_as_float((unsigned)(mine >> 32));
    }
}

__global__ __launch_bounds__(256) void gather_half_queries_kernel(const _Float16 *__restrict__ Qh,
                                                                  const float *__restrict__ theta_all,
                                                                  const int *__restrict__ list, int first,
                                                                  int count, int DP,
                                                                  _Float16 *__restrict__ Qc,
                                                                  float *__restrict__ theta_c,
                                                                  int *__restrict__ cnt) {
    const int i = blockIdx.x;
    if (i >= count) return;
    const int q = list[first + i];
    for (int c = threadIdx.x; c < DP; c += 256) Qc[(size_t)i * DP + c] = Qh[(size_t)q * DP + c];
    if (threadIdx.x == 0) {
        theta_c[i] = theta_all[q];
        cnt[i] = 0;
    }
}

// The neighbours of an all-zero query do not depend on the query: the all-zero targets at distance 0
// in index order, then every other target at distance 1 in index order.  One workgroup scans the
// zero flags for the first K rows of each kind (stops as soon as K zero rows are known).
__global__ __launch_bounds__(1024) void zero_answer_kernel(const unsigned *__restrict__ tzbits, int nt,
                                                           int t_base, int K, int *__restrict__ zidx,
                                                           float *__restrict__ zdist) {
    __shared__ int wz[16], wnz[16];
    __shared__ int zlist[FDR_MAX_K], nzlist[FDR_MAX_K];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwords = (nt + 31) >> 5;
    int zc = 0, nzc = 0;  // found so far (uniform)
    for (int base = 0; base < nwords && zc < K; base += 1024) {  // 32768 rows per pass
        const int w = base + tid;
        unsigned zb = 0, nzb = 0;
        if (w < nwords) {
            const int rows = min(32, nt - 32 * w);
            const unsigned valid = rows == 32 ? ~0u : (1u << rows) - 1u;
            zb = tzbits[w] & valid;
            nzb = valid & ~zb;
        }
        const int z = __popc(zb), nz = __popc(nzb);
        int sz = z, snz = nz;  // inclusive scans over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int a = __shfl_up(sz, off), c = __shfl_up(snz, off);
            if (lane >= off) {
                sz += a;
                snz += c;
            }
        }
        if (lane == 63) {
            wz[wave] = sz;
            wnz[wave] = snz;
        }
        __syncthreads();
        int pz = zc + sz - z, pnz = nzc + snz - nz, tz = 0, tnz = 0;
        for (int w2 = 0; w2 < 16; ++w2) {
            if (w2 < wave) {
                pz += wz[w2];
                pnz += wnz[w2];
            }
            tz += wz[w2];
            tnz += wnz[w2];
        }
        for (unsigned m = zb; m && pz < K; m &= m - 1) zlist[pz++] = 32 * w + __ffs(m) - 1;
        for (unsigned m = nzb; m && pnz < K; m &= m - 1) nzlist[pnz++] = 32 * w + __ffs(m) - 1;
        zc += tz;
        nzc += tnz;
        __syncthreads();
    }
    if (zc > K) zc = K;
    if (tid < K) {
        const bool from_zero = tid < zc;
        zidx[tid] = t_base + (from_zero ? zlist[tid] : nzlist[tid - zc]);
        zdist[tid] = from_zero ? 0.0f : 1.0f;
    }
}

// (launched before the host knows how many all-zero queries there are: the count is read on the device;
// they are the last counter[1] entries of `flagged`, which has nq slots)
__global__ __launch_bounds__(256) void scatter_zero_answer_kernel(const int *__restrict__ zidx,
                                                                  const float *__restrict__ zdist,
                                                                  const int *__restrict__ flagged, int nq,
                                                                  const int *__restrict__ counter, int K,
                                                                  int *__restrict__ idx_out,
                                                                  float *__restrict__ dist_out) {
    const int count = counter[1];
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)count * K) return;
    const int i = (int)(t / K), e = (int)(t - (long long)i * K);
    const int q = flagged[nq - count + i];
    idx_out[(size_t)q * K + e] = zidx[e];
    dist_out[(size_t)q * K + e] = zdist[e];
}

__global__ __launch_bounds__(256) void gather_queries_kernel(const float *__restrict__ Qhat,
                                                             const unsigned char *__restrict__ qzero,
                                                             const int *__restrict__ list, int first,
                                                             int count, int DP, float *__restrict__ Qc,
                                                             unsigned char *__restrict__ qzc) {
    const int i = blockIdx.x;  // one block per flagged query
    if (i >= count) return;
    const int q = list[first + i];
    for (int c = threadIdx.x; c < DP; c += 256) Qc[(size_t)i * DP + c] = Qhat[(size_t)q * DP + c];
    if (threadIdx.x == 0) qzc[i] = qzero[q];
}

__global__ __launch_bounds__(256) void scatter_results_kernel(const int *__restrict__ idxc,
                                                              const float *__restrict__ distc,
                                                              const int *__restrict__ list, int first,
                                                              int count, int K, int *__restrict__ idx_out,
                                                              float *__restrict__ dist_out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= count * K) return;
    const int i = t / K, e = t - i * K;
    const int q = list[first + i];
    idx_out[(size_t)q * K + e] = idxc[t];
    dist_out[(size_t)q * K + e] = distc[t];
}

// ------------------------------------------------------------------------------------------
// Duplicate-row classes.
//
// Sparse embeddings repeat: rows with one non-zero component are all +-e_a after normalisation, all
// zero rows are identical, overlapping reads often hit the same few projected features (4 M synthetic
// reads: 61 % unique rows, classes of ~1400 rows).  Bitwise-identical rows have bitwise-identical
// distances to everything, so the search runs over UNIQUE query rows x UNIQUE target rows and classes
// are expanded afterwards.  If classes are ordered by (dist, smallest member index), the exact top-K
// of the expanded set lies inside the members of the first K classes: an element outside them is
// preceded by K class representatives.  So: k-NN over representatives (stored in ascending index
// order, so the kernels' (dist, row) order is (dist, representative)), then per query take the first
// K members of each of its K classes and keep the K smallest (dist, index).
// Classes come from a 64-bit row hash, a stable radix sort (members stay in index order) and a
// full-row comparison of sorted neighbours (a hash collision only splits a class: harmless).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 mix64(u64 x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

// 16 lanes per row: position-salted, order-independent combination
__global__ __launch_bounds__(256) void hash_rows_kernel(const float *__restrict__ X, int n, int DP,
                                                        u64 *__restrict__ hash, int *__restrict__ idx) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int row = t >> 4, sub = t & 15;
    u64 h = 0;
    if (row < n) {
        const uint4 *p = reinterpret_cast<const uint4 *>(X + (size_t)row * DP);
        for (int i = sub; i < DP / 4; i += 16) {
            const uint4 v = p[i];
            h += mix64(((u64)v.x | ((u64)v.y << 32)) ^ (0x9e3779b97f4a7c15ull * (u64)(2 * i + 1)));
            h += mix64(((u64)v.z | ((u64)v.w << 32)) ^ (0x9e3779b97f4a7c15ull * (u64)(2 * i + 2)));
        }
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) h += __shfl_xor(h, off);
    if (row < n && sub == 0) {
        hash[row] = mix64(h);
        idx[row] = row;
    }
}

// Cheap estimate of the number of duplicate rows, to decide whether the class machinery is worth
// running at all: every row hash goes into an open-addressing table (linear probing, at most 16
// steps); a row that finds its own hash already present counts as a duplicate.  counter[0] += count.
__global__ __launch_bounds__(256) void dedup_probe_kernel(const u64 *__restrict__ hash, int n,
                                                          u64 *__restrict__ table, unsigned tmask,
                                                          int *__restrict__ counter) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool dup = false;
    if (i < n) {
        const u64 hv = hash[i] | 1ull;  // (0 = empty slot)
        unsigned slot = (unsigned)(hv >> 20) & tmask;
        for (int step = 0; step < 16; ++step) {
            const u64 old = atomicCAS(reinterpret_cast<unsigned long long *>(table + slot), 0ull,
                                      (unsigned long long)hv);
            if (old == 0ull) break;
            if (old == hv) {
                dup = true;
                break;
            }
            slot = (slot + 1) & tmask;
        }
    }
    const int c = __popcll(__ballot(dup));
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(counter, c);
}

// flag[p] = 1 if sorted position p starts a new class (hash differs or the rows differ)
__global__ __launch_bounds__(256) void mark_class_starts_kernel(const float *__restrict__ X, int n, int DP,
                                                                const u64 *__restrict__ hash_s,
                                                                const int *__restrict__ idx_s,
                                                                int *__restrict__ flag) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    int f = 1;
    if (p > 0 && hash_s[p] == hash_s[p - 1]) {
        const uint4 *a = reinterpret_cast<const uint4 *>(X + (size_t)idx_s[p] * DP);
        const uint4 *b = reinterpret_cast<const uint4 *>(X + (size_t)idx_s[p - 1] * DP);
        bool same = true;
        for (int i = 0; i < DP / 4 && same; ++i) {
            const uint4 u = a[i], v = b[i];
            same = u.x == v.x && u.y == v.y && u.z == v.z && u.w == v.w;
        }
        f = same ? 0 : 1;
    }
    flag[p] = f;
}

// cid = inclusive scan of flag.  Per sorted position: class of the row, class start, representative mark.
__global__ __launch_bounds__(256) void class_tables_kernel(int n, const int *__restrict__ flag,
                                                           const int *__restrict__ cid_incl,
                                                           const int *__restrict__ idx_s,
                                                           int *__restrict__ cls_of_row,
                                                           int *__restrict__ class_start,
                                                           int *__restrict__ isrep) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int c = cid_incl[p] - 1;
    const int r = idx_s[p];
    cls_of_row[r] = c;
    isrep[r] = flag[p];  // the first member in (stable) sorted order is the smallest index of its class
    if (flag[p]) class_start[c] = p;
    if (p == n - 1) class_start[c + 1] = n;
}

// upos = inclusive scan of isrep over ROW order: representative r becomes unique row upos[r]-1
// (unique rows are therefore in ascending representative order).  Also marks unique rows that have a
// member among the query rows [q0, q0+nq).
__global__ __launch_bounds__(256) void unique_tables_kernel(int n, const int *__restrict__ isrep,
                                                            const int *__restrict__ upos,
                                                            const int *__restrict__ cls_of_row,
                                                            int *__restrict__ u_of_class,
                                                            int *__restrict__ class_of_u) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n || !isrep[r]) return;
    const int u = upos[r] - 1;
    const int c = cls_of_row[r];
    u_of_class[c] = u;
    class_of_u[u] = c;
}

__global__ __launch_bounds__(256) void gather_unique_rows_kernel(const float *__restrict__ X,
                                                                 const unsigned char *__restrict__ zero,
                                                                 int n, int DP, const int *__restrict__ isrep,
                                                                 const int *__restrict__ upos,
                                                                 float *__restrict__ U,
                                                                 unsigned char *__restrict__ uzero) {
    const int t = blockIdx.x * 256 + threadIdx.x;  // 16 lanes per row
    const int r = t >> 4, sub = t & 15;
    if (r >= n || !isrep[r]) return;
    const int u = upos[r] - 1;
    const uint4 *src = reinterpret_cast<const uint4 *>(X + (size_t)r * DP);
    uint4 *dst = reinterpret_cast<uint4 *>(U + (size_t)u * DP);
    for (int i = sub; i < DP / 4; i += 16) dst[i] = src[i];
    if (sub == 0) uzero[u] = zero[r];
}

// which unique rows are needed as queries: those with a member in [q0, q0+nq)
__global__ __launch_bounds__(256) void mark_query_classes_kernel(int q0, int nq,
                                                                 const int *__restrict__ cls_of_row,
                                                                 const int *__restrict__ u_of_class,
                                                                 int *__restrict__ uqflag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nq) return;
    uqflag[u_of_class[cls_of_row[q0 + i]]] = 1;  // (benign race: every writer stores 1)
}

// uqpos = inclusive scan of uqflag: unique row u is unique query uqpos[u]-1
__global__ __launch_bounds__(256) void gather_unique_queries_kernel(const float *__restrict__ U,
                                                                    const unsigned char *__restrict__ uzero,
                                                                    int nu, int DP,
                                                                    const int *__restrict__ uqflag,
                                                                    const int *__restrict__ uqpos,
                                                                    float *__restrict__ Q,
                                                                    unsigned char *__restrict__ qz) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int u = t >> 4, sub = t & 15;
    if (u >= nu || !uqflag[u]) return;
    const int j = uqpos[u] - 1;
    const uint4 *src = reinterpret_cast<const uint4 *>(U + (size_t)u * DP);
    uint4 *dst = reinterpret_cast<uint4 *>(Q + (size_t)j * DP);
    for (int i = sub; i < DP / 4; i += 16) dst[i] = src[i];
    if (sub == 0) qz[j] = uzero[u];
}

// One wave per QUERY row: look up its class's unique-query result (K classes by (dist, representative)),
// take the first K members of each class (ascending index) and keep the K smallest (dist, index).
// Lane r < K owns class r of the list (the four dependent table look-ups run once, in parallel);
// dynamic LDS = K * K keys (K * min(K, class size) <= K * K).
__global__ __launch_bounds__(64) void expand_classes_kernel(
    int q0, int nq, int K, int t_base, const int *__restrict__ cls_of_row,
    const int *__restrict__ u_of_class, const int *__restrict__ uqpos, const int *__restrict__ idx_u,
    const float *__restrict__ dist_u, const int *__restrict__ class_of_u,
    const int *__restrict__ class_start, const int *__restrict__ idx_s, int *__restrict__ idx_out,
    float *__restrict__ dist_out) {
    extern __shared__ u64 keys[];
    const int lane = threadIdx.x;
    const int i = blockIdx.x;
    if (i >= nq) return;
    const int j = uqpos[u_of_class[cls_of_row[q0 + i]]] - 1;  // this query's unique-query number
    int s0 = 0, m = 0;
    unsigned db = 0;
    if (lane < K) {
        const int u = idx_u[(size_t)j * K + lane];
        db = __float_as_uint(dist_u[(size_t)j * K + lane]);
        const int c = class_of_u[u];
        s0 = class_start[c];
        m = min(K, class_start[c + 1] - s0);
    }
    int incl = m;  // inclusive prefix sum of m over the lanes
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    const int n = __shfl(incl, 63);
    const int base = incl - m;
    for (int r = 0; r < K; ++r) {
        const int rs0 = __shfl(s0, r), rm = __shfl(m, r), rb = __shfl(base, r);
        const unsigned rdb = (unsigned)__shfl((int)db, r);
        if (lane < rm) keys[rb + lane] = ((u64)rdb << 32) | (unsigned)(t_base + idx_s[rs0 + lane]);
    }
    __syncthreads();
    u64 prev1 = 0, mine = 0;
    for (int r = 0; r < K; ++r) {
        u64 best = ~0ull;
        for (int e = lane; e < n; e += 64) {
            const u64 kv = keys[e];
            if (kv + 1 > prev1 && kv < best) best = kv;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const u64 o = __shfl_xor(best, off);
            best = o < best ? o : best;
        }
        prev1 = best + 1;
        if (lane == r) mine = best;
    }
    if (lane < K) {
        idx_out[(size_t)i * K + lane] = (int)(unsigned)(mine & 0xffffffffull);
        dist_out[(size_t)i * K + lane] = __uint_as_float((unsigned)(mine >> 32));
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return FDR_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(FDR_E_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        }
        cap = bytes;
        return FDR_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct fdr_ctx {
    int device = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop;
    // projection
    long long n_features = 0;
    int d = 0;
    long long p_nnz = 0, p_rows = 0;
    DevBuf ftab, crow, ent;
    // scratch for the host-pointer API
    DevBuf a_indptr, a_indices, E, Ehat, zero, idx, dist, ws;
    // k-mer search (kmer_search.inc)
    DevBuf ks_seq, ks_off, ks_codes, ks_keys, ks_vals, ks_bloom, ks_counter, ks_pairs, ks_pairs2, ks_flag, ks_pos,
        ks_idx, ks_rows, ks_indptr, ks_tmp, kc_counts;
    long long ks_nnz = 0, kc_n = 0;
    // timing: when enabled, every launch of kernel kind i gets its own hipEvent pair on the launch
    // stream; fdr_timing_read() sums the elapsed times of all launches since the last read
    int knn_mode = FDR_MODE_AUTO;
    int last_flagged = 0;  // prefilter mode: queries of the last call that took the exact path
    int last_unique_targets = 0, last_unique_queries = 0;  // duplicate-row classes of the last call
    bool timing = false;
    std::vector<hipEvent_t> ev_pool[FDR_NUM_KERNELS];  // start, stop, start, stop, ...
    size_t ev_used[FDR_NUM_KERNELS] = {};
};

static int timing_begin(fdr_ctx *ctx, int kind, hipStream_t st) {
    if (!ctx->timing) return FDR_OK;
    std::vector<hipEvent_t> &pool = ctx->ev_pool[kind];
    if (ctx->ev_used[kind] + 2 > pool.size()) {
        hipEvent_t a, b;
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&b));
        pool.push_back(a);
        pool.push_back(b);
    }
    HIP_TRY(hipEventRecord(pool[ctx->ev_used[kind]], st));
    return FDR_OK;
}

static int timing_end(fdr_ctx *ctx, int kind, hipStream_t st) {
    if (!ctx->timing) return FDR_OK;
    HIP_TRY(hipEventRecord(ctx->ev_pool[kind][ctx->ev_used[kind] + 1], st));
    ctx->ev_used[kind] += 2;
    return FDR_OK;
}

static int use_device(fdr_ctx *ctx) {
    if (!ctx) return fail(FDR_E_ARG, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    return FDR_OK;
}

FDR_EXPORT const char *fdr_last_error(void) { return g_err; }

FDR_EXPORT int fdr_padded_dim(int d) {
    if (d <= 0) return FDR_E_ARG;
    if (d <= 128) return 128;
    if (d <= 256) return 256;
    if (d <= 512) return 512;
    return FDR_E_ARG;
}

FDR_EXPORT int fdr_create(int device_id, fdr_ctx **out) {
    if (!out) return fail(FDR_E_ARG, "fdr_create: out is null");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (n <= 0) return fail(FDR_E_HIP, "no HIP device visible: the MI355X path cannot run");
    if (device_id < 0 || device_id >= n)
        return fail(FDR_E_ARG, "device %d out of range (have %d)", device_id, n);
    fdr_ctx *c = new (std::nothrow) fdr_ctx();
    if (!c) return fail(FDR_E_NOMEM, "out of host memory");
    c->device = device_id;
    HIP_TRY(hipSetDevice(device_id));
    HIP_TRY(hipGetDeviceProperties(&c->prop, device_id));
    c->num_cus = c->prop.multiProcessorCount > 0 ? c->prop.multiProcessorCount : 256;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    *out = c;
    return FDR_OK;
}

FDR_EXPORT int fdr_destroy(fdr_ctx *ctx) {
    if (!ctx) return FDR_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    DevBuf *bufs[] = {&ctx->ftab, &ctx->crow, &ctx->ent, &ctx->a_indptr, &ctx->a_indices, &ctx->E,
                      &ctx->Ehat, &ctx->zero, &ctx->idx, &ctx->dist, &ctx->ws,
                      &ctx->ks_seq, &ctx->ks_off, &ctx->ks_codes, &ctx->ks_keys, &ctx->ks_vals, &ctx->ks_bloom,
                      &ctx->ks_counter, &ctx->ks_pairs, &ctx->ks_pairs2, &ctx->ks_flag, &ctx->ks_pos,
                      &ctx->ks_idx, &ctx->ks_rows, &ctx->ks_indptr, &ctx->ks_tmp, &ctx->kc_counts};
    for (DevBuf *b : bufs) b->release();
    for (int i = 0; i < FDR_NUM_KERNELS; ++i)
        for (hipEvent_t e : ctx->ev_pool[i]) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return FDR_OK;
}

FDR_EXPORT int fdr_device_info(fdr_ctx *ctx, char *buf, int buflen) {
    if (!ctx || !buf || buflen <= 0) return fail(FDR_E_ARG, "fdr_device_info: bad argument");
    snprintf(buf, (size_t)buflen, "%s|%s|%d|%zu", ctx->prop.name, ctx->prop.gcnArchName,
             ctx->prop.multiProcessorCount, (size_t)ctx->prop.totalGlobalMem);
    return FDR_OK;
}

FDR_EXPORT int fdr_last_uncertified(fdr_ctx *ctx) { return ctx ? ctx->last_flagged : 0; }

FDR_EXPORT int fdr_last_unique(fdr_ctx *ctx, int *unique_targets, int *unique_queries) {
    if (!ctx || !unique_targets || !unique_queries) return fail(FDR_E_ARG, "bad argument");
    *unique_targets = ctx->last_unique_targets;
    *unique_queries = ctx->last_unique_queries;
    return FDR_OK;
}

FDR_EXPORT int fdr_set_knn_mode(fdr_ctx *ctx, int mode) {
    if (!ctx || mode < FDR_MODE_AUTO || mode > FDR_MODE_PREFILTER) return fail(FDR_E_ARG, "bad k-NN mode");
    ctx->knn_mode = mode;
    return FDR_OK;
}

FDR_EXPORT int fdr_timing(fdr_ctx *ctx, int enable) {
    if (!ctx) return fail(FDR_E_ARG, "null context");
    ctx->timing = enable != 0;
    for (int i = 0; i < FDR_NUM_KERNELS; ++i) ctx->ev_used[i] = 0;
    return FDR_OK;
}

FDR_EXPORT int fdr_timing_read(fdr_ctx *ctx, int which, int *count_out, float *total_ms_out) {
    int rc = use_device(ctx);
    if (rc) return rc;
    if (which < 0 || which >= FDR_NUM_KERNELS || !count_out || !total_ms_out)
        return fail(FDR_E_ARG, "fdr_timing_read: bad argument");
    float total = 0.f;
    const size_t used = ctx->ev_used[which];
    for (size_t i = 0; i + 1 < used; i += 2) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(ctx->ev_pool[which][i + 1]));
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_pool[which][i], ctx->ev_pool[which][i + 1]));
        total += ms;
    }
    *count_out = (int)(used / 2);
    *total_ms_out = total;
    ctx->ev_used[which] = 0;
    return FDR_OK;
}

// ---- projection ------------------------------------------------------------------------------
FDR_EXPORT int fdr_projection_load(fdr_ctx *ctx, int64_t n_features, int32_t d,
                                   const int64_t *p_indptr, const int32_t *p_cols,
                                   const float *p_vals) {
    int rc = use_device(ctx);
    if (rc) return rc;
    if (n_features <= 0 || n_features > 0x7fffffffll || !p_indptr)
        return fail(FDR_E_ARG, "projection: bad n_features %lld", (long long)n_features);
    if (fdr_padded_dim(d) < 0)
        return fail(FDR_E_ARG, "projection: embedding dimension %d unsupported (1..%d)", d,
                    FDR_MAX_DIM);
    const int64_t nnz = p_indptr[n_features];
    if (p_indptr[0] != 0 || nnz < 0 || nnz > 0x7fffffffll || (nnz > 0 && (!p_cols || !p_vals)))
        return fail(FDR_E_ARG, "projection: bad CSR arrays");
    const int64_t nwords = (n_features + 31) / 32;
    std::vector<uint2> ftab((size_t)nwords);
    std::vector<uint4> crow;  // rowinfo
    std::vector<uint2> ent((size_t)std::max<int64_t>(nnz, 1));
    crow.reserve(1024);
    unsigned rows = 0;
    for (int64_t w = 0; w < nwords; ++w) {
        unsigned bits = 0;
        const unsigned prefix = rows;
        const int64_t f0 = w * 32, f1 = std::min<int64_t>(n_features, f0 + 32);
        for (int64_t f = f0; f < f1; ++f) {
            const int64_t s = p_indptr[f], e = p_indptr[f + 1];
            if (e < s || e > nnz) return fail(FDR_E_ARG, "projection: indptr not monotone at %lld", (long long)f);
            if (e > s) {
                bits |= 1u << (unsigned)(f - f0);
                uint32_t fv;
                memcpy(&fv, &p_vals[s], 4);
                crow.push_back(make_uint4((unsigned)s, (unsigned)(e - s), (unsigned)p_cols[s], fv));
                ++rows;
                for (int64_t q = s; q < e; ++q) {
                    if (p_cols[q] < 0 || p_cols[q] >= d)
                        return fail(FDR_E_ARG, "projection: column %d out of range at nnz %lld",
                                    p_cols[q], (long long)q);
                    uint32_t vb;
                    memcpy(&vb, &p_vals[q], 4);
                    ent[(size_t)q] = make_uint2((unsigned)p_cols[q], vb);
                }
            }
        }
        ftab[(size_t)w] = make_uint2(bits, prefix);
    }
    if (crow.empty()) crow.push_back(make_uint4(0u, 0u, 0u, 0u));
    if ((rc = ctx->ftab.reserve(ftab.size() * sizeof(uint2)))) return rc;
    if ((rc = ctx->crow.reserve(crow.size() * sizeof(uint4)))) return rc;
    if ((rc = ctx->ent.reserve(ent.size() * sizeof(uint2)))) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->ftab.p, ftab.data(), ftab.size() * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->crow.p, crow.data(), crow.size() * sizeof(uint4), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->ent.p, ent.data(), ent.size() * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->n_features = n_features;
    ctx->d = d;
    ctx->p_nnz = nnz;
    ctx->p_rows = rows;
    return FDR_OK;
}

// ---- launches --------------------------------------------------------------------------------
static int launch_embed(fdr_ctx *ctx, int64_t n_rows, const int64_t *d_indptr,
                        const int32_t *d_indices, float *d_E, hipStream_t st) {
    if (ctx->n_features <= 0) return fail(FDR_E_STATE, "embed: no projection loaded");
    if (n_rows < 0) return fail(FDR_E_ARG, "embed: n_rows < 0");
    if (n_rows == 0) return FDR_OK;
    const int dp = fdr_padded_dim(ctx->d);
    const long long blocks_needed = (n_rows + 3) / 4;
    const int grid = (int)std::min<long long>(blocks_needed, (long long)ctx->num_cus * 8 * 4);
    int trc = timing_begin(ctx, FDR_KERNEL_EMBED, st);
    if (trc) return trc;
#define FDR_LAUNCH_EMBED(DP_)                                                                   \
    hipLaunchKernelGGL(embed_csr_kernel<DP_>, dim3(grid), dim3(256), 0, st, (long long)n_rows,  \
                       (const long long *)d_indptr, d_indices, ctx->n_features,                 \
                       (const uint2 *)ctx->ftab.p, (const uint4 *)ctx->crow.p,                  \
                       (const uint2 *)ctx->ent.p, ctx->d, d_E)
    if (dp == 128)
        FDR_LAUNCH_EMBED(128);
    else if (dp == 256)
        FDR_LAUNCH_EMBED(256);
    else
        FDR_LAUNCH_EMBED(512);
#undef FDR_LAUNCH_EMBED
    HIP_TRY(hipGetLastError());
    return timing_end(ctx, FDR_KERNEL_EMBED, st);
}

static int launch_normalize(fdr_ctx *ctx, const float *d_E, int64_t n_rows, int d, float *d_Ehat,
                            uint8_t *d_zero, hipStream_t st) {
    const int dp = fdr_padded_dim(d);
    if (dp < 0) return fail(FDR_E_ARG, "normalize: dimension %d unsupported (1..%d)", d, FDR_MAX_DIM);
    if (n_rows < 0) return fail(FDR_E_ARG, "normalize: n_rows < 0");
    if (n_rows == 0) return FDR_OK;
    const int rb = dp == 128 ? 64 : (dp == 256 ? 32 : 16);
    const long long grid = (n_rows + rb - 1) / rb;
    if (grid > 0x7fffffffll) return fail(FDR_E_ARG, "normalize: too many rows");
    int trc = timing_begin(ctx, FDR_KERNEL_NORMALIZE, st);
    if (trc) return trc;
    if (dp == 128)
        hipLaunchKernelGGL((normalize_rows_kernel<128, 64>), dim3((unsigned)grid), dim3(64), 0, st,
                           d_E, (long long)n_rows, d, d_Ehat, d_zero);
    else if (dp == 256)
        hipLaunchKernelGGL((normalize_rows_kernel<256, 32>), dim3((unsigned)grid), dim3(64), 0, st,
                           d_E, (long long)n_rows, d, d_Ehat, d_zero);
    else
        hipLaunchKernelGGL((normalize_rows_kernel<512, 16>), dim3((unsigned)grid), dim3(64), 0, st,
                           d_E, (long long)n_rows, d, d_Ehat, d_zero);
    HIP_TRY(hipGetLastError());
    return timing_end(ctx, FDR_KERNEL_NORMALIZE, st);
}

// Kernel shapes.  queries/workgroup QW = 32*NQ*NW; slots = workgroups resident per CU (LDS- and
// register-limited).  Shape choice: see knn_choose_shape().
struct KnnShape {
    int dp, nq, nw, wps;
    int tps;  // > 0: fp16 prefilter shape whose LDS ring holds tps one-tile (8 KB) stages
};
static const KnnShape kShapes[] = {
    {128, 1, 4, 3, 0},  // 128 queries/WG, <=168 VGPRs, up to 3 WG/CU
    {128, 1, 8, 4, 0},  // 256 queries/WG, <=128 VGPRs, up to 2 WG/CU
    {128, 2, 4, 2, 0},  // 256 queries/WG, <=256 VGPRs, up to 2 WG/CU
    {256, 1, 8, 2, 0},  // 256 queries/WG, <=256 VGPRs, 1 WG/CU
    {512, 1, 4, 1, 0},  // 128 queries/WG, one wave per SIMD: 512 registers per lane (256 of them queries)
    // fp16 prefilter (LDS = the ring only): 128 queries/WG, two-unit stages (32 KB ring).  One-unit
    // stages (a barrier per 8 MFMAs) cost +25 % at d <= 128, three / four-unit stages lose workgroups per
    // CU, two query sets per wave (NQ = 2, 192 VGPRs) lost 10-25 % to the lower occupancy.
    {128, 1, 4, 4, 4},  // d <= 128: <= 128 VGPRs, 4 WG/CU
    {256, 1, 4, 3, 4},  // d <= 256: <= 168 VGPRs, one tile per stage
    {512, 1, 4, 2, 4},  // d <= 512: 128 VGPRs of queries, half a tile per stage
};
#define FDR_SHAPE_PREFILTER 5  // + 0 / 1 / 2 for d <= 128 / 256 / 512
static int range_shape(int dp) { return FDR_SHAPE_PREFILTER + (dp == 128 ? 0 : dp == 256 ? 1 : 2); }
static int prefilter_shape(int dp) { return range_shape(dp); }

static size_t knn_lds_bytes_q(const KnnShape &sh, int k, int qcap) {
    const size_t qw = (size_t)32 * sh.nq * sh.nw, nt = (size_t)64 * sh.nw;
    if (sh.tps > 0) return (size_t)sh.tps * 32 * 256;  // fp16 prefilter: the ring only (lists in registers)
    const size_t ring = (size_t)2 * 32 * 64 * 4;
    return ring + (size_t)k * qw * 8 + (size_t)qcap * sh.nq * nt * 8;
}

// Entries per lane append queue: 4, or 2 when that lets one more workgroup share the CU's LDS
// (co-resident workgroups matter more than the flush rate).
static int knn_qcap(const KnnShape &sh, int k) {
    const size_t lds = 160 * 1024;
    const int with4 = (int)(lds / knn_lds_bytes_q(sh, k, 4)), with2 = (int)(lds / knn_lds_bytes_q(sh, k, 2));
    const int by_regs = std::max(1, (sh.wps * 4) / sh.nw);
    return std::min(with2, by_regs) > std::min(with4, by_regs) ? 2 : QCAP;
}

static size_t knn_lds_bytes(const KnnShape &sh, int k) { return knn_lds_bytes_q(sh, k, knn_qcap(sh, k)); }

static int knn_wg_per_cu(const KnnShape &sh, int k) {
    const int by_lds = (int)((size_t)160 * 1024 / knn_lds_bytes(sh, k));
    const int by_regs = (sh.wps * 4) / sh.nw;  // waves per CU the register budget allows / waves per WG
    return std::max(0, std::min(by_lds, std::max(by_regs, 1)));
}

static int knn_choose_shape(int dp, int k) {
    if (const char *e = getenv("FDR_KNN_SHAPE")) {  // development knob: index into kShapes
        const int i = atoi(e);
        if (i >= 0 && i < (int)(sizeof(kShapes) / sizeof(kShapes[0])) && kShapes[i].dp == dp &&
            knn_wg_per_cu(kShapes[i], k) > 0)
            return i;
    }
    if (dp == 256) return 3;
    if (dp == 512) return 4;
    // d <= 128: the 4-wave / 128-query shape (no spills at 168 VGPRs, finest work granularity) while
    // at least two workgroups fit in LDS; for larger k the 2-chain 256-VGPR shape
    if (knn_wg_per_cu(kShapes[0], k) >= 2) return 0;
    return 2;
}

struct KnnPlan {
    int shape, qw, nqb, nseg, nq_pad;
    SegBounds segs;
    size_t bits_bytes;     // packed zero-target flags, at the start of the workspace
    size_t shared_bytes;   // one cross-segment bound word per (padded) query
    size_t partial_bytes;  // per-segment top-k lists
    size_t total_bytes;
};

// Makespan (in 32-row tiles) of dispatching, in order, nqb workgroups per segment onto `slots`
// concurrently resident workgroups; every workgroup costs its segment's tiles + `ov` tiles of fixed
// work (query load, list set-up, final flush and write-out).
static double simulate_makespan(const std::vector<int> &seg_tiles, int nqb, int slots, double ov) {
    std::vector<double> heap((size_t)slots, 0.0);  // min-heap of slot free times
    auto cmp = [](double a, double b) { return a > b; };
    double last = 0.0;
    for (int t : seg_tiles)
        for (int q = 0; q < nqb; ++q) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            const double done = heap.back() + (double)t + ov;
            heap.back() = done;
            std::push_heap(heap.begin(), heap.end(), cmp);
            last = std::max(last, done);
        }
    return last;
}

static KnnPlan knn_plan_compute(const fdr_ctx *ctx, int64_t nq, int64_t nt, int d, int k, int shape);

// the plan search simulates a few hundred dispatch orders: remember the last few results
struct PlanCacheEntry {
    int64_t nq, nt;
    int dp, k, cus, shape;
    KnnPlan plan;
};
static thread_local std::vector<PlanCacheEntry> g_plan_cache;

static KnnPlan knn_plan(const fdr_ctx *ctx, int64_t nq, int64_t nt, int d, int k, int shape = -1) {
    const int dp = fdr_padded_dim(d);
    for (const PlanCacheEntry &e : g_plan_cache)
        if (e.nq == nq && e.nt == nt && e.dp == dp && e.k == k && e.cus == ctx->num_cus && e.shape == shape)
            return e.plan;
    PlanCacheEntry e{nq, nt, dp, k, ctx->num_cus, shape, knn_plan_compute(ctx, nq, nt, d, k, shape)};
    if (g_plan_cache.size() >= 16) g_plan_cache.erase(g_plan_cache.begin());
    g_plan_cache.push_back(e);
    if (const char *dbg = getenv("FDR_KNN_DEBUG")) {
        if (atoi(dbg) & 8) {
            fprintf(stderr, "[fdr plan] nq=%lld nt=%lld shape=%d nqb=%d nseg=%d tiles:", (long long)nq,
                    (long long)nt, e.plan.shape, e.plan.nqb, e.plan.nseg);
            for (int i = 0; i < e.plan.nseg; ++i)
                fprintf(stderr, " %d", (e.plan.segs.b[i + 1] - e.plan.segs.b[i]) / 32);
            fprintf(stderr, "\n");
        }
    }
    return e.plan;
}

static KnnPlan knn_plan_compute(const fdr_ctx *ctx, int64_t nq, int64_t nt, int d, int k, int shape) {
    KnnPlan p;
    p.shape = shape >= 0 ? shape : knn_choose_shape(fdr_padded_dim(d), k);
    const KnnShape &sh = kShapes[p.shape];
    p.qw = 32 * sh.nq * sh.nw;
    p.nqb = (int)((nq + p.qw - 1) / p.qw);
    p.nq_pad = p.nqb * p.qw;
    // Work split.  The grid is nqb query blocks x nseg target segments; workgroups are dispatched
    // segment by segment onto `slots` resident workgroups.  Equal segments leave the last "round"
    // mostly empty whenever nqb * nseg is not just below a multiple of `slots`, so the plan is a few
    // long segments followed by shorter ones that fill the tail (guided self-scheduling), chosen by
    // simulating the dispatch.  Segments of one query block share their bound (topk_share), so the
    // extra segments cost little more than their fixed set-up (`ov`, in tiles).
    int slots = ctx->num_cus * std::max(1, knn_wg_per_cu(sh, k));
    // fp16 prefilter: a fourth workgroup on a CU hides latency but shares the same MFMA / LDS pipes; the
    // dispatch model that matched the measurements best counts three (100 k rows: 3.3 ms vs 3.9 ms)
    if (sh.tps > 0) slots = std::min(slots, ctx->num_cus * 3);
    if (const char *e = getenv("FDR_KNN_SLOTS")) slots = ctx->num_cus * std::max(1, atoi(e));  // development knob
    const int T = (int)((nt + 31) / 32);  // tiles
    double ov = sh.tps > 0 ? 96.0 : 16.0;  // fp16 tiles are 16x shorter: the fixed cost weighs more
    if (const char *e = getenv("FDR_KNN_OV")) ov = atof(e);  // development knob
    // fp16 prefilter keys index rows inside a segment with at most FDR_PREFILTER_MAX_IB bits
    const int cap_tiles = sh.tps > 0 ? (1 << FDR_PREFILTER_MAX_IB) / 32 : T;
    std::vector<int> best;
    {
        const int c0 = (T + cap_tiles - 1) / cap_tiles, l0 = (T + c0 - 1) / c0;
        for (int left = T; left > 0; left -= l0) best.push_back(std::min(left, l0));
    }
    double best_cost = simulate_makespan(best, p.nqb, slots, ov);
    const int min_tiles = sh.tps > 0 ? 128 : 24;  // never cut segments shorter than 768 (4096) rows
    for (int cmain = 1; cmain <= 24; ++cmain)
        for (int tf = 0; tf <= 4; ++tf)          // share of the tiles given to the short tail
            for (int div = 2; div <= 8; div *= 2) {  // tail segments are 1/div of a main segment
                const double tail_frac = 0.08 * tf;
                if (tf == 0 && div != 2) continue;
                const int main_total = (int)((1.0 - tail_frac) * T);
                const int lm = std::max(min_tiles, (main_total + cmain - 1) / cmain);
                std::vector<int> segs;
                int left = T;
                for (int i = 0; i < cmain && left > 0; ++i) {
                    const int t = std::min(left, lm);
                    segs.push_back(t);
                    left -= t;
                }
                const int ls = std::max(min_tiles, lm / div);
                while (left > 0) {
                    const int t = (left < ls + min_tiles) ? left : ls;
                    segs.push_back(t);
                    left -= t;
                }
                if ((int)segs.size() > FDR_MAX_SEG) continue;
                if (*std::max_element(segs.begin(), segs.end()) > cap_tiles) continue;
                const double cost = simulate_makespan(segs, p.nqb, slots, ov);
                if (cost < best_cost * 0.999) {
                    best_cost = cost;
                    best = segs;
                }
            }
    if (const char *e = getenv("FDR_KNN_NSEG")) {  // development knob: nseg equal segments
        const int c = std::max(1, std::min(atoi(e), FDR_MAX_SEG));
        if (atoi(e) > 0) {
            best.clear();
            const int l = (T + c - 1) / c;
            for (int left = T; left > 0; left -= l) best.push_back(std::min(left, l));
        }
    }
    p.nseg = (int)best.size();
    int row = 0;
    for (int i = 0; i < p.nseg; ++i) {
        p.segs.b[i] = row;
        row += best[(size_t)i] * 32;
    }
    for (int i = p.nseg; i <= FDR_MAX_SEG; ++i) p.segs.b[i] = row;
    p.partial_bytes = (size_t)p.nseg * p.nq_pad * (size_t)k * sizeof(u64);
    p.bits_bytes = ((size_t)((nt + 31) / 32) * 4 + 255) / 256 * 256;
    p.shared_bytes = ((size_t)p.nq_pad * 4 + 255) / 256 * 256;
    p.total_bytes = p.bits_bytes + p.shared_bytes + p.partial_bytes;
    return p;
}

static size_t knn_workspace_bytes_impl(const fdr_ctx *ctx, int64_t nq, int64_t nt, int d, int k);

// ---- prefilter mode: workspace layout -------------------------------------------------------
// mode: FDR_MODE_AUTO uses the fp16 prefilter whenever it applies (d <= 128, k + 8 <= 64) and the
// target set is large enough to pay for it; FDR_KNN_MODE=exact|prefilter|auto overrides the context.
static bool knn_prefilter_wanted(const fdr_ctx *ctx, int dp, int64_t nt, int k) {
    int mode = ctx->knn_mode;
    if (const char *e = getenv("FDR_KNN_MODE")) {
        if (strcmp(e, "exact") == 0) mode = FDR_MODE_EXACT;
        else if (strcmp(e, "prefilter") == 0) mode = FDR_MODE_PREFILTER;
        else if (strcmp(e, "auto") == 0) mode = FDR_MODE_AUTO;
    }
    if (mode == FDR_MODE_EXACT) return false;
    const int kp = (k + prefilter_extra() + 1) & ~1;
    if (!(kp <= FDR_MAX_K && nt >= kp)) return false;
    if (nt > (int64_t)FDR_MAX_SEG << FDR_PREFILTER_MAX_IB) return false;  // segments too long for the keys
    return mode == FDR_MODE_PREFILTER || nt >= 8192;
}

struct PrefilterLayout {
    int kp, chunk;
    size_t knn_bytes;  // region shared (in stream order) by the prefilter pass and the exact passes
    size_t off_ht, off_hq, off_cand, off_counter, off_flagged, off_qc, off_qzc, off_idxc, off_distc;
    size_t off_rlist, off_theta, off_hqc, off_thetac, off_cnt, off_rcand, total;  // range pass
    int rchunk;
};

static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

static PrefilterLayout prefilter_layout(const fdr_ctx *ctx, int64_t nq, int64_t nt, int d, int k) {
    PrefilterLayout L;
    L.kp = (k + prefilter_extra() + 1) & ~1;
    L.chunk = (int)std::min<int64_t>(nq, 16384);
    const size_t exact_all = knn_plan(ctx, nq, nt, d, k).total_bytes;
    const int dp = fdr_padded_dim(d);
    const size_t pre = knn_plan(ctx, nq, nt, d, L.kp, prefilter_shape(dp)).total_bytes;
    // any exact plan for <= chunk queries: bits + bound words + at most FDR_MAX_SEG segments of lists
    const size_t chunk_bound = align256((size_t)((nt + 31) / 32) * 4) + align256((size_t)(L.chunk + 128) * 4) +
                               (size_t)FDR_MAX_SEG * (L.chunk + 128) * (size_t)k * 8;
    L.knn_bytes = align256(std::max(exact_all, std::max(pre, chunk_bound)));
    size_t o = L.knn_bytes;
    L.off_ht = o;       o += align256((size_t)nt * dp * 2);
    L.off_hq = o;       o += align256((size_t)nq * dp * 2);
    L.off_cand = o;     o += align256((size_t)nq * L.kp * 8);
    L.off_counter = o;  o += 1024;  // [0] exact list, [1] all-zero queries, [2] range list; zero answer at +256 / +512
    L.off_flagged = o;  o += align256((size_t)nq * 4);
    L.off_qc = o;       o += align256((size_t)L.chunk * dp * 4);
    L.off_qzc = o;      o += align256((size_t)L.chunk);
    L.off_idxc = o;     o += align256((size_t)L.chunk * k * 4);
    L.off_distc = o;    o += align256((size_t)L.chunk * k * 4);
    L.rchunk = (int)std::min<int64_t>(nq, 32768);  // range pass: queries per launch
    L.off_rlist = o;    o += align256((size_t)nq * 4);
    L.off_theta = o;    o += align256((size_t)nq * 4);
    L.off_hqc = o;      o += align256((size_t)L.rchunk * dp * 2);
    L.off_thetac = o;   o += align256((size_t)L.rchunk * 4);
    L.off_cnt = o;      o += align256((size_t)L.rchunk * 4);
    L.off_rcand = o;    o += align256((size_t)L.rchunk * RANGE_CAP * 4);
    L.total = o;
    return L;
}

FDR_EXPORT size_t fdr_knn_workspace_bytes(fdr_ctx *ctx, int64_t nq, int64_t nt, int32_t d,
                                          int32_t k) {
    if (!ctx || nq <= 0 || nt <= 0 || k <= 0 || k > FDR_MAX_K || fdr_padded_dim(d) < 0) return 0;
    return knn_workspace_bytes_impl(ctx, nq, nt, d, k);
}

static int launch_knn_exact(fdr_ctx *ctx, const float *d_Qhat, const uint8_t *d_qzero, int64_t nq,
                      const float *d_That, const uint8_t *d_tzero, int64_t nt, int64_t t_base,
                      int d, int k, int32_t *d_idx, float *d_dist, void *d_ws, size_t ws_bytes,
                      hipStream_t st) {
    const int dp = fdr_padded_dim(d);
    if (dp < 0) return fail(FDR_E_ARG, "knn: dimension %d unsupported (1..%d)", d, FDR_MAX_DIM);
    if (k < 1 || k > FDR_MAX_K) return fail(FDR_E_ARG, "knn: k=%d unsupported (1..%d)", k, FDR_MAX_K);
    if (nq < 0 || nt < k) return fail(FDR_E_ARG, "knn: need n_targets (%lld) >= k (%d)", (long long)nt, k);
    if (nt + t_base > 0x7fffffffll || nq > 0x7fffffffll)
        return fail(FDR_E_ARG, "knn: row numbers exceed int32");
    if (nq == 0) return FDR_OK;
    if (!d_Qhat || !d_qzero || !d_That || !d_tzero || !d_idx || !d_dist || !d_ws)
        return fail(FDR_E_ARG, "knn: null device pointer");
    const KnnPlan p = knn_plan(ctx, nq, nt, d, k);
    if (ws_bytes < p.total_bytes)
        return fail(FDR_E_ARG, "knn: workspace %zu < required %zu bytes", ws_bytes, p.total_bytes);
    unsigned *d_bits = reinterpret_cast<unsigned *>(d_ws);
    unsigned *d_shared = reinterpret_cast<unsigned *>(static_cast<char *>(d_ws) + p.bits_bytes);
    u64 *d_partial = reinterpret_cast<u64 *>(static_cast<char *>(d_ws) + p.bits_bytes + p.shared_bytes);
    hipLaunchKernelGGL(pack_zero_bits_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st,
                       d_tzero, (int)nt, d_bits, d_shared, p.nq_pad);
    HIP_TRY(hipGetLastError());
    const KnnShape &sh = kShapes[p.shape];
    const size_t lds = knn_lds_bytes(sh, k);
    if (lds > 160 * 1024) return fail(FDR_E_ARG, "knn: k=%d, d=%d needs %zu B of LDS (> 160 KiB)", k, d, lds);
    dim3 grid((unsigned)p.nqb, (unsigned)p.nseg);
    const char *dbg_env = getenv("FDR_KNN_DEBUG");  // development knob, see DESIGN.md
    const int dbg = dbg_env ? atoi(dbg_env) : 0;
    int trc = timing_begin(ctx, FDR_KERNEL_KNN_TILE, st);
    if (trc) return trc;
#define FDR_LAUNCH_KNN(DP_, NQ_, NW_, WPS_)                                                          \
    do {                                                                                             \
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_tile_kernel<DP_, NQ_, NW_, WPS_>), \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));          \
        hipLaunchKernelGGL((knn_tile_kernel<DP_, NQ_, NW_, WPS_>), grid, dim3(64 * NW_), lds, st, d_Qhat, \
                           d_qzero, (int)nq, d_That, d_bits, (int)nt, (int)t_base, p.segs, k,        \
                           p.nq_pad, d_partial, d_shared, knn_qcap(sh, k), dbg);                     \
    } while (0)
    switch (p.shape) {
        case 0: FDR_LAUNCH_KNN(128, 1, 4, 3); break;
        case 1: FDR_LAUNCH_KNN(128, 1, 8, 4); break;
        case 2: FDR_LAUNCH_KNN(128, 2, 4, 2); break;
        case 3: FDR_LAUNCH_KNN(256, 1, 8, 2); break;
        default: FDR_LAUNCH_KNN(512, 1, 4, 1); break;
    }
#undef FDR_LAUNCH_KNN
    HIP_TRY(hipGetLastError());
#ifdef FDR_DEBUG_COUNTERS
    if (dbg & 2) {
        unsigned long long c[8];
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_dbg_counters), sizeof(c)));
        fprintf(stderr, "[fdr debug] grid %d x %d  calls(with tile) %llu  hot-episodes... update_calls=%llu rounds=%llu flushes=%llu flush_iters=%llu rescans=%llu\n",
                p.nqb, p.nseg, c[0], c[0], c[1], c[2], c[3], c[4]);
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_counters), z, sizeof(z)));
    }
#endif
    if ((trc = timing_end(ctx, FDR_KERNEL_KNN_TILE, st))) return trc;
    if ((trc = timing_begin(ctx, FDR_KERNEL_KNN_MERGE, st))) return trc;
    hipLaunchKernelGGL(knn_merge_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, st,
                       (const u64 *)d_partial, p.nseg, (int)nq, p.nq_pad, k, d_idx, d_dist);
    HIP_TRY(hipGetLastError());
    return timing_end(ctx, FDR_KERNEL_KNN_MERGE, st);
}

// ---- prefilter mode: fp16 pass -> certificate + exact re-rank -> exact pass for the rest -------
static int launch_knn_prefilter(fdr_ctx *ctx, const float *d_Qhat, const uint8_t *d_qzero, int64_t nq,
                                const float *d_That, const uint8_t *d_tzero, int64_t nt, int64_t t_base,
                                int d, int k, int32_t *d_idx, float *d_dist, void *d_ws, size_t ws_bytes,
                                hipStream_t st) {
    const PrefilterLayout L = prefilter_layout(ctx, nq, nt, d, k);
    if (ws_bytes < L.total)
        return fail(FDR_E_ARG, "knn: workspace %zu < required %zu bytes", ws_bytes, L.total);
    char *ws = static_cast<char *>(d_ws);
    _Float16 *d_ht = reinterpret_cast<_Float16 *>(ws + L.off_ht);
    _Float16 *d_hq = reinterpret_cast<_Float16 *>(ws + L.off_hq);
    u64 *d_cand = reinterpret_cast<u64 *>(ws + L.off_cand);
    int *d_counter = reinterpret_cast<int *>(ws + L.off_counter);
    int *d_flagged = reinterpret_cast<int *>(ws + L.off_flagged);
    float *d_qc = reinterpret_cast<float *>(ws + L.off_qc);
    uint8_t *d_qzc = reinterpret_cast<uint8_t *>(ws + L.off_qzc);
    int32_t *d_idxc = reinterpret_cast<int32_t *>(ws + L.off_idxc);
    float *d_distc = reinterpret_cast<float *>(ws + L.off_distc);
    const int kp = L.kp;

    const int dp = fdr_padded_dim(d);
    const KnnPlan p = knn_plan(ctx, nq, nt, d, kp, prefilter_shape(dp));
    const KnnShape &sh = kShapes[prefilter_shape(dp)];
    unsigned *d_bits = reinterpret_cast<unsigned *>(ws);
    unsigned *d_shared = reinterpret_cast<unsigned *>(ws + p.bits_bytes);
    u64 *d_partial = reinterpret_cast<u64 *>(ws + p.bits_bytes + p.shared_bytes);

    int trc = timing_begin(ctx, FDR_KERNEL_KNN_RERANK, st);  // conversion + set-up count as "rerank"
    if (trc) return trc;
    hipLaunchKernelGGL(to_half_kernel, dim3((unsigned)((nt * (dp / 8) + 255) / 256)), dim3(256), 0, st, d_That,
                       (long long)nt * (dp / 8), d_ht);
    hipLaunchKernelGGL(to_half_kernel, dim3((unsigned)((nq * (dp / 8) + 255) / 256)), dim3(256), 0, st, d_Qhat,
                       (long long)nq * (dp / 8), d_hq);
    hipLaunchKernelGGL(pack_zero_bits_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st,
                       d_tzero, (int)nt, d_bits, d_shared, p.nq_pad);
    HIP_TRY(hipGetLastError());
    const size_t lds = knn_lds_bytes(sh, kp);
    int max_seg = 1;
    for (int i = 0; i < p.nseg; ++i) max_seg = std::max(max_seg, p.segs.b[i + 1] - p.segs.b[i]);
    const int ib = prefilter_index_bits(max_seg);
    if (ib > FDR_PREFILTER_MAX_IB) return fail(FDR_E_ARG, "knn prefilter: segment of %d rows", max_seg);
    if ((trc = timing_end(ctx, FDR_KERNEL_KNN_RERANK, st))) return trc;
    if ((trc = timing_begin(ctx, FDR_KERNEL_KNN_PREFILTER, st))) return trc;
    const int pdbg = getenv("FDR_KNN_DEBUG") ? atoi(getenv("FDR_KNN_DEBUG")) : 0;
#define FDR_LAUNCH_PRE2(DP_, NQ_, NW_, WPS_, U_, LH_)                                                   \
    do { /* (the ring is at most 32 KB: no dynamic-LDS attribute needed) */                             \
        hipLaunchKernelGGL((knn_prefilter_kernel<DP_, NQ_, NW_, WPS_, U_, LH_>),                        \
                           dim3((unsigned)p.nqb, (unsigned)p.nseg), dim3(64 * NW_), lds, st, d_hq, (int)nq, \
                           d_ht, (int)nt, (int)t_base, p.segs, kp, p.nq_pad, d_partial, d_shared, ib,   \
                           pdbg);                                                                       \
    } while (0)
#define FDR_LAUNCH_PRE(DP_, NQ_, NW_, WPS_, U_)                                                         \
    do {                                                                                                \
        if (kp <= 32) FDR_LAUNCH_PRE2(DP_, NQ_, NW_, WPS_, U_, 16);                                     \
        else FDR_LAUNCH_PRE2(DP_, NQ_, NW_, WPS_, U_, 32);                                              \
    } while (0)
    if (dp == 128 && kp <= 32 && !(getenv("FDR_KNN_PAIR") && atoi(getenv("FDR_KNN_PAIR")) == 0)) {
        // d <= 128, K' <= 32: the stage's two tiles as two interleaved MFMA chains (1-4 % faster; still
        // <= 128 VGPRs.  FDR_KNN_PAIR=0: development knob, one chain)
        hipLaunchKernelGGL((knn_prefilter_kernel<128, 1, 4, 4, 2, 16, true>), dim3((unsigned)p.nqb, (unsigned)p.nseg),
                           dim3(256), lds, st, d_hq, (int)nq, d_ht, (int)nt, (int)t_base, p.segs, kp, p.nq_pad,
                           d_partial, d_shared, ib, pdbg);
    } else if (dp == 128) FDR_LAUNCH_PRE(128, 1, 4, 4, 2);
    else if (dp == 256) FDR_LAUNCH_PRE(256, 1, 4, 3, 2);
    else FDR_LAUNCH_PRE(512, 1, 4, 2, 2);
#undef FDR_LAUNCH_PRE2
#undef FDR_LAUNCH_PRE
    HIP_TRY(hipGetLastError());
#ifdef FDR_DEBUG_COUNTERS
    if (pdbg & 2) {
        unsigned long long c[8];
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_dbg_counters), sizeof(c)));
        fprintf(stderr, "[fdr debug] prefilter grid %d x %d  wave-tiles %llu  cold %llu  groups %llu  rounds %llu  "
                        "second rounds %llu  candidates %llu\n", p.nqb, p.nseg, c[0], c[1], c[2], c[3], c[4], c[5]);
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_counters), z, sizeof(z)));
    }
#endif
    if ((trc = timing_end(ctx, FDR_KERNEL_KNN_PREFILTER, st))) return trc;

    if ((trc = timing_begin(ctx, FDR_KERNEL_KNN_RERANK, st))) return trc;
    hipLaunchKernelGGL(knn_merge_keys_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, st,
                       (const u64 *)d_partial, p.nseg, (int)nq, p.nq_pad, kp, d_cand);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(d_counter, 0, 16, st));
    int *d_rlist = reinterpret_cast<int *>(ws + L.off_rlist);
    float *d_theta = reinterpret_cast<float *>(ws + L.off_theta);
    const bool use_range = !(getenv("FDR_KNN_RANGE") && atoi(getenv("FDR_KNN_RANGE")) == 0);  // dev knob
    const float margin = 2.0f * prefilter_eps(ib) + 4.0e-7f;
    hipLaunchKernelGGL(knn_rerank_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, st,
                       (const u64 *)d_cand, kp, k, d_Qhat, d_qzero, d_That, (int)nq, dp, (int)t_base, margin,
                       d_idx, d_dist, d_counter, d_flagged, use_range ? d_rlist : (int *)nullptr, d_theta);
    {   // all-zero queries share one closed-form answer (their number is only known on the device yet)
        int *d_zidx = d_counter + 64;
        float *d_zdist = reinterpret_cast<float *>(d_counter + 128);
        hipLaunchKernelGGL(zero_answer_kernel, dim3(1), dim3(1024), 0, st, (const unsigned *)d_bits, (int)nt,
                           (int)t_base, k, d_zidx, d_zdist);
        hipLaunchKernelGGL(scatter_zero_answer_kernel, dim3((unsigned)(((int64_t)nq * k + 255) / 256)),
                           dim3(256), 0, st, (const int *)d_zidx, (const float *)d_zdist,
                           (const int *)d_flagged, (int)nq, (const int *)d_counter, k, d_idx, d_dist);
    }
    HIP_TRY(hipGetLastError());
    if ((trc = timing_end(ctx, FDR_KERNEL_KNN_RERANK, st))) return trc;

    // how many queries could not be certified / are all-zero / need a range pass?  (one 12-byte
    // read-back; the passes below are sized from it)
    int counts[3] = {0, 0, 0};
    HIP_TRY(hipMemcpyAsync(counts, d_counter, 12, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    int count = counts[0];
    const int zcount = counts[1], rcount = counts[2];
    ctx->last_flagged = count + rcount;
    if (rcount > 0) {  // plateau queries: collect {d~ <= theta} with a second fp16 pass, rank it exactly
        if ((trc = timing_begin(ctx, FDR_KERNEL_KNN_RERANK, st))) return trc;
        _Float16 *d_hqc = reinterpret_cast<_Float16 *>(ws + L.off_hqc);
        float *d_thetac = reinterpret_cast<float *>(ws + L.off_thetac);
        int *d_cnt = reinterpret_cast<int *>(ws + L.off_cnt);
        int *d_rcand = reinterpret_cast<int *>(ws + L.off_rcand);
        for (int first = 0; first < rcount; first += L.rchunk) {
            const int c = std::min(L.rchunk, rcount - first);
            hipLaunchKernelGGL(gather_half_queries_kernel, dim3((unsigned)c), dim3(256), 0, st,
                               (const _Float16 *)d_hq, (const float *)d_theta, (const int *)d_rlist, first, c,
                               dp, d_hqc, d_thetac, d_cnt);
            HIP_TRY(hipGetLastError());
            const KnnPlan rp = knn_plan(ctx, c, nt, d, 1, range_shape(dp));  // (k = 1: ring-only LDS)
            const size_t rlds = (size_t)2 * 32 * 256;
#define FDR_LAUNCH_RANGE(DP_, WPS_)                                                                     \
    hipLaunchKernelGGL((knn_range_kernel<DP_, 4, WPS_>), dim3((unsigned)rp.nqb, (unsigned)rp.nseg),        \
                       dim3(256), rlds, st, (const _Float16 *)d_hqc, (const float *)d_thetac, c,          \
                       (const _Float16 *)d_ht, (int)nt, (int)t_base, rp.segs, d_cnt, d_rcand)
            if (dp == 128) FDR_LAUNCH_RANGE(128, 4);
            else if (dp == 256) FDR_LAUNCH_RANGE(256, 2);
            else FDR_LAUNCH_RANGE(512, 2);
#undef FDR_LAUNCH_RANGE
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(knn_rerank_long_kernel, dim3((unsigned)((c + 3) / 4)), dim3(256), 0, st,
                               (const int *)(d_rlist + first), c, (const int *)d_cnt, (const int *)d_rcand, k,
                               d_Qhat, d_That, dp, (int)t_base, d_idx, d_dist, d_counter, d_flagged);
            HIP_TRY(hipGetLastError());
        }
        if ((trc = timing_end(ctx, FDR_KERNEL_KNN_RERANK, st))) return trc;
        // ranges that overflowed were appended to the exact list: read its final length
        HIP_TRY(hipMemcpyAsync(counts, d_counter, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        count = counts[0];
    }
    if (count <= 0) return FDR_OK;
    if ((int64_t)count * 2 > nq - zcount)  // the prefilter did not help on this input: exact pass for everyone
        return launch_knn_exact(ctx, d_Qhat, d_qzero, nq, d_That, d_tzero, nt, t_base, d, k, d_idx, d_dist,
                                d_ws, L.knn_bytes, st);
    for (int first = 0; first < count; first += L.chunk) {
        const int c = std::min(L.chunk, count - first);
        hipLaunchKernelGGL(gather_queries_kernel, dim3((unsigned)c), dim3(256), 0, st, d_Qhat, d_qzero,
                           (const int *)d_flagged, first, c, dp, d_qc, d_qzc);
        HIP_TRY(hipGetLastError());
        int rc = launch_knn_exact(ctx, d_qc, d_qzc, c, d_That, d_tzero, nt, t_base, d, k, d_idxc, d_distc,
                                  d_ws, L.knn_bytes, st);
        if (rc) return rc;
        hipLaunchKernelGGL(scatter_results_kernel, dim3((unsigned)(((int64_t)c * k + 255) / 256)), dim3(256),
                           0, st, (const int *)d_idxc, (const float *)d_distc, (const int *)d_flagged, first,
                           c, k, d_idx, d_dist);
        HIP_TRY(hipGetLastError());
    }
    return FDR_OK;
}

static size_t knn_mode_workspace_bytes(const fdr_ctx *ctx, int64_t nq, int64_t nt, int d, int k) {
    if (knn_prefilter_wanted(ctx, fdr_padded_dim(d), nt, k)) return prefilter_layout(ctx, nq, nt, d, k).total;
    return knn_plan(ctx, nq, nt, d, k).total_bytes;
}

static int launch_knn_mode(fdr_ctx *ctx, const float *d_Qhat, const uint8_t *d_qzero, int64_t nq,
                           const float *d_That, const uint8_t *d_tzero, int64_t nt, int64_t t_base, int d,
                           int k, int32_t *d_idx, float *d_dist, void *d_ws, size_t ws_bytes, hipStream_t st) {
    const int dp = fdr_padded_dim(d);
    if (dp > 0 && k >= 1 && k <= FDR_MAX_K && nq > 0 && nt >= k && knn_prefilter_wanted(ctx, dp, nt, k) &&
        d_Qhat && d_qzero && d_That && d_tzero && d_idx && d_dist && d_ws)
        return launch_knn_prefilter(ctx, d_Qhat, d_qzero, nq, d_That, d_tzero, nt, t_base, d, k, d_idx,
                                    d_dist, d_ws, ws_bytes, st);
    ctx->last_flagged = 0;  // (exact mode certifies nothing)
    return launch_knn_exact(ctx, d_Qhat, d_qzero, nq, d_That, d_tzero, nt, t_base, d, k, d_idx, d_dist,
                            d_ws, ws_bytes, st);
}

// ---- duplicate-row classes: search unique queries x unique targets, expand --------------------
struct DedupLayout {
    size_t inner_bytes;  // workspace of the inner k-NN call (sized for the un-deduplicated problem)
    size_t off_hash, off_hash_s, off_idx, off_idx_s, off_flag, off_cid, off_cls, off_cstart, off_isrep,
        off_upos, off_uofc, off_cofu, off_uqflag, off_uqpos, off_U, off_uzero, off_Uq, off_uqz, off_idxu,
        off_distu, off_tmp, tmp_bytes, total;
};

static bool knn_dedup_wanted(int64_t nq, int64_t nt) {
    if (const char *e = getenv("FDR_KNN_DEDUP")) return atoi(e) != 0;  // development knob
    return nt >= 16384 && nq >= 1024;
}

static DedupLayout dedup_layout(const fdr_ctx *ctx, int64_t nq, int64_t nt, int d, int k) {
    DedupLayout L;
    const int dp = fdr_padded_dim(d);
    L.inner_bytes = align256(knn_mode_workspace_bytes(ctx, nq, nt, d, k));
    size_t o = L.inner_bytes;
    auto take = [&](size_t bytes) { const size_t at = o; o += align256(bytes); return at; };
    L.off_hash = take((size_t)nt * 8);
    L.off_hash_s = take((size_t)nt * 8);
    L.off_idx = take((size_t)nt * 4);
    L.off_idx_s = take((size_t)nt * 4);
    L.off_flag = take((size_t)nt * 4);
    L.off_cid = take((size_t)nt * 4);
    L.off_cls = take((size_t)nt * 4);
    L.off_cstart = take((size_t)(nt + 1) * 4);
    L.off_isrep = take((size_t)nt * 4);
    L.off_upos = take((size_t)nt * 4);
    L.off_uofc = take((size_t)nt * 4);
    L.off_cofu = take((size_t)nt * 4);
    L.off_uqflag = take((size_t)nt * 4);
    L.off_uqpos = take((size_t)nt * 4);
    L.off_U = take((size_t)nt * dp * 4);
    L.off_uzero = take((size_t)nt);
    L.off_Uq = take((size_t)nq * dp * 4);
    L.off_uqz = take((size_t)nq);
    L.off_idxu = take((size_t)nq * k * 4);
    L.off_distu = take((size_t)nq * k * 4);
    size_t t_sort = 0, t_scan = 0;
    (void)rocprim::radix_sort_pairs(nullptr, t_sort, (u64 *)nullptr, (u64 *)nullptr, (int *)nullptr,
                                    (int *)nullptr, (size_t)nt, 0, 64, (hipStream_t) nullptr);
    (void)rocprim::inclusive_scan(nullptr, t_scan, (int *)nullptr, (int *)nullptr, (size_t)nt,
                                  rocprim::plus<int>(), (hipStream_t) nullptr);
    L.tmp_bytes = align256(std::max(t_sort, t_scan));
    L.off_tmp = take(L.tmp_bytes);
    L.total = o;
    return L;
}

static int launch_knn(fdr_ctx *ctx, const float *d_Qhat, const uint8_t *d_qzero, int64_t nq,
                      const float *d_That, const uint8_t *d_tzero, int64_t nt, int64_t t_base, int d,
                      int k, int32_t *d_idx, float *d_dist, void *d_ws, size_t ws_bytes, hipStream_t st) {
    const int dp = fdr_padded_dim(d);
    // the queries must be a block of the target rows (they are in every caller of this library)
    const bool q_in_t = d_Qhat && d_That && dp > 0 && d_Qhat >= d_That &&
                        d_Qhat + (size_t)nq * dp <= d_That + (size_t)nt * dp &&
                        ((d_Qhat - d_That) % dp) == 0;
    if (!(dp > 0 && k >= 1 && k <= FDR_MAX_K && nq > 0 && nt >= k && q_in_t && knn_dedup_wanted(nq, nt) &&
          d_qzero && d_tzero && d_idx && d_dist && d_ws)) {
        ctx->last_unique_targets = (int)nt;
        ctx->last_unique_queries = (int)nq;
        return launch_knn_mode(ctx, d_Qhat, d_qzero, nq, d_That, d_tzero, nt, t_base, d, k, d_idx, d_dist,
                               d_ws, ws_bytes, st);
    }
    const DedupLayout L = dedup_layout(ctx, nq, nt, d, k);
    if (ws_bytes < L.total) return fail(FDR_E_ARG, "knn: workspace %zu < required %zu bytes", ws_bytes, L.total);
    char *ws = static_cast<char *>(d_ws);
    u64 *hash = (u64 *)(ws + L.off_hash), *hash_s = (u64 *)(ws + L.off_hash_s);
    int *idx = (int *)(ws + L.off_idx), *idx_s = (int *)(ws + L.off_idx_s), *flag = (int *)(ws + L.off_flag);
    int *cid = (int *)(ws + L.off_cid), *cls = (int *)(ws + L.off_cls), *cstart = (int *)(ws + L.off_cstart);
    int *isrep = (int *)(ws + L.off_isrep), *upos = (int *)(ws + L.off_upos), *uofc = (int *)(ws + L.off_uofc);
    int *cofu = (int *)(ws + L.off_cofu), *uqflag = (int *)(ws + L.off_uqflag), *uqpos = (int *)(ws + L.off_uqpos);
    float *U = (float *)(ws + L.off_U), *Uq = (float *)(ws + L.off_Uq);
    uint8_t *uzero = (uint8_t *)(ws + L.off_uzero), *uqz = (uint8_t *)(ws + L.off_uqz);
    int32_t *idx_u = (int32_t *)(ws + L.off_idxu);
    float *dist_u = (float *)(ws + L.off_distu);
    void *tmp = ws + L.off_tmp;
    const int n = (int)nt;
    const int q0 = (int)((d_Qhat - d_That) / dp);
    const unsigned g16 = (unsigned)(((size_t)n * 16 + 255) / 256), g1 = (unsigned)((n + 255) / 256);

    int trc = timing_begin(ctx, FDR_KERNEL_KNN_DEDUP, st);
    if (trc) return trc;
    hipLaunchKernelGGL(hash_rows_kernel, dim3(g16), dim3(256), 0, st, d_That, n, dp, hash, idx);
    HIP_TRY(hipGetLastError());
    const char *knob = getenv("FDR_KNN_DEDUP");
    const bool always = knob && atoi(knob) == 2;  // development knob: expand even without duplicates
    if (!always) {
        // a hash-table probe (~15 us) tells whether enough rows repeat to pay for the sort and the tables;
        // the table borrows the (still unused) unique-row buffer
        unsigned tsize = 1024;
        while (tsize < 2u * (unsigned)n && tsize < (1u << 30)) tsize <<= 1;
        if ((size_t)tsize * 8 + 256 <= (size_t)nt * dp * 4) {
            u64 *table = reinterpret_cast<u64 *>(U);
            int *d_cnt = reinterpret_cast<int *>(table + tsize);
            HIP_TRY(hipMemsetAsync(table, 0, (size_t)tsize * 8 + 4, st));
            hipLaunchKernelGGL(dedup_probe_kernel, dim3(g1), dim3(256), 0, st, (const u64 *)hash, n, table,
                               tsize - 1, d_cnt);
            HIP_TRY(hipGetLastError());
            int dups = 0;
            HIP_TRY(hipMemcpyAsync(&dups, d_cnt, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if ((double)dups < 0.05 * (double)n) {  // (unique share)^2 > 0.9: not worth it
                if ((trc = timing_end(ctx, FDR_KERNEL_KNN_DEDUP, st))) return trc;
                ctx->last_unique_targets = (int)nt;
                ctx->last_unique_queries = (int)nq;
                return launch_knn_mode(ctx, d_Qhat, d_qzero, nq, d_That, d_tzero, nt, t_base, d, k, d_idx,
                                       d_dist, d_ws, L.inner_bytes, st);
            }
        }
    }
    size_t tb = L.tmp_bytes;
    HIP_TRY(rocprim::radix_sort_pairs(tmp, tb, hash, hash_s, idx, idx_s, (size_t)n, 0, 64, st));
    hipLaunchKernelGGL(mark_class_starts_kernel, dim3(g1), dim3(256), 0, st, d_That, n, dp,
                       (const u64 *)hash_s, (const int *)idx_s, flag);
    tb = L.tmp_bytes;
    HIP_TRY(rocprim::inclusive_scan(tmp, tb, flag, cid, (size_t)n, rocprim::plus<int>(), st));
    hipLaunchKernelGGL(class_tables_kernel, dim3(g1), dim3(256), 0, st, n, (const int *)flag,
                       (const int *)cid, (const int *)idx_s, cls, cstart, isrep);
    tb = L.tmp_bytes;
    HIP_TRY(rocprim::inclusive_scan(tmp, tb, isrep, upos, (size_t)n, rocprim::plus<int>(), st));
    hipLaunchKernelGGL(unique_tables_kernel, dim3(g1), dim3(256), 0, st, n, (const int *)isrep,
                       (const int *)upos, (const int *)cls, uofc, cofu);
    HIP_TRY(hipMemsetAsync(uqflag, 0, (size_t)n * 4, st));
    hipLaunchKernelGGL(mark_query_classes_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, q0,
                       (int)nq, (const int *)cls, (const int *)uofc, uqflag);
    tb = L.tmp_bytes;
    HIP_TRY(rocprim::inclusive_scan(tmp, tb, uqflag, uqpos, (size_t)n, rocprim::plus<int>(), st));
    HIP_TRY(hipGetLastError());
    int nu = 0, nuq = 0;
    HIP_TRY(hipMemcpyAsync(&nu, cid + (n - 1), 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&nuq, uqpos + (n - 1), 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    ctx->last_unique_targets = nu;
    ctx->last_unique_queries = nuq;
    const bool worth = nu >= k && (always || (double)nu * nuq <= 0.9 * (double)nt * (double)nq);
    size_t inner_need = worth ? knn_mode_workspace_bytes(ctx, nuq, nu, d, k) : 0;
    if (!worth || inner_need > L.inner_bytes) {  // few duplicates (or, never seen, no room): plain search
        if ((trc = timing_end(ctx, FDR_KERNEL_KNN_DEDUP, st))) return trc;
        ctx->last_unique_targets = (int)nt;
        ctx->last_unique_queries = (int)nq;
        return launch_knn_mode(ctx, d_Qhat, d_qzero, nq, d_That, d_tzero, nt, t_base, d, k, d_idx, d_dist,
                               d_ws, L.inner_bytes, st);
    }
    hipLaunchKernelGGL(gather_unique_rows_kernel, dim3(g16), dim3(256), 0, st, d_That, d_tzero, n, dp,
                       (const int *)isrep, (const int *)upos, U, uzero);
    hipLaunchKernelGGL(gather_unique_queries_kernel, dim3((unsigned)(((size_t)nu * 16 + 255) / 256)), dim3(256),
                       0, st, (const float *)U, (const unsigned char *)uzero, nu, dp, (const int *)uqflag,
                       (const int *)uqpos, Uq, uqz);
    HIP_TRY(hipGetLastError());
    if ((trc = timing_end(ctx, FDR_KERNEL_KNN_DEDUP, st))) return trc;
    // unique rows are stored in ascending representative order, and the unique queries are a subsequence
    // of them; the inner search numbers targets 0..nu-1
    int rc = launch_knn_mode(ctx, Uq, uqz, nuq, U, uzero, nu, 0, d, k, idx_u, dist_u, d_ws, L.inner_bytes, st);
    if (rc) return rc;
    if ((trc = timing_begin(ctx, FDR_KERNEL_KNN_DEDUP, st))) return trc;
    hipLaunchKernelGGL(expand_classes_kernel, dim3((unsigned)nq), dim3(64), (size_t)k * k * 8, st, q0, (int)nq, k, (int)t_base,
                       (const int *)cls, (const int *)uofc, (const int *)uqpos, (const int *)idx_u,
                       (const float *)dist_u, (const int *)cofu, (const int *)cstart, (const int *)idx_s, d_idx,
                       d_dist);
    HIP_TRY(hipGetLastError());
    return timing_end(ctx, FDR_KERNEL_KNN_DEDUP, st);
}

static size_t knn_workspace_bytes_impl(const fdr_ctx *ctx, int64_t nq, int64_t nt, int d, int k) {
    if (knn_dedup_wanted(nq, nt)) return dedup_layout(ctx, nq, nt, d, k).total;
    return knn_mode_workspace_bytes(ctx, nq, nt, d, k);
}

// ---- device-pointer API ----------------------------------------------------------------------
FDR_EXPORT int fdr_embed_dev(fdr_ctx *ctx, int64_t n_rows, const int64_t *d_indptr,
                             const int32_t *d_indices, float *d_E, void *stream) {
    int rc = use_device(ctx);
    if (rc) return rc;
    if (n_rows > 0 && (!d_indptr || !d_E)) return fail(FDR_E_ARG, "embed: null device pointer");
    return launch_embed(ctx, n_rows, d_indptr, d_indices, d_E, (hipStream_t)stream);
}

FDR_EXPORT int fdr_normalize_dev(fdr_ctx *ctx, const float *d_E, int64_t n_rows, int32_t d,
                                 float *d_Ehat, uint8_t *d_zero, void *stream) {
    int rc = use_device(ctx);
    if (rc) return rc;
    if (n_rows > 0 && (!d_E || !d_Ehat || !d_zero)) return fail(FDR_E_ARG, "normalize: null device pointer");
    return launch_normalize(ctx, d_E, n_rows, d, d_Ehat, d_zero, (hipStream_t)stream);
}

FDR_EXPORT int fdr_knn_dev(fdr_ctx *ctx, const float *d_Qhat, const uint8_t *d_qzero, int64_t nq,
                           const float *d_That, const uint8_t *d_tzero, int64_t nt, int64_t t_base,
                           int32_t d, int32_t k, int32_t *d_idx, float *d_dist, void *d_workspace,
                           size_t workspace_bytes, void *stream) {
    int rc = use_device(ctx);
    if (rc) return rc;
    return launch_knn(ctx, d_Qhat, d_qzero, nq, d_That, d_tzero, nt, t_base, d, k, d_idx, d_dist,
                      d_workspace, workspace_bytes, (hipStream_t)stream);
}

// ---- host-pointer API ------------------------------------------------------------------------
static int check_csr(int64_t n_rows, const int64_t *a_indptr, const int32_t *a_indices) {
    if (n_rows < 0 || !a_indptr) return fail(FDR_E_ARG, "embed: bad CSR (n_rows=%lld)", (long long)n_rows);
    if (a_indptr[0] != 0 || a_indptr[n_rows] < 0) return fail(FDR_E_ARG, "embed: bad indptr");
    if (a_indptr[n_rows] > 0 && !a_indices) return fail(FDR_E_ARG, "embed: indices is null");
    return FDR_OK;
}

static int upload_csr(fdr_ctx *ctx, int64_t n_rows, const int64_t *a_indptr,
                      const int32_t *a_indices) {
    int rc;
    const int64_t nnz = a_indptr[n_rows];
    if ((rc = ctx->a_indptr.reserve((size_t)(n_rows + 1) * 8))) return rc;
    if ((rc = ctx->a_indices.reserve((size_t)std::max<int64_t>(nnz, 1) * 4))) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->a_indptr.p, a_indptr, (size_t)(n_rows + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    if (nnz > 0)
        HIP_TRY(hipMemcpyAsync(ctx->a_indices.p, a_indices, (size_t)nnz * 4, hipMemcpyHostToDevice, ctx->stream));
    return FDR_OK;
}

FDR_EXPORT int fdr_embed(fdr_ctx *ctx, int64_t n_rows, const int64_t *a_indptr,
                         const int32_t *a_indices, float *E_out) {
    int rc = use_device(ctx);
    if (rc) return rc;
    if ((rc = check_csr(n_rows, a_indptr, a_indices))) return rc;
    if (ctx->n_features <= 0) return fail(FDR_E_STATE, "embed: no projection loaded");
    if (n_rows == 0) return FDR_OK;
    if (!E_out) return fail(FDR_E_ARG, "embed: E_out is null");
    if ((rc = upload_csr(ctx, n_rows, a_indptr, a_indices))) return rc;
    const size_t ebytes = (size_t)n_rows * ctx->d * 4;
    if ((rc = ctx->E.reserve(ebytes))) return rc;
    if ((rc = launch_embed(ctx, n_rows, (const int64_t *)ctx->a_indptr.p,
                           (const int32_t *)ctx->a_indices.p, (float *)ctx->E.p, ctx->stream)))
        return rc;
    HIP_TRY(hipMemcpyAsync(E_out, ctx->E.p, ebytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FDR_OK;
}

// E (device, [n,d]) -> idx/dist on the host
static int knn_from_device_E(fdr_ctx *ctx, const float *d_E, int64_t n, int d, int k,
                             int32_t *idx_out, float *dist_out) {
    int rc;
    const int dp = fdr_padded_dim(d);
    if (dp < 0) return fail(FDR_E_ARG, "knn: dimension %d unsupported (1..%d)", d, FDR_MAX_DIM);
    if (k < 1 || k > FDR_MAX_K) return fail(FDR_E_ARG, "knn: k=%d unsupported (1..%d)", k, FDR_MAX_K);
    if (n < k) return fail(FDR_E_ARG, "knn: need n (%lld) >= k (%d)", (long long)n, k);
    if (!idx_out || !dist_out) return fail(FDR_E_ARG, "knn: null output pointer");
    if ((rc = ctx->Ehat.reserve((size_t)n * dp * 4))) return rc;
    if ((rc = ctx->zero.reserve((size_t)n))) return rc;
    if ((rc = ctx->idx.reserve((size_t)n * k * 4))) return rc;
    if ((rc = ctx->dist.reserve((size_t)n * k * 4))) return rc;
    const size_t wsb = fdr_knn_workspace_bytes(ctx, n, n, d, k);
    if ((rc = ctx->ws.reserve(wsb))) return rc;
    if ((rc = launch_normalize(ctx, d_E, n, d, (float *)ctx->Ehat.p, (uint8_t *)ctx->zero.p, ctx->stream)))
        return rc;
    if ((rc = launch_knn(ctx, (const float *)ctx->Ehat.p, (const uint8_t *)ctx->zero.p, n,
                         (const float *)ctx->Ehat.p, (const uint8_t *)ctx->zero.p, n, 0, d, k,
                         (int32_t *)ctx->idx.p, (float *)ctx->dist.p, ctx->ws.p, wsb, ctx->stream)))
        return rc;
    HIP_TRY(hipMemcpyAsync(idx_out, ctx->idx.p, (size_t)n * k * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dist_out, ctx->dist.p, (size_t)n * k * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FDR_OK;
}

FDR_EXPORT int fdr_knn(fdr_ctx *ctx, const float *E, int64_t n, int32_t d, int32_t k,
                       int32_t *idx_out, float *dist_out) {
    int rc = use_device(ctx);
    if (rc) return rc;
    if (!E || n <= 0) return fail(FDR_E_ARG, "knn: empty input");
    if (fdr_padded_dim(d) < 0) return fail(FDR_E_ARG, "knn: dimension %d unsupported (1..%d)", d, FDR_MAX_DIM);
    if ((rc = ctx->E.reserve((size_t)n * d * 4))) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->E.p, E, (size_t)n * d * 4, hipMemcpyHostToDevice, ctx->stream));
    return knn_from_device_E(ctx, (const float *)ctx->E.p, n, d, k, idx_out, dist_out);
}

FDR_EXPORT int fdr_embed_knn(fdr_ctx *ctx, int64_t n_rows, const int64_t *a_indptr,
                             const int32_t *a_indices, int32_t k, int32_t *idx_out, float *dist_out,
                             float *E_out) {
    int rc = use_device(ctx);
    if (rc) return rc;
    if ((rc = check_csr(n_rows, a_indptr, a_indices))) return rc;
    if (ctx->n_features <= 0) return fail(FDR_E_STATE, "embed: no projection loaded");
    if (n_rows <= 0) return fail(FDR_E_ARG, "embed_knn: empty input");
    if ((rc = upload_csr(ctx, n_rows, a_indptr, a_indices))) return rc;
    const size_t ebytes = (size_t)n_rows * ctx->d * 4;
    if ((rc = ctx->E.reserve(ebytes))) return rc;
    if ((rc = launch_embed(ctx, n_rows, (const int64_t *)ctx->a_indptr.p,
                           (const int32_t *)ctx->a_indices.p, (float *)ctx->E.p, ctx->stream)))
        return rc;
    if (E_out) HIP_TRY(hipMemcpyAsync(E_out, ctx->E.p, ebytes, hipMemcpyDeviceToHost, ctx->stream));
    return knn_from_device_E(ctx, (const float *)ctx->E.p, n_rows, ctx->d, k, idx_out, dist_out);
}

#include "kmer_search.inc"
#include "kmer_output_loader.inc"
