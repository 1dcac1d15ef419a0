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
#include <utility>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_merge.hpp>
#include <rocprim/device/device_reduce_by_key.hpp>

#include <algorithm>
#include <cstdarg>
#include <limits>
#include <string>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/fedrann_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#include "host_common.inc"  // error plumbing, FDR_EXPORT, the development knobs (plain C++: shared with the sanitizer build)

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
// A wave streams a row's column ids in coalesced 64-id chunks, tests the bitmap words (L2 resident) and
// fetches rowinfo for the hits; several rows' (short rows: eight, one chunk each) or chunks' (long rows: two
// rows, four chunks each) dependent id -> bitmap -> rowinfo chains are in flight at once.  Hits are then
// applied in ascending feature order by the whole wave (lane l owns columns l, l+64, ...), which reproduces
// scipy's sequential fp32 sums bit for bit.
// ------------------------------------------------------------------------------------------
template <int DP>
__global__ __launch_bounds__(256) void embed_csr_kernel(
    long long n_rows, const long long *__restrict__ a_indptr, const int *__restrict__ a_indices,
    long long n_features, const uint2 *__restrict__ ftab, const uint4 *__restrict__ rowinfo,
    const uint2 *__restrict__ ent, int d, float *__restrict__ E) {
    constexpr int NACC = DP / 64;
    static_assert(DP <= 512, "d > 512: embed_csr_wide_kernel");
    constexpr int GR = 8;   // rows a wave takes per turn
    constexpr int NB = 4;   // long rows: 64-id chunks in flight per row
    constexpr int RPW = 2;  // long rows: rows in flight
    // The id -> bitmap -> rowinfo chain is three dependent loads (~2 us a row) and the rows in flight are all that hides
    // it (stubbed lookups: the kernel's time does not depend on where rowinfo comes from, and the ordered application below
    // is 18 % of it).  SHORT rows -- every row of a turn within 64 ids: a compacted CSR's ~17 live ids -- run eight chains
    // side by side with one chunk each; LONG rows (a raw CSR's ~170 ids) two chains with four chunks each.
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
    // hits in ascending lane = ascending feature order, applied by the whole wave (lane l owns columns l, l + 64, ...)
    auto apply = [&](float (&acc)[NACC], const uint4 &inf, const bool h) __attribute__((always_inline)) {
        u64 m = __ballot(h);
        while (m) {  // wave-uniform
            const int src = __builtin_ctzll(m);
            m &= m - 1;
            int q = __builtin_amdgcn_readlane((int)inf.x, src);
            const int ee = q + __builtin_amdgcn_readlane((int)inf.y, src);
            unsigned c = (unsigned)__builtin_amdgcn_readlane((int)inf.z, src);
            float v = __int_as_float(__builtin_amdgcn_readlane((int)inf.w, src));
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
    };
    auto store_row = [&](const long long row, const float (&acc)[NACC]) __attribute__((always_inline)) {
        float *out = E + row * (long long)d;
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            if (lane + 64 * i < d) out[lane + 64 * i] = acc[i];
    };
    for (long long row0 = wave0 * GR; row0 < n_rows; row0 += nwaves * GR) {
        long long beg[GR], len_max = 0;
        int len[GR];
#pragma unroll
        for (int r = 0; r < GR; ++r) {
            const long long row = row0 + r < n_rows ? row0 + r : n_rows - 1;
            beg[r] = a_indptr[row];
            const long long l = row0 + r < n_rows ? a_indptr[row + 1] - beg[r] : 0;
            len[r] = (int)(l < 64 ? l : 64);
            len_max = l > len_max ? l : len_max;
        }
        if (len_max <= 64) {  // (wave-uniform)
            int f[GR];
            uint2 w[GR];
            uint4 info[GR];
            bool hit[GR];
#pragma unroll
            for (int r = 0; r < GR; ++r) f[r] = lane < len[r] ? a_indices[beg[r] + lane] : -1;
#pragma unroll
            for (int r = 0; r < GR; ++r) {
                w[r] = make_uint2(0u, 0u);
                if (f[r] >= 0 && (long long)f[r] < n_features) w[r] = ftab[f[r] >> 5];
            }
#pragma unroll
            for (int r = 0; r < GR; ++r) {
                const unsigned bit = 1u << (f[r] & 31);
                hit[r] = (w[r].x & bit) != 0u;
                info[r] = make_uint4(0u, 0u, 0u, 0u);
                if (hit[r]) info[r] = rowinfo[w[r].y + __popc(w[r].x & (bit - 1u))];
            }
#pragma unroll
            for (int r = 0; r < GR; ++r) {
                if (row0 + r >= n_rows) continue;  // (wave-uniform)
                float acc[NACC];
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = 0.0f;
                apply(acc, info[r], hit[r]);
                store_row(row0 + r, acc);
            }
            continue;
        }
        for (int p0 = 0; p0 < GR && row0 + p0 < n_rows; p0 += RPW) {  // long rows, pair by pair
            const long long prow0 = row0 + p0;
            long long pbeg[RPW], pend[RPW];
            int f[RPW][NB];
            uint2 w[RPW][NB];
            uint4 info[RPW][NB];
            bool hit[RPW][NB];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const long long row = prow0 + r < n_rows ? prow0 + r : n_rows - 1;
                pbeg[r] = a_indptr[row];
                pend[r] = prow0 + r < n_rows ? a_indptr[row + 1] : pbeg[r];
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const long long pos = pbeg[r] + 64 * u + lane;
                    f[r][u] = pos < pend[r] ? a_indices[pos] : -1;
                }
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    w[r][u] = make_uint2(0u, 0u);
                    if (f[r][u] >= 0 && (long long)f[r][u] < n_features) w[r][u] = ftab[f[r][u] >> 5];
                }
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const unsigned bit = 1u << (f[r][u] & 31);
                    hit[r][u] = (w[r][u].x & bit) != 0u;
                    info[r][u] = make_uint4(0u, 0u, 0u, 0u);
                    if (hit[r][u]) info[r][u] = rowinfo[w[r][u].y + __popc(w[r][u].x & (bit - 1u))];
                }
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                if (prow0 + r >= n_rows) continue;  // (wave-uniform)
                float acc[NACC];
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = 0.0f;
#pragma unroll
                for (int u = 0; u < NB; ++u) apply(acc, info[r][u], hit[r][u]);
                // rows longer than the 256 ids in flight (the full CSR of a long read): the rest, chunk by chunk
                for (long long base = pbeg[r] + 64 * NB; base < pend[r]; base += 64 * NB) {
                    int f2[NB];
                    uint2 w2[NB];
#pragma unroll
                    for (int u = 0; u < NB; ++u) {
                        const long long pos = base + 64 * u + lane;
                        f2[u] = pos < pend[r] ? a_indices[pos] : -1;
                    }
#pragma unroll
                    for (int u = 0; u < NB; ++u) {
                        w2[u] = make_uint2(0u, 0u);
                        if (f2[u] >= 0 && (long long)f2[u] < n_features) w2[u] = ftab[f2[u] >> 5];
                    }
#pragma unroll
                    for (int u = 0; u < NB; ++u) {
                        const unsigned bit = 1u << (f2[u] & 31);
                        const bool h2 = (w2[u].x & bit) != 0u;
                        uint4 i2 = make_uint4(0u, 0u, 0u, 0u);
                        if (h2) i2 = rowinfo[w2[u].y + __popc(w2[u].x & (bit - 1u))];
                        apply(acc, i2, h2);
                    }
                }
                store_row(prow0 + r, acc);
            }
        }
    }
}

// d > 512 (16+ accumulators per lane and row): two rows per wave and turn, four chunks each -- round 3's kernel; with the
// eight-row turns above hipcc no longer unrolls its loops at DP = 2048 (a stack array, 144 B of scratch).
template <int DP>
__global__ __launch_bounds__(256) void embed_csr_wide_kernel(
    long long n_rows, const long long *__restrict__ a_indptr, const int *__restrict__ a_indices,
    long long n_features, const uint2 *__restrict__ ftab, const uint4 *__restrict__ rowinfo,
    const uint2 *__restrict__ ent, int d, float *__restrict__ E) {
    constexpr int NACC = DP / 64;
    constexpr int NB = 4;   // 64-id chunks in flight per wave and row
    constexpr int RPW = 2;  // rows a wave has in flight: the id -> bitmap -> rowinfo chain is three dependent loads (~2 us a
                            // row), and the occupancy that hides it is all there is -- two rows' chains side by side
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
    for (long long row0 = wave0 * RPW; row0 < n_rows; row0 += nwaves * RPW) {
        long long beg[RPW], end[RPW];
        int f[RPW][NB];
        uint2 w[RPW][NB];
        uint4 info[RPW][NB];
        bool hit[RPW][NB];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const long long row = row0 + r < n_rows ? row0 + r : n_rows - 1;
            beg[r] = a_indptr[row];
            end[r] = row0 + r < n_rows ? a_indptr[row + 1] : beg[r];
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const long long pos = beg[r] + 64 * u + lane;
                f[r][u] = pos < end[r] ? a_indices[pos] : -1;
            }
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                w[r][u] = make_uint2(0u, 0u);
                if (f[r][u] >= 0 && (long long)f[r][u] < n_features) w[r][u] = ftab[f[r][u] >> 5];
            }
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const unsigned bit = 1u << (f[r][u] & 31);
                hit[r][u] = (w[r][u].x & bit) != 0u;
                info[r][u] = make_uint4(0u, 0u, 0u, 0u);
                if (hit[r][u]) info[r][u] = rowinfo[w[r][u].y + __popc(w[r][u].x & (bit - 1u))];
            }
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            if (row0 + r >= n_rows) break;  // (wave-uniform)
            float acc[NACC];
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = 0.0f;
            // hits in ascending lane = ascending feature order, applied by the whole wave (lane l owns columns l, l + 64, ...)
            auto apply = [&](const uint4 &inf, const bool h) {
                u64 m = __ballot(h);
                while (m) {  // wave-uniform
                    const int src = __builtin_ctzll(m);
                    m &= m - 1;
                    int q = __builtin_amdgcn_readlane((int)inf.x, src);
                    const int ee = q + __builtin_amdgcn_readlane((int)inf.y, src);
                    unsigned c = (unsigned)__builtin_amdgcn_readlane((int)inf.z, src);
                    float v = __int_as_float(__builtin_amdgcn_readlane((int)inf.w, src));
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
            };
#pragma unroll
            for (int u = 0; u < NB; ++u) apply(info[r][u], hit[r][u]);
            // rows longer than the 256 ids in flight (the full CSR of a long read): the rest, chunk by chunk
            for (long long base = beg[r] + 64 * NB; base < end[r]; base += 64 * NB) {
                int f2[NB];
                uint2 w2[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const long long pos = base + 64 * u + lane;
                    f2[u] = pos < end[r] ? a_indices[pos] : -1;
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    w2[u] = make_uint2(0u, 0u);
                    if (f2[u] >= 0 && (long long)f2[u] < n_features) w2[u] = ftab[f2[u] >> 5];
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const unsigned bit = 1u << (f2[u] & 31);
                    const bool h2 = (w2[u].x & bit) != 0u;
                    uint4 i2 = make_uint4(0u, 0u, 0u, 0u);
                    if (h2) i2 = rowinfo[w2[u].y + __popc(w2[u].x & (bit - 1u))];
                    apply(i2, h2);
                }
            }
            float *out = E + (row0 + r) * (long long)d;
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                if (lane + 64 * i < d) out[lane + 64 * i] = acc[i];
        }
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
template <int DP, int RB, bool VEC>
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float *__restrict__ E,
                                                             long long n_rows, int d,
                                                             float *__restrict__ Ehat,
                                                             unsigned char *__restrict__ zero) {
    // 256 threads move the RB rows in and out (VEC: 16 bytes per lane and access -- d a multiple of 4, both bases 16-byte
    // aligned: 32 instead of 128 dependent-latency round trips per row block, four waves of them instead of one), the first
    // RB threads run the rows' chains.
    __shared__ float tile[RB][DP + 1];  // +1: the per-lane row walk below is bank-conflict free
    const int tid = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * RB;
    const int nr = (int)min((long long)RB, n_rows - row0);
    const float *src = E + row0 * (long long)d;
    if (d < DP || nr < RB) {
        for (int i = tid; i < RB * DP; i += 256) tile[i / DP][i % DP] = 0.0f;
        __syncthreads();
    }
    const int total = nr * d;
    if constexpr (VEC) {
        const f32x4 *src4 = reinterpret_cast<const f32x4 *>(src);
        for (int i4 = tid; i4 < total / 4; i4 += 256) {
            const f32x4 v = src4[i4];
            const int i = 4 * i4, r = i / d, c = i - r * d;  // (d % 4 == 0: the four stay in one row)
            tile[r][c] = v[0];
            tile[r][c + 1] = v[1];
            tile[r][c + 2] = v[2];
            tile[r][c + 3] = v[3];
        }
    } else {
        for (int i = tid; i < total; i += 256) tile[i / d][i % d] = src[i];
    }
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
    if constexpr (VEC) {
        f32x4 *dst4 = reinterpret_cast<f32x4 *>(dst);
        for (int i4 = tid; i4 < nr * DP / 4; i4 += 256) {
            const int i = 4 * i4, r = i / DP, p = i % DP;  // (p % 4 == 0: s = 0 .. 3 below)
            const int g = p >> 3, hh = (p >> 2) & 1;
            const float *t = &tile[r][8 * g + hh];
            dst4[i4] = f32x4{t[0], t[2], t[4], t[6]};
        }
    } else {
        for (int i = tid; i < nr * DP; i += 256) {
            const int r = i / DP, p = i % DP;
            const int g = p >> 3, hh = (p >> 2) & 1, s = p & 3;
            dst[i] = tile[r][8 * g + 2 * s + hh];
        }
    }
}

#include "knn_plan.inc"           // shapes, LDS budgets, target segments (plain C++)
#include "projection_tables.inc"  // the embed kernel's lookup tables (plain C++)
#include "csr_compact.inc"        // dead-feature filter (plain C++)
#include "host_upload.inc"        // host CSR -> device in chunks: raw over PCIe and compacted on the host at once
#include "knn_exact.inc"      // K3 / K4: fp32 MFMA tile kernel with LDS top-k lists, merge
#include "knn_prefilter.inc"  // P1 / P2: fp16 MFMA candidate pass, merges, certificate + re-rank, range pass
#include "knn_prefilter_pp.inc"  // P1 for the 256-register shapes: the two waves of a SIMD take turns at the matrix pipe
#include "knn_order.inc"      // P1: scan order by chunk mask (sort keys, ordered fp16 copy)
#include "dedup_classes.inc"  // duplicate-row classes: hash, tables, gathers, expansion
#include "knn_generic.inc"    // d > 512 or k > 64: every pair on the vector ALU

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
    hipStream_t aux_stream[3] = {nullptr, nullptr, nullptr};  // further queues for the prefilter pass's launches
    hipEvent_t aux_ev[4] = {nullptr, nullptr, nullptr, nullptr};  // [0] fork, [1..3] joins
    hipDeviceProp_t prop;
    // projection
    long long n_features = 0;
    int d = 0;
    long long p_nnz = 0, p_rows = 0;
    DevBuf ftab, crow, ent;
    std::vector<uint32_t> h_bits;  // host copy of ftab's bitmap words
    // scratch for the host-pointer API
    DevBuf a_indptr, a_indices, E, Ehat, zero, idx, dist, ws;
    DevBuf c_indptr, c_indices;             // ... the chunks the host compacted (upload_embed_pipelined)
    hup::PinnedBuf stage_ids, stage_ptr;    // ... their pinned staging
    hipEvent_t up_ev[2] = {nullptr, nullptr};  // ... the two raw runs in flight
    hup::WorkerPool up_pool;                // ... the helpers
    // k-mer search (kmer_search.inc)
    DevBuf ks_seq, ks_off, ks_codes, ks_keys, ks_vals, ks_bloom, ks_counter, ks_pairs, ks_pairs2, ks_flag, ks_pos,
        ks_idx, ks_rows, ks_indptr, ks_tmp, kc_counts;
    DevBuf kc_a0, kc_a1, kc_c0, kc_c1, kc_mk, kc_mv, kc_rc;  // counting in blocks: accumulated table (ping / pong), merge buffers
    long long ks_nnz = 0, kc_n = 0;
    int64_t kc_block_chars = 0;  // fdr_set_kmer_count_block
    int kc_blocks = 0;           // blocks of the last fdr_kmer_count
    int kc_k = 0, kc_acc = 0;    // incremental counting: k (0: not begun), which of the ping / pong tables is current
    long long kc_na = 0;         // ... its entries
    // timing: when enabled, every launch of kernel kind i gets its own hipEvent pair on the launch
    // stream; fdr_timing_read() sums the elapsed times of all launches since the last read
    int knn_mode = FDR_MODE_AUTO;
    int dedup_mode = FDR_DEDUP_AUTO;
    int last_flagged = 0;  // prefilter mode: queries of the last call that took the exact path
    int last_unique_targets = 0, last_unique_queries = 0;  // duplicate-row classes of the last call
    int last_pass_launches = 0, last_pass_queues = 0;      // prefilter pass of the last call
    // fdr_last_query_paths: the last call's per-query path codes (in that call's workspace), or one code for all rows
    struct {
        const uint8_t *dev = nullptr;
        int64_t n = 0;  // 0: nothing recorded
        uint8_t all = 0;
        hipStream_t stream = nullptr;
    } paths;
    // duplicate-row classes built by fdr_knn_classes_dev for the calls that follow it (fdr_knn_unique_dev /
    // fdr_knn_expand_dev): the tables live in the caller's workspace
    struct {
        bool valid = false;
        const float *That = nullptr;
        const uint8_t *tzero = nullptr;
        int64_t nt = 0, nq_max = 0;
        int d = 0, k = 0, nu = 0;
        void *ws = nullptr;
        size_t ws_bytes = 0;
    } cls;
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

FDR_EXPORT int fdr_padded_dim(int d) { return padded_dim(d); }

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
    hipError_t e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipGetDeviceProperties(&c->prop, device_id);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(FDR_E_HIP, "fdr_create: device %d: %s", device_id, hipGetErrorString(e));
    }
    c->num_cus = c->prop.multiProcessorCount > 0 ? c->prop.multiProcessorCount : 256;
    // the one environment variable of the release library, read once: the context's initial k-NN mode
    if (const char *m = getenv("FDR_KNN_MODE")) {
        if (strcmp(m, "exact") == 0) c->knn_mode = FDR_MODE_EXACT;
        else if (strcmp(m, "prefilter") == 0) c->knn_mode = FDR_MODE_PREFILTER;
    }
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
                      &ctx->ks_idx, &ctx->ks_rows, &ctx->ks_indptr, &ctx->ks_tmp, &ctx->kc_counts,
                      &ctx->kc_a0, &ctx->kc_a1, &ctx->kc_c0, &ctx->kc_c1, &ctx->kc_mk, &ctx->kc_mv, &ctx->kc_rc};
    for (DevBuf *b : bufs) b->release();
    ctx->up_pool.stop();
    ctx->c_indptr.release();
    ctx->c_indices.release();
    ctx->stage_ids.release();
    ctx->stage_ptr.release();
    for (hipEvent_t e : ctx->up_ev)
        if (e) (void)hipEventDestroy(e);
    for (int i = 0; i < FDR_NUM_KERNELS; ++i)
        for (hipEvent_t e : ctx->ev_pool[i]) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(ctx->stream);
    for (hipStream_t a : ctx->aux_stream)
        if (a) {
            (void)hipStreamSynchronize(a);
            (void)hipStreamDestroy(a);
        }
    for (hipEvent_t e : ctx->aux_ev)
        if (e) (void)hipEventDestroy(e);
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

FDR_EXPORT int fdr_last_query_paths(fdr_ctx *ctx, uint8_t *paths, int64_t n_queries) {
    int rc = use_device(ctx);
    if (rc) return rc;
    if (!paths || n_queries <= 0) return fail(FDR_E_ARG, "last_query_paths: bad argument");
    if (ctx->paths.n != n_queries)
        return fail(FDR_E_STATE, "last_query_paths: the last k-NN call recorded codes for %lld query rows, not %lld",
                    (long long)ctx->paths.n, (long long)n_queries);
    if (!ctx->paths.dev) {
        memset(paths, ctx->paths.all, (size_t)n_queries);
        return FDR_OK;
    }
    HIP_TRY(hipMemcpyAsync(paths, ctx->paths.dev, (size_t)n_queries, hipMemcpyDeviceToHost, ctx->paths.stream));
    HIP_TRY(hipStreamSynchronize(ctx->paths.stream));
    return FDR_OK;
}

FDR_EXPORT int fdr_last_unique(fdr_ctx *ctx, int *unique_targets, int *unique_queries) {
    if (!ctx || !unique_targets || !unique_queries) return fail(FDR_E_ARG, "bad argument");
    *unique_targets = ctx->last_unique_targets;
    *unique_queries = ctx->last_unique_queries;
    return FDR_OK;
}

FDR_EXPORT int fdr_last_prefilter_launches(fdr_ctx *ctx, int *launches, int *queues) {
    if (!ctx || !launches || !queues) return fail(FDR_E_ARG, "bad argument");
    *launches = ctx->last_pass_launches;
    *queues = ctx->last_pass_queues;
    return FDR_OK;
}

FDR_EXPORT int fdr_set_knn_mode(fdr_ctx *ctx, int mode) {
    if (!ctx || mode < FDR_MODE_AUTO || mode > FDR_MODE_PREFILTER) return fail(FDR_E_ARG, "bad k-NN mode");
    ctx->knn_mode = mode;
    return FDR_OK;
}

FDR_EXPORT int fdr_set_dedup_mode(fdr_ctx *ctx, int mode) {
    if (!ctx || mode < FDR_DEDUP_AUTO || mode > FDR_DEDUP_FORCE) return fail(FDR_E_ARG, "bad duplicate-row mode");
    ctx->dedup_mode = mode;
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
    ProjectionTables T;
    if ((rc = build_projection_tables(n_features, d, p_indptr, p_cols, p_vals, T))) return rc;
    const std::vector<PU2> &ftab = T.ftab, &ent = T.ent;
    const std::vector<PU4> &crow = T.rowinfo;
    const int64_t nnz = p_indptr[n_features];
    const unsigned rows = T.rows;
    static_assert(sizeof(PU2) == sizeof(uint2) && sizeof(PU4) == sizeof(uint4), "table records = the kernel's uint2 / uint4");
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
    ctx->h_bits.resize(ftab.size());  // host copy of the bitmap: fdr_csr_compact
    for (size_t w = 0; w < ftab.size(); ++w) ctx->h_bits[w] = ftab[w].x;
    return FDR_OK;
}

FDR_EXPORT int fdr_csr_compact(fdr_ctx *ctx, int64_t n_rows, const int64_t *a_indptr, const int32_t *a_indices,
                               int64_t *out_indptr, int32_t *out_indices, int64_t out_capacity, int32_t n_threads) {
    if (!ctx) return fail(FDR_E_ARG, "null context");
    if (ctx->n_features <= 0) return fail(FDR_E_STATE, "csr_compact: no projection loaded");
    return csrc::compact(ctx->h_bits, ctx->n_features, n_rows, a_indptr, a_indices, out_indptr, out_indices,
                         out_capacity, n_threads);
}

FDR_EXPORT int fdr_host_register(fdr_ctx *ctx, void *ptr, size_t bytes) {
    int rc = use_device(ctx);
    if (rc) return rc;
    if (!ptr || bytes == 0) return fail(FDR_E_ARG, "host_register: empty buffer");
    HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return FDR_OK;
}

FDR_EXPORT int fdr_host_unregister(fdr_ctx *ctx, void *ptr) {
    int rc = use_device(ctx);
    if (rc) return rc;
    if (!ptr) return fail(FDR_E_ARG, "host_unregister: null pointer");
    HIP_TRY(hipHostUnregister(ptr));
    return FDR_OK;
}

// ---- launches --------------------------------------------------------------------------------
static int launch_embed(fdr_ctx *ctx, int64_t n_rows, const int64_t *d_indptr,
                        const int32_t *d_indices, float *d_E, hipStream_t st) {
    if (ctx->n_features <= 0) return fail(FDR_E_STATE, "embed: no projection loaded");
    if (n_rows < 0) return fail(FDR_E_ARG, "embed: n_rows < 0");
    if (n_rows == 0) return FDR_OK;
    const int dp = fdr_padded_dim(ctx->d);
    const long long blocks_needed = (n_rows + 7) / 8;  // (four waves per block, two to eight rows per wave and turn; grid-stride)
    const int grid = (int)std::min<long long>(blocks_needed, (long long)ctx->num_cus * 8 * 4);
    int trc = timing_begin(ctx, FDR_KERNEL_EMBED, st);
    if (trc) return trc;
#define FDR_LAUNCH_EMBED(DP_)                                                                   \
    hipLaunchKernelGGL(FDR_EMBED_KERNEL<DP_>, dim3(grid), dim3(256), 0, st, (long long)n_rows,  \
                       (const long long *)d_indptr, d_indices, ctx->n_features,                 \
                       (const uint2 *)ctx->ftab.p, (const uint4 *)ctx->crow.p,                  \
                       (const uint2 *)ctx->ent.p, ctx->d, d_E)
#define FDR_EMBED_KERNEL embed_csr_kernel
    if (dp == 128)
        FDR_LAUNCH_EMBED(128);
    else if (dp == 256)
        FDR_LAUNCH_EMBED(256);
    else if (dp == 512)
        FDR_LAUNCH_EMBED(512);
#undef FDR_EMBED_KERNEL
#define FDR_EMBED_KERNEL embed_csr_wide_kernel
    else if (dp == 1024)
        FDR_LAUNCH_EMBED(1024);
    else
        FDR_LAUNCH_EMBED(2048);
#undef FDR_EMBED_KERNEL
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
    const int rb = dp == 128 ? 64 : dp == 256 ? 32 : dp == 512 ? 16 : dp == 1024 ? 8 : 4;
    const long long grid = (n_rows + rb - 1) / rb;
    if (grid > 0x7fffffffll) return fail(FDR_E_ARG, "normalize: too many rows");
    int trc = timing_begin(ctx, FDR_KERNEL_NORMALIZE, st);
    if (trc) return trc;
    const bool vec = d % 4 == 0 && (reinterpret_cast<uintptr_t>(d_E) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_Ehat) & 15) == 0;
#define FDR_LAUNCH_NORM(DP_, RB_)                                                                                  \
    do {                                                                                                            \
        if (vec)                                                                                                    \
            hipLaunchKernelGGL((normalize_rows_kernel<DP_, RB_, true>), dim3((unsigned)grid), dim3(256), 0, st, d_E, \
                               (long long)n_rows, d, d_Ehat, d_zero);                                              \
        else                                                                                                        \
            hipLaunchKernelGGL((normalize_rows_kernel<DP_, RB_, false>), dim3((unsigned)grid), dim3(256), 0, st,    \
                               d_E, (long long)n_rows, d, d_Ehat, d_zero);                                         \
    } while (0)
    if (dp == 128) FDR_LAUNCH_NORM(128, 64);
    else if (dp == 256) FDR_LAUNCH_NORM(256, 32);
    else if (dp == 512) FDR_LAUNCH_NORM(512, 16);
    else if (dp == 1024) FDR_LAUNCH_NORM(1024, 8);
    else FDR_LAUNCH_NORM(2048, 4);
#undef FDR_LAUNCH_NORM
    HIP_TRY(hipGetLastError());
    return timing_end(ctx, FDR_KERNEL_NORMALIZE, st);
}

static size_t knn_workspace_bytes_impl(const fdr_ctx *ctx, int64_t nq, int64_t nt, int d, int k);
static bool knn_generic_wanted(int dp, int k) { return dp > FDR_FAST_MAX_DIM || k > FDR_FAST_MAX_K; }

// ---- prefilter mode: workspace layout -------------------------------------------------------
// mode: FDR_MODE_AUTO uses the fp16 prefilter whenever it applies (d <= 128, k + 8 <= 64) and the
// target set is large enough to pay for it; FDR_KNN_MODE=exact|prefilter|auto overrides the context.
static bool knn_prefilter_wanted(const fdr_ctx *ctx, int dp, int64_t nt, int k) {
    const int mode = ctx->knn_mode;
    if (mode == FDR_MODE_EXACT) return false;
    const int kp = (k + prefilter_extra(k) + 1) & ~1;
    if (!(kp <= FDR_FAST_MAX_K && nt >= kp)) return false;
    if (nt > (int64_t)FDR_MAX_SEG << FDR_PREFILTER_MAX_IB) return false;  // segments too long for the keys
    return mode == FDR_MODE_PREFILTER || nt >= 8192;
}

struct PrefilterLayout {
    int kp, chunk;
    size_t knn_bytes;  // region shared (in stream order) by the prefilter pass and the exact passes
    size_t off_ht, off_hq, off_cand, off_counter, off_flagged, off_qc, off_qzc, off_idxc, off_distc;
    size_t off_rlist, off_theta, off_hqc, off_thetac, off_cnt, off_rcand, total;  // range pass
    size_t off_path;  // per-query path codes (fdr_last_query_paths)
    int rchunk;
    int ordered;  // ordered scan possible: sort keys, order tables, ordered fp16 copies
    size_t off_okeys, off_okeys_s, off_ovals, off_perm_t, off_perm_q, off_ho_t, off_ho_q, off_otmp, otmp_bytes;
};

static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

static PrefilterLayout prefilter_layout(const fdr_ctx *ctx, int64_t nq, int64_t nt, int d, int k) {
    PrefilterLayout L;
    L.kp = (k + prefilter_extra(k) + 1) & ~1;
    L.chunk = (int)std::min<int64_t>(nq, 16384);
    const size_t exact_all = knn_plan(ctx->num_cus, nq, nt, d, k).total_bytes;
    const int dp = fdr_padded_dim(d);
    const KnnPlan pp = knn_plan(ctx->num_cus, nq, nt, d, L.kp, prefilter_shape(dp, L.kp, nq, ctx->num_cus, nt));
    // (a later call on fewer unique rows may plan more, shorter segments: room for the largest such plan)
    // (the bound words and the lists padded for the widest query block: a later call may choose the other shape)
    const size_t pre = std::max(pp.total_bytes, pp.bits_bytes + align256((size_t)(nq + 512) * 4) +
                                                    prefilter_partial_bound(nq, nt, L.kp, pp.qw));
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
    L.off_path = o;     o += align256((size_t)nq);
    L.ordered = pp.cohort > 0 || dev_knobs().ordered != 0;  // (the sizes at which the pass runs in synchronised rounds)
    L.off_okeys = L.off_okeys_s = L.off_ovals = L.off_perm_t = L.off_perm_q = L.off_ho_t = L.off_ho_q = 0;
    L.off_otmp = L.otmp_bytes = 0;
    if (L.ordered) {
        const size_t nmax = (size_t)std::max(nq, nt);  // (the keys / sort scratch serve the targets, then the queries)
        L.off_okeys = o;    o += align256(nmax * 8);
        L.off_okeys_s = o;  o += align256(nmax * 8);
        L.off_ovals = o;    o += align256(nmax * 4);
        L.off_perm_t = o;   o += align256((size_t)nt * 4);
        L.off_perm_q = o;   o += align256((size_t)nq * 4);
        L.off_ho_t = o;     o += align256((size_t)nt * dp * 2);
        L.off_ho_q = o;     o += align256((size_t)nq * dp * 2);
        size_t t_sort = 0;
        (void)rocprim::radix_sort_pairs(nullptr, t_sort, (u64 *)nullptr, (u64 *)nullptr, (int *)nullptr, (int *)nullptr,
                                        nmax, 0, 40, (hipStream_t) nullptr);
        L.otmp_bytes = align256(t_sort);
        L.off_otmp = o;     o += L.otmp_bytes;
    }
    L.total = o;
    return L;
}

FDR_EXPORT size_t fdr_knn_workspace_bytes(fdr_ctx *ctx, int64_t nq, int64_t nt, int32_t d,
                                          int32_t k) {
    if (!ctx || nq <= 0 || nt <= 0 || k <= 0 || k > FDR_MAX_K || fdr_padded_dim(d) < 0) return 0;
    if (knn_generic_wanted(fdr_padded_dim(d), k)) return 256;  // (the generic kernel needs no scratch)
    return knn_workspace_bytes_impl(ctx, nq, nt, d, k);
}

static int launch_knn_exact(fdr_ctx *ctx, const float *d_Qhat, const uint8_t *d_qzero, int64_t nq,
                      const float *d_That, const uint8_t *d_tzero, int64_t nt, int64_t t_base,
                      int d, int k, int32_t *d_idx, float *d_dist, void *d_ws, size_t ws_bytes,
                      hipStream_t st) {
    const int dp = fdr_padded_dim(d);
    if (dp < 0) return fail(FDR_E_ARG, "knn: dimension %d unsupported (1..%d)", d, FDR_MAX_DIM);
    if (k < 1 || k > FDR_FAST_MAX_K || dp > FDR_FAST_MAX_DIM)
        return fail(FDR_E_ARG, "knn: k=%d, d=%d outside the MFMA kernels' shapes", k, d);
    if (nq < 0 || nt < k) return fail(FDR_E_ARG, "knn: need n_targets (%lld) >= k (%d)", (long long)nt, k);
    if (nt + t_base > 0x7fffffffll || nq > 0x7fffffffll)
        return fail(FDR_E_ARG, "knn: row numbers exceed int32");
    if (nq == 0) return FDR_OK;
    if (!d_Qhat || !d_qzero || !d_That || !d_tzero || !d_idx || !d_dist || !d_ws)
        return fail(FDR_E_ARG, "knn: null device pointer");
    const KnnPlan p = knn_plan(ctx->num_cus, nq, nt, d, k);
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
    const int dbg = dev_knobs().debug;  // (development builds only; 0 in the release library)
    (void)dbg;
    // the nqb * nseg work items in one launch, or (p.cohort > 0: knn_plan_compute) in synchronised rounds dealt to
    // p.queues queues, like the prefilter pass
    const long long n_items = (long long)p.nqb * p.nseg;
    const long long per_launch = p.cohort > 0 ? p.cohort : n_items;
    const int nqueues = p.cohort > 0 && n_items > per_launch ? std::max(1, std::min(p.queues, 2)) : 1;
    hipStream_t qs[2] = {st, st};
    if (nqueues > 1) {
        if (!ctx->aux_ev[0]) {
            for (hipStream_t &a : ctx->aux_stream) HIP_TRY(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
            for (hipEvent_t &e : ctx->aux_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        qs[1] = ctx->aux_stream[0];
    }
    int trc = timing_begin(ctx, FDR_KERNEL_KNN_TILE, st);
    if (trc) return trc;
    if (nqueues > 1) {
        HIP_TRY(hipEventRecord(ctx->aux_ev[0], st));
        HIP_TRY(hipStreamWaitEvent(qs[1], ctx->aux_ev[0], 0));
    }
#define FDR_LAUNCH_KNN(DP_, NQ_, NW_, WPS_)                                                          \
    do {                                                                                             \
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_tile_kernel<DP_, NQ_, NW_, WPS_>), \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));          \
        int li_ = 0;                                                                                 \
        for (long long base_ = 0; base_ < n_items; base_ += per_launch, ++li_)                       \
            hipLaunchKernelGGL((knn_tile_kernel<DP_, NQ_, NW_, WPS_>),                               \
                               dim3((unsigned)std::min(per_launch, n_items - base_)), dim3(64 * NW_), lds, \
                               qs[li_ % nqueues], d_Qhat, d_qzero, (int)nq, d_That, d_bits, (int)nt, (int)t_base, \
                               p.segs, k, p.nq_pad, d_partial, d_shared, knn_qcap(sh, k), (int)base_, p.nqb \
                               FDR_DBG_ARG(dbg));                                                    \
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
    if (nqueues > 1) {
        HIP_TRY(hipEventRecord(ctx->aux_ev[1], qs[1]));
        HIP_TRY(hipStreamWaitEvent(st, ctx->aux_ev[1], 0));
    }
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

// Which of the prefilter pass's shapes are compiled (-DFDR_SHAPE_MASK=<bits of the FDR_SHAPE_CASE numbers below>,
// -DFDR_LH_MASK=<16 | 32 | 48>): the release library holds the shapes prefilter_shape() chooses by itself, a development
// build all of them (the round-3 shapes the knobs D256 / PP can put back), or a subset to save minutes of hipcc.
#ifndef FDR_SHAPE_MASK
#ifdef FDR_DEV
#define FDR_SHAPE_MASK 0xffff
#else
#define FDR_SHAPE_MASK 0xA4CE  // the shapes prefilter_shape() can choose without a development knob: cases 1-3, 6, 7, 10, 13, 15
#endif
#endif
#ifndef FDR_LH_MASK
#define FDR_LH_MASK 48
#endif
#define FDR_SEL_1(...) __VA_ARGS__
#define FDR_SEL_0(...) return fail(FDR_E_STATE, "this development build was compiled without the kernel shape this call needs")
#define FDR_CAT2(a, b) a##b
#define FDR_CAT(a, b) FDR_CAT2(a, b)
#if (FDR_SHAPE_MASK >> 0) & 1
#define FDR_SHAPE_ON_0 1
#else
#define FDR_SHAPE_ON_0 0
#endif
#if (FDR_SHAPE_MASK >> 1) & 1
#define FDR_SHAPE_ON_1 1
#else
#define FDR_SHAPE_ON_1 0
#endif
#if (FDR_SHAPE_MASK >> 2) & 1
#define FDR_SHAPE_ON_2 1
#else
#define FDR_SHAPE_ON_2 0
#endif
#if (FDR_SHAPE_MASK >> 3) & 1
#define FDR_SHAPE_ON_3 1
#else
#define FDR_SHAPE_ON_3 0
#endif
#if (FDR_SHAPE_MASK >> 4) & 1
#define FDR_SHAPE_ON_4 1
#else
#define FDR_SHAPE_ON_4 0
#endif
#if (FDR_SHAPE_MASK >> 5) & 1
#define FDR_SHAPE_ON_5 1
#else
#define FDR_SHAPE_ON_5 0
#endif
#if (FDR_SHAPE_MASK >> 6) & 1
#define FDR_SHAPE_ON_6 1
#else
#define FDR_SHAPE_ON_6 0
#endif
#if (FDR_SHAPE_MASK >> 7) & 1
#define FDR_SHAPE_ON_7 1
#else
#define FDR_SHAPE_ON_7 0
#endif
#if (FDR_SHAPE_MASK >> 8) & 1
#define FDR_SHAPE_ON_8 1
#else
#define FDR_SHAPE_ON_8 0
#endif
#if (FDR_SHAPE_MASK >> 9) & 1
#define FDR_SHAPE_ON_9 1
#else
#define FDR_SHAPE_ON_9 0
#endif
#if (FDR_SHAPE_MASK >> 10) & 1
#define FDR_SHAPE_ON_10 1
#else
#define FDR_SHAPE_ON_10 0
#endif
#if (FDR_SHAPE_MASK >> 11) & 1
#define FDR_SHAPE_ON_11 1
#else
#define FDR_SHAPE_ON_11 0
#endif
#if (FDR_SHAPE_MASK >> 12) & 1
#define FDR_SHAPE_ON_12 1
#else
#define FDR_SHAPE_ON_12 0
#endif
#if (FDR_SHAPE_MASK >> 13) & 1
#define FDR_SHAPE_ON_13 1
#else
#define FDR_SHAPE_ON_13 0
#endif
#if (FDR_SHAPE_MASK >> 14) & 1
#define FDR_SHAPE_ON_14 1
#else
#define FDR_SHAPE_ON_14 0
#endif
#if (FDR_SHAPE_MASK >> 15) & 1
#define FDR_SHAPE_ON_15 1
#else
#define FDR_SHAPE_ON_15 0
#endif
#if FDR_LH_MASK & 16
#define FDR_LH_ON_16 1
#else
#define FDR_LH_ON_16 0
#endif
#if FDR_LH_MASK & 32
#define FDR_LH_ON_32 1
#else
#define FDR_LH_ON_32 0
#endif
#define FDR_SHAPE_CASE(n, ...) FDR_CAT(FDR_SEL_, FDR_SHAPE_ON_##n)(__VA_ARGS__)
#define FDR_LH_CASE(n, ...) FDR_CAT(FDR_SEL_, FDR_LH_ON_##n)(__VA_ARGS__)
// ---- prefilter mode: fp16 pass -> certificate + exact re-rank -> exact pass for the rest -------
static int launch_knn_prefilter(fdr_ctx *ctx, const float *d_Qhat, const uint8_t *d_qzero, int64_t nq,
                                const float *d_That, const uint8_t *d_tzero, int64_t nt, int64_t t_base,
                                int d, int k, int32_t *d_idx, float *d_dist, void *d_ws, size_t ws_bytes,
                                hipStream_t st) {
    // the queries ARE the targets (row i of one is row i of the other): one fp16 copy serves both sides
    const bool self = d_Qhat == d_That && d_qzero == d_tzero && nq == nt;
    const PrefilterLayout L = prefilter_layout(ctx, nq, nt, d, k);
    if (ws_bytes < L.total)
        return fail(FDR_E_ARG, "knn: workspace %zu < required %zu bytes", ws_bytes, L.total);
    char *ws = static_cast<char *>(d_ws);
    _Float16 *d_ht = reinterpret_cast<_Float16 *>(ws + L.off_ht);
    _Float16 *d_hq = self ? d_ht : reinterpret_cast<_Float16 *>(ws + L.off_hq);
    u64 *d_cand = reinterpret_cast<u64 *>(ws + L.off_cand);
    int *d_counter = reinterpret_cast<int *>(ws + L.off_counter);
    int *d_flagged = reinterpret_cast<int *>(ws + L.off_flagged);
    float *d_qc = reinterpret_cast<float *>(ws + L.off_qc);
    uint8_t *d_qzc = reinterpret_cast<uint8_t *>(ws + L.off_qzc);
    int32_t *d_idxc = reinterpret_cast<int32_t *>(ws + L.off_idxc);
    float *d_distc = reinterpret_cast<float *>(ws + L.off_distc);
    const int kp = L.kp;
    uint8_t *d_path = reinterpret_cast<uint8_t *>(ws + L.off_path);
    ctx->paths.dev = d_path;
    ctx->paths.n = nq;
    ctx->paths.stream = st;

    const int dp = fdr_padded_dim(d);
    const int pshape = prefilter_shape(dp, kp, nq, ctx->num_cus, nt);
    const KnnPlan p = knn_plan(ctx->num_cus, nq, nt, d, kp, pshape);
    const KnnShape &sh = kShapes[pshape];
    unsigned *d_bits = reinterpret_cast<unsigned *>(ws);
    unsigned *d_shared = reinterpret_cast<unsigned *>(ws + p.bits_bytes);
    u64 *d_partial = reinterpret_cast<u64 *>(ws + p.bits_bytes + p.shared_bytes);

    int trc = timing_begin(ctx, FDR_KERNEL_KNN_RERANK, st);  // conversion + set-up count as "rerank"
    if (trc) return trc;
    hipLaunchKernelGGL(to_half_kernel, dim3((unsigned)((nt * (dp / 8) + 255) / 256)), dim3(256), 0, st, d_That,
                       (long long)nt * (dp / 8), d_ht);
    if (!self)
        hipLaunchKernelGGL(to_half_kernel, dim3((unsigned)((nq * (dp / 8) + 255) / 256)), dim3(256), 0, st, d_Qhat,
                           (long long)nq * (dp / 8), d_hq);
    hipLaunchKernelGGL(pack_zero_bits_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st,
                       d_tzero, (int)nt, d_bits, d_shared, p.nq_pad);
    HIP_TRY(hipGetLastError());
    // Ordered scan (knn_order.inc): rows by chunk mask, fp16 copies in that order
    OrderArgs ord = {nullptr, nullptr};
    const _Float16 *p1_q = d_hq, *p1_t = d_ht;  // what the candidate pass streams
    if (L.ordered) {
        u64 *okeys = reinterpret_cast<u64 *>(ws + L.off_okeys), *okeys_s = reinterpret_cast<u64 *>(ws + L.off_okeys_s);
        int *ovals = reinterpret_cast<int *>(ws + L.off_ovals);
        int *perm_t = reinterpret_cast<int *>(ws + L.off_perm_t), *perm_q = reinterpret_cast<int *>(ws + L.off_perm_q);
        _Float16 *ho_t = reinterpret_cast<_Float16 *>(ws + L.off_ho_t);
        _Float16 *ho_q = self ? ho_t : reinterpret_cast<_Float16 *>(ws + L.off_ho_q);
        auto order = [&](const float *X, int64_t n, int *perm, _Float16 *out) -> int {
            hipLaunchKernelGGL(row_chunk_keys_kernel, dim3((unsigned)(((size_t)n * 16 + 255) / 256)), dim3(256), 0, st, X,
                               (int)n, dp, okeys, ovals);
            size_t tb = L.otmp_bytes;
            HIP_TRY(rocprim::radix_sort_pairs(ws + L.off_otmp, tb, okeys, okeys_s, ovals, perm, (size_t)n, 0, 40, st));
            const long long groups = (long long)n * (dp / 8);
            hipLaunchKernelGGL(to_half_ordered_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, X,
                               (const int *)perm, groups, dp / 8, out);
            HIP_TRY(hipGetLastError());
            return FDR_OK;
        };
        int orc;
        if ((orc = order(d_That, nt, perm_t, ho_t))) return orc;
        if (!self && (orc = order(d_Qhat, nq, perm_q, ho_q))) return orc;
        ord.perm_t = perm_t;
        p1_t = ho_t;
        ord.perm_q = self ? perm_t : perm_q;
        p1_q = ho_q;
    }
    const size_t lds = knn_lds_bytes(sh, kp);
    int max_seg = 1;
    for (int i = 0; i < p.nseg; ++i) max_seg = std::max(max_seg, p.segs.b[i + 1] - p.segs.b[i]);
    const int ib = prefilter_index_bits(max_seg);
    if (ib > FDR_PREFILTER_MAX_IB) return fail(FDR_E_ARG, "knn prefilter: segment of %d rows", max_seg);
    if ((trc = timing_end(ctx, FDR_KERNEL_KNN_RERANK, st))) return trc;
    const int pdbg = dev_knobs().debug;
    (void)pdbg;
    // the nqb * nseg work items in launches of p.cohort workgroups (0: one launch): see knn_plan_compute
    const long long n_items = (long long)p.nqb * p.nseg;
    const long long per_launch = p.cohort > 0 ? p.cohort : n_items;
    // Several queues: with the launches of the synchronised rounds dealt round-robin to the caller's stream and
    // further ones, the workgroups of a later launch take the slots the stragglers of an earlier one have
    // freed (one queue: every launch ends with its slowest workgroup while the rest of the chip idles).
    const int nqueues = p.cohort > 0 && n_items > per_launch ? std::max(1, std::min(p.queues, 4)) : 1;
    // one timed span for the whole pass when launches overlap (their own spans would count the same time twice)
    if (nqueues > 1 && (trc = timing_begin(ctx, FDR_KERNEL_KNN_PREFILTER, st))) return trc;
    hipStream_t qs[4] = {st, st, st, st};
    if (nqueues > 1) {
        if (!ctx->aux_ev[0]) {
            for (hipStream_t &a : ctx->aux_stream) HIP_TRY(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
            for (hipEvent_t &e : ctx->aux_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        for (int q = 1; q < nqueues; ++q) qs[q] = ctx->aux_stream[q - 1];
    }
    auto fork = [&]() -> int {  // what `st` has queued so far is visible to the other queues' next launches
        if (nqueues > 1) {
            HIP_TRY(hipEventRecord(ctx->aux_ev[0], st));
            for (int q = 1; q < nqueues; ++q) HIP_TRY(hipStreamWaitEvent(qs[q], ctx->aux_ev[0], 0));
        }
        return FDR_OK;
    };
    auto join = [&]() -> int {  // `st` waits for everything the other queues have been given
        for (int q = 1; q < nqueues; ++q) {
            HIP_TRY(hipEventRecord(ctx->aux_ev[q], qs[q]));
            HIP_TRY(hipStreamWaitEvent(st, ctx->aux_ev[q], 0));
        }
        return FDR_OK;
    };
    int li = 0;  // launches so far (launch li goes to queue li % nqueues)
    auto launch_items = [&](long long it_lo, long long it_hi) -> int {
#define FDR_LAUNCH_PRE3(KERNEL_, THREADS_)                                                              \
    do {                                                                                                \
        for (long long base_ = it_lo; base_ < it_hi; base_ += per_launch, ++li) { /* (one queue: every launch its own timed span) */ \
            hipStream_t ls_ = qs[li % nqueues];                                                         \
            if (lds > 32768) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(KERNEL_), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            if (nqueues == 1 && (trc = timing_begin(ctx, FDR_KERNEL_KNN_PREFILTER, ls_))) return trc;   \
            hipLaunchKernelGGL(KERNEL_, dim3((unsigned)std::min(per_launch, it_hi - base_)), dim3(THREADS_), lds, \
                               ls_, p1_q, (int)nq, p1_t, (int)nt, (int)t_base, p.segs, kp, p.nq_pad, d_partial, \
                               d_shared, ib, (int)base_, p.nqb, ord FDR_DBG_ARG(pdbg));             \
            if (nqueues == 1 && (trc = timing_end(ctx, FDR_KERNEL_KNN_PREFILTER, ls_))) return trc;     \
        }                                                                                               \
    } while (0)
#define FDR_LAUNCH_PRE2(DP_, NQ_, NW_, WPS_, U_, LH_) \
    FDR_LAUNCH_PRE3((knn_prefilter_kernel<DP_, NQ_, NW_, WPS_, U_, LH_>), 64 * NW_)
#define FDR_LAUNCH_PRE(DP_, NQ_, NW_, WPS_, U_)                                                         \
    do {                                                                                                \
        if (kp <= 32) FDR_LH_CASE(16, FDR_LAUNCH_PRE2(DP_, NQ_, NW_, WPS_, U_, 16));                    \
        else FDR_LH_CASE(32, FDR_LAUNCH_PRE2(DP_, NQ_, NW_, WPS_, U_, 32));                             \
    } while (0)
        // (FDR_SHAPE_CASE: a development build may compile a subset of the shapes, -DFDR_SHAPE_MASK=bits: 10 minutes of hipcc otherwise)
        if (dp == 128 && kp <= 32) {
            // d <= 128, K' <= 32: the stage's two tiles as two interleaved MFMA chains (1-4 % faster; still
            // <= 128 VGPRs)
            if (sh.nw == 8 && sh.tps == 8) FDR_SHAPE_CASE(15, FDR_LAUNCH_PRE3((knn_prefilter_kernel<128, 1, 8, 4, 4, 16, true>), 512));
            else if (sh.nw == 8) FDR_SHAPE_CASE(0, FDR_LAUNCH_PRE3((knn_prefilter_kernel<128, 1, 8, 4, 2, 16, true>), 512));
            else FDR_SHAPE_CASE(1, FDR_LAUNCH_PRE3((knn_prefilter_kernel<128, 1, 4, 4, 2, 16, true>), 256));
        } else if (dp == 128) FDR_SHAPE_CASE(2, FDR_LAUNCH_PRE(128, 1, 4, 4, 2));
        else if (sh.nw == 8 && sh.wps == 2 && (sh.tps == 16 || (dp == 512 && sh.tps == 8)) && dev_knobs().pp != 0) {
            // the ping-pong kernel (knn_prefilter_pp.inc)
            if (dp == 256) {
                if (kp <= 32) FDR_SHAPE_CASE(3, FDR_LH_CASE(16, FDR_LAUNCH_PRE3((knn_prefilter_pp_kernel<256, 8, 16>), 512)));
                else FDR_SHAPE_CASE(3, FDR_LH_CASE(32, FDR_LAUNCH_PRE3((knn_prefilter_pp_kernel<256, 8, 32>), 512)));
            } else if (dp == 512 && sh.tps == 16 && kp <= 32) {
                FDR_SHAPE_CASE(13, FDR_LH_CASE(16, FDR_LAUNCH_PRE3((knn_prefilter_pp_kernel<512, 8, 16>), 512)));
            } else if (dp == 512 && sh.tps == 8 && kp > 32) {
                FDR_SHAPE_CASE(13, FDR_LH_CASE(32, FDR_LAUNCH_PRE3((knn_prefilter_pp_kernel<512, 4, 32>), 512)));
            } else {
                return fail(FDR_E_STATE, "knn prefilter: no ping-pong kernel for d = %d, K' = %d, ring of %d units", d, kp, sh.tps);
            }
        } else if (dp == 256 && sh.tps == 16 && sh.nw == 8) FDR_SHAPE_CASE(14, FDR_LAUNCH_PRE(256, 1, 8, 2, 8));
        else if (dp == 256 && sh.tps == 8 && sh.nw == 8) FDR_SHAPE_CASE(4, FDR_LAUNCH_PRE(256, 1, 8, 2, 4));
        else if (dp == 512 && sh.tps == 8 && sh.nw == 8) FDR_SHAPE_CASE(5, FDR_LAUNCH_PRE(512, 1, 8, 2, 4));
        else if (dp == 256 && sh.tps == 8) FDR_SHAPE_CASE(6, FDR_LAUNCH_PRE(256, 1, 4, 2, 4));
        else if (dp == 512 && sh.tps == 8) FDR_SHAPE_CASE(7, FDR_LAUNCH_PRE(512, 1, 4, 2, 4));
        else if (dp == 256 && sh.wps == 2 && sh.nw == 8) FDR_SHAPE_CASE(8, FDR_LAUNCH_PRE(256, 1, 8, 2, 2));
        else if (dp == 256 && sh.wps == 2) FDR_SHAPE_CASE(9, FDR_LAUNCH_PRE(256, 1, 4, 2, 2));
        else if (dp == 256) FDR_SHAPE_CASE(10, FDR_LAUNCH_PRE(256, 1, 4, 3, 2));
        else FDR_SHAPE_CASE(11, FDR_LAUNCH_PRE(512, 1, 4, 2, 2));
#undef FDR_LAUNCH_PRE3
#undef FDR_LAUNCH_PRE2
#undef FDR_LAUNCH_PRE
        HIP_TRY(hipGetLastError());
        return FDR_OK;
    };
    int lrc;
    if ((lrc = fork())) return lrc;
    if ((lrc = launch_items(0, n_items))) return lrc;
    ctx->last_pass_launches = li;
    ctx->last_pass_queues = nqueues;
    if ((lrc = join())) return lrc;  // the merge below (on `st`) needs the other queues' launches too
    if (nqueues > 1 && (trc = timing_end(ctx, FDR_KERNEL_KNN_PREFILTER, st))) return trc;
#ifdef FDR_STAMPS
    {
        unsigned long long c[16][8];
        HIP_TRY(hipStreamSynchronize(st));
        for (int q = 1; q < nqueues; ++q) HIP_TRY(hipStreamSynchronize(qs[q]));
        HIP_TRY(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_stamps), sizeof(c)));
        for (int w = 0; w < sh.nw; ++w) {
            const double st_n = (double)std::max<unsigned long long>(1, c[w][5]);
            // (knn_prefilter_kernel: 0 dma-issue, 1 mfma+score, 2 share, 3 vmcnt(0), 4 barrier; knn_prefilter_pp_kernel: 0 pipe
            // turn, 1 dma-issue, 2 scoring, 3 / 4 barrier after the pipe / the other turn, 6 long other turns)
            fprintf(stderr, "[fdr stamps] wave %d: stages %llu  per stage (s_memtime ticks): ph0 %.0f  ph1 %.0f  ph2 %.0f  ph3 %.0f  "
                            "ph4 %.0f  long turns %llu\n", w, c[w][5], c[w][0] / st_n, c[w][1] / st_n, c[w][2] / st_n,
                    c[w][3] / st_n, c[w][4] / st_n, c[w][6]);
        }
        unsigned long long z[16][8] = {};
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)));
    }
#endif
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

    if ((trc = timing_begin(ctx, FDR_KERNEL_KNN_RERANK, st))) return trc;
    const int64_t mq = p.nseg * kp <= 64 ? 4 * MERGE_QPW : 4;  // queries per workgroup of knn_merge_keys_kernel
    hipLaunchKernelGGL(knn_merge_keys_kernel, dim3((unsigned)((nq + mq - 1) / mq)), dim3(256), 0, st,
                       (const u64 *)d_partial, p.nseg, (int)nq, p.nq_pad, kp, d_cand);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(d_counter, 0, 16, st));
    int *d_rlist = reinterpret_cast<int *>(ws + L.off_rlist);
    float *d_theta = reinterpret_cast<float *>(ws + L.off_theta);
    const float margin = 2.0f * prefilter_eps(ib) + 4.0e-7f;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_rerank_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, RERANK_LDS_BYTES));
    hipLaunchKernelGGL(knn_rerank_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), RERANK_LDS_BYTES, st,
                       (const u64 *)d_cand, kp, k, d_Qhat, d_qzero, d_That, (int)nq, dp, (int)t_base, margin,
                       d_idx, d_dist, d_counter, d_flagged, d_rlist, d_theta, d_path);
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
            // From 4096 plateau queries (16 blocks of 256, times the plan's segments) the pass runs on the ping-pong
            // skeleton (knn_range_pp_kernel: eight waves, one workgroup per CU, eight-unit stages); a handful of them (1 M
            // rows: 1100) keeps round 3's kernel with its many short segments.
            // (d <= 128, many plateau queries: round 3's eight-wave form, development knob RANGE8 with RANGEPP = 0)
            const int r8 = dev_knobs().range8, rpp = dev_knobs().rangepp;
            const bool use_pp = rpp == 1 || (rpp < 0 && c >= 4096);
            const bool wide = !use_pp && dp == 128 && (r8 == 1 || (r8 < 0 && rcount >= 65536));
            const KnnPlan rp = knn_plan(ctx->num_cus, c, nt, d, 1, use_pp ? FDR_SHAPE_PP256 : wide ? FDR_SHAPE_PREFILTER_W8 : range_shape(dp));  // (k = 1: ring-only LDS)
            const size_t rlds = use_pp ? (size_t)16 * 32 * 256 + (size_t)RANGE_LANE_BUF * 512 * 4
                                       : (size_t)RANGE_STAGES * 32 * 256 + (size_t)RANGE_LANE_BUF * (wide ? 512 : 256) * 4;  // ring + lane buffers
#define FDR_LAUNCH_RANGE(DP_, NW_, WPS_)                                                                \
    hipLaunchKernelGGL((knn_range_kernel<DP_, NW_, WPS_>), dim3((unsigned)rp.nqb, (unsigned)rp.nseg),      \
                       dim3(64 * NW_), rlds, st, (const _Float16 *)d_hqc, (const float *)d_thetac, c,     \
                       (const _Float16 *)d_ht, (int)nt, (int)t_base, rp.segs, d_cnt, d_rcand)
#define FDR_LAUNCH_RANGE_PP(DP_)                                                                        \
    do {                                                                                                \
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_range_pp_kernel<DP_, 8>),         \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds));            \
        hipLaunchKernelGGL((knn_range_pp_kernel<DP_, 8>), dim3((unsigned)rp.nqb, (unsigned)rp.nseg), dim3(512), rlds, \
                           st, (const _Float16 *)d_hqc, (const float *)d_thetac, c, (const _Float16 *)d_ht, (int)nt,  \
                           (int)t_base, rp.segs, d_cnt, d_rcand);                                       \
    } while (0)
            if (use_pp && dp == 128) FDR_LAUNCH_RANGE_PP(128);
            else if (use_pp && dp == 256) FDR_LAUNCH_RANGE_PP(256);
            else if (use_pp) FDR_LAUNCH_RANGE_PP(512);
            else if (wide) FDR_LAUNCH_RANGE(128, 8, 4);
            else if (dp == 128) FDR_LAUNCH_RANGE(128, 4, 4);
            else if (dp == 256) FDR_LAUNCH_RANGE(256, 4, 2);
            else FDR_LAUNCH_RANGE(512, 4, 2);
#undef FDR_LAUNCH_RANGE
#undef FDR_LAUNCH_RANGE_PP
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(knn_rerank_long_kernel, dim3((unsigned)((c + 3) / 4)), dim3(256), 0, st,
                               (const int *)(d_rlist + first), c, (const int *)d_cnt, (const int *)d_rcand, k,
                               d_Qhat, d_That, dp, (int)t_base, d_idx, d_dist, d_counter, d_flagged, d_path);
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
    return knn_plan(ctx->num_cus, nq, nt, d, k).total_bytes;
}

static int launch_knn_mode(fdr_ctx *ctx, const float *d_Qhat, const uint8_t *d_qzero, int64_t nq,
                           const float *d_That, const uint8_t *d_tzero, int64_t nt, int64_t t_base, int d,
                           int k, int32_t *d_idx, float *d_dist, void *d_ws, size_t ws_bytes, hipStream_t st) {
    const int dp = fdr_padded_dim(d);
    if (dp > 0 && k >= 1 && k <= FDR_FAST_MAX_K && nq > 0 && nt >= k && knn_prefilter_wanted(ctx, dp, nt, k) &&
        d_Qhat && d_qzero && d_That && d_tzero && d_idx && d_dist && d_ws)
        return launch_knn_prefilter(ctx, d_Qhat, d_qzero, nq, d_That, d_tzero, nt, t_base, d, k, d_idx,
                                    d_dist, d_ws, ws_bytes, st);
    ctx->last_flagged = 0;  // (exact mode certifies nothing)
    ctx->last_pass_launches = ctx->last_pass_queues = 0;
    ctx->paths.dev = nullptr;
    ctx->paths.n = nq;
    ctx->paths.all = FDR_PATH_EXACT;
    return launch_knn_exact(ctx, d_Qhat, d_qzero, nq, d_That, d_tzero, nt, t_base, d, k, d_idx, d_dist,
                            d_ws, ws_bytes, st);
}

// ---- duplicate-row classes: search unique queries x unique targets, expand --------------------
struct DedupLayout {
    size_t inner_bytes;  // workspace of the inner k-NN call (sized for the un-deduplicated problem)
    size_t off_hash, off_hash_s, off_idx, off_idx_s, off_flag, off_cid, off_cls, off_cstart, off_isrep,
        off_upos, off_uofc, off_cofu, off_uqflag, off_uqpos, off_U, off_uzero, off_Uq, off_uqz, off_idxu,
        off_distu, off_rowpath, off_tmp, tmp_bytes, total;
};

#define FDR_DEDUP_PROBE_BELOW (1 << 18)
static bool knn_dedup_wanted(const fdr_ctx *ctx, int64_t nq, int64_t nt) {
    if (ctx->dedup_mode != FDR_DEDUP_AUTO) return ctx->dedup_mode != FDR_DEDUP_OFF;
    return nt >= 8192 && nq >= 1024;  // (from the size at which the prefilter mode engages)
}

static DedupLayout dedup_layout(const fdr_ctx *ctx, int64_t nq, int64_t nt, int d, int k) {
    DedupLayout L;
    const int dp = fdr_padded_dim(d);
    L.inner_bytes = align256(knn_mode_workspace_bytes(ctx, nq, nt, d, k));
    size_t o = L.inner_bytes;
    auto take = [&](size_t bytes) { const size_t at = o; o += align256(bytes); return at; };
    // (one block: the sorted hashes follow the unsorted ones, and the (representative, size, start) table of
    // expand_classes_kernel, 16 bytes per unique row, takes the place of both once the classes are marked)
    L.off_hash = take(align256((size_t)nt * 8) + (size_t)nt * 8);
    L.off_hash_s = L.off_hash + align256((size_t)nt * 8);
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
    L.off_rowpath = take((size_t)nq);  // path codes of the original rows (fdr_last_query_paths)
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

// queries (waves) per workgroup of expand_classes_kernel: four while their K * K keys stay within 32 KiB of LDS
static int expand_waves_per_block(int k) { return (size_t)4 * k * k * 8 <= 32768 ? 4 : (size_t)2 * k * k * 8 <= 32768 ? 2 : 1; }

static int launch_knn(fdr_ctx *ctx, const float *d_Qhat, const uint8_t *d_qzero, int64_t nq,
                      const float *d_That, const uint8_t *d_tzero, int64_t nt, int64_t t_base, int d,
                      int k, int32_t *d_idx, float *d_dist, void *d_ws, size_t ws_bytes, hipStream_t st) {
    const int dp = fdr_padded_dim(d);
    if (dp > 0 && k >= 1 && k <= FDR_MAX_K && knn_generic_wanted(dp, k)) {  // beyond the MFMA kernels' shapes
        if (nq < 0 || nt < k) return fail(FDR_E_ARG, "knn: need n_targets (%lld) >= k (%d)", (long long)nt, k);
        if (nt + t_base > 0x7fffffffll || nq > 0x7fffffffll) return fail(FDR_E_ARG, "knn: row numbers exceed int32");
        if (nq == 0) return FDR_OK;
        if (!d_Qhat || !d_qzero || !d_That || !d_tzero || !d_idx || !d_dist) return fail(FDR_E_ARG, "knn: null device pointer");
        ctx->last_unique_targets = (int)nt;
        ctx->last_unique_queries = (int)nq;
        ctx->last_flagged = 0;
        ctx->last_pass_launches = ctx->last_pass_queues = 0;
        ctx->paths.dev = nullptr;
        ctx->paths.n = nq;
        ctx->paths.all = FDR_PATH_GENERIC;
        int trc = timing_begin(ctx, FDR_KERNEL_KNN_TILE, st);
        if (trc) return trc;
        hipLaunchKernelGGL(knn_generic_kernel, dim3((unsigned)((nq + GEN_QPB - 1) / GEN_QPB)), dim3(256),
                           (size_t)GEN_QPB * dp * 4, st, d_Qhat, d_qzero, (int)nq, d_That, d_tzero, (int)nt, (int)t_base, dp, k,
                           d_idx, d_dist);
        HIP_TRY(hipGetLastError());
        return timing_end(ctx, FDR_KERNEL_KNN_TILE, st);
    }
    // the queries must be a block of the target rows (they are in every caller of this library)
    const bool q_in_t = d_Qhat && d_That && dp > 0 && d_Qhat >= d_That &&
                        d_Qhat + (size_t)nq * dp <= d_That + (size_t)nt * dp &&
                        ((d_Qhat - d_That) % dp) == 0;
    if (!(dp > 0 && k >= 1 && k <= FDR_FAST_MAX_K && nq > 0 && nt >= k && q_in_t && knn_dedup_wanted(ctx, nq, nt) &&
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
    const bool always = ctx->dedup_mode == FDR_DEDUP_FORCE;  // expand even without duplicates (tests)
    // From 2^18 targets the sort and the tables (< 1 ms at 1 M rows) are noise next to the search (~ n^2): no probe,
    // no read-back for it; the decision falls on the exact unique counts below.
    if (!always && nt < FDR_DEDUP_PROBE_BELOW) {
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
    // (rep_m takes the hashes' place: nothing reads them once the classes are marked)
    hipLaunchKernelGGL(unique_tables_kernel, dim3(g1), dim3(256), 0, st, n, (const int *)isrep,
                       (const int *)upos, (const int *)cls, (const int *)cstart, uofc, cofu, (int4 *)hash);
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
    if (nuq != nu)  // (else the unique queries are the unique rows: no copy, see below)
        hipLaunchKernelGGL(gather_unique_queries_kernel, dim3((unsigned)(((size_t)nu * 16 + 255) / 256)), dim3(256),
                           0, st, (const float *)U, (const unsigned char *)uzero, nu, dp, (const int *)uqflag,
                           (const int *)uqpos, Uq, uqz);
    HIP_TRY(hipGetLastError());
    if ((trc = timing_end(ctx, FDR_KERNEL_KNN_DEDUP, st))) return trc;
    // unique rows are stored in ascending representative order, and the unique queries are a subsequence
    // of them; the inner search numbers targets 0..nu-1
    // (every unique row is a query: Uq would be a copy of U -- the same pointers let the prefilter mode see that
    // the queries are the targets)
    const bool all_q = nuq == nu;
    int rc = launch_knn_mode(ctx, all_q ? U : Uq, all_q ? uzero : uqz, nuq, U, uzero, nu, 0, d, k, idx_u, dist_u, d_ws,
                             L.inner_bytes, st);
    if (rc) return rc;
    if ((trc = timing_begin(ctx, FDR_KERNEL_KNN_DEDUP, st))) return trc;
    const int xw = expand_waves_per_block(k), xq = xw * (64 / k) * EXPAND_UNROLL;  // (k <= FDR_FAST_MAX_K = 64 here)
    hipLaunchKernelGGL(expand_classes_kernel, dim3((unsigned)((nq + xq - 1) / xq)), dim3(64 * xw), (size_t)xw * k * k * 8, st, q0, (int)nq, k, 64 / k,
                       (int)t_base, (const int *)cls, (const int *)uofc, (const int *)uqpos, (const int *)idx_u,
                       (const float *)dist_u, (const int *)idx_s, (const int4 *)hash, d_idx, d_dist, k,
                       ctx->paths.n == nuq ? ctx->paths.dev : nullptr, ctx->paths.all, (uint8_t *)(ws + L.off_rowpath));
    HIP_TRY(hipGetLastError());
    ctx->paths.dev = (const uint8_t *)(ws + L.off_rowpath);  // (the unique rows' codes, carried to the rows of their classes)
    ctx->paths.n = nq;
    ctx->paths.stream = st;
    return timing_end(ctx, FDR_KERNEL_KNN_DEDUP, st);
}

static size_t knn_workspace_bytes_impl(const fdr_ctx *ctx, int64_t nq, int64_t nt, int d, int k) {
    if (knn_dedup_wanted(ctx, nq, nt)) return dedup_layout(ctx, nq, nt, d, k).total;
    return knn_mode_workspace_bytes(ctx, nq, nt, d, k);
}

// ---- duplicate-row classes across ranks ---------------------------------------------------------
// A row-sharded run searches a duplicate QUERY row once per rank that holds a member of its class (8 ranks
// at 1 M reads: 7816 query blocks instead of 6494).  These three calls let the ranks split the UNIQUE rows
// instead: every rank builds the classes of the (all-gathered) target set -- the same tables on every rank --
// searches its share of the unique rows, the shares are exchanged (nu x k indices and distances), and every
// rank expands its own rows from the complete result.
FDR_EXPORT int fdr_knn_classes_dev(fdr_ctx *ctx, const float *d_That, const uint8_t *d_tzero, int64_t nt, int32_t d,
                                   int32_t k, int64_t nq_max, void *d_ws, size_t ws_bytes, void *stream,
                                   int32_t *n_unique_out) {
    int rc = use_device(ctx);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    ctx->cls.valid = false;
    if (!n_unique_out) return fail(FDR_E_ARG, "knn_classes: n_unique_out is null");
    *n_unique_out = 0;
    const int dp = fdr_padded_dim(d);
    if (dp < 0 || k < 1 || k > FDR_MAX_K || nt < k || nq_max <= 0 || nq_max > nt || !d_That || !d_tzero || !d_ws ||
        nt > 0x7fffffffll)
        return fail(FDR_E_ARG, "knn_classes: bad argument");
    if (knn_generic_wanted(dp, k)) return FDR_OK;  // (no classes beyond the MFMA kernels' shapes: the callers use fdr_knn_dev)
    if (!knn_dedup_wanted(ctx, nq_max, nt)) return FDR_OK;  // (small sets: the callers use fdr_knn_dev)
    const DedupLayout L = dedup_layout(ctx, nq_max, nt, d, k);
    if (ws_bytes < L.total) return fail(FDR_E_ARG, "knn_classes: workspace %zu < required %zu bytes", ws_bytes, L.total);
    char *ws = static_cast<char *>(d_ws);
    u64 *hash = (u64 *)(ws + L.off_hash), *hash_s = (u64 *)(ws + L.off_hash_s);
    int *idx = (int *)(ws + L.off_idx), *idx_s = (int *)(ws + L.off_idx_s), *flag = (int *)(ws + L.off_flag);
    int *cid = (int *)(ws + L.off_cid), *cls = (int *)(ws + L.off_cls), *cstart = (int *)(ws + L.off_cstart);
    int *isrep = (int *)(ws + L.off_isrep), *upos = (int *)(ws + L.off_upos), *uofc = (int *)(ws + L.off_uofc);
    int *cofu = (int *)(ws + L.off_cofu);
    float *U = (float *)(ws + L.off_U);
    uint8_t *uzero = (uint8_t *)(ws + L.off_uzero);
    void *tmp = ws + L.off_tmp;
    const int n = (int)nt;
    const unsigned g16 = (unsigned)(((size_t)n * 16 + 255) / 256), g1 = (unsigned)((n + 255) / 256);
    int trc = timing_begin(ctx, FDR_KERNEL_KNN_DEDUP, st);
    if (trc) return trc;
    hipLaunchKernelGGL(hash_rows_kernel, dim3(g16), dim3(256), 0, st, d_That, n, dp, hash, idx);
    HIP_TRY(hipGetLastError());
    // No hash-table probe here (launch_knn's shortcut for small sets): its count depends on the order in which
    // the table fills, and every rank of a row-sharded run must take the SAME decision from the same gathered
    // rows or their collectives no longer match.  The exact unique count after the sort is deterministic.
    size_t tb = L.tmp_bytes;
    HIP_TRY(rocprim::radix_sort_pairs(tmp, tb, hash, hash_s, idx, idx_s, (size_t)n, 0, 64, st));
    hipLaunchKernelGGL(mark_class_starts_kernel, dim3(g1), dim3(256), 0, st, d_That, n, dp, (const u64 *)hash_s,
                       (const int *)idx_s, flag);
    tb = L.tmp_bytes;
    HIP_TRY(rocprim::inclusive_scan(tmp, tb, flag, cid, (size_t)n, rocprim::plus<int>(), st));
    hipLaunchKernelGGL(class_tables_kernel, dim3(g1), dim3(256), 0, st, n, (const int *)flag, (const int *)cid,
                       (const int *)idx_s, cls, cstart, isrep);
    tb = L.tmp_bytes;
    HIP_TRY(rocprim::inclusive_scan(tmp, tb, isrep, upos, (size_t)n, rocprim::plus<int>(), st));
    hipLaunchKernelGGL(unique_tables_kernel, dim3(g1), dim3(256), 0, st, n, (const int *)isrep, (const int *)upos,
                       (const int *)cls, (const int *)cstart, uofc, cofu, (int4 *)hash);  // (rep_m over the hashes)
    HIP_TRY(hipGetLastError());
    int nu = 0;
    HIP_TRY(hipMemcpyAsync(&nu, cid + (n - 1), 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    // (fewer unique rows than neighbours asked for; or too few repeats to pay: the rule of launch_knn, on exact counts)
    if (nu < k || (ctx->dedup_mode != FDR_DEDUP_FORCE && (double)nu * (double)nu > 0.9 * (double)nt * (double)nt))
        return timing_end(ctx, FDR_KERNEL_KNN_DEDUP, st);
    hipLaunchKernelGGL(gather_unique_rows_kernel, dim3(g16), dim3(256), 0, st, d_That, d_tzero, n, dp,
                       (const int *)isrep, (const int *)upos, U, uzero);
    HIP_TRY(hipGetLastError());
    if ((trc = timing_end(ctx, FDR_KERNEL_KNN_DEDUP, st))) return trc;
    ctx->cls.valid = true;
    ctx->cls.That = d_That;
    ctx->cls.tzero = d_tzero;
    ctx->cls.nt = nt;
    ctx->cls.nq_max = nq_max;
    ctx->cls.d = d;
    ctx->cls.k = k;
    ctx->cls.nu = nu;
    ctx->cls.ws = d_ws;
    ctx->cls.ws_bytes = ws_bytes;
    ctx->last_unique_targets = nu;
    *n_unique_out = nu;
    return FDR_OK;
}

FDR_EXPORT int fdr_knn_unique_dev(fdr_ctx *ctx, int64_t u_lo, int64_t u_hi, int32_t *d_idx_u, float *d_dist_u,
                                  void *stream) {
    int rc = use_device(ctx);
    if (rc) return rc;
    if (!ctx->cls.valid) return fail(FDR_E_STATE, "knn_unique: no classes (call fdr_knn_classes_dev first)");
    if (u_lo < 0 || u_hi < u_lo || u_hi > ctx->cls.nu || u_hi - u_lo > ctx->cls.nq_max)
        return fail(FDR_E_ARG, "knn_unique: bad range [%lld, %lld) of %d unique rows (at most %lld per call)",
                    (long long)u_lo, (long long)u_hi, ctx->cls.nu, (long long)ctx->cls.nq_max);
    if (u_hi == u_lo) return FDR_OK;
    if (!d_idx_u || !d_dist_u) return fail(FDR_E_ARG, "knn_unique: null output");
    const int dp = fdr_padded_dim(ctx->cls.d);
    const DedupLayout L = dedup_layout(ctx, ctx->cls.nq_max, ctx->cls.nt, ctx->cls.d, ctx->cls.k);
    char *ws = static_cast<char *>(ctx->cls.ws);
    const float *U = (const float *)(ws + L.off_U);
    const uint8_t *uzero = (const uint8_t *)(ws + L.off_uzero);
    ctx->last_unique_queries = (int)(u_hi - u_lo);
    struct NoPaths {  // (fdr_last_query_paths covers whole calls; a share of the unique rows is not one)
        fdr_ctx *c;
        ~NoPaths() { c->paths.n = 0; }
    } no_paths{ctx};
    // the unique rows are stored in ascending representative order; a share of them is a block of U
    return launch_knn_mode(ctx, U + (size_t)u_lo * dp, uzero + u_lo, u_hi - u_lo, U, uzero, ctx->cls.nu, 0, ctx->cls.d,
                           ctx->cls.k, d_idx_u, d_dist_u, ctx->cls.ws, L.inner_bytes, (hipStream_t)stream);
}

FDR_EXPORT int fdr_knn_expand_dev(fdr_ctx *ctx, int64_t q0, int64_t nq, int64_t t_base, const int32_t *d_idx_u_all,
                                  const float *d_dist_u_all, int64_t u_row_stride, int32_t *d_idx, float *d_dist,
                                  void *stream) {
    int rc = use_device(ctx);
    if (rc) return rc;
    if (!ctx->cls.valid) return fail(FDR_E_STATE, "knn_expand: no classes (call fdr_knn_classes_dev first)");
    if (q0 < 0 || nq < 0 || q0 + nq > ctx->cls.nt) return fail(FDR_E_ARG, "knn_expand: bad row range");
    if (nq == 0) return FDR_OK;
    if (!d_idx_u_all || !d_dist_u_all || !d_idx || !d_dist) return fail(FDR_E_ARG, "knn_expand: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int k = ctx->cls.k;
    if (u_row_stride == 0) u_row_stride = k;
    if (u_row_stride < k || u_row_stride > 0x7fffffffll) return fail(FDR_E_ARG, "knn_expand: bad row stride");
    const DedupLayout L = dedup_layout(ctx, ctx->cls.nq_max, ctx->cls.nt, ctx->cls.d, k);
    char *ws = static_cast<char *>(ctx->cls.ws);
    int trc = timing_begin(ctx, FDR_KERNEL_KNN_DEDUP, st);
    if (trc) return trc;
    const int xw = expand_waves_per_block(k), xq = xw * (64 / k) * EXPAND_UNROLL;
    hipLaunchKernelGGL(expand_classes_kernel, dim3((unsigned)((nq + xq - 1) / xq)), dim3(64 * xw), (size_t)xw * k * k * 8, st, (int)q0, (int)nq, k,
                       64 / k, (int)t_base, (const int *)(ws + L.off_cls), (const int *)(ws + L.off_uofc), (const int *)nullptr,
                       (const int *)d_idx_u_all, d_dist_u_all, (const int *)(ws + L.off_idx_s),
                       (const int4 *)(ws + L.off_hash), d_idx, d_dist, (int)u_row_stride, (const uint8_t *)nullptr,
                       (uint8_t)0, (uint8_t *)nullptr);
    HIP_TRY(hipGetLastError());
    ctx->paths.n = 0;  // (the unique rows were searched by several ranks: no codes)
    return timing_end(ctx, FDR_KERNEL_KNN_DEDUP, st);
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
    if (d_workspace == ctx->cls.ws) ctx->cls.valid = false;  // (this call overwrites the tables fdr_knn_classes_dev left there)
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

// Host CSR -> E (device, [n_rows, d]) on ctx->stream: see host_upload.inc.  Small inputs take the plain path.
static int upload_embed_pipelined(fdr_ctx *ctx, int64_t n_rows, const int64_t *a_indptr, const int32_t *a_indices,
                                  float *d_E) {
    int rc;
    const int64_t nnz = a_indptr[n_rows];
    const int d = ctx->d;
    const int64_t min_ids = 1 << 20;  // below 4 MB of ids there is nothing to overlap
    const int nthr = std::max(1, std::min(host_cpu_budget() - 1, 31));  // (the calling thread drives the link)
    if (nnz < min_ids || ctx->h_bits.empty()) {
        if ((rc = upload_csr(ctx, n_rows, a_indptr, a_indices))) return rc;
        return launch_embed(ctx, n_rows, (const int64_t *)ctx->a_indptr.p, (const int32_t *)ctx->a_indices.p, d_E, ctx->stream);
    }
    // chunks of ~T ids, cut at row boundaries by bisection of the (monotone) row pointers
    const int64_t T = 1 << 18;  // (1 MB of ids: ~0.25 ms for a helper, 18 us on the link; helpers' chunks cost no launch each)
    const int RUN = 8;          // chunks the link takes at once
    std::vector<hup::Chunk> chunks;
    for (int64_t r = 0; r < n_rows;) {
        const int64_t want = a_indptr[r] + T;
        int64_t r1 = std::upper_bound(a_indptr + r + 1, a_indptr + n_rows + 1, want) - a_indptr;  // first row END beyond `want`
        r1 = std::min(n_rows, std::max(r + 1, r1 - 1 > r ? r1 - 1 : r + 1));
        hup::Chunk c;
        c.r0 = r;
        c.r1 = r1;
        chunks.push_back(c);
        r = r1;
    }
    const int64_t nch = (int64_t)chunks.size();
    for (const hup::Chunk &c : chunks)
        if (a_indptr[c.r0] < 0 || a_indptr[c.r1] < a_indptr[c.r0] || a_indptr[c.r1] > nnz)
            return fail(FDR_E_ARG, "embed: indptr not monotone near row %lld", (long long)c.r0);
    const int64_t stage_cap = std::max<int64_t>(1 << 20, nnz / 6);  // ids of pinned staging (P keeps ~5 % of them)
    if ((rc = ctx->a_indptr.reserve((size_t)(n_rows + 1) * 8))) return rc;
    if ((rc = ctx->a_indices.reserve((size_t)nnz * 4))) return rc;
    if ((rc = ctx->c_indptr.reserve((size_t)(n_rows + nch + 1) * 8))) return rc;
    if ((rc = ctx->c_indices.reserve((size_t)stage_cap * 4))) return rc;
    if ((rc = ctx->stage_ids.reserve((size_t)stage_cap * 4))) return rc;
    if ((rc = ctx->stage_ptr.reserve((size_t)(n_rows + nch + 1) * 8))) return rc;
    for (hipEvent_t &e : ctx->up_ev)
        if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    int32_t *stage_ids = (int32_t *)ctx->stage_ids.p;
    int64_t *stage_ptr = (int64_t *)ctx->stage_ptr.p;
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(ctx->a_indptr.p, a_indptr, (size_t)(n_rows + 1) * 8, hipMemcpyHostToDevice, st));

    // the two fronts: chunks [front, back) are unclaimed
    std::atomic<uint64_t> ends{(uint64_t)nch};  // front << 32 | back
    auto claim = [&](bool from_back, int want, int &got) -> int64_t {  // up to `want` chunks from one end; returns the first
        uint64_t v = ends.load();
        for (;;) {
            const uint64_t f = v >> 32, b = v & 0xffffffffull;
            if (f >= b) return -1;
            const uint64_t m = std::min<uint64_t>((uint64_t)want, b - f);
            const uint64_t nv = from_back ? (f << 32 | (b - m)) : ((f + m) << 32 | b);
            if (ends.compare_exchange_weak(v, nv)) {
                got = (int)m;
                return (int64_t)(from_back ? b - m : f);
            }
        }
    };
    // The helpers' chunks end up in ROW ORDER at the END of the staging buffer, without gaps -- chunk ci right below
    // chunk ci + 1 -- so that what they produced is one copy of ids, one of row pointers and ONE embed launch, however
    // small the chunks.  A helper that has compacted chunk ci (into a buffer of its own) waits for cum[ci + 1], the ids
    // of all chunks above it (a chained scan: that chunk was claimed just before this one and takes as long), publishes
    // cum[ci] and copies its ids to stage_ids + stage_cap - cum[ci]; its rows' pointers are absolute positions there.
    std::vector<std::atomic<int64_t>> cum((size_t)nch + 1);
    for (auto &c : cum) c.store(-1, std::memory_order_relaxed);
    cum[(size_t)nch].store(0);
    std::atomic<bool> overflow{false};
    std::atomic<int64_t> bad_row{-1};
    const uint32_t *bw = ctx->h_bits.data();
    const uint64_t F = (uint64_t)ctx->n_features;
    auto worker = [&]() {
        std::vector<int32_t> scratch;
        std::vector<int64_t> lptr;
        for (;;) {
            int got = 0;
            const int64_t ci = claim(true, 1, got);
            if (ci < 0) break;
            hup::Chunk &c = chunks[(size_t)ci];
            const int64_t raw = a_indptr[c.r1] - a_indptr[c.r0], rows = c.r1 - c.r0;
            int64_t n = 0;
            bool ok = !overflow.load() && bad_row.load() < 0;
            for (int64_t r = c.r0; ok && r < c.r1; ++r)
                if (a_indptr[r + 1] < a_indptr[r]) {
                    int64_t exp = -1;
                    bad_row.compare_exchange_strong(exp, r);
                    ok = false;
                }
            if (ok) {
                if ((int64_t)scratch.size() < raw + 16) scratch.resize((size_t)raw + 16);  // (+ 16: the vector loop stores whole registers)
                if ((int64_t)lptr.size() < rows + 1) lptr.resize((size_t)rows + 1);
                n = csrc::compact_chunk(bw, F, a_indptr, a_indices, c.r0, c.r1, scratch.data(), raw, lptr.data());
            }
            int64_t above;  // (every claimed chunk publishes, whatever happened to it: the chain must not break)
            while ((above = cum[(size_t)ci + 1].load(std::memory_order_acquire)) < 0) std::this_thread::yield();
            const int64_t mine = above + (ok ? n : 0);
            if (ok && mine > stage_cap) {  // (P keeps far more ids than expected: the helpers' chunks go raw after all)
                overflow.store(true);
                ok = false;
            }
            cum[(size_t)ci].store(ok ? mine : above, std::memory_order_release);
            if (!ok) {
                c.staged_off = -2;
                continue;
            }
            const int64_t base = stage_cap - mine;
            memcpy(stage_ids + base, scratch.data(), (size_t)n * 4);
            for (int64_t r = 0; r < rows; ++r) stage_ptr[c.r0 + r] = base + lptr[(size_t)r];
            c.staged_off = base;
            c.staged_n = n;
        }
    };
    ctx->up_pool.ensure(nthr);  // (fewer helpers than hoped: the link carries more)
    ctx->up_pool.start(worker);
    auto send_raw = [&](int64_t r0, int64_t r1, int slot) -> int {  // rows [r0, r1) as they are, the embed kernel behind them
        const int64_t o = a_indptr[r0], len = a_indptr[r1] - o;
        if (len > 0)
            HIP_TRY(hipMemcpyAsync((int32_t *)ctx->a_indices.p + o, a_indices + o, (size_t)len * 4, hipMemcpyHostToDevice, st));
        int erc = launch_embed(ctx, r1 - r0, (const int64_t *)ctx->a_indptr.p + r0, (const int32_t *)ctx->a_indices.p,
                               d_E + (size_t)r0 * d, st);
        if (erc) return erc;
        if (slot >= 0) HIP_TRY(hipEventRecord(ctx->up_ev[slot], st));
        return FDR_OK;
    };
    int urc = FDR_OK;
    int64_t sent = 0;
    for (; urc == FDR_OK; ++sent) {
        if (sent >= 2) {  // two runs in flight: the next is claimed when the link has taken the one before the last
            hipError_t e = hipEventSynchronize(ctx->up_ev[sent & 1]);
            if (e != hipSuccess) {
                urc = fail(FDR_E_HIP, "hipEventSynchronize failed: %s", hipGetErrorString(e));
                break;
            }
        }
        int got = 0;
        const int64_t ci = claim(false, RUN, got);
        if (ci < 0) break;
        urc = send_raw(chunks[(size_t)ci].r0, chunks[(size_t)(ci + got - 1)].r1, (int)(sent & 1));
    }
    ctx->up_pool.wait();
    if (urc) return urc;
    if (bad_row.load() >= 0) return fail(FDR_E_ARG, "embed: indptr not monotone at row %lld", (long long)bad_row.load());
    // what the helpers left: chunks [first, nch), in row order at the end of the staging buffer -- two copies and one
    // launch; if the staging buffer overflowed (a dense P) or a chunk was dropped, their rows go raw after all
    const int64_t first = (int64_t)(ends.load() & 0xffffffffull);  // (the helpers claimed downwards from nch)
    if (first < nch) {
        bool all_staged = !overflow.load();
        for (int64_t ci = first; ci < nch && all_staged; ++ci) all_staged = chunks[(size_t)ci].staged_off >= 0;
        const int64_t rb = chunks[(size_t)first].r0;
        if (!all_staged) {
            if ((rc = send_raw(rb, n_rows, -1))) return rc;
        } else {
            const int64_t used = cum[(size_t)first].load();
            stage_ptr[n_rows] = stage_cap;
            if (used > 0)
                HIP_TRY(hipMemcpyAsync((int32_t *)ctx->c_indices.p + (stage_cap - used), stage_ids + (stage_cap - used),
                                       (size_t)used * 4, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync((int64_t *)ctx->c_indptr.p + rb, stage_ptr + rb, (size_t)(n_rows - rb + 1) * 8,
                                   hipMemcpyHostToDevice, st));
            if ((rc = launch_embed(ctx, n_rows - rb, (const int64_t *)ctx->c_indptr.p + rb, (const int32_t *)ctx->c_indices.p,
                                   d_E + (size_t)rb * d, st)))
                return rc;
        }
    }
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
    const size_t ebytes = (size_t)n_rows * ctx->d * 4;
    if ((rc = ctx->E.reserve(ebytes))) return rc;
    if ((rc = upload_embed_pipelined(ctx, n_rows, a_indptr, a_indices, (float *)ctx->E.p))) return rc;
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
    const size_t ebytes = (size_t)n_rows * ctx->d * 4;
    if ((rc = ctx->E.reserve(ebytes))) return rc;
    if ((rc = upload_embed_pipelined(ctx, n_rows, a_indptr, a_indices, (float *)ctx->E.p))) return rc;
    if (E_out) HIP_TRY(hipMemcpyAsync(E_out, ctx->E.p, ebytes, hipMemcpyDeviceToHost, ctx->stream));
    return knn_from_device_E(ctx, (const float *)ctx->E.p, n_rows, ctx->d, k, idx_out, dist_out);
}

#include "kmer_search.inc"
#include "kmer_output_loader.inc"
#include "reads_parser.inc"
#include "overlaps_writer.inc"
