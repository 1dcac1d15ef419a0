// Development micro-benchmark (not part of the library): what the candidate pass's inner loop can sustain with
// v_mfma_f32_32x32x16_f16 against v_mfma_f32_16x16x32_f16 -- same flops, same LDS fragment reads (8 x ds_read_b128 per
// 32 targets x 32 queries x 128 components), same 16 accumulators and the same max tree per tile, four waves per
// SIMD, run long enough (hundreds of ms) for the clocks to settle under the power limit.  No staging from global
// memory (the LDS tile is re-read), no lists: the upper bound of a re-cut kernel, shape against shape.
//   hipcc --offload-arch=gfx950 -O3 devtools/mfma_shape_bench.hip -o gpurun_out/mfma_shape_bench && ./gpurun_out/mfma_shape_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// operand values like the embeddings' (DENSITY of the components non-zero, magnitudes of a few tenths) or dense
// random ones: the matrix pipe's power, and with it the clock, depends on the data
__device__ __host__ inline float sparse_value(unsigned h) {
#ifdef DENSE_DATA
    return ((float)(h >> 8 & 0xffff) / 65536.0f - 0.5f);
#else
    return (h >> 24) < 12 ? ((float)(h >> 8 & 0xffff) / 65536.0f - 0.5f) : 0.0f;  // ~4.7 % non-zero
#endif
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// 32x32x16: per tile 8 fragments (k-steps) of 32 rows; two tiles = two chains (the shipped kernel's paired loop)
__global__ __launch_bounds__(256, 4) void shape32(const _Float16 *__restrict__ q, int tiles, int *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) f16x8 lds[2 * 8 * 64];  // two tiles x 8 k-steps x 64 lanes
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2 * 8 * 64; i += 256) {
        f16x8 v;
        for (int c = 0; c < 8; ++c) v[c] = (_Float16)(sparse_value((unsigned)(i * 8 + c) * 2654435761u + blockIdx.x));
        lds[i] = v;
    }
    f16x8 b[8];
    for (int s = 0; s < 8; ++s) b[s] = reinterpret_cast<const f16x8 *>(q)[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 512 + s * 64 + lane];
    __syncthreads();
    int best = 0;
    for (int t = 0; t < tiles; t += 2) {
        f32x16 a0 = {}, a1 = {};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds[s * 64 + lane], b[s], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds[512 + s * 64 + lane], b[s], a1, 0, 0, 0);
        }
        int m0 = 0, m1 = 0;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            m0 = max(max(m0, __float_as_int(a0[r])), __float_as_int(a0[r + 1]));
            m1 = max(max(m1, __float_as_int(a1[r])), __float_as_int(a1[r + 1]));
        }
        best = max(best, max(m0, m1));
        if (__any(best == 0x7fffffff)) break;  // (never: keeps the tree and a vector -> scalar test in the loop)
    }
    if (best == 12345) out[0] = best;
}

// 32x32x16 + NWR ds_write_b128 per wave and two tiles into a scratch area: what LDS write traffic of the size of the
// staging's (4 per wave and two tiles) costs a loop whose fragment reads already take 128 B per clock and CU
template <int NWR>
__global__ __launch_bounds__(256, 4) void shape32_writes(const _Float16 *__restrict__ q, int tiles, int *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) f16x8 lds[2 * 8 * 64 + 4 * 4 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * 8 * 64; i += 256) {
        f16x8 v;
        for (int c = 0; c < 8; ++c) v[c] = (_Float16)(sparse_value((unsigned)(i * 8 + c) * 2654435761u + blockIdx.x));
        lds[i] = v;
    }
    f16x8 b[8];
    for (int s = 0; s < 8; ++s) b[s] = reinterpret_cast<const f16x8 *>(q)[(blockIdx.x * 4 + wave) * 512 + s * 64 + lane];
    __syncthreads();
    int best = 0;
    f16x8 *scratch = lds + 1024 + wave * 256;
    for (int t = 0; t < tiles; t += 2) {
#pragma unroll
        for (int w = 0; w < NWR; ++w) scratch[w * 64 + lane] = b[w];
        f32x16 a0 = {}, a1 = {};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds[s * 64 + lane], b[s], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds[512 + s * 64 + lane], b[s], a1, 0, 0, 0);
        }
        int m0 = 0, m1 = 0;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            m0 = max(max(m0, __float_as_int(a0[r])), __float_as_int(a0[r + 1]));
            m1 = max(max(m1, __float_as_int(a1[r])), __float_as_int(a1[r + 1]));
        }
        best = max(best, max(m0, m1));
        if (__any(best == 0x7fffffff)) break;
    }
    if (best == 12345) out[0] = best + (int)scratch[lane][0];
}

// 32x32x16 + NV extra vector instructions and NS extra scalar instructions per tile (the shipped kernel issues ~44
// vector and ~12 scalar instructions per wave and tile beside its 8 MFMAs)
template <int NV, int NS>
__global__ __launch_bounds__(256, 4) void shape32_valu(const _Float16 *__restrict__ q, int tiles, int *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) f16x8 lds[2 * 8 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * 8 * 64; i += 256) {
        f16x8 v;
        for (int c = 0; c < 8; ++c) v[c] = (_Float16)(sparse_value((unsigned)(i * 8 + c) * 2654435761u + blockIdx.x));
        lds[i] = v;
    }
    f16x8 b[8];
    for (int s = 0; s < 8; ++s) b[s] = reinterpret_cast<const f16x8 *>(q)[(blockIdx.x * 4 + wave) * 512 + s * 64 + lane];
    __syncthreads();
    int best = 0;
    unsigned x0 = lane, x1 = lane * 3u, x2 = lane * 5u, x3 = lane * 7u;  // four independent chains
    unsigned s0 = blockIdx.x;
    for (int t = 0; t < tiles; t += 2) {
        f32x16 a0 = {}, a1 = {};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds[s * 64 + lane], b[s], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds[512 + s * 64 + lane], b[s], a1, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < NV / 4; ++e) {  // 2 tiles x NV / 8 steps / 4 chains... = NV per tile in total
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x0) : "v"(x1));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(x1) : "v"(x2));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x2) : "v"(x3));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(x3) : "v"(x0));
            }
#pragma unroll
            for (int e = 0; e < NS / 4; ++e) asm volatile("s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 5\n s_add_u32 %0, %0, 3\n s_xor_b32 %0, %0, 9" : "+s"(s0) : : "scc");
        }
        int m0 = 0, m1 = 0;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            m0 = max(max(m0, __float_as_int(a0[r])), __float_as_int(a0[r + 1]));
            m1 = max(max(m1, __float_as_int(a1[r])), __float_as_int(a1[r + 1]));
        }
        best = max(best, max(m0, m1));
        if (__any(best == 0x7fffffff)) break;
    }
    if (best == 12345) out[0] = best + (int)(x0 + x1 + x2 + x3 + s0);
}

// 32x32x16 with the shipped kernel's staging: a ring of two stages of two tiles (32 KiB), every wave fetches four
// 1-KiB pieces of the NEXT stage with LDS-DMA while the current one is multiplied, one barrier per stage.
// MODE 0: barrier only (no DMA), 1: DMA + barrier, 2: DMA + barrier, stages of four tiles (64 KiB ring, 2 WG/CU)
template <int MODE, int NW = 4>
__global__ __launch_bounds__(64 * NW, 4) void shape32_staged(const _Float16 *__restrict__ q, const _Float16 *__restrict__ tg,
                                                         int tiles, int *__restrict__ out, int twrap = 0x7fffffff) {
    constexpr int TPS = MODE == 2 ? 4 : 2;  // (MODE 3: two tiles, staged through registers)  // tiles per stage
    extern __shared__ __attribute__((aligned(16))) f16x8 lds[];  // 2 stages x TPS tiles x 8 k-steps x 64 lanes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * TPS * 512; i += 64 * NW) {
        f16x8 v;
        for (int c = 0; c < 8; ++c) v[c] = (_Float16)(sparse_value((unsigned)(i * 8 + c) * 2654435761u + blockIdx.x));
        lds[i] = v;
    }
    f16x8 b[8];
    for (int s = 0; s < 8; ++s) b[s] = reinterpret_cast<const f16x8 *>(q)[((blockIdx.x * NW + wave) & 32767) * 512 + s * 64 + lane];
    __syncthreads();
    int best = 0;
    const char *src = reinterpret_cast<const char *>(tg) + lane * 16;
    for (int t = 0, par = 0; t < tiles; t += TPS, par ^= 1) {
        f16x8 stg[8 * TPS / NW];  // (MODE 3: the next stage's pieces travel through registers)
        if (MODE == 3) {
#pragma unroll
            for (int u = 0; u < 8 * TPS / NW; ++u) {
                const int piece = wave + NW * u;
                stg[u] = *reinterpret_cast<const f16x8 *>(src + ((size_t)((t + TPS) & twrap) * 8 + piece) * 1024);
            }
        } else if (MODE >= 1) {
#pragma unroll
            for (int u = 0; u < 8 * TPS / NW; ++u) {  // the next stage: TPS x 8 pieces, wave w takes pieces w, w + NW, ...
                const int piece = wave + NW * u;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(src + ((size_t)((t + TPS) & twrap) * 8 + piece) * 1024),
                    (__attribute__((address_space(3))) void *)(reinterpret_cast<char *>(lds) + ((par ^ 1) * TPS * 8 + piece) * 1024),
                    16, 0, 0);
            }
        }
#pragma unroll
        for (int pr = 0; pr < TPS; pr += 2) {
            f32x16 a0 = {}, a1 = {};
            const f16x8 *base = lds + (par * TPS + pr) * 512;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(base[s * 64 + lane], b[s], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(base[512 + s * 64 + lane], b[s], a1, 0, 0, 0);
            }
            int m0 = 0, m1 = 0;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                m0 = max(max(m0, __float_as_int(a0[r])), __float_as_int(a0[r + 1]));
                m1 = max(max(m1, __float_as_int(a1[r])), __float_as_int(a1[r + 1]));
            }
            best = max(best, max(m0, m1));
        }
        if (__any(best == 0x7fffffff)) break;
        if (MODE == 3) {
#pragma unroll
            for (int u = 0; u < 8 * TPS / NW; ++u) {
                const int piece = wave + NW * u;
                lds[((par ^ 1) * TPS * 8 + piece) * 64 + lane] = stg[u];
            }
        }
        __syncthreads();
    }
    if (best == 12345) out[0] = best;
}

// 32x32x16 with a DEEPER ring in the same 32 KiB: RING stages of TPS tiles each (RING * TPS = 4 tiles), the stage that is
// fetched is RING - 1 stages ahead of the one that is multiplied; one barrier per stage
template <int RING, int NW>
__global__ __launch_bounds__(64 * NW, 4) void shape32_ring(const _Float16 *__restrict__ q, const _Float16 *__restrict__ tg,
                                                       int tiles, int *__restrict__ out) {
    constexpr int TPS = 4 / RING;
    extern __shared__ __attribute__((aligned(16))) f16x8 lds[];  // 4 tiles x 8 k-steps x 64 lanes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 b[8];
    for (int s = 0; s < 8; ++s) b[s] = reinterpret_cast<const f16x8 *>(q)[((blockIdx.x * NW + wave) & 32767) * 512 + s * 64 + lane];
    const char *src = reinterpret_cast<const char *>(tg) + lane * 16;
    auto issue = [&](int stage) {  // stage number -> ring slot stage % RING
#pragma unroll
        for (int u = 0; u < 8 * TPS / NW; ++u) {
            const int piece = wave + NW * u;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(src + ((size_t)stage * TPS * 8 + piece) * 1024),
                (__attribute__((address_space(3))) void *)(reinterpret_cast<char *>(lds) + ((stage % RING) * TPS * 8 + piece) * 1024),
                16, 0, 0);
        }
    };
    for (int st = 0; st < RING - 1; ++st) issue(st);
    int best = 0;
    const int nstages = tiles / TPS;
    for (int st = 0; st < nstages; ++st) {
        // the stage multiplied now was issued RING - 1 stages ago: wait for all but the RING - 2 younger ones
        if (RING == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (8 / 4 / NW > 0 ? 8 / 4 / NW : 1)) : "memory");
        else if (RING == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        issue(st + RING - 1);  // (its slot was read RING stages... one stage ago at the latest: safe after the barrier)
        const f16x8 *base = lds + (st % RING) * TPS * 512;
#pragma unroll
        for (int tl = 0; tl < TPS; ++tl) {
            f32x16 a0 = {};
#pragma unroll
            for (int s = 0; s < 8; ++s) a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(base[tl * 512 + s * 64 + lane], b[s], a0, 0, 0, 0);
            int m0 = 0;
#pragma unroll
            for (int r = 0; r < 16; r += 2) m0 = max(max(m0, __float_as_int(a0[r])), __float_as_int(a0[r + 1]));
            best = max(best, m0);
        }
        if (__any(best == 0x7fffffff)) break;
    }
    if (best == 12345) out[0] = best;
}

// 16x16x32: the same 32 x 32 x 128 block as 2 row blocks x 2 column blocks x 4 k-steps = 16 MFMAs on 4 accumulators of
// 4 registers; 8 A fragments from LDS, each used for both column blocks
__global__ __launch_bounds__(256, 4) void shape16(const _Float16 *__restrict__ q, int tiles, int *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) f16x8 lds[2 * 8 * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2 * 8 * 64; i += 256) {
        f16x8 v;
        for (int c = 0; c < 8; ++c) v[c] = (_Float16)(sparse_value((unsigned)(i * 8 + c) * 2654435761u + blockIdx.x));
        lds[i] = v;
    }
    f16x8 b[2][4];
    for (int cb = 0; cb < 2; ++cb)
        for (int s = 0; s < 4; ++s)
            b[cb][s] = reinterpret_cast<const f16x8 *>(q)[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 512 + (cb * 4 + s) * 64 + lane];
    __syncthreads();
    int best = 0;
    for (int t = 0; t < tiles; t += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {  // two tiles
            f32x4 acc[2][2] = {};
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    const f16x8 a = lds[u * 512 + (rb * 4 + s) * 64 + lane];
                    acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[0][s], acc[rb][0], 0, 0, 0);
                    acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[1][s], acc[rb][1], 0, 0, 0);
                }
            int m = 0;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
                    m = max(max(m, max(__float_as_int(acc[rb][cb][0]), __float_as_int(acc[rb][cb][1]))),
                            max(__float_as_int(acc[rb][cb][2]), __float_as_int(acc[rb][cb][3])));
            best = max(best, m);
        }
        if (__any(best == 0x7fffffff)) break;
    }
    if (best == 12345) out[0] = best;
}

int main(int argc, char **argv) {
    const int first = argc > 1 ? atoi(argv[1]) : 0, reps = argc > 2 ? atoi(argv[2]) : 3;
    const int wgs = 1024 * 8;  // 8 rounds of 4 workgroups per CU
    const int tiles = 26000;   // one scan of 831 k rows
    _Float16 *q;
    int *out;
    CHECK(hipMalloc(&q, (size_t)wgs * 4 * 512 * 16));
    {
        std::vector<_Float16> hq((size_t)wgs * 4 * 512 * 8);
        for (size_t i = 0; i < hq.size(); ++i) hq[i] = (_Float16)sparse_value((unsigned)i * 2246822519u + 12345u);
        CHECK(hipMemcpy(q, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    }
    CHECK(hipMalloc(&out, 4));
    _Float16 *tg;  // "targets": (tiles + 8) x 8 KiB, streamed by every workgroup
    {
        const size_t n = (size_t)(tiles + 8) * 4096;
        std::vector<_Float16> ht(n);
        for (size_t i = 0; i < n; ++i) ht[i] = (_Float16)sparse_value((unsigned)i * 2654435761u + 77u);
        CHECK(hipMalloc(&tg, n * 2));
        CHECK(hipMemcpy(tg, ht.data(), n * 2, hipMemcpyHostToDevice));
    }
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(shape32_staged<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const double flops = (double)wgs * 4 * tiles * 32.0 * 32.0 * 128.0 * 2.0;
    for (int rep = 0; rep < reps; ++rep)
        for (int which = first; which < 22; ++which) {
            CHECK(hipEventRecord(e0));
            if (which == 0) hipLaunchKernelGGL(shape32, dim3(wgs), dim3(256), 0, 0, q, tiles, out);
            else if (which == 1) hipLaunchKernelGGL(shape16, dim3(wgs), dim3(256), 0, 0, q, tiles, out);
            else if (which == 2) hipLaunchKernelGGL(shape32_staged<0>, dim3(wgs), dim3(256), 32768, 0, q, tg, tiles, out);
            else if (which == 3) hipLaunchKernelGGL(shape32_staged<1>, dim3(wgs), dim3(256), 32768, 0, q, tg, tiles, out);
            else if (which == 4) hipLaunchKernelGGL(shape32_staged<2>, dim3(wgs), dim3(256), 65536, 0, q, tg, tiles, out);
            else if (which == 8) hipLaunchKernelGGL((shape32_staged<1, 8>), dim3(wgs / 2), dim3(512), 32768, 0, q, tg, tiles, out);
            else if (which == 9) hipLaunchKernelGGL((shape32_staged<1, 16>), dim3(wgs / 4), dim3(1024), 32768, 0, q, tg, tiles, out);
            else if (which == 10) hipLaunchKernelGGL((shape32_staged<3, 4>), dim3(wgs), dim3(256), 32768, 0, q, tg, tiles, out);
            else if (which == 11) hipLaunchKernelGGL((shape32_staged<3, 8>), dim3(wgs / 2), dim3(512), 32768, 0, q, tg, tiles, out);
            else if (which == 12) hipLaunchKernelGGL((shape32_staged<1, 4>), dim3(wgs), dim3(256), 32768, 0, q, tg, tiles, out, 255);
            else if (which == 13) hipLaunchKernelGGL((shape32_staged<1, 8>), dim3(wgs / 2), dim3(512), 32768, 0, q, tg, tiles, out, 255);
            else if (which == 14) hipLaunchKernelGGL((shape32_staged<1, 4>), dim3(wgs), dim3(256), 32768, 0, q, tg, tiles, out, 4095);
            else if (which == 15) hipLaunchKernelGGL((shape32_ring<2, 4>), dim3(wgs), dim3(256), 32768, 0, q, tg, tiles, out);
            else if (which == 16) hipLaunchKernelGGL((shape32_ring<4, 4>), dim3(wgs), dim3(256), 32768, 0, q, tg, tiles, out);
            else if (which == 17) hipLaunchKernelGGL((shape32_ring<4, 8>), dim3(wgs / 2), dim3(512), 32768, 0, q, tg, tiles, out);
            else if (which == 18) hipLaunchKernelGGL((shape32_valu<16, 0>), dim3(wgs), dim3(256), 0, 0, q, tiles, out);
            else if (which == 19) hipLaunchKernelGGL((shape32_valu<32, 0>), dim3(wgs), dim3(256), 0, 0, q, tiles, out);
            else if (which == 20) hipLaunchKernelGGL((shape32_valu<64, 0>), dim3(wgs), dim3(256), 0, 0, q, tiles, out);
            else if (which == 21) hipLaunchKernelGGL((shape32_valu<32, 32>), dim3(wgs), dim3(256), 0, 0, q, tiles, out);
            else if (which == 6) hipLaunchKernelGGL(shape32_writes<2>, dim3(wgs), dim3(256), 0, 0, q, tiles, out);
            else if (which == 7) hipLaunchKernelGGL(shape32_writes<4>, dim3(wgs), dim3(256), 0, 0, q, tiles, out);
            else {  // the shipped launch pattern: rounds of 512 workgroups dealt to two queues
                static hipStream_t st2[2] = {nullptr, nullptr};
                if (!st2[0]) { CHECK(hipStreamCreate(&st2[0])); CHECK(hipStreamCreate(&st2[1])); }
                CHECK(hipEventRecord(e0, st2[0]));
                CHECK(hipStreamWaitEvent(st2[1], e0, 0));
                for (int l = 0; l < wgs / 512; ++l)
                    hipLaunchKernelGGL(shape32_staged<1>, dim3(512), dim3(256), 32768, st2[l & 1], q + (size_t)l * 512 * 4 * 512 * 8, tg, tiles, out);
                static hipEvent_t ej = nullptr;
                if (!ej) CHECK(hipEventCreate(&ej));
                CHECK(hipEventRecord(ej, st2[1]));
                CHECK(hipStreamWaitEvent(st2[0], ej, 0));
                CHECK(hipEventRecord(e1, st2[0]));
                CHECK(hipEventSynchronize(e1));
                float ms2 = 0;
                CHECK(hipEventElapsedTime(&ms2, e0, e1));
                printf("%-56s %.1f ms  %.3f PFLOP/s  (%.3f of 2.5)\n", "32x32x16 + LDS-DMA + barrier, rounds of 512 on two queues", ms2,
                       flops / ms2 / 1e12, flops / ms2 / 1e12 / 2.5);
                continue;
            }
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            static const char *names[] = {"32x32x16", "16x16x32", "32x32x16 + barrier per two tiles", "32x32x16 + LDS-DMA + barrier per two tiles",
                                          "32x32x16 + LDS-DMA + barrier per four tiles (2 WG/CU)", "", "32x32x16 + 2 ds_write_b128 per wave and two tiles",
                                          "32x32x16 + 4 ds_write_b128 per wave and two tiles",
                                          "32x32x16 + LDS-DMA + barrier, 8 waves per workgroup (2 WG/CU)",
                                          "32x32x16 + LDS-DMA + barrier, 16 waves per workgroup (1 WG/CU)",
                                          "32x32x16 + global_load -> VGPR -> ds_write + barrier, 4 waves",
                                          "32x32x16 + global_load -> VGPR -> ds_write + barrier, 8 waves",
                                          "32x32x16 + LDS-DMA from a 2 MB window (L2-resident), 4 waves",
                                          "32x32x16 + LDS-DMA from a 2 MB window (L2-resident), 8 waves",
                                          "32x32x16 + LDS-DMA from a 32 MB window, 4 waves",
                                          "ring of 2 two-tile stages, barrier first (reference for the next two), 4 waves",
                                          "ring of 4 one-tile stages, fetched 3 stages ahead, 4 waves",
                                          "ring of 4 one-tile stages, fetched 3 stages ahead, 8 waves",
                                          "32x32x16 + 16 extra vector instructions per tile", "32x32x16 + 32 extra vector instructions per tile",
                                          "32x32x16 + 64 extra vector instructions per tile",
                                          "32x32x16 + 32 extra vector + 32 extra scalar instructions per tile"};
            printf("%-56s %.1f ms  %.3f PFLOP/s  (%.3f of 2.5)\n", names[which], ms, flops / ms / 1e12, flops / ms / 1e12 / 2.5);
            fflush(stdout);
        }
    return 0;
}
