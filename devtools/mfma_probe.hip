// tools/mfma_probe.hip -- development probe (not part of the product): how fast can one CU issue
// v_mfma_f32_32x32x2_f32 under the in-wave structures the k-NN kernel could use?
//   NQ   query sets per wave (independent accumulators sharing one A fragment)
//   PF   LDS fragment prefetch depth (0 = read, wait, use)
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NQ, int PF, int NT>
__global__ __launch_bounds__(NT) void probe(const float *__restrict__ q, float *__restrict__ out, int ntiles) {
    __shared__ __attribute__((aligned(16))) float tile[2][32 * 128];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    for (int i = tid; i < 2 * 32 * 128; i += NT) (&tile[0][0])[i] = (float)((i * 7 + 3) % 13) * 0.01f;
    float b[NQ][64];
#pragma unroll
    for (int s = 0; s < NQ; ++s)
#pragma unroll
        for (int i = 0; i < 64; ++i) b[s][i] = q[(tid * NQ + s) * 64 + i];
    __syncthreads();
    float sum = 0.f;
    for (int t = 0; t < ntiles; ++t) {
        const f32x4 *sb = reinterpret_cast<const f32x4 *>(tile[t & 1]) + j * 32;
        const int sw = j & 15;
        f32x16 acc[NQ];
#pragma unroll
        for (int s = 0; s < NQ; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
        f32x4 a[PF + 1];
#pragma unroll
        for (int p = 0; p < PF; ++p) a[p] = sb[(2 * p + h) ^ sw];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (g + PF < 16) a[(g + PF) % (PF + 1)] = sb[(2 * (g + PF) + h) ^ sw];
            const f32x4 av = a[g % (PF + 1)];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int s = 0; s < NQ; ++s)
                    acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b[s][4 * g + e], acc[s], 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < NQ; ++s) {
            float mx = acc[s][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[s][r]);
            sum += mx;
        }
    }
    out[blockIdx.x * NT + tid] = sum;
}

template <int NQ, int PF, int NT>
void run(const char *name, int blocks_per_cu, const float *dq, float *dout) {
    const int ntiles = 400;
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<NQ, PF, NT><<<grid, NT>>>(dq, dout, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NQ, PF, NT><<<grid, NT>>>(dq, dout, ntiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * (NT / 64) * ntiles * NQ * 64.0 * (32 * 32 * 2 * 2);
    printf("%-34s NQ=%d PF=%d waves/blk=%d blk/CU=%d  %.3f ms  %.1f TFLOP/s  (%s)\n", name, NQ, PF, NT / 64,
           blocks_per_cu, ms, flops / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

int main() {
    float *dq, *dout;
    hipMalloc(&dq, 512 * 2 * 64 * 4 * 4);
    hipMalloc(&dout, 256 * 8 * 512 * 4);
    std::vector<float> hq(512 * 2 * 64 * 4, 0.37f);
    hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice);
    run<1, 0, 256>("1 chain, no prefetch, 1w/SIMD", 1, dq, dout);
    run<1, 0, 512>("1 chain, no prefetch, 2w/SIMD", 1, dq, dout);
    run<1, 0, 512>("1 chain, no prefetch, 4w/SIMD", 2, dq, dout);
    run<1, 2, 256>("1 chain, prefetch 2, 1w/SIMD", 1, dq, dout);
    run<1, 2, 512>("1 chain, prefetch 2, 2w/SIMD", 1, dq, dout);
    run<1, 2, 512>("1 chain, prefetch 2, 4w/SIMD", 2, dq, dout);
    run<2, 0, 256>("2 chains, no prefetch, 1w/SIMD", 1, dq, dout);
    run<2, 0, 512>("2 chains, no prefetch, 2w/SIMD", 1, dq, dout);
    run<2, 2, 256>("2 chains, prefetch 2, 1w/SIMD", 1, dq, dout);
    run<2, 2, 512>("2 chains, prefetch 2, 2w/SIMD", 1, dq, dout);
    run<2, 2, 256>("2 chains, prefetch 2, 2w/SIMD (2 blk)", 2, dq, dout);
    return 0;
}
