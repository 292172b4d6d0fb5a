// Microbenchmark: how much MFMA throughput does a stream of LDS-DMA instructions (global_load_lds_dwordx4) cost on gfx950?
// Each wave runs ITERS iterations of { NDMA x global_load_lds (L2-resident source), NMFMA x v_mfma_f32_32x32x2_f32 on 4
// independent accumulators, optional NREAD x ds_read_b128, optional workgroup barrier }.  2 workgroups x 4 waves per CU.
// Build: hipcc -O3 --offload-arch=gfx950 -o dma_mfma_probe tools/probes/dma_mfma_probe.hip ; run: ./dma_mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int NDMA, int NREAD, bool BARRIER>
__global__ void __launch_bounds__(256, 2) probe(const float* __restrict__ src, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // 64 KB
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = (float)lane, b = 1.0f;
    const float* g = src + (size_t)(blockIdx.x % 64) * 16384 + wave * 2048 + lane * 4;  // 64 KB per block slot: L2 resident
    f32x4 rd = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < NDMA; ++d)
            __builtin_amdgcn_global_load_lds((gptr_t)(g + d * 256), (lptr_t)(smem + ((it & 1) * 8192) + wave * 2048 + d * 256), 16, 0, 0);
#pragma unroll
        for (int r = 0; r < NREAD; ++r) rd += *reinterpret_cast<const f32x4*>(smem + (((it + 1) & 1) * 8192) + ((wave * 2048 + r * 256 + lane * 4) & 8191));
#pragma unroll
        for (int m = 0; m < 32; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
        if (BARRIER) __syncthreads();
    }
    float s = rd[0] + rd[1] + rd[2] + rd[3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NDMA, int NREAD, bool BARRIER>
void run(const float* src, float* out, const char* label) {
    const int iters = 2000, blocks = 512;
    auto k = probe<NDMA, NREAD, BARRIER>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 65536, 0, src, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 65536, 0, src, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 32 * 4096.0;  // waves x iters x MFMAs x flops
    printf("%-44s %7.3f ms  %6.1f TFLOP/s  (%.0f KB DMA per wave-iter)\n", label, ms, flops / ms / 1e9, NDMA * 1.0);
}

int main() {
    float *src, *out;
    hipMalloc(&src, 64 * 65536 * sizeof(float));
    hipMemset(src, 0, 64 * 65536 * sizeof(float));
    hipMalloc(&out, 4096);
    run<0, 0, false>(src, out, "MFMA only");
    run<0, 3, false>(src, out, "MFMA + 3 ds_read_b128 / 32 MFMA");
    run<0, 12, false>(src, out, "MFMA + 12 ds_read_b128 / 32 MFMA");
    run<2, 0, false>(src, out, "MFMA + 2 DMA / 32 MFMA");
    run<4, 0, false>(src, out, "MFMA + 4 DMA / 32 MFMA");
    run<6, 0, false>(src, out, "MFMA + 6 DMA / 32 MFMA");
    run<8, 0, false>(src, out, "MFMA + 8 DMA / 32 MFMA");
    run<6, 12, false>(src, out, "MFMA + 6 DMA + 12 reads");
    run<6, 12, true>(src, out, "MFMA + 6 DMA + 12 reads + barrier");
    run<0, 12, true>(src, out, "MFMA + 12 reads + barrier");
    return 0;
}
