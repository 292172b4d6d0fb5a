// Microbenchmark for the bf16x3 conv kernel (gfx950): what does each ingredient of the K loop cost next to the bf16 MFMAs?
// One iteration = one 16-k group of a 64x64 wave tile: 24 x v_mfma_f32_32x32x16_bf16 on 4 accumulators, plus optionally
//   NREAD ds_read_b128 (operand fragments), NPIECE split pieces (9 VALU each: the fp32 -> 3 x bf16 split of one float pair),
//   NDMA global_load_lds_dwordx4 (L2-resident source), a workgroup barrier every second iteration.
// 4 waves per workgroup, 80 KB of LDS => 2 workgroups per CU, like the real kernel.
// Build: hipcc -O3 --offload-arch=gfx950 -o x3_mfma_probe tools/probes/x3_mfma_probe.hip ; run: ./x3_mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(const float a0, const float a1, unsigned& hi, unsigned& mid, unsigned& lo) {
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a0, a1}, bf16x2));
    const float r0 = a0 - __uint_as_float(h << 16), r1 = a1 - __uint_as_float(h & 0xffff0000u);
    const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    hi = h; mid = m;
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
}

template <int NREAD, int NPIECE, int NDMA, bool BARRIER, bool INTERLEAVE, int WGS>
__global__ void __launch_bounds__(256, WGS) probe(const float* __restrict__ src, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 op[2][6];  // [slot][operand]: 2 A tiles x 3 terms (B reuses them: only the instruction mix matters)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int o = 0; o < 6; ++o) op[s][o] = u32x4{0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    const float* g = src + (size_t)(blockIdx.x % 64) * 16384 + wave * 2048 + lane * 4;  // 64 KB per block slot: L2 resident
    f32x4 raw[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) raw[r] = f32x4{1.f + lane, 2.f, 3.f, 4.f};
    // one group with compile-time operand slots (a runtime slot index would turn into v_cndmask chains)
    auto group = [&](auto slot_c, int it) __attribute__((always_inline)) {
        constexpr int slot = decltype(slot_c)::value;
#pragma unroll
        for (int d = 0; d < NDMA; ++d)
            __builtin_amdgcn_global_load_lds((gptr_t)(g + d * 256), (lptr_t)(smem + (slot * 8192) + wave * 2048 + d * 256), 16, 0, 0);
#pragma unroll
        for (int r = 0; r < NREAD; ++r)
            raw[r] = *reinterpret_cast<const f32x4*>(smem + ((slot ^ 1) * 8192) + ((wave * 2048 + r * 256 + lane * 4) & 8191));
        __builtin_amdgcn_sched_barrier(0);
        int piece = 0;
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, op[slot][m % 6]),
                                                                 __builtin_bit_cast(bf16x8, op[slot][(m + 1) % 6]), acc[m & 3], 0, 0, 0);
            if (INTERLEAVE) {
                __builtin_amdgcn_sched_barrier(0);
                if (m >= 1 && piece < NPIECE) {
                    unsigned h, mm, l;
                    const int f = piece / 4, e = piece % 4;
                    split_pair(raw[(f * 2 + e / 2) % 16][(e % 2) * 2], raw[(f * 2 + e / 2) % 16][(e % 2) * 2 + 1], h, mm, l);
                    op[slot ^ 1][(f * 3) % 6][e] = h; op[slot ^ 1][(f * 3 + 1) % 6][e] = mm; op[slot ^ 1][(f * 3 + 2) % 6][e] = l;
                    ++piece;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (!INTERLEAVE) {
#pragma unroll
            for (; piece < NPIECE; ++piece) {
                unsigned h, mm, l;
                const int f = piece / 4, e = piece % 4;
                split_pair(raw[(f * 2 + e / 2) % 16][(e % 2) * 2], raw[(f * 2 + e / 2) % 16][(e % 2) * 2 + 1], h, mm, l);
                op[slot ^ 1][(f * 3) % 6][e] = h; op[slot ^ 1][(f * 3 + 1) % 6][e] = mm; op[slot ^ 1][(f * 3 + 2) % 6][e] = l;
            }
        }
    };
    for (int it = 0; it < iters; it += 2) {
        group(std::integral_constant<int, 0>{}, it);
        group(std::integral_constant<int, 1>{}, it);
        if (BARRIER) __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += raw[r][0];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NREAD, int NPIECE, int NDMA, bool BARRIER, bool INTERLEAVE, int WGS>
void run(const float* src, float* out, const char* label) {
    const int iters = 4000, blocks = 256 * WGS;
    const int lds = WGS == 2 ? 80 * 1024 : 150 * 1024;
    auto k = probe<NREAD, NPIECE, NDMA, BARRIER, INTERLEAVE, WGS>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, src, out, 10);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, src, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double flops = (double)blocks * 4 * iters * 24 * 32768.0;  // waves x iters x MFMAs x flops
    printf("%-58s %8.3f ms  %7.1f TF bf16  (%4.1f %% of 2500)  fp32-equivalent %6.1f TF\n", label, best, flops / best / 1e9,
           flops / best / 1e9 / 25.0, flops / best / 1e9 / 6.0);
}

int main() {
    float *src, *out;
    hipMalloc(&src, 64 * 16384 * sizeof(float) + 65536);
    hipMemset(src, 0, 64 * 16384 * sizeof(float) + 65536);
    hipMalloc(&out, 4096);
    run<0, 0, 0, false, false, 2>(src, out, "MFMA only, 2 WG/CU");
    run<0, 0, 0, false, false, 1>(src, out, "MFMA only, 1 WG/CU");
    run<10, 0, 0, false, false, 2>(src, out, "+ 10 ds_read_b128");
    run<10, 8, 0, false, false, 2>(src, out, "+ 10 reads + 8 pieces (72 VALU) after the MFMAs");
    run<10, 8, 0, false, true, 2>(src, out, "+ 10 reads + 8 pieces interleaved");
    run<16, 16, 0, false, false, 2>(src, out, "+ 16 reads + 16 pieces (144 VALU) after the MFMAs");
    run<16, 16, 0, false, true, 2>(src, out, "+ 16 reads + 16 pieces interleaved");
    run<10, 8, 5, false, true, 2>(src, out, "+ 10 reads + 8 pieces interleaved + 5 DMA");
    run<10, 8, 5, true, true, 2>(src, out, "+ 10 reads + 8 pieces interleaved + 5 DMA + barrier");
    run<10, 8, 5, true, false, 2>(src, out, "+ 10 reads + 8 pieces after + 5 DMA + barrier");
    run<10, 8, 5, true, true, 1>(src, out, "same (interleaved), 1 WG/CU");
    run<0, 8, 0, false, true, 2>(src, out, "MFMA + 8 pieces interleaved, no LDS");
    run<0, 0, 5, true, false, 2>(src, out, "MFMA + 5 DMA + barrier");
    return 0;
}
