// Minimal reproducer attempt for the round-5 finding: v_pk_mul_f32 / v_pk_fma_f32 whose second source selects the ODD element of a
// register pair (op_sel:[0,1] op_sel_hi:[1,1]) returned wrong products in lanes 16-31 / 48-63 of the f16x2 weight-gradient kernel whenever two waves
// shared a SIMD.  Here: the same instruction forms in isolation, next to v_mfma_f32_32x32x16_f16 traffic, with one and with two workgroups per CU.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/pk_opsel_probe tools/probes/pk_opsel_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <bool MFMA, int NOPS = 0>
__global__ void __launch_bounds__(256, 2) probe(const float* __restrict__ in, unsigned* __restrict__ bad, float* __restrict__ sink, int iters) {
    extern __shared__ float pad[];
    const int tid = threadIdx.x, gid = blockIdx.x * 256 + tid;
    f32x2 a = {in[(gid * 4 + 0) & 65535], in[(gid * 4 + 1) & 65535]};
    f32x2 s = {in[(gid * 4 + 2) & 65535] + 2.f, in[(gid * 4 + 3) & 65535] + 3.f};  // two per-lane scales in one 64-bit register pair
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f16x8 fa, fb;
    for (int k = 0; k < 8; ++k) { fa[k] = (_Float16)(a[0] * 0.01f + k); fb[k] = (_Float16)(a[1] * 0.01f - k); }
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        f32x2 p_odd, p_even, f_odd;
        if (MFMA) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
        if (NOPS == 1) asm volatile("s_nop 15");
        if (NOPS == 2) asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15");
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(p_odd) : "v"(a), "v"(s));      // {a0 * s1, a1 * s1}
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p_even) : "v"(a), "v"(s));                   // {a0 * s0, a1 * s0}
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(f_odd) : "v"(a), "v"(s), "v"(p_even));  // a * s1 + p_even
        if (MFMA) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fa, acc, 0, 0, 0);
        const float r0 = a[0] * s[1], r1 = a[1] * s[1], e0 = a[0] * s[0], e1 = a[1] * s[0];
        const float g0 = __builtin_fmaf(a[0], s[1], e0), g1 = __builtin_fmaf(a[1], s[1], e1);
        nbad += (p_odd[0] != r0) + (p_odd[1] != r1) + (p_even[0] != e0) + (p_even[1] != e1) + (f_odd[0] != g0) + (f_odd[1] != g1);
        a[0] += 0.125f; a[1] -= 0.0625f; s[0] += 0.5f; s[1] -= 0.25f;
    }
    if (nbad) atomicAdd(&bad[tid & 63], nbad);
    float t = 0.f;
    for (int r = 0; r < 16; ++r) t += acc[r];
    if (t == 12345.678f) sink[gid] = t + pad[tid];
}

int main() {
    std::vector<float> h(65536);
    for (int i = 0; i < 65536; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xFFFF) / 4096.f - 8.f;
    float *in, *sink; unsigned* bad;
    hipMalloc(&in, 65536 * 4); hipMalloc(&sink, 4096 * 256 * 4); hipMalloc(&bad, 64 * 4);
    hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice);
    for (int cfg = 0; cfg < 8; ++cfg) {
        const int mfma = cfg >= 2, two = cfg & 1, nops = cfg >= 6 ? 2 : cfg >= 4 ? 1 : 0;
        hipMemset(bad, 0, 64 * 4);
        const size_t lds = two ? 1024 : 100 * 1024;  // 100 KB of LDS: one workgroup per CU (one wave per SIMD)
        auto run = [&](auto kern) {
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(4096), dim3(256), lds, 0, in, bad, sink, 4000);
        };
        if (!mfma) run(probe<false>); else if (nops == 0) run(probe<true, 0>); else if (nops == 1) run(probe<true, 1>); else run(probe<true, 2>);
        unsigned hb[64];
        hipError_t e = hipMemcpy(hb, bad, 64 * 4, hipMemcpyDeviceToHost);
        unsigned long long tot = 0; for (int i = 0; i < 64; ++i) tot += hb[i];
        printf("mfma traffic %d, s_nop after the first mfma %d x 16 cycles, workgroups per CU %d: %s, mismatching results %llu", mfma, nops == 2 ? 4 : nops, two ? 2 : 1, hipGetErrorString(e), tot);
        if (tot) { printf("  lanes:"); for (int i = 0; i < 64; ++i) if (hb[i]) printf(" %d(%u)", i, hb[i]); }
        printf("\n");
    }
    return 0;
}
