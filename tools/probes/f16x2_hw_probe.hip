// Hardware facts the f16x2 contraction mode rests on, checked on gfx950:
//   1. does v_mfma_f32_32x32x16_f16 keep f16 DENORMAL inputs (the low piece of small elements)?
//   2. v_cvt_pk_f16_f32: round-to-nearest-even, overflow -> inf, denormal results kept?
//   3. v_permlane32_swap: lane l <-> lane l ^ 32 exchange as the kernels use it
//   4. ds_bpermute row fetch in the 32x32 accumulator layout: row(r, lh) = (r & 3) + 8 (r >> 2) + 4 lh
//   5. timing: 12 f16 MFMAs + f16x2 split/max-tracking per 16-k group vs 24 bf16 MFMAs + bf16x3 split (one wave tile 32 x 128)
// Build: hipcc -O3 --offload-arch=gfx950 -o f16x2_hw_probe tools/probes/f16x2_hw_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void mfma_denorm(float* out, float aval, float bval) {
    const int lane = threadIdx.x;
    f16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)aval; b[e] = (_Float16)bval; }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
}

__global__ void cvt_probe(const float* in, unsigned* out, int n) {
    const int i = threadIdx.x;
    if (i < n) {
        f16x2 h = __builtin_convertvector(f32x2{in[i], -in[i]}, f16x2);
        out[i] = __builtin_bit_cast(unsigned, h);
    }
}

__global__ void swap_probe(int* out) {
    const int lane = threadIdx.x;
    int v = lane * 10;
    // returns {new old-operand, new src-operand}: lanes 32..63 of arg0 <-> lanes 0..31 of arg1
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    out[lane] = r[0];
    out[64 + lane] = r[1];
    // accumulator-row fetch: every lane holds value 1000 + lane; row(r, lh) lives in lane row
    const int lh = lane >> 5;
    int acc = 0;
    for (int r2 = 0; r2 < 16; ++r2) {
        const int row = (r2 & 3) + 8 * (r2 >> 2) + 4 * lh;
        const int got = __builtin_amdgcn_ds_bpermute(row * 4, 1000 + lane);
        acc += (got == 1000 + row) ? 1 : 0;
    }
    out[128 + lane] = acc;
}

// ---- timing ----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split_bf16x3(const float a0, const float a1, unsigned& hi, unsigned& mid, unsigned& lo) {
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a0, a1}, bf16x2));
    float r0 = a0 - __uint_as_float(h << 16);
    asm volatile("" : "+v"(r0));
    float r1 = a1 - __uint_as_float(h & 0xffff0000u);
    const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
    float s0 = r0 - __uint_as_float(m << 16);
    asm volatile("" : "+v"(s0));
    float s1 = r1 - __uint_as_float(m & 0xffff0000u);
    hi = h; mid = m;
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
}
__device__ __forceinline__ void split_f16x2(const float a0, const float a1, const float sc, unsigned& hi, unsigned& lo) {
    const float x0 = a0 * sc, x1 = a1 * sc;
    const f16x2 h = __builtin_convertvector(f32x2{x0, x1}, f16x2);
    const float r0 = x0 - (float)h[0], r1 = x1 - (float)h[1];
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, f16x2));
}

// MODE 0: 24 bf16 MFMAs + 4 bf16x3 split pieces per group; 1: 12 f16 MFMAs + 4 f16x2 pieces + running-max tracking; 2: 12 f16 MFMAs only;
// 3: 24 bf16 MFMAs only
template <int MODE>
__global__ void __launch_bounds__(512, 1) timing(const float* __restrict__ src, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 sa[3], sb[3][4];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        sa[t] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
#pragma unroll
        for (int j = 0; j < 4; ++j) sb[t][j] = u32x4{0x3c003c00u + (unsigned)lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    }
    f32x4 ra[2];
    ra[0] = *reinterpret_cast<const f32x4*>(src + lane * 8);
    ra[1] = *reinterpret_cast<const f32x4*>(src + lane * 8 + 4);
    float sc = 1024.f, lim = 65472.f / 1024.f;
    int resc = 0;
    for (int it = 0; it < iters; ++it) {
        u32x4 na[3];
        if (MODE == 1) {
            float m = fmaxf(fmaxf(fabsf(ra[0][0]), fabsf(ra[0][1])), fabsf(ra[0][2]));
            m = fmaxf(fmaxf(m, fabsf(ra[0][3])), fabsf(ra[1][0]));
            m = fmaxf(fmaxf(m, fabsf(ra[1][1])), fabsf(ra[1][2]));
            m = fmaxf(m, fabsf(ra[1][3]));
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
            m = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            if (__builtin_amdgcn_ballot_w64(m > lim) != 0) {  // rare path
                sc *= 0.25f; lim *= 4.f; ++resc;
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] *= 0.25f;
            }
        }
        int piece = 0;
        constexpr int NM = (MODE == 0 || MODE == 3) ? 24 : 12;
#pragma unroll
        for (int mth = 0; mth < NM; ++mth) {
            if (MODE == 0 || MODE == 3)
                acc[mth & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, sa[mth % 3]), __builtin_bit_cast(bf16x8, sb[(mth / 4) % 3][mth & 3]), acc[mth & 3], 0, 0, 0);
            else
                acc[mth & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, sa[mth % 2]), __builtin_bit_cast(f16x8, sb[(mth / 4) % 2][mth & 3]), acc[mth & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (mth >= 2 && piece < 4 && MODE <= 1) {
                const int e = piece;
                if (MODE == 0) {
                    unsigned h, m2, l;
                    split_bf16x3(ra[e / 2][(e % 2) * 2], ra[e / 2][(e % 2) * 2 + 1], h, m2, l);
                    na[0][e] = h; na[1][e] = m2; na[2][e] = l;
                } else {
                    unsigned h, l;
                    split_f16x2(ra[e / 2][(e % 2) * 2], ra[e / 2][(e % 2) * 2 + 1], sc, h, l);
                    na[0][e] = h; na[1][e] = l; na[2][e] = l;
                }
                ++piece;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MODE <= 1) {
#pragma unroll
            for (int t = 0; t < 3; ++t) sa[t] = na[t];
            ra[0] += ra[1] * 1e-9f;  // keep the raw data live and changing
        }
    }
    float s = (float)resc;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static float run_timing(const float* src, float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    timing<MODE><<<256, 512>>>(src, out, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    timing<MODE><<<256, 512>>>(src, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* d; unsigned* du; int* di;
    hipMalloc(&d, 1 << 20); hipMalloc(&du, 4096); hipMalloc(&di, 4096);
    printf("== 1. v_mfma_f32_32x32x16_f16 with denormal f16 inputs (expected 16 * a * b)\n");
    const float avals[] = {1.f, 6.103515625e-05f /*2^-14 normal min*/, 3.0517578125e-05f /*2^-15*/, 5.9604644775390625e-08f /*2^-24 smallest denormal*/, 1.e-6f};
    for (float a : avals) {
        mfma_denorm<<<1, 64>>>(d, a, 1.f);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("   a = %.6e (f16 %s): result %.9e, expected %.9e  -> %s\n", a, a < 6.1e-5f ? "denormal" : "normal", h, 16.0 * (double)(float)(_Float16)a,
               h == 16.f * (float)(_Float16)a ? "kept" : (h == 0.f ? "FLUSHED" : "other"));
    }
    mfma_denorm<<<1, 64>>>(d, 3.0517578125e-05f, 3.0517578125e-05f);
    { float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); printf("   denormal x denormal (2^-15 * 2^-15 * 16 = %.6e): %.6e\n", 16.0 * pow(2.0, -30), h); }

    printf("== 2. v_cvt_pk_f16_f32\n");
    const float cv[] = {1.00048828125f /*1 + 2^-11: tie -> even (1.0)*/, 1.00146484375f /*1 + 3*2^-11: tie -> 1 + 2^-9... even*/, 65504.f, 65519.9f, 65520.f, 70000.f, 1e-5f, 5.9604644775390625e-08f, 2.98e-08f, 3.1e-08f};
    float* dc; hipMalloc(&dc, sizeof(cv)); hipMemcpy(dc, cv, sizeof(cv), hipMemcpyHostToDevice);
    const int ncv = sizeof(cv) / 4;
    cvt_probe<<<1, 64>>>(dc, du, ncv);
    unsigned hu[64]; hipMemcpy(hu, du, ncv * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < ncv; ++i) printf("   %.9e -> 0x%04x (neg 0x%04x)\n", cv[i], hu[i] & 0xffff, hu[i] >> 16);

    printf("== 3/4. permlane32_swap and ds_bpermute row fetch\n");
    swap_probe<<<1, 64>>>(di);
    int hi_[192]; hipMemcpy(hi_, di, 192 * 4, hipMemcpyDeviceToHost);
    printf("   swap r[0]: lane0 %d lane1 %d lane32 %d lane33 %d | r[1]: lane0 %d lane1 %d lane32 %d lane33 %d\n", hi_[0], hi_[1], hi_[32], hi_[33], hi_[64], hi_[65], hi_[96], hi_[97]);
    int okrows = 0; for (int l = 0; l < 64; ++l) okrows += hi_[128 + l] == 16;
    printf("   bpermute rows correct in %d of 64 lanes\n", okrows);

    printf("== 5. timing, 256 workgroups x 8 waves (2 per SIMD), per 16-k group of a 32 x 128 wave tile\n");
    float* src; hipMalloc(&src, 4096);
    float hs[1024]; for (int i = 0; i < 1024; ++i) hs[i] = (float)((i * 7919) % 1000) / 500.f - 1.f;
    hipMemcpy(src, hs, 4096, hipMemcpyHostToDevice);
    const int iters = 20000;
    const float t3 = run_timing<3>(src, d, iters), t0 = run_timing<0>(src, d, iters), t2 = run_timing<2>(src, d, iters), t1 = run_timing<1>(src, d, iters);
    printf("   24 bf16 MFMAs only            %.3f ms  (%.1f ns/group)\n", t3, t3 * 1e6 / iters);
    printf("   24 bf16 MFMAs + bf16x3 split  %.3f ms  (%.1f ns/group)\n", t0, t0 * 1e6 / iters);
    printf("   12 f16 MFMAs only             %.3f ms  (%.1f ns/group)\n", t2, t2 * 1e6 / iters);
    printf("   12 f16 MFMAs + f16x2 split + max tracking  %.3f ms  (%.1f ns/group)   ratio to bf16x3: %.3f\n", t1, t1 * 1e6 / iters, t1 / t0);
    return 0;
}
