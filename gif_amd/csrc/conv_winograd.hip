// Winograd F(2x2, 3x3) path for the stride-1 / pad-1 3x3 convolutions (fp32, NHWC, gfx950).
//
// The same layers as conv_igemm.hip serves (ModulatedConv2d :343-347, EqualConv2d :176 of
// model/stylegan2_common_layers.py, and their stride-1 data gradients), computed with 16 instead of 36 multiplies per
// 2x2 output tile:   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A.
// Kernels:
//   0. wino_gy_transform    : (wgrad only, F(3x3,2x2)) 2x2 tiles of the output gradient -> Mg [16][tiles_pad][C_pad]; the
//      16 plane GEMMs of the weight gradient run in conv_wgrad.hip (conv_wgrad_mfma, "planes" mode)
//   1. wino_input_transform : x [B,H,W,C] (times the per-sample modulation s[b,c] — free here) -> V [16][tiles_pad][C_pad]
//   2. wino_weight_transform: canonical weight view (any strides, optional 180-degree flip for the dgrad) -> U [16][RP][CP]
//   3. wino_gemm_mfma       : for each of the 16 positions p a dense GEMM  M_p[tile,co] = sum_ci V_p[tile,ci] U_p[co,ci]
//      on v_mfma_f32_32x32x2_f32 with LDS-DMA staging (same unpadded-row + XOR-swizzle scheme as conv_gather_mfma_glds;
//      no gather, no bounds: V is a dense padded matrix; 3-stage ring), and the OUTPUT TRANSFORM FUSED: the accumulators of
//      position p are folded, in the MFMA shadow of position p+1 (ping-pong accumulator sets), into the four 2x2-output
//      accumulators with the +-1/0 coefficients of A^T (x) A^T, so the 16x-larger M tensor never exists.  The epilogue (demodulation, noise residual, bias, leaky ReLU) is the shared
//      LDS-transposed float4 epilogue, run once per output position (a,b) of the 2x2 tile.
// HBM traffic: V is 4x the input (written once, read once); MFMA work is 4/9 of the direct convolution.
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    int xcd = bid % nx, idx = bid / nx;
    int q = nwg / nx, r = nwg % nx;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// GEMM tile: 256 threads = 4 waves as 2 (M) x 2 (N); wave tile 64 tiles x 32 couts; block tile 128 x 64; BK = 32.
constexpr int WBM = 128, WBN = 64, WBK = 32, WNSTAGE = 3;
constexpr int WPAD = 256;  // row padding of the V / Mg planes: the bf16x3 GEMM walks them in 256-row blocks

// ------------------------------------------------------------------------------------------------ input transform
// one lane = one 4x4 input patch of one tile x 4 channels; B^T d B with B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1].
// V is [16][ntiles_pad][CP]: the channel padding (C..CP) is written as zeros, the row padding (ntiles..ntiles_pad) is
// never written and never matters (GEMM row m only feeds output row m, and rows >= ntiles are not stored), so the GEMM
// streams V without any bounds logic.
__global__ void __launch_bounds__(256) wino_input_transform(const float* __restrict__ x, const float* __restrict__ scale,
                                                            float* __restrict__ V, int B, int H, int W, int C, int CP,
                                                            long ntiles_pad) {
    const int C4 = CP >> 2, TH = H >> 1, TW = W >> 1;
    const long ntiles = (long)B * TH * TW;
    const long total = ntiles * C4;
    const size_t plane = (size_t)ntiles_pad * CP;
    // 32-bit index math (callers guarantee ntiles_pad * CP < 2^31): 64-bit div/mod costs ~60 instructions each
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < (unsigned)total; idx += gridDim.x * blockDim.x) {
        const unsigned tile = idx / (unsigned)C4;
        const int c4 = (int)(idx - tile * (unsigned)C4);
        float* vout = V + (size_t)tile * CP + c4 * 4;
        if (c4 * 4 >= C) {  // channel padding
#pragma unroll
            for (int q = 0; q < 16; ++q) *reinterpret_cast<f32x4*>(vout + q * plane) = (f32x4)(0.f);
            continue;
        }
        const unsigned t2 = tile / (unsigned)TW;
        const int tx = (int)(tile - t2 * (unsigned)TW);
        const int b = (int)(t2 / (unsigned)TH), ty = (int)(t2 - (unsigned)b * (unsigned)TH);
        const float* xb = x + ((size_t)b * H * W) * C + c4 * 4;
        f32x4 d[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int iy = 2 * ty - 1 + r;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ix = 2 * tx - 1 + c;
                const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                d[r][c] = ok ? *reinterpret_cast<const f32x4*>(xb + ((size_t)iy * W + ix) * C) : (f32x4)(0.f);
            }
        }
        f32x4 s = (f32x4)(1.f);
        if (scale) s = *reinterpret_cast<const f32x4*>(scale + (size_t)b * C + c4 * 4);
        f32x4 t[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            t[0][c] = d[0][c] - d[2][c];
            t[1][c] = d[1][c] + d[2][c];
            t[2][c] = d[2][c] - d[1][c];
            t[3][c] = d[1][c] - d[3][c];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x4 v0 = (t[r][0] - t[r][2]) * s, v1 = (t[r][1] + t[r][2]) * s;
            f32x4 v2 = (t[r][2] - t[r][1]) * s, v3 = (t[r][1] - t[r][3]) * s;
            *reinterpret_cast<f32x4*>(vout + (r * 4 + 0) * plane) = v0;
            *reinterpret_cast<f32x4*>(vout + (r * 4 + 1) * plane) = v1;
            *reinterpret_cast<f32x4*>(vout + (r * 4 + 2) * plane) = v2;
            *reinterpret_cast<f32x4*>(vout + (r * 4 + 3) * plane) = v3;
        }
    }
}

// Variant with vertical reuse (round 4, review item 6 "measure, don't argue"): the 4x4 patches of vertically adjacent tiles share two of
// their four input rows; wino_input_transform fetches them again (PMC: 1.6x the ideal fetch, served by L2 / Infinity Cache).  Here
// a lane walks TYB tiles down one tile column and keeps the two shared rows in registers: 8 instead of 16 float4 loads per tile.
template <int TYB>
__global__ void __launch_bounds__(256) wino_input_transform_rows(const float* __restrict__ x, const float* __restrict__ scale,
                                                                 float* __restrict__ V, int B, int H, int W, int C, int CP,
                                                                 long ntiles_pad) {
    const int C4 = CP >> 2, TH = H >> 1, TW = W >> 1;
    const int NYB = (TH + TYB - 1) / TYB;
    const unsigned total = (unsigned)B * NYB * TW * C4;
    const size_t plane = (size_t)ntiles_pad * CP;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % (unsigned)C4);
        unsigned r = idx / (unsigned)C4;
        const int tx = (int)(r % (unsigned)TW);
        r /= (unsigned)TW;
        const int yb = (int)(r % (unsigned)NYB), b = (int)(r / (unsigned)NYB);
        const int ty0 = yb * TYB, ty1 = min(ty0 + TYB, TH);
        const bool pad = c4 * 4 >= C;
        const float* xb = x + ((size_t)b * H * W) * C + c4 * 4;
        f32x4 s = (f32x4)(1.f);
        if (scale && !pad) s = *reinterpret_cast<const f32x4*>(scale + (size_t)b * C + c4 * 4);
        auto load_row = [&](int iy, f32x4 (&d)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ix = 2 * tx - 1 + c;
                const bool ok = !pad && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                d[c] = ok ? *reinterpret_cast<const f32x4*>(xb + ((size_t)iy * W + ix) * C) : (f32x4)(0.f);
            }
        };
        f32x4 d[4][4];
        load_row(2 * ty0 - 1, d[0]);
        load_row(2 * ty0, d[1]);
        for (int ty = ty0; ty < ty1; ++ty) {
            load_row(2 * ty + 1, d[2]);
            load_row(2 * ty + 2, d[3]);
            float* vout = V + ((size_t)(b * TH + ty) * TW + tx) * CP + c4 * 4;
            f32x4 t[4][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                t[0][c] = d[0][c] - d[2][c];
                t[1][c] = d[1][c] + d[2][c];
                t[2][c] = d[2][c] - d[1][c];
                t[3][c] = d[1][c] - d[3][c];
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                *reinterpret_cast<f32x4*>(vout + (rr * 4 + 0) * plane) = (t[rr][0] - t[rr][2]) * s;
                *reinterpret_cast<f32x4*>(vout + (rr * 4 + 1) * plane) = (t[rr][1] + t[rr][2]) * s;
                *reinterpret_cast<f32x4*>(vout + (rr * 4 + 2) * plane) = (t[rr][2] - t[rr][1]) * s;
                *reinterpret_cast<f32x4*>(vout + (rr * 4 + 3) * plane) = (t[rr][1] - t[rr][3]) * s;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) { d[0][c] = d[2][c]; d[1][c] = d[3][c]; }
        }
    }
}

// ------------------------------------------------------------------------------------------------ gy transform (wgrad)
// F(3x3,2x2): the 2x2 tile of the output gradient is the "filter": Mg = G g G^T with G = [1 0; .5 .5; .5 -.5; 0 1].
// Same [16][ntiles_pad][CP] layout as V (channel padding zeroed, row padding untouched: the wgrad GEMM bounds K itself).
__global__ void __launch_bounds__(256) wino_gy_transform(const float* __restrict__ gy, const float* __restrict__ scale,
                                                         float* __restrict__ Mg, int B, int H, int W, int C, int CP,
                                                         long ntiles_pad) {
    const int C4 = CP >> 2, TH = H >> 1, TW = W >> 1;
    const long ntiles = (long)B * TH * TW;
    const long total = ntiles * C4;
    const size_t plane = (size_t)ntiles_pad * CP;
    // 32-bit index math (callers guarantee ntiles_pad * CP < 2^31): 64-bit div/mod costs ~60 instructions each
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < (unsigned)total; idx += gridDim.x * blockDim.x) {
        const unsigned tile = idx / (unsigned)C4;
        const int c4 = (int)(idx - tile * (unsigned)C4);
        float* out = Mg + (size_t)tile * CP + c4 * 4;
        if (c4 * 4 >= C) {
#pragma unroll
            for (int q = 0; q < 16; ++q) *reinterpret_cast<f32x4*>(out + q * plane) = (f32x4)(0.f);
            continue;
        }
        const unsigned t2 = tile / (unsigned)TW;
        const int tx = (int)(tile - t2 * (unsigned)TW);
        const int b = (int)(t2 / (unsigned)TH), ty = (int)(t2 - (unsigned)b * (unsigned)TH);
        const float* g0 = gy + (((size_t)b * H + 2 * ty) * W + 2 * tx) * C + c4 * 4;
        f32x4 s = (f32x4)(1.f);
        if (scale) s = *reinterpret_cast<const f32x4*>(scale + (size_t)b * C + c4 * 4);
        f32x4 g[2][2];
        g[0][0] = *reinterpret_cast<const f32x4*>(g0) * s;
        g[0][1] = *reinterpret_cast<const f32x4*>(g0 + C) * s;
        g[1][0] = *reinterpret_cast<const f32x4*>(g0 + (size_t)W * C) * s;
        g[1][1] = *reinterpret_cast<const f32x4*>(g0 + (size_t)W * C + C) * s;
        f32x4 m[4][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            m[0][j] = g[0][j];
            m[1][j] = 0.5f * (g[0][j] + g[1][j]);
            m[2][j] = 0.5f * (g[0][j] - g[1][j]);
            m[3][j] = g[1][j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(out + (i * 4 + 0) * plane) = m[i][0];
            *reinterpret_cast<f32x4*>(out + (i * 4 + 1) * plane) = 0.5f * (m[i][0] + m[i][1]);
            *reinterpret_cast<f32x4*>(out + (i * 4 + 2) * plane) = 0.5f * (m[i][0] - m[i][1]);
            *reinterpret_cast<f32x4*>(out + (i * 4 + 3) * plane) = m[i][1];
        }
    }
}

// ------------------------------------------------------------------------------------------------ weight transform
// U = G g G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]; rows = op output channels, cols = op input channels
__global__ void wino_weight_transform(const float* __restrict__ w, float* __restrict__ U, int R, int C, int RP, int CP,
                                      long sr, long sc, long sky, long skx, int flip, float scale) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)RP * CP) return;
    int c = (int)(idx % CP), r = (int)(idx / CP);
    float g[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            int sy = flip ? 2 - ky : ky, sx = flip ? 2 - kx : kx;
            g[ky][kx] = (r < R && c < C) ? scale * w[r * sr + c * sc + sy * sky + sx * skx] : 0.f;
        }
    float u[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        u[0][j] = g[0][j];
        u[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
        u[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
        u[3][j] = g[2][j];
    }
    const size_t plane = (size_t)RP * CP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        U[(i * 4 + 0) * plane + idx] = u[i][0];
        U[(i * 4 + 1) * plane + idx] = 0.5f * (u[i][0] + u[i][1] + u[i][2]);
        U[(i * 4 + 2) * plane + idx] = 0.5f * (u[i][0] - u[i][1] + u[i][2]);
        U[(i * 4 + 3) * plane + idx] = u[i][2];
    }
}

// ------------------------------------------------------------------------------------------------ GEMM + output transform
struct WinoParams {
    const float* V;  // [16][ntiles_pad][CP]
    const float* U;  // [16][RP][CP]
    float* y;        // [B,H,W,Co]
    const float* out_scale;
    const float* bias;
    const float* residual;
    int B, H, W, Co, RP, CP;
    int ntiles, ntiles_pad, TH, TW;
    int act;
    float slope, gain;
    int tiles_m, tiles_n;
    // gradient-producer fusions (gif_conv_epilogue ABI 2): see conv_igemm.hip; partial rows = GEMM row tiles
    const float* mask_src;
    const float* dot_src;
    float mask_slope, mask_gain;
    float* part_cs;
    float* part_dot;
    // f16x2 (common.h "h2"): U = the f16 planes, uexp = their header ([RP] row exponents, then [RP] row flags); gate: raised by the
    // f16x2 GEMM when a K group leaves the precision window; wino_gemm_x3 WITH a gate is the guarded fallback (runs iff *gate == gate_gen)
    const int* uexp;
    unsigned* gate;
    unsigned gate_gen;
    unsigned* h2_stats;
};

// The row loop of both GEMM kernels' epilogues for one output position (oa, ob): modulation-gradient dot product, out_scale,
// residual, bias, activation, leaky-ReLU-backward mask, running column sums.  Rows go in batches of PF: first every global load
// of the batch, then the arithmetic and the stores — in one loop each row's loads would sit behind the previous row's store
// (y may alias the sources as far as the compiler knows) and pay a full memory round trip on their own (conv_igemm.hip).
// rtab != nullptr (wino_gemm_h2): the tile's row table in LDS — per GEMM row the 64-bit offset of its 2x2 output patch's first element (or ~0:
// padding row) and its sample, filled once per tile by wino_row_table; without it every lane repeats two div/mod pairs and a 64-bit multiply
// chain per row and output position (4 x E_IT times per tile).
template <bool FUSED, int E_IT, int EROWS, int LDC, int PF>
__device__ __forceinline__ void wino_epilogue_rows(const WinoParams& p, const float* Cs, int m0, int e_row0, int e_c, int n, int oa, int ob,
                                                   const f32x4& bias4, f32x4& cs, f32x4& ds, const unsigned long long* rtab = nullptr,
                                                   const int* rsmp = nullptr) {
    static_assert(E_IT % PF == 0, "epilogue batches");
    const bool two_src = FUSED && p.mask_src && p.dot_src && p.mask_src != p.dot_src;
    const size_t pos_off = ((size_t)oa * p.W + ob) * p.Co + n;
#pragma unroll 1
    for (int it0 = 0; it0 < E_IT; it0 += PF) {
        size_t off[PF];
        bool ok[PF];
        f32x4 rv[PF], dv[PF], xa[PF], xb[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            int b;
            if (rtab) {  // (compile-time per call site)
                const int row = e_row0 + (it0 + k) * EROWS;
                const unsigned long long o = rtab[row];
                ok[k] = o != ~0ull;
                b = rsmp[row];
                off[k] = (size_t)o + pos_off;
            } else {
                const int m = m0 + e_row0 + (it0 + k) * EROWS;
                ok[k] = m < p.ntiles;
                const int mm = ok[k] ? m : 0;
                const int tx = mm % p.TW, t2 = mm / p.TW;
                const int ty = t2 % p.TH;
                b = t2 / p.TH;
                off[k] = (((size_t)b * p.H + 2 * ty + oa) * p.W + 2 * tx + ob) * p.Co + n;
            }
            rv[k] = dv[k] = xa[k] = xb[k] = (f32x4)(0.f);
            if (ok[k]) {
                if (p.residual) rv[k] = *reinterpret_cast<const f32x4*>(p.residual + off[k]);
                if (p.out_scale) dv[k] = *reinterpret_cast<const f32x4*>(p.out_scale + (size_t)b * p.Co + n);
                if (FUSED && p.dot_src) xa[k] = *reinterpret_cast<const f32x4*>(p.dot_src + off[k]);
                if (FUSED && p.mask_src && !p.dot_src) xa[k] = *reinterpret_cast<const f32x4*>(p.mask_src + off[k]);
                if (two_src) xb[k] = *reinterpret_cast<const f32x4*>(p.mask_src + off[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            if (!ok[k]) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(Cs + (e_row0 + (it0 + k) * EROWS) * LDC + e_c);
            if (FUSED && p.dot_src) ds += v * xa[k];
            if (p.out_scale) v *= dv[k];
            if (p.residual) v += rv[k];
            v += bias4;
            if (p.act) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (v[e] > 0.f ? v[e] : v[e] * p.slope) * p.gain;
            }
            if (FUSED) {
                if (p.mask_src) {
                    const f32x4 xs = two_src ? xb[k] : xa[k];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= p.mask_gain * (xs[e] > 0.f ? 1.f : p.mask_slope);
                }
                cs += v;
            }
            *reinterpret_cast<f32x4*>(p.y + off[k]) = v;
        }
    }
}

// per-tile partial sums of one workgroup -> row `prow` of part_cs / part_dot (fixed order; smem is free by now)
template <int THREADS, int C4_ROW, int EROWS>
__device__ __forceinline__ void wino_write_partials(const WinoParams& p, float* smem, int tid, int n, int prow, f32x4 cs, f32x4 ds) {
    if (!p.part_cs && !p.part_dot) return;  // workgroup-uniform
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(smem);  // [2][THREADS]
    red[tid] = cs;
    red[THREADS + tid] = ds;
    __syncthreads();
    if (tid < C4_ROW && n < p.Co) {
        f32x4 a = red[tid], b = red[THREADS + tid];
#pragma unroll
        for (int k = 1; k < EROWS; ++k) {
            a += red[k * C4_ROW + tid];
            b += red[THREADS + k * C4_ROW + tid];
        }
        const size_t o = (size_t)prow * p.Co + n;
        if (p.part_cs) *reinterpret_cast<f32x4*>(p.part_cs + o) = a;
        if (p.part_dot) *reinterpret_cast<f32x4*>(p.part_dot + o) = b;
    }
}

// A^T = [1 1 1 0; 0 1 -1 -1]: coefficient of M[xi][nu] in Y[a][b] is cA(a,xi) * cA(b,nu)
__device__ __forceinline__ float wino_coef(int a, int xi) {
    return a == 0 ? (xi < 3 ? 1.f : 0.f) : (xi == 0 ? 0.f : (xi == 1 ? 1.f : -1.f));
}

// WN = waves along N: 2 -> 256 threads, block 128 x 64, two workgroups per CU; 4 -> 512 threads, block 128 x 128, one
// workgroup per CU (same 2 waves per SIMD) with a third fewer DMA bytes per MFMA (V tile shared by 128 couts).
template <int WN>
__global__ void __launch_bounds__(128 * WN, WN == 2 ? 2 : 1) wino_gemm_mfma(const WinoParams p) {
    constexpr int THREADS = 128 * WN, BN = 32 * WN, PROWS = THREADS / 8;  // PROWS = tile rows one DMA pass covers
    constexpr int LD = WBK, CH = WBK / 4, RB = 64 / WBK, RPW = 64 / CH;
    constexpr int MT = 2;
    constexpr int A_IT = WBM / PROWS, B_IT = BN / PROWS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [NSTAGE][WBM][LD]
    float* Bs = smem + WNSTAGE * WBM * LD;  // [NSTAGE][BN][LD]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm0 = (wave / WN) * 64, wn0 = (wave % WN) * 32;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
    const int m0 = tm * WBM, n0 = tn * BN;
    const int t_row = tid / CH;
    const int src_c4 = ((tid % CH) ^ ((t_row / RB) % CH)) * 4;
    const int fsw = (li / RB) % CH;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    // Both operands are dense, padded matrices: per-lane 32-bit element offsets + one wave-uniform pointer per stage.
    const unsigned a_off = (unsigned)(m0 + t_row) * (unsigned)p.CP + (unsigned)src_c4;
    const unsigned b_off = (unsigned)(n0 + t_row) * (unsigned)p.CP + (unsigned)src_c4;
    const size_t pass_stride = (size_t)PROWS * p.CP;  // tile rows per DMA pass
    const size_t planeV = (size_t)p.ntiles_pad * p.CP, planeU = (size_t)p.RP * p.CP;
    const int kchunks = p.CP / WBK;
    const int nsteps = 16 * kchunks;
    const float* vptr = p.V;  // wave-uniform cursors of the stage being loaded
    const float* uptr = p.U;
    int ld_kc = 0;

    auto issue = [&](int buf) __attribute__((always_inline)) {
        float* Ad = As + buf * WBM * LD + wave * RPW * LD;
        float* Bd = Bs + buf * BN * LD + wave * RPW * LD;
#pragma unroll
        for (int it = 0; it < A_IT; ++it)
            __builtin_amdgcn_global_load_lds((gptr_t)((vptr + it * pass_stride) + a_off), (lptr_t)(Ad + it * PROWS * LD), 16, 0, 0);
#pragma unroll
        for (int it = 0; it < B_IT; ++it)
            __builtin_amdgcn_global_load_lds((gptr_t)((uptr + it * pass_stride) + b_off), (lptr_t)(Bd + it * PROWS * LD), 16, 0, 0);
        vptr += WBK;
        uptr += WBK;
        ld_kc += WBK;
        if (ld_kc >= p.CP) {  // next position: same rows of the next plane
            ld_kc = 0;
            vptr += planeV - p.CP;
            uptr += planeU - p.CP;
        }
    };

    // Two accumulator sets: while the MFMAs of position q run into one, the finished M_{q-1} in the other is folded into
    // the four 2x2-output accumulators (and re-zeroed) by VALU instructions issued in the MFMAs' shadow.
    f32x16 accA[MT], accB[MT];
    f32x16 yo[4][MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accA[i][r] = 0.f;
            accB[i][r] = 0.f;
#pragma unroll
            for (int o = 0; o < 4; ++o) yo[o][i][r] = 0.f;
        }

    auto compute = [&](int buf, f32x16(&acc)[MT]) __attribute__((always_inline)) {
        const float* Ab = As + buf * WBM * LD + (wm0 + li) * LD;
        const float* Bb = Bs + buf * BN * LD + (wn0 + li) * LD;
#pragma unroll
        for (int kk = 0; kk < WBK / 8; ++kk) {
            const int c = ((kk * 2 + lh) ^ fsw) * 4;
            f32x4 av[MT], bv;
#pragma unroll
            for (int i = 0; i < MT; ++i) av[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LD + c);
            bv = *reinterpret_cast<const f32x4*>(Bb + c);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][t], bv[t], acc[i], 0, 0, 0);
        }
    };
    // branch-free (coefficients 0 / +-1 as data) so that the scheduler can place it between the MFMAs of `compute`
    auto fold = [&](int pos, f32x16(&acc)[MT]) __attribute__((always_inline)) {
        const int xi = pos >> 2, nu = pos & 3;
        float cf[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) cf[o] = wino_coef(o >> 1, xi) * wino_coef(o & 1, nu);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int o = 0; o < 4; ++o) yo[o][i] += cf[o] * acc[i];
            acc[i] = (f32x16)(0.f);
        }
    };

    // scheduling hint for the stages that carry a fold: 32 x { 1 MFMA, 3 VALU } so the fold's v_pk_fma run in the MFMAs' shadow
    auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 32; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
    };

    // 3-stage ring: the DMA of stage s+2 is issued before the MFMAs of stage s.  `s_waitcnt vmcnt(NI)` (NI = DMA
    // instructions per stage and wave) lets the newest stage stay in flight; the last two stages drain with vmcnt(0).
    constexpr int NI = A_IT + B_IT;
    static_assert((WN == 2 && NI == 6) || (WN == 4 && NI == 4), "the s_waitcnt immediates below encode vmcnt(NI)");
    int cur = 0, step = 0;
    auto advance = [&]() __attribute__((always_inline)) {
        if (step + 2 < nsteps) {
            if (WN == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        cur = cur == 2 ? 0 : cur + 1;
        ++step;
    };
    auto prefetch = [&]() __attribute__((always_inline)) {
        if (step + 2 < nsteps) issue(cur >= 1 ? cur - 1 : 2);  // (cur + 2) % 3
    };
    issue(0);
    issue(1);  // nsteps >= 16
    if (WN == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    for (int pp = 0; pp < 8; ++pp) {
        // even position 2pp -> accA; its first stage also folds accB (= position 2pp-1)
        prefetch();
        compute(cur, accA);
        fold(2 * pp - 1, accB);  // pp == 0: accB is still zero
        interleave();
        advance();
        for (int kc = 1; kc < kchunks; ++kc) {
            prefetch();
            compute(cur, accA);
            advance();
        }
        // odd position 2pp+1 -> accB; its first stage folds accA (= position 2pp)
        prefetch();
        compute(cur, accB);
        fold(2 * pp, accA);
        interleave();
        if (step + 1 < nsteps) advance();
        for (int kc = 1; kc < kchunks; ++kc) {
            prefetch();
            compute(cur, accB);
            if (step + 1 < nsteps) advance();
        }
    }
    fold(15, accB);

    // ---- epilogue: for each output position (a,b): transpose through LDS, then coalesced float4 rows
    constexpr int LDC = BN + 4;
    float* Cs = smem;  // [WBM][LDC]  (34.8 of the 72 KB / 67.6 of the 96 KB of staging)
    constexpr int C4_ROW = BN / 4, EROWS = THREADS / C4_ROW, E_IT = WBM / EROWS;
    const int e_row0 = tid / C4_ROW, e_c = (tid % C4_ROW) * 4;
    const int n = n0 + e_c;
    f32x4 bias4 = (f32x4)(0.f);
    if (p.bias && n < p.Co) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
    f32x4 cs = (f32x4)(0.f), ds = (f32x4)(0.f);
    const bool fused = p.mask_src || p.dot_src || p.part_cs || p.part_dot;  // workgroup-uniform
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                Cs[row * LDC + wn0 + li] = yo[o][i][r];
            }
        __syncthreads();
        if (n < p.Co) {
            if (fused) wino_epilogue_rows<true, E_IT, EROWS, LDC, (E_IT < 4 ? E_IT : 4)>(p, Cs, m0, e_row0, e_c, n, o >> 1, o & 1, bias4, cs, ds);
            else wino_epilogue_rows<false, E_IT, EROWS, LDC, (E_IT < 4 ? E_IT : 4)>(p, Cs, m0, e_row0, e_c, n, o >> 1, o & 1, bias4, cs, ds);
        }
    }
    wino_write_partials<THREADS, C4_ROW, EROWS>(p, smem, tid, n, tm, cs, ds);
}


// ------------------------------------------------------------------------------------------------ bf16x3 GEMM + output transform
// The same 16 position GEMMs with the fp32 operands on the bf16 matrix cores (common.h split_pair): V stays fp32 in HBM and LDS
// and is split after the LDS read; U arrives PRE-SPLIT (wino_weight_transform_x3: U3 [16][3 terms][RP][CP] bf16) in three
// [128][32] bf16 tiles of 64-byte rows per stage (16-byte chunks XOR-swizzled with (row >> 2) & 3, as in conv_igemm.hip), so a
// weight operand is one ds_read_b128.  512 threads = 8 waves as 4 (M) x 2 (N); wave tile 32 tiles x 64 couts (ONE V fragment
// to split per 12 MFMAs); block 128 x 128; BK = 32 = two 16-k groups; 3-stage ring (120 KB) => one workgroup per CU, two waves
// per SIMD.  Registers: 4 x 2 output accumulators (128) + ONE position accumulator set (32): the fold into the 2x2 outputs runs
// at the end of each position, un-overlapped inside the wave (the SIMD's other wave keeps the matrix pipe busy).
// Per group: the V fragments of the NEXT group are read first and split piecewise between this group's MFMAs; the weight
// operands are reloaded term by term as soon as the last MFMA that uses a term has issued (lo after 2, mid after 6, hi after 12).
// DBG (ablation builds only, GIF_WINO_DBG; results are wrong): 1 = no fold, 2 = no split of V, 4 = no weight-operand reloads
// Block BM x BN = 128 x 128 (4 x 2 waves; default) or 256 x 64 (8 x 1 waves: every V fragment is split by exactly one wave).
template <int DBG = 0, int BM = 256, int BN = 64>
__global__ void __launch_bounds__(512, 1) wino_gemm_x3(const WinoParams p) {
    if (p.gate) {  // guarded fallback of an f16x2 launch: nothing to do unless that launch raised the gate
        if (*p.gate < p.gate_gen) return;
        if (blockIdx.x == 0 && threadIdx.x == 0 && p.h2_stats) atomicAdd(p.h2_stats, 1u);
    }
    constexpr int THREADS = 512, PROWS = THREADS / 8;
    constexpr int LD = WBK, CH = WBK / 4, RB = 64 / WBK, RPW = 64 / CH;
    constexpr int NT = 2, WAVES_N = BN / 64;
    static_assert(BM * BN == 128 * 128 && (8 / WAVES_N) * 32 == BM, "8 waves of 32 tiles x 64 couts");
    constexpr int A_IT = BM / PROWS;                  // 4 / 2
    constexpr int B3_BLK = 3 * BN / 16;               // 12 / 24 one-KiB blocks (16 rows x 64 B) per stage
    constexpr int B3_IT = (B3_BLK + 7) / 8;           // 2 (waves 4-7: 1) / 3 per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                                                    // [NSTAGE][BM][LD] fp32
    unsigned short* B3 = reinterpret_cast<unsigned short*>(smem + WNSTAGE * BM * LD);   // [NSTAGE][3][BN][32] bf16
    const unsigned short* const U3 = reinterpret_cast<const unsigned short*>(p.U);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm0 = (wave / WAVES_N) * 32, wn0 = (wave % WAVES_N) * 64;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int t_row = tid / CH;
    const int src_c4 = ((tid % CH) ^ ((t_row / RB) % CH)) * 4;
    const int fsw = (li / RB) % CH;
    const int b3_sw = (li >> 2) & 3;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const unsigned a_off = (unsigned)(m0 + t_row) * (unsigned)p.CP + (unsigned)src_c4;
    const size_t pass_stride = (size_t)PROWS * p.CP;
    const size_t planeV = (size_t)p.ntiles_pad * p.CP, planeU = (size_t)p.RP * p.CP;  // planeU: one TERM plane of one position
    unsigned b3_off[B3_IT];
#pragma unroll
    for (int it = 0; it < B3_IT; ++it) {
        const int blk = wave + it * 8;
        const int term = blk / (BN / 16), r = (blk % (BN / 16)) * 16 + (lane >> 2);
        b3_off[it] = (unsigned)((size_t)term * planeU + (size_t)(n0 + r) * p.CP + (((lane & 3) ^ ((lane >> 4) & 3)) << 3));
    }
    const int kchunks = p.CP / WBK;
    const int nsteps = 16 * kchunks;
    const float* vptr = p.V;
    const unsigned short* uptr = U3;
    int ld_kc = 0;

    auto issue = [&](int buf) __attribute__((always_inline)) {
        float* Ad = As + buf * BM * LD + wave * RPW * LD;
#pragma unroll
        for (int it = 0; it < A_IT; ++it)
            __builtin_amdgcn_global_load_lds((gptr_t)((vptr + it * pass_stride) + a_off), (lptr_t)(Ad + it * PROWS * LD), 16, 0, 0);
#pragma unroll
        for (int it = 0; it < B3_IT; ++it)
            if (B3_BLK % 8 == 0 || wave + it * 8 < B3_BLK)  // wave-uniform
                __builtin_amdgcn_global_load_lds((gptr_t)(uptr + b3_off[it]), (lptr_t)(B3 + (buf * B3_BLK + wave + it * 8) * 512), 16, 0, 0);
        vptr += WBK;
        uptr += WBK;
        ld_kc += WBK;
        if (ld_kc >= p.CP) {  // next position: same rows of the next plane (U3: skip the three term planes)
            ld_kc = 0;
            vptr += planeV - p.CP;
            uptr += 3 * planeU - p.CP;
        }
    };

    f32x16 acc[NT];
    f32x16 yo[4][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[j][r] = 0.f;
#pragma unroll
            for (int o = 0; o < 4; ++o) yo[o][j][r] = 0.f;
        }

    gif::u32x4_t sa[2][3];   // [slot][hi, mid, lo] of the V fragment
    gif::u32x4_t sb[3][NT];  // [hi, mid, lo][cout tile]
    f32x4 ra[2];             // raw V fragments of the group being split

    auto read_a = [&](int buf, int q) __attribute__((always_inline)) {
        const float* Ab = As + buf * BM * LD + (wm0 + li) * LD;
#pragma unroll
        for (int u = 0; u < 2; ++u) ra[u] = *reinterpret_cast<const f32x4*>(Ab + (((q * 4 + lh * 2 + u) ^ fsw) << 2));
    };
    auto read_b = [&](int buf, int q, int t) __attribute__((always_inline)) {
        const unsigned short* Bb = B3 + (buf * 3 + t) * BN * 32 + (wn0 + li) * 32 + (((q * 2 + lh) ^ b3_sw) << 3);
#pragma unroll
        for (int j = 0; j < NT; ++j) sb[t][j] = *reinterpret_cast<const gif::u32x4_t*>(Bb + j * 32 * 32);
    };
    auto split_piece = [&](int slot, int e) __attribute__((always_inline)) {
        unsigned h, m, l;
        gif::split_pair_scalar(ra[e / 2][(e % 2) * 2], ra[e / 2][(e % 2) * 2 + 1], h, m, l);
        sa[slot][0][e] = h; sa[slot][1][e] = m; sa[slot][2][e] = l;
    };
    auto mma = [&](int slot, int ta, int tb) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gif::bf16x8_t, sa[slot][ta]),
                                                             __builtin_bit_cast(gif::bf16x8_t, sb[tb][j]), acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // one 16-k group: MFMAs of (slot) + operand preparation of the next group (V fragments of (rbuf, nq) -> slot ^ 1, weights)
    auto group = [&](int slot, int rbuf, int nq) __attribute__((always_inline)) {
        read_a(rbuf, nq);
        __builtin_amdgcn_sched_barrier(0);
        if (GIF_X3_FIRST_TERM == 0) mma(slot, 0, 2);  // hi * lo
        if (!(DBG & 4)) read_b(rbuf, nq, 2);
        __builtin_amdgcn_sched_barrier(0);
        if (GIF_X3_FIRST_TERM == 0) mma(slot, 1, 1);  // mid * mid
        if (!(DBG & 2)) split_piece(slot ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(slot, 0, 1);  // hi * mid
        if (!(DBG & 4)) read_b(rbuf, nq, 1);
        if (!(DBG & 2)) split_piece(slot ^ 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (GIF_X3_FIRST_TERM == 0) mma(slot, 2, 0);  // lo * hi
        if (!(DBG & 2)) split_piece(slot ^ 1, 2);
        __builtin_amdgcn_sched_barrier(0);
        mma(slot, 1, 0);  // mid * hi
        if (!(DBG & 2)) split_piece(slot ^ 1, 3);
        __builtin_amdgcn_sched_barrier(0);
        mma(slot, 0, 0);  // hi * hi
        if (!(DBG & 4)) read_b(rbuf, nq, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto fold = [&](int pos) __attribute__((always_inline)) {
        const int xi = pos >> 2, nu = pos & 3;
        float cf[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) cf[o] = wino_coef(o >> 1, xi) * wino_coef(o & 1, nu);
        // 36 of the 64 (position, output) coefficients are non-zero: skip the others (wave-uniform branches)
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (cf[o] != 0.f) {
#pragma unroll
                for (int j = 0; j < NT; ++j) yo[o][j] += cf[o] * acc[j];
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = (f32x16)(0.f);
    };

    // DMA instructions per stage and wave: 256x64: 4 + 2 (waves 0-3) / 4 + 1 (waves 4-7); 128x128: 2 + 3
    static_assert((BM == 256 && A_IT == 4 && B3_BLK == 12) || (BM == 128 && A_IT == 2 && B3_BLK == 24), "the s_waitcnt immediates");
    auto wait_newest_in_flight = [&]() __attribute__((always_inline)) {
        if (BM == 128) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (wave < 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    };
    issue(0);
    issue(1);  // nsteps >= 16
    wait_newest_in_flight();
    __builtin_amdgcn_s_barrier();
    read_a(0, 0);
#pragma unroll
    for (int t = 0; t < 3; ++t) read_b(0, 0, t);
#pragma unroll
    for (int e = 0; e < 4; ++e) split_piece(0, e);

    int cur = 0, kc_in_pos = 0, pos = 0;
    for (int step = 0; step < nsteps; ++step) {
        if (step + 2 < nsteps) issue(cur >= 1 ? cur - 1 : 2);  // (cur + 2) % 3: last read before the previous stage's barrier
        __builtin_amdgcn_sched_barrier(0);
        group(0, cur, 1);
        // stage step+1 must have landed before its operands are read (this stage's are all in registers by now)
        if (step + 2 < nsteps) wait_newest_in_flight();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur = cur == 2 ? 0 : cur + 1;
        group(1, cur, 0);  // (after the last stage this prepares operands nobody uses: the reads stay inside the ring)
        if (++kc_in_pos == kchunks) {
            if (!(DBG & 1)) fold(pos);
            kc_in_pos = 0;
            ++pos;
        }
    }
    if (DBG & 1) fold(5);

    // ---- epilogue: for each output position (a,b): transpose through LDS, then coalesced float4 rows
    constexpr int LDC = BN + 4;
    float* Cs = smem;  // [BM][LDC]
    constexpr int C4_ROW = BN / 4, EROWS = THREADS / C4_ROW, E_IT = BM / EROWS;
    const int e_row0 = tid / C4_ROW, e_c = (tid % C4_ROW) * 4;
    const int n = n0 + e_c;
    f32x4 bias4 = (f32x4)(0.f);
    if (p.bias && n < p.Co) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
    f32x4 cs = (f32x4)(0.f), ds = (f32x4)(0.f);
    const bool fused = p.mask_src || p.dot_src || p.part_cs || p.part_dot;  // workgroup-uniform
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = wm0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                Cs[row * LDC + wn0 + j * 32 + li] = yo[o][j][r];
            }
        __syncthreads();
        if (n < p.Co) {
            if (fused) wino_epilogue_rows<true, E_IT, EROWS, LDC, (E_IT < 4 ? E_IT : 4)>(p, Cs, m0, e_row0, e_c, n, o >> 1, o & 1, bias4, cs, ds);
            else wino_epilogue_rows<false, E_IT, EROWS, LDC, (E_IT < 4 ? E_IT : 4)>(p, Cs, m0, e_row0, e_c, n, o >> 1, o & 1, bias4, cs, ds);
        }
    }
    wino_write_partials<THREADS, C4_ROW, EROWS>(p, smem, tid, n, tm, cs, ds);
}

// ------------------------------------------------------------------------------------------------ f16x2 GEMM (round 5)
// The same GEMM + fused output transform on v_mfma_f32_32x32x16_f16 with THREE products per fp32 product (common.h "h2").
// V stays fp32 in HBM and LDS (the weight gradient reuses it); a V fragment is split into two f16 terms under the RUNNING exponent of
// its row (= Winograd tile), which persists over all 16 positions: the four 2x2-output accumulators then all live in ONE scale per
// row and are brought back to fp32's own scale once, in the epilogue (a per-position exponent would cost 32 ldexp + 16 lane
// exchanges per position: a quarter of a 128-channel position's MFMA time).  When a row outgrows its exponent the position
// accumulator AND the four output accumulators of that row are multiplied by the exact power of two (160 registers, rare).
// U2 = gif_winograd_weight_f32h2: [16][2][RP][CP] f16 planes of G g G^T * 2^e_row, one exponent per output channel over all positions.
// Block 128 x 128, 8 waves as 4 (M) x 2 (N), wave tile 32 tiles x 64 couts, 3-stage ring: 48 KB of fp32 V + 48 KB of f16 planes.
template <int BM = 128, int BN = 128, int NST = 3>
__global__ void __launch_bounds__(512, 1) wino_gemm_h2(const WinoParams p) {
    constexpr int THREADS = 512, PROWS = THREADS / 8;
    constexpr int LD = WBK, CH = WBK / 4, RB = 64 / WBK, RPW = 64 / CH;
    constexpr int NT = 2, WAVES_N = BN / 64;
    static_assert(BM == 128 && BN == 128, "8 waves of 32 tiles x 64 couts as 4 x 2");
    constexpr int A_IT = BM / PROWS;        // 2
    constexpr int B2_BLK = 2 * BN / 16;     // 16 one-KiB blocks (16 rows x 64 B) per stage
    constexpr int B2_IT = B2_BLK / 8;       // 2 per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                                                    // [NST][BM][LD] fp32
    unsigned short* B2 = reinterpret_cast<unsigned short*>(smem + NST * BM * LD);       // [NST][2][BN][32] f16
    const unsigned short* const U2 = reinterpret_cast<const unsigned short*>(p.U);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm0 = (wave / WAVES_N) * 32, wn0 = (wave % WAVES_N) * 64;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int t_row = tid / CH;
    const int src_c4 = ((tid % CH) ^ ((t_row / RB) % CH)) * 4;
    const int fsw = (li / RB) % CH;
    const int b2_sw = (li >> 2) & 3;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const unsigned a_off = (unsigned)(m0 + t_row) * (unsigned)p.CP + (unsigned)src_c4;
    const size_t pass_stride = (size_t)PROWS * p.CP;
    const size_t planeV = (size_t)p.ntiles_pad * p.CP, planeU = (size_t)p.RP * p.CP;  // planeU: one TERM plane of one position
    unsigned b2_off[B2_IT];
#pragma unroll
    for (int it = 0; it < B2_IT; ++it) {
        const int blk = wave + it * 8;
        const int term = blk / (BN / 16), r = (blk % (BN / 16)) * 16 + (lane >> 2);
        b2_off[it] = (unsigned)((size_t)term * planeU + (size_t)(n0 + r) * p.CP + (((lane & 3) ^ ((lane >> 4) & 3)) << 3));
    }
    const int kchunks = p.CP / WBK;
#ifdef GIF_WINO_F4_PROBE  // timing probe (tools/probes/wino_f4_probe.py; results are meaningless): the GEMM of an UNFUSED F(4x4,3x3) — 36
    // position GEMMs over a quarter of the rows, every position's accumulator stored as its own M plane (through p.residual) instead of
    // being folded into 2x2 outputs; no output transform, no epilogue
    const int nsteps = 36 * kchunks;
#else
    const int nsteps = 16 * kchunks;
#endif
    const float* vptr = p.V;
    const unsigned short* uptr = U2;
    int ld_kc = 0;

    auto issue = [&](int buf) __attribute__((always_inline)) {
        float* Ad = As + buf * BM * LD + wave * RPW * LD;
#pragma unroll
        for (int it = 0; it < A_IT; ++it)
            __builtin_amdgcn_global_load_lds((gptr_t)((vptr + it * pass_stride) + a_off), (lptr_t)(Ad + it * PROWS * LD), 16, 0, 0);
#pragma unroll
        for (int it = 0; it < B2_IT; ++it)
            __builtin_amdgcn_global_load_lds((gptr_t)(uptr + b2_off[it]), (lptr_t)(B2 + (buf * B2_BLK + wave + it * 8) * 512), 16, 0, 0);
        vptr += WBK;
        uptr += WBK;
        ld_kc += WBK;
        if (ld_kc >= p.CP) {  // next position: same rows of the next plane (U2: skip the two term planes)
            ld_kc = 0;
            vptr += planeV - p.CP;
            uptr += 2 * planeU - p.CP;
        }
    };

    f32x16 acc[NT];
    f32x16 yo[4][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[j][r] = 0.f;
#pragma unroll
            for (int o = 0; o < 4; ++o) yo[o][j][r] = 0.f;
        }

    gif::u32x4_t sa[2][2];   // [slot][hi, lo] of the V fragment
    gif::u32x4_t sb[2][NT];  // [hi, lo][cout tile]
    f32x4 ra[2];             // raw V fragments of the group being split
    // running exponent of this lane's row (lanes li and li + 32 agree) and the guard's statistics (common.h)
    int h_ex = 126, h_dl = 0;
    float h_sc = gif::h2_pow2(126), h_lim = gif::kH2Limit * gif::h2_pow2(-126), h_max = 0.f;
    unsigned h_gmin = 0xFFFFFFFFu;
    bool h_need = false;

    auto read_a = [&](int buf, int q) __attribute__((always_inline)) {
        const float* Ab = As + buf * BM * LD + (wm0 + li) * LD;
#pragma unroll
        for (int u = 0; u < 2; ++u) ra[u] = *reinterpret_cast<const f32x4*>(Ab + (((q * 4 + lh * 2 + u) ^ fsw) << 2));
    };
    auto read_b = [&](int buf, int q, int t) __attribute__((always_inline)) {
        const unsigned short* Bb = B2 + (buf * 2 + t) * BN * 32 + (wn0 + li) * 32 + (((q * 2 + lh) ^ b2_sw) << 3);
#pragma unroll
        for (int j = 0; j < NT; ++j) sb[t][j] = *reinterpret_cast<const gif::u32x4_t*>(Bb + j * 32 * 32);
    };
    auto track = [&]() __attribute__((always_inline)) {
        float m = fmaxf(fmaxf(fabsf(ra[0][0]), fabsf(ra[0][1])), fabsf(ra[0][2]));
        m = fmaxf(fmaxf(m, fabsf(ra[0][3])), fabsf(ra[1][0]));
        m = fmaxf(fmaxf(m, fabsf(ra[1][1])), fabsf(ra[1][2]));
        m = fmaxf(m, fabsf(ra[1][3]));
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
        m = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        h_max = fmaxf(h_max, m);
        h_gmin = min(h_gmin, __float_as_uint(m) - 1u);
        h_dl = 0;
        h_need = false;
        if (__builtin_amdgcn_ballot_w64(m > h_lim) != 0) {  // wave-uniform, rare
            const int ne = m > h_lim ? gif::h2_exp_for(__float_as_uint(m), gif::kH2Target) : h_ex;
            h_dl = ne - h_ex;
            h_ex = ne;
            h_sc = gif::h2_pow2(ne);
            h_lim = ldexpf(gif::kH2Limit, -ne);
            h_need = true;
        }
    };
    // rows whose exponent changed: the position accumulator and the four output accumulators *= 2^(e' - e) (exact)
    auto rescale = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = __builtin_amdgcn_ds_bpermute(((r & 3) + 8 * (r >> 2) + 4 * lh) * 4, h_dl);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                acc[j][r] = ldexpf(acc[j][r], d);
#pragma unroll
                for (int o = 0; o < 4; ++o) yo[o][j][r] = ldexpf(yo[o][j][r], d);
            }
        }
    };
    auto split_piece = [&](int slot, int e) __attribute__((always_inline)) {
        unsigned h, l;
        gif::split_pair_h2(ra[e / 2][(e % 2) * 2], ra[e / 2][(e % 2) * 2 + 1], h_sc, h, l);
        sa[slot][0][e] = h; sa[slot][1][e] = l;
    };
    auto mma = [&](int slot, int ta, int tb) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gif::f16x8_t, sa[slot][ta]),
                                                            __builtin_bit_cast(gif::f16x8_t, sb[tb][j]), acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // one 16-k group: the 6 MFMAs of (slot) + operand preparation of the next group (V fragments of (rbuf, nq) -> slot ^ 1, weights)
    auto group = [&](int slot, int rbuf, int nq) __attribute__((always_inline)) {
        read_a(rbuf, nq);
        __builtin_amdgcn_sched_barrier(0);
        mma(slot, 1, 0);  // lo * hi
        track();
        __builtin_amdgcn_sched_barrier(0);
        split_piece(slot ^ 1, 0);
        split_piece(slot ^ 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(slot, 0, 1);  // hi * lo
        read_b(rbuf, nq, 1);
        split_piece(slot ^ 1, 2);
        split_piece(slot ^ 1, 3);
        __builtin_amdgcn_sched_barrier(0);
        mma(slot, 0, 0);  // hi * hi
        read_b(rbuf, nq, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (h_need) rescale();
    };
    auto fold = [&](int pos) __attribute__((always_inline)) {
        const int xi = pos >> 2, nu = pos & 3;
        float cf[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) cf[o] = wino_coef(o >> 1, xi) * wino_coef(o & 1, nu);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (cf[o] != 0.f) {
#pragma unroll
                for (int j = 0; j < NT; ++j) yo[o][j] += cf[o] * acc[j];
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = (f32x16)(0.f);
    };

    // DMA instructions per stage and wave: 2 (V) + 2 (planes); NST - 2 stages stay in flight across the barrier (ring of NST stages)
    auto wait_newest_in_flight = [&]() __attribute__((always_inline)) {
        if (NST == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (NST == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    };
    static_assert(NST >= 3 && NST <= 5, "ring depth");
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) issue(st);  // nsteps >= 16
    wait_newest_in_flight();
    __builtin_amdgcn_s_barrier();
    read_a(0, 0);
    read_b(0, 0, 0);
    read_b(0, 0, 1);
    track();  // first exponents (the accumulators are zero)
#pragma unroll
    for (int e = 0; e < 4; ++e) split_piece(0, e);

    int cur = 0, kc_in_pos = 0, pos = 0;
    for (int step = 0; step < nsteps; ++step) {
        if (step + NST - 1 < nsteps) issue(cur >= 1 ? cur - 1 : NST - 1);  // (cur + NST - 1) % NST: last read before the previous stage's barrier
        __builtin_amdgcn_sched_barrier(0);
        group(0, cur, 1);
        // stage step + 1 must have landed before its operands are read (this stage's are all in registers by now)
        if (step + NST - 1 < nsteps) wait_newest_in_flight();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur = cur == NST - 1 ? 0 : cur + 1;
        group(1, cur, 0);  // (after the last stage this prepares operands nobody uses: the reads stay inside the ring)
        if (++kc_in_pos == kchunks) {
#ifdef GIF_WINO_F4_PROBE
            {
                float* Mp = const_cast<float*>(p.residual) + (size_t)pos * p.ntiles_pad * p.RP;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int er = __builtin_amdgcn_ds_bpermute(((r & 3) + 8 * (r >> 2) + 4 * lh) * 4, h_ex);
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        Mp[(size_t)row * p.RP + n0 + wn0 + j * 32 + li] = ldexpf(acc[j][r], -er);
                        acc[j][r] = 0.f;
                    }
                }
            }
#else
            fold(pos);
#endif
            kc_in_pos = 0;
            ++pos;
        }
    }
#ifdef GIF_WINO_F4_PROBE
    return;
#endif

    // guard (common.h): a 16-element K group more than 2^kH2Window below its row's maximum meeting a weight row the packing flagged
    // raises the launch's gate
    {
        // (the tracking step after the last stage looked at ring data nobody uses: its statistics may only widen the window check,
        // never narrow it — a spurious fallback at worst; rows >= ntiles are padding and may hold anything: same remark)
        const bool wide = m0 + wm0 + li < p.ntiles && (int)(__float_as_uint(h_max) >> 23) - (int)((h_gmin + 1u) >> 23) > gif::kH2Window;
        bool wflag = false;
#pragma unroll
        for (int j = 0; j < NT; ++j) wflag |= p.uexp[p.RP + n0 + wn0 + j * 32 + li] != 0;
        // (both operands narrow somewhere: common.h — smooth activations give near-zero groups at the differencing positions of B^T d B
        // all the time, which costs nothing against in-window weights)
        if (p.gate && __builtin_amdgcn_ballot_w64(wide) != 0 && __builtin_amdgcn_ballot_w64(wflag) != 0 && lane == 0) atomicMax(p.gate, p.gate_gen);
    }
    // back to fp32's own scale: yo[o][row][col] *= 2^-(e_row + e_col)
    {
        int wex[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) wex[j] = p.uexp[n0 + wn0 + j * 32 + li];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int er = __builtin_amdgcn_ds_bpermute(((r & 3) + 8 * (r >> 2) + 4 * lh) * 4, h_ex);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int o = 0; o < 4; ++o) yo[o][j][r] = ldexpf(yo[o][j][r], -(er + wex[j]));
        }
    }

    // ---- epilogue: for each output position (a,b): transpose through LDS, then coalesced float4 rows
    constexpr int LDC = BN + 4;
    float* Cs = smem;  // [BM][LDC]
    constexpr int C4_ROW = BN / 4, EROWS = THREADS / C4_ROW, E_IT = BM / EROWS;
    const int e_row0 = tid / C4_ROW, e_c = (tid % C4_ROW) * 4;
    const int n = n0 + e_c;
    f32x4 bias4 = (f32x4)(0.f);
    if (p.bias && n < p.Co) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
    f32x4 cs = (f32x4)(0.f), ds = (f32x4)(0.f);
    const bool fused = p.mask_src || p.dot_src || p.part_cs || p.part_dot;  // workgroup-uniform
    // row table behind the staging tile (the ring is larger): written between the first pair of barriers, read by all four positions
    static_assert((BM * LDC + 3 * BM) * 4 <= NST * (BM * LD * 4 + 2 * BN * 64), "row table must fit the ring");
    unsigned long long* const rtab = reinterpret_cast<unsigned long long*>(Cs + BM * LDC);  // [BM]
    int* const rsmp = reinterpret_cast<int*>(rtab + BM);                                      // [BM]
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        __syncthreads();
        if (o == 0 && tid < BM) {
            const int m = m0 + tid;
            const bool okr = m < p.ntiles;
            const int mm = okr ? m : 0;
            const int tx = mm % p.TW, t2 = mm / p.TW;
            const int ty = t2 % p.TH, b = t2 / p.TH;
            rtab[tid] = okr ? (unsigned long long)((((size_t)b * p.H + 2 * ty) * p.W + 2 * tx) * p.Co) : ~0ull;
            rsmp[tid] = b;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = wm0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                Cs[row * LDC + wn0 + j * 32 + li] = yo[o][j][r];
            }
        __syncthreads();
        if (n < p.Co) {
            if (fused) wino_epilogue_rows<true, E_IT, EROWS, LDC, (E_IT < 4 ? E_IT : 4)>(p, Cs, m0, e_row0, e_c, n, o >> 1, o & 1, bias4, cs, ds, rtab, rsmp);
            else wino_epilogue_rows<false, E_IT, EROWS, LDC, (E_IT < 4 ? E_IT : 4)>(p, Cs, m0, e_row0, e_c, n, o >> 1, o & 1, bias4, cs, ds, rtab, rsmp);
        }
    }
    wino_write_partials<THREADS, C4_ROW, EROWS>(p, smem, tid, n, tm, cs, ds);
}

// U2 = the two f16 terms of G g G^T * 2^e_row: header [RP] exponents + [RP] flags (int32), then planes [16][2][RP][CP] (f16 bits).
// One workgroup per output-channel row: the row maximum runs over all 16 positions and all channels (common.h "h2"); a 16-channel
// group of one position more than 2^kH2WindowW below it sets the row's flag.
// U3 != NULL: the bf16x3 transform [16][3][RP][CP] of the same weights (the guarded fallback's operand) in the same pass.
__global__ void __launch_bounds__(256) wino_weight_transform_h2(const float* __restrict__ w, int* __restrict__ hdr, unsigned short* __restrict__ planes,
                                                                unsigned short* __restrict__ U3, int R, int C, int RP, int CP, long sr, long sc,
                                                                long sky, long skx, int flip, float scale) {
    __shared__ float red[256];
    __shared__ int s_flag;
    const int r = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) s_flag = 0;
    auto transform = [&](int c, float (&v)[16]) {
        float g[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                int sy = flip ? 2 - ky : ky, sx = flip ? 2 - kx : kx;
                g[ky][kx] = (r < R && c < C) ? scale * w[r * sr + c * sc + sy * sky + sx * skx] : 0.f;
            }
        float u[4][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            u[0][j] = g[0][j];
            u[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
            u[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
            u[3][j] = g[2][j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i * 4 + 0] = u[i][0];
            v[i * 4 + 1] = 0.5f * (u[i][0] + u[i][1] + u[i][2]);
            v[i * 4 + 2] = 0.5f * (u[i][0] - u[i][1] + u[i][2]);
            v[i * 4 + 3] = u[i][2];
        }
    };
    float m = 0.f;
    for (int c = tid; c < CP; c += 256) {
        float v[16];
        transform(c, v);
#pragma unroll
        for (int q = 0; q < 16; ++q) m = fmaxf(m, fabsf(v[q]));
    }
    red[tid] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
        __syncthreads();
    }
    const float rowmax = red[0];
    const int e = gif::h2_exp_for(__float_as_uint(rowmax), gif::kH2TargetW);
    const float sc2 = gif::h2_pow2(e);
    const size_t plane = (size_t)RP * CP;
    bool narrow = false;
    for (int c = tid; c < CP; c += 256) {  // (CP is a multiple of 32: whole 16-lane groups stay together)
        float v[16];
        transform(c, v);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            unsigned h, l;
            gif::split_pair_h2(v[q], 0.f, sc2, h, l);
            unsigned short* o = planes + (size_t)q * 2 * plane + (size_t)r * CP + c;
            o[0] = (unsigned short)(h & 0xffffu);
            o[plane] = (unsigned short)(l & 0xffffu);
            if (U3) {
                unsigned m3;
                gif::split_pair(v[q], 0.f, h, m3, l);
                unsigned short* o3 = U3 + (size_t)q * 3 * plane + (size_t)r * CP + c;
                o3[0] = (unsigned short)(h & 0xffffu);
                o3[plane] = (unsigned short)(m3 & 0xffffu);
                o3[2 * plane] = (unsigned short)(l & 0xffffu);
            }
            float gm = fabsf(v[q]);  // maximum over the 16 channels of this lane's group
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) gm = fmaxf(gm, __shfl_xor(gm, d, 16));
            if (gm > 0.f && (int)(__float_as_uint(rowmax) >> 23) - (int)(__float_as_uint(gm) >> 23) > gif::kH2WindowW) narrow = true;
        }
    }
    if (narrow) atomicOr(&s_flag, 1);
    __syncthreads();
    if (tid == 0) {
        hdr[r] = e;
        hdr[RP + r] = s_flag;
    }
}

// U3 = the three bf16 terms of G g G^T: [16][3][RP][CP]
__global__ void wino_weight_transform_x3(const float* __restrict__ w, unsigned short* __restrict__ U3, int R, int C, int RP, int CP,
                                         long sr, long sc, long sky, long skx, int flip, float scale) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)RP * CP) return;
    int c = (int)(idx % CP), r = (int)(idx / CP);
    float g[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            int sy = flip ? 2 - ky : ky, sx = flip ? 2 - kx : kx;
            g[ky][kx] = (r < R && c < C) ? scale * w[r * sr + c * sc + sy * sky + sx * skx] : 0.f;
        }
    float u[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        u[0][j] = g[0][j];
        u[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
        u[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
        u[3][j] = g[2][j];
    }
    const size_t plane = (size_t)RP * CP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float v4[4] = {u[i][0], 0.5f * (u[i][0] + u[i][1] + u[i][2]), 0.5f * (u[i][0] - u[i][1] + u[i][2]), u[i][2]};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned h, m, l;
            gif::split_pair(v4[q], 0.f, h, m, l);
            unsigned short* o = U3 + (size_t)(i * 4 + q) * 3 * plane + idx;
            o[0] = (unsigned short)(h & 0xffffu);
            o[plane] = (unsigned short)(m & 0xffffu);
            o[2 * plane] = (unsigned short)(l & 0xffffu);
        }
    }
}

}  // namespace

namespace gif {

void winograd_padded_dims(long ntiles, int C, long* ntiles_pad, int* CP) {
    *ntiles_pad = (ntiles + WPAD - 1) / WPAD * WPAD;
    *CP = (C + WBK - 1) / WBK * WBK;
}

static unsigned transform_blocks(long total) {
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    return (unsigned)blocks;
}

int winograd_input_transform(const float* x, const float* scale, float* V, int B, int H, int W, int C, hipStream_t s) {
    const long ntiles = (long)B * (H / 2) * (W / 2);
    long ntiles_pad;
    int CP;
    winograd_padded_dims(ntiles, C, &ntiles_pad, &CP);
    // The vertical-reuse variant with 8 tiles per lane is the default (profiles/r4_wino_transform_rows.txt: 23.8 -> 23.0 ms/step, HBM bytes
    // of the transform family 1.10x -> 1.065x the algorithmic count); GIF_WINO_XFORM=tile selects the one-patch-per-lane kernel,
    // rows4 / rows16 the other depths (A/B).
    static const int rows = [] {
        const char* e = gif::knob("GIF_WINO_XFORM");
        if (!e) return 8;
        return !strncmp(e, "rows", 4) ? atoi(e + 4) : 0;
    }();
    if (rows == 4 || rows == 8 || rows == 16) {
        const long lanes = (long)B * ((H / 2 + rows - 1) / rows) * (W / 2) * (CP / 4);
        if (rows == 4) wino_input_transform_rows<4><<<transform_blocks(lanes), 256, 0, s>>>(x, scale, V, B, H, W, C, CP, ntiles_pad);
        else if (rows == 8) wino_input_transform_rows<8><<<transform_blocks(lanes), 256, 0, s>>>(x, scale, V, B, H, W, C, CP, ntiles_pad);
        else wino_input_transform_rows<16><<<transform_blocks(lanes), 256, 0, s>>>(x, scale, V, B, H, W, C, CP, ntiles_pad);
        return check_launch("winograd_input_transform(rows)");
    }
    wino_input_transform<<<transform_blocks(ntiles * (CP / 4)), 256, 0, s>>>(x, scale, V, B, H, W, C, CP, ntiles_pad);
    return check_launch("winograd_input_transform");
}

int winograd_gy_transform(const float* gy, const float* scale, float* Mg, int B, int H, int W, int C, hipStream_t s) {
    const long ntiles = (long)B * (H / 2) * (W / 2);
    long ntiles_pad;
    int CP;
    winograd_padded_dims(ntiles, C, &ntiles_pad, &CP);
    wino_gy_transform<<<transform_blocks(ntiles * (CP / 4)), 256, 0, s>>>(gy, scale, Mg, B, H, W, C, CP, ntiles_pad);
    return check_launch("winograd_gy_transform");
}

}  // namespace gif

extern "C" {

// RP / CP of the transformed weight U [16][RP][CP] for `cout` / `cin` channels
int gif_winograd_pack_dims(int cout, int cin, int* RP, int* CP) {
    GIF_REQUIRE(cout > 0 && cin > 0 && RP && CP, "winograd_pack_dims: bad arguments");
    *RP = (cout + WBN - 1) / WBN * WBN;
    *CP = (cin + WBK - 1) / WBK * WBK;
    return 0;
}

// floats of the V scratch of gif_conv3x3_winograd_f32: 16 planes x (tiles padded to the GEMM's M block) x padded channels
int64_t gif_winograd_workspace_floats(int B, int H, int W, int C) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    const int64_t ntiles = (int64_t)B * (H / 2) * (W / 2);
    const int64_t ntiles_pad = (ntiles + WPAD - 1) / WPAD * WPAD;
    const int64_t CP = (C + WBK - 1) / WBK * WBK;
    return 16 * ntiles_pad * CP;
}

int gif_winograd_weight_f32(const float* w, float* U, int R, int C, int RP, int CP, int64_t sr, int64_t sc, int64_t sky,
                            int64_t skx, int flip, float scale, gif_stream_t stream) {
    GIF_REQUIRE(w && U && R > 0 && C > 0 && RP >= R && CP >= C, "winograd_weight: bad arguments");
    long total = (long)RP * CP;
    wino_weight_transform<<<gif::cdiv(total, 256), 256, 0, gif::as_stream(stream)>>>(w, U, R, C, RP, CP, sr, sc, sky, skx,
                                                                                      flip, scale);
    return gif::check_launch("winograd_weight");
}

}  // extern "C"

namespace {

// gradient-producer fusions of a Winograd launch: partial-sum rows = GEMM row tiles of `bm` Winograd tiles each
struct WinoSums {
    float *colsum = nullptr, *dot = nullptr, *tmp = nullptr;
    int begin(WinoParams& p, const gif_conv_epilogue* e, int B, int H, int W, int Co, int bm, const char* who) {
        p.mask_src = e ? static_cast<const float*>(e->mask_src) : nullptr;
        p.mask_slope = e ? e->mask_slope : 1.f;
        p.mask_gain = e ? e->mask_gain : 1.f;
        p.dot_src = e ? static_cast<const float*>(e->dot_src) : nullptr;
        if (!e || (!e->colsum && !e->dot)) return 0;
        GIF_REQUIRE(e->red_ws, "%s: colsum / dot need the red_ws workspace (gif_conv_epilogue_ws_floats)", who);
        const long per_sample = (long)(H / 2) * (W / 2);
        GIF_REQUIRE(!e->dot || (e->dot_src && per_sample % bm == 0),
                    "%s: the dot fusion needs dot_src and a multiple of %d 2x2 tiles per sample (got %ld)", who, bm, per_sample);
        const long cap = (long)B * H * W / 64 + 16;
        GIF_REQUIRE(p.tiles_m <= cap, "%s: partial-sum rows exceed the workspace", who);
        colsum = e->colsum;
        dot = e->dot;
        p.part_cs = colsum ? e->red_ws : nullptr;
        p.part_dot = dot ? e->red_ws + cap * Co : nullptr;
        tmp = e->red_ws + 2 * cap * Co;
        return 0;
    }
    int finish(const WinoParams& p, int B, int H, int W, int Co, int bm, hipStream_t s) {
        if (colsum)
            if (int rc = gif::reduce_partials(p.part_cs, colsum, 1, p.tiles_m, Co, tmp, s)) return rc;
        if (dot) {
            // the row padding of the last tile block holds no pixels: ntiles = B * per_sample is a multiple of bm here
            const int per = (int)((long)(H / 2) * (W / 2) / bm);
            if (int rc = gif::reduce_partials(p.part_dot, dot, B, per, Co, tmp, s)) return rc;
        }
        return 0;
    }
};

}  // namespace

extern "C" {

// y [B,H,W,Co] = act(out_scale * conv3x3_s1_p1(in_scale * x [B,H,W,C], weights behind U) + residual + bias).
// V is scratch of gif_winograd_workspace_floats(B,H,W,C) floats.  H, W even; C, Co multiples of 4.
int gif_conv3x3_winograd_f32(const float* x, const float* U, float* y, float* V, int B, int H, int W, int C, int Co,
                             const gif_conv_epilogue* e, gif_stream_t stream) {
    GIF_REQUIRE(x && U && y && V && B >= 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "winograd: bad dims (H, W must be even)");
    GIF_REQUIRE(C > 0 && Co > 0 && C % 4 == 0 && Co % 4 == 0, "winograd: channels must be multiples of 4");
    if (B == 0) return 0;
    const long ntiles = (long)B * (H / 2) * (W / 2);
    WinoParams p{};
    gif_winograd_pack_dims(Co, C, &p.RP, &p.CP);
    const long ntiles_pad = (ntiles + WPAD - 1) / WPAD * WPAD;
    GIF_REQUIRE(ntiles_pad * p.CP < (1L << 31) && (long)B * H * W * Co < (1L << 31) && (long)B * H * W * C < (1L << 31),
                "winograd: tensor too large for 32-bit offsets");
    hipStream_t s = gif::as_stream(stream);
    double flops = 2.0 * B * H * W * 9.0 * C * Co;  // ALGORITHMIC (direct-convolution) FLOPs
    {
        // x read once + V written once (the 4 overlapping patch reads of a pixel hit in L2)
        gif::ProfScope prof_t(4, 4.0 * ((double)B * H * W * C + 16.0 * ntiles_pad * p.CP), s, (int)((long)B * H * W), C, 0, 1);
        if (int rc = gif::winograd_input_transform(x, e ? e->in_scale : nullptr, V, B, H, W, C, s)) return rc;
    }
    gif::ProfScope prof(2, flops, s, (int)((long)B * H * W), Co, C, 1091);
    p.V = V; p.U = U; p.y = y;
    p.out_scale = e ? e->out_scale : nullptr;
    p.bias = e ? e->bias : nullptr;
    p.residual = e ? static_cast<const float*>(e->residual) : nullptr;
    p.act = e ? e->act : 0;
    p.slope = e ? e->slope : 0.f;
    p.gain = e ? e->gain : 1.f;
    p.B = B; p.H = H; p.W = W; p.Co = Co;
    p.ntiles = (int)ntiles; p.ntiles_pad = (int)ntiles_pad; p.TH = H / 2; p.TW = W / 2;
    p.tiles_m = (int)(ntiles_pad / WBM);
    // 128-wide N tile (8 waves) whenever the output channels fill it; GIF_WINO_WN=2 / 4 forces a variant (benchmarking)
    static int force_wn = -1;
    if (force_wn < 0) {
        const char* env = gif::knob("GIF_WINO_WN");
        force_wn = env ? atoi(env) : 0;
    }
    // measured: +1-2 % on the >= 32^2 layers, slower when the launch has fewer than ~512 workgroups of 512 threads
    const bool wide = force_wn == 4 ? (p.RP % 128 == 0)
                                    : (force_wn == 2 ? false : (Co % 128 == 0 && (long)p.tiles_m * (p.RP / 128) >= 512));
    const int bn = wide ? 128 : 64;
    p.tiles_n = p.RP / bn;
    WinoSums sums;
    if (int rc = sums.begin(p, e, B, H, W, Co, WBM, "conv3x3_winograd")) return rc;
    const size_t lds = (size_t)WNSTAGE * (WBM + bn) * WBK * sizeof(float);
    static gif::LdsAttr attr2, attr4;
    if (wide) attr4.ensure(reinterpret_cast<const void*>(wino_gemm_mfma<4>), lds);
    else attr2.ensure(reinterpret_cast<const void*>(wino_gemm_mfma<2>), lds);
    const dim3 grid((unsigned)(p.tiles_m * p.tiles_n));
    if (wide) hipLaunchKernelGGL(wino_gemm_mfma<4>, grid, dim3(512), lds, s, p);
    else hipLaunchKernelGGL(wino_gemm_mfma<2>, grid, dim3(256), lds, s, p);
    if (int rc = sums.finish(p, B, H, W, Co, WBM, s)) return rc;
    return gif::check_launch("conv3x3_winograd");
}
// ---- bf16x3 variants: same contract; U3 = [16][3][RP][CP] bf16 from gif_winograd_weight_f32x3, RP a multiple of 128
int gif_winograd_pack_dims_x3(int cout, int cin, int* RP, int* CP) {
    GIF_REQUIRE(cout > 0 && cin > 0 && RP && CP, "winograd_pack_dims_x3: bad arguments");
    *RP = (cout + 127) / 128 * 128;
    *CP = (cin + WBK - 1) / WBK * WBK;
    return 0;
}

int gif_winograd_weight_f32x3(const float* w, void* U3, int R, int C, int RP, int CP, int64_t sr, int64_t sc, int64_t sky,
                              int64_t skx, int flip, float scale, gif_stream_t stream) {
    GIF_REQUIRE(w && U3 && R > 0 && C > 0 && RP >= R && CP >= C && RP % 128 == 0, "winograd_weight_f32x3: bad arguments");
    long total = (long)RP * CP;
    wino_weight_transform_x3<<<gif::cdiv(total, 256), 256, 0, gif::as_stream(stream)>>>(w, static_cast<unsigned short*>(U3), R, C, RP,
                                                                                         CP, sr, sc, sky, skx, flip, scale);
    return gif::check_launch("winograd_weight_f32x3");
}

}  // extern "C"

// U2 != NULL: the f16x2 GEMM on U2 followed by its guarded bf16x3 twin on U3 (a no-op unless the gate was raised; U3 may be NULL: unguarded)
static int conv3x3_winograd_x3_impl(const float* x, const void* U2, const void* U3, float* y, float* V, int B, int H, int W, int C, int Co,
                                    const gif_conv_epilogue* e, gif_stream_t stream) {
    GIF_REQUIRE(x && (U3 || U2) && y && V && B >= 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "winograd_x3: bad dims (H, W must be even)");
    GIF_REQUIRE(C > 0 && Co > 0 && C % 4 == 0 && Co % 4 == 0, "winograd_x3: channels must be multiples of 4");
    if (B == 0) return 0;
    const long ntiles = (long)B * (H / 2) * (W / 2);
    WinoParams p{};
    gif_winograd_pack_dims_x3(Co, C, &p.RP, &p.CP);
    const long ntiles_pad = (ntiles + WPAD - 1) / WPAD * WPAD;
    GIF_REQUIRE(ntiles_pad * p.CP < (1L << 31) && (long)B * H * W * Co < (1L << 31) && (long)B * H * W * C < (1L << 31) &&
                    3L * p.RP * p.CP < (1L << 31),
                "winograd_x3: tensor too large for 32-bit offsets");
    hipStream_t s = gif::as_stream(stream);
    double flops = 2.0 * B * H * W * 9.0 * C * Co;  // ALGORITHMIC (direct-convolution) FLOPs
    {
        gif::ProfScope prof_t(4, 4.0 * ((double)B * H * W * C + 16.0 * ntiles_pad * p.CP), s, (int)((long)B * H * W), C, 0, 1);
        if (int rc = gif::winograd_input_transform(x, e ? e->in_scale : nullptr, V, B, H, W, C, s)) return rc;
    }
    gif::ProfScope prof(U2 ? 14 : 10, flops, s, (int)((long)B * H * W), Co, C, 2091);
    p.V = V; p.U = static_cast<const float*>(U3); p.y = y;
    p.out_scale = e ? e->out_scale : nullptr;
    p.bias = e ? e->bias : nullptr;
    p.residual = e ? static_cast<const float*>(e->residual) : nullptr;
    p.act = e ? e->act : 0;
    p.slope = e ? e->slope : 0.f;
    p.gain = e ? e->gain : 1.f;
    p.B = B; p.H = H; p.W = W; p.Co = Co;
    p.ntiles = (int)ntiles; p.ntiles_pad = (int)ntiles_pad; p.TH = H / 2; p.TW = W / 2;
    static const int dbg = gif::knob("GIF_WINO_DBG") ? atoi(gif::knob("GIF_WINO_DBG")) : 0;
    // 128 x 128 blocks (4 x 2 waves) by default; GIF_WINO_X3_TILE=256 selects 256 x 64 (8 x 1 waves: every V fragment split
    // once instead of twice) — measured equal (2.998 / 2.141 / 1.790 ms vs 2.993 / 2.133 / 1.757 ms on the three big layers)
    static const int sq = gif::knob("GIF_WINO_X3_TILE") ? atoi(gif::knob("GIF_WINO_X3_TILE")) != 256 : 1;
    const int bm = (sq || U2) ? 128 : 256, bn = (sq || U2) ? 128 : 64;
    p.tiles_m = (int)(ntiles_pad / bm);
    p.tiles_n = p.RP / bn;
    WinoSums sums;
    if (int rc = sums.begin(p, e, B, H, W, Co, bm, "conv3x3_winograd_f32x3")) return rc;
    const size_t lds = (size_t)WNSTAGE * ((size_t)bm * WBK * sizeof(float) + 3 * (size_t)bn * 64);
    const dim3 grid((unsigned)(p.tiles_m * p.tiles_n));
    if (U2) {
        const int* hdr = static_cast<const int*>(U2);
        WinoParams q = p;
        q.uexp = hdr;
        q.U = reinterpret_cast<const float*>(static_cast<const char*>(U2) + gif::h2_header_bytes(p.RP));
        if (U3) {
            const gif::H2Gate gt = gif::h2_next_gate(s);
            if (gt.err) return gt.err;
            q.gate = gt.word; q.gate_gen = gt.gen;
        }
        // ring depth 3 / 4 / 5 (96 / 128 / 160 KB): measured equal — 2.648 / 2.638 / 2.686 ms on 128 -> 128 at 256^2, batch 32: the GEMM does
        // not wait for its DMA; GIF_WINO_H2_STAGES keeps the A/B
        static const int nst = gif::knob("GIF_WINO_H2_STAGES") ? atoi(gif::knob("GIF_WINO_H2_STAGES")) : 3;
        const size_t lds2 = (size_t)(nst == 3 ? 3 : nst == 5 ? 5 : 4) * ((size_t)128 * WBK * sizeof(float) + 2 * (size_t)128 * 64);
        static gif::LdsAttr attr2[3];
        if (nst == 3) {
            attr2[0].ensure(reinterpret_cast<const void*>(wino_gemm_h2<128, 128, 3>), lds2);
            hipLaunchKernelGGL((wino_gemm_h2<128, 128, 3>), grid, dim3(512), lds2, s, q);
        } else if (nst == 5) {
            attr2[2].ensure(reinterpret_cast<const void*>(wino_gemm_h2<128, 128, 5>), lds2);
            hipLaunchKernelGGL((wino_gemm_h2<128, 128, 5>), grid, dim3(512), lds2, s, q);
        } else {
            attr2[1].ensure(reinterpret_cast<const void*>(wino_gemm_h2<128, 128, 4>), lds2);
            hipLaunchKernelGGL((wino_gemm_h2<128, 128, 4>), grid, dim3(512), lds2, s, q);
        }
        if (!q.gate) {  // unguarded
            if (int rc = sums.finish(p, B, H, W, Co, bm, s)) return rc;
            return gif::check_launch("conv3x3_winograd_f32h2");
        }
        p.gate = q.gate; p.gate_gen = q.gate_gen; p.h2_stats = gif::h2_stats_words();
    }
#define GIF_WINO_X3_LAUNCH(D, BM_, BN_)                                                        \
    {                                                                                          \
        static gif::LdsAttr attr;                                                              \
        attr.ensure(reinterpret_cast<const void*>(wino_gemm_x3<D, BM_, BN_>), lds);            \
        hipLaunchKernelGGL((wino_gemm_x3<D, BM_, BN_>), grid, dim3(512), lds, s, p);           \
    }
    if (!sq && !U2) GIF_WINO_X3_LAUNCH(0, 256, 64)
    else if (dbg == 1) GIF_WINO_X3_LAUNCH(1, 128, 128)
    else if (dbg == 2) GIF_WINO_X3_LAUNCH(2, 128, 128)
    else if (dbg == 4) GIF_WINO_X3_LAUNCH(4, 128, 128)
    else if (dbg == 7) GIF_WINO_X3_LAUNCH(7, 128, 128)
    else GIF_WINO_X3_LAUNCH(0, 128, 128)
#undef GIF_WINO_X3_LAUNCH
    if (int rc = sums.finish(p, B, H, W, Co, bm, s)) return rc;
    return gif::check_launch("conv3x3_winograd_f32x3");
}

extern "C" {

int gif_conv3x3_winograd_f32x3(const float* x, const void* U3, float* y, float* V, int B, int H, int W, int C, int Co,
                               const gif_conv_epilogue* e, gif_stream_t stream) {
    return conv3x3_winograd_x3_impl(x, nullptr, U3, y, V, B, H, W, C, Co, e, stream);
}

/* f16x2 (ABI 4): U2 from gif_winograd_weight_f32h2 (gif_winograd_weight_f32h2_bytes), U3 the bf16x3 transform of the same weights for
 * the guarded fallback (NULL: unguarded); same pack dims as the bf16x3 GEMM */
int64_t gif_winograd_weight_f32h2_bytes(int RP, int CP) {
    if (RP <= 0 || CP <= 0) return 0;
    return (int64_t)gif::h2_header_bytes(RP) + 16LL * 2 * RP * CP * 2;
}

/* U3 (optional): also write the bf16x3 transform of the same weights (gif_winograd_weight_f32x3's output) in the same launch */
int gif_winograd_weight_f32h2(const float* w, void* U2, void* U3, int R, int C, int RP, int CP, int64_t sr, int64_t sc, int64_t sky,
                              int64_t skx, int flip, float scale, gif_stream_t stream) {
    GIF_REQUIRE(w && U2 && R > 0 && C > 0 && RP >= R && CP >= C && RP % 128 == 0 && CP % 32 == 0, "winograd_weight_f32h2: bad arguments");
    int* hdr = static_cast<int*>(U2);
    unsigned short* planes = reinterpret_cast<unsigned short*>(static_cast<char*>(U2) + gif::h2_header_bytes(RP));
    wino_weight_transform_h2<<<RP, 256, 0, gif::as_stream(stream)>>>(w, hdr, planes, static_cast<unsigned short*>(U3), R, C, RP, CP, sr, sc,
                                                                     sky, skx, flip, scale);
    return gif::check_launch("winograd_weight_f32h2");
}

int gif_conv3x3_winograd_f32h2(const float* x, const void* U2, const void* U3, float* y, float* V, int B, int H, int W, int C, int Co,
                               const gif_conv_epilogue* e, gif_stream_t stream) {
    GIF_REQUIRE(U2, "conv3x3_winograd_f32h2: null U2");
    return conv3x3_winograd_x3_impl(x, U2, U3, y, V, B, H, W, C, Co, e, stream);
}
}
