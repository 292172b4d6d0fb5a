// Skinny fp32 GEMMs for the linear layers of the G/D step (EqualLinear, stylegan2_common_layers.py:193-235: the 8-layer
// mapping network, every modulation linear, the demodulation contraction and the discriminator head), gfx950.
//
// These are [rows x 512..8192] x [512..8192 x 512] products with rows = batch (4..64): a few MFLOP each, pure latency.
// On the implicit-GEMM conv kernels (a 1x1 conv over a [rows, K, 1, 1] image) one launch costs 16 K-steps of
// DMA -> barrier -> MFMA on 8 workgroups = 20 us (240 us for the 8192-wide head).  Here every operand goes straight from
// global memory (L2 resident: the activations are <= 256 KB, a weight matrix 1 MB) into v_mfma_f32_32x32x2_f32 operands,
// all loads of a wave are issued before its first MFMA, and the reduction axis is split over the waves of a workgroup
// (fixed-order LDS reduction, deterministic).  Three access patterns, closed under differentiation:
//   NT  C[m][n] = act(scale * sum_k A[m][k] * B[n][k] + bias[n])      forward            (A, B k-contiguous: 16-byte loads)
//   NN  C[m][k] = scale * sum_n A[m][n] * B[n][k]                      data gradient      (B rows read 128 B per half wave)
//   TN  C[n][k] = scale * sum_m A[m][n] * B[m][k]                      weight gradient    (both operands coalesced dwords)
// MFMA operand convention (lane l: i = l & 31, h = l >> 5): A[i][kk], B[kk][i] with kk = 2*step + h; any K permutation
// applied to BOTH operands is legal, which is what lets a lane consume a float4 (k = 8c + 4h + t, t = 0..3) as 4 steps.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SkinnyParams {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int M, N, K;        // NT: C[M][N], reduce K.  NN: C[M][K], reduce N.  TN: C[N][K], reduce M.
    int lda, ldb, ldc;  // row strides in floats
    int cpad;           // NT / NN: columns [valid, cpad) of C are written as zero (channel padding of the caller)
    float scale;
    int act;
    float slope, gain;
    // ---- style path (ModulatedConv2d's demodulation, stylegan2_common_layers.py:311-320, forward and backward as GEMM pro-/epilogues)
    const float* A2;  // a_mode 2: second factor of the A operand, same layout as A
    int a_mode;       // 0: A   1: A^2 (s^2)   2: A * A2^3 (gd * d^3: the demodulation gradient w.r.t. its accumulator, up to -scale^2/2)
    int b_sq;         // TN: B^2
    int epi;          // 0: bias / activation   1: rsqrt(v + eps)   2: E1 + 2 * E2 * v   (E1 may be NULL)
    float eps, pad_value;
    const float* E1;
    const float* E2;
    int lde;
    float* colsum;    // TN: colsum[n] = sum_m A[m][n] (the bias gradient of a linear layer), written by the k-tile 0 wave of each n-tile
};

constexpr int SK_WAVES = 16;  // waves per workgroup = ways the reduction axis is split

// fixed-order reduction of the SK_WAVES accumulator tiles through LDS + epilogue; C/D map of the 32x32 MFMA:
// col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
__device__ __forceinline__ void reduce_store(const SkinnyParams& p, f32x16 acc, float (*red)[16][64], int wave, int lane, int row0,
                                             int col0, int rows_valid, int cols_valid, int cols_pad, bool epilogue) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    const int li = lane & 31, lh = lane >> 5;
    const int r = wave;  // SK_WAVES == 16 accumulator registers: wave w finishes register w
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < SK_WAVES; ++w) s += red[w][r][lane];
    const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * lh, col = col0 + li;
    if (row < rows_valid && col < cols_pad) {
        float v = p.pad_value;
        if (col < cols_valid) {
            v = s * p.scale;
            if (p.epi == 1) {
                v = rsqrtf(v + p.eps);
            } else if (p.epi == 2) {
                v = 2.f * p.E2[(size_t)row * p.lde + col] * v;
                if (p.E1) v += p.E1[(size_t)row * p.lde + col];
            } else if (epilogue) {
                if (p.bias) v += p.bias[col];
                if (p.act) v = (v > 0.f ? v : v * p.slope) * p.gain;
            }
        }
        p.C[(size_t)row * p.ldc + col] = v;
    }
}

__device__ __forceinline__ void skinny_nt_body(const SkinnyParams& p, int bx, int by, float (*red)[16][64]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = bx * 32, m0 = by * 32;
    const bool a_ok = m0 + li < p.M, b_ok = n0 + li < p.N;
    const float* ap = p.A + (size_t)(a_ok ? m0 + li : 0) * p.lda + 4 * lh;
    const float* bp = p.B + (size_t)(b_ok ? n0 + li : 0) * p.ldb + 4 * lh;
    const int nchunks = (p.K + 7) / 8;  // chunk c: k = 8c .. 8c+7, this lane half takes 8c + 4h .. +3
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int U = 4;  // chunks in flight per wave iteration
    for (int c0 = wave * U; c0 < nchunks; c0 += SK_WAVES * U) {
        f32x4 av[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = 8 * (c0 + u) + 4 * lh;
            const bool ok = k < p.K;  // K % 4 == 0: a float4 is inside or outside as a whole
            av[u] = (ok && a_ok) ? *reinterpret_cast<const f32x4*>(ap + 8 * (c0 + u)) : (f32x4)(0.f);
            bv[u] = (ok && b_ok) ? *reinterpret_cast<const f32x4*>(bp + 8 * (c0 + u)) : (f32x4)(0.f);
            if (p.a_mode == 1) av[u] *= av[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][t], bv[u][t], acc, 0, 0, 0);
    }
    reduce_store(p, acc, red, wave, lane, m0, n0, p.M, p.N, p.cpad, true);
}

__global__ void __launch_bounds__(64 * SK_WAVES) skinny_nt_kernel(const SkinnyParams p) {
    __shared__ float red[SK_WAVES][16][64];
    skinny_nt_body(p, blockIdx.x, blockIdx.y, red);
}

__global__ void __launch_bounds__(64 * SK_WAVES) skinny_nn_kernel(const SkinnyParams p) {
    __shared__ float red[SK_WAVES][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int k0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const bool a_ok = m0 + li < p.M, b_ok = k0 + li < p.K;
    const float* ap = p.A + (size_t)(a_ok ? m0 + li : 0) * p.lda + 4 * lh;  // A[m][n]: float4 along n
    const float* ap2 = p.a_mode == 2 ? p.A2 + (size_t)(a_ok ? m0 + li : 0) * p.lda + 4 * lh : ap;
    const float* bp = p.B + (b_ok ? k0 + li : 0);                           // B[n][k0 + li]: one dword per n
    const int nchunks = (p.N + 7) / 8;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int U = 4;
    for (int c0 = wave * U; c0 < nchunks; c0 += SK_WAVES * U) {
        f32x4 av[U];
        float bv[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int n = 8 * (c0 + u) + 4 * lh;
            av[u] = (n < p.N && a_ok) ? *reinterpret_cast<const f32x4*>(ap + 8 * (c0 + u)) : (f32x4)(0.f);  // in-row: lda >= pad4(N)
            if (p.a_mode == 2) {
                const f32x4 d = (n < p.N && a_ok) ? *reinterpret_cast<const f32x4*>(ap2 + 8 * (c0 + u)) : (f32x4)(0.f);
                av[u] *= d * d * d;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) bv[u][t] = (n + t < p.N && b_ok) ? bp[(size_t)(n + t) * p.ldb] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][t], bv[u][t], acc, 0, 0, 0);
    }
    reduce_store(p, acc, red, wave, lane, m0, k0, p.M, p.K, p.cpad, false);
}

// one wave = one 32x32 tile of C[N][K]; the reduction axis (rows m) is short, no split
__device__ __forceinline__ void skinny_tn_body(const SkinnyParams& p, int bx) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int tiles_k = (p.K + 31) / 32;
    const int tile = bx * 4 + wave;
    const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
    const int n0 = tn * 32, k0 = tk * 32;
    if (n0 >= p.N) return;
    const bool a_ok = n0 + li < p.N, b_ok = k0 + li < p.K;
    const float* ap = p.A + (a_ok ? n0 + li : 0);  // A[m][n0 + li]
    const float* ap2 = p.a_mode == 2 ? p.A2 + (a_ok ? n0 + li : 0) : ap;
    const float* bp = p.B + (b_ok ? k0 + li : 0);  // B[m][k0 + li]
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int U = 8;
    float asum = 0.f;  // this lane's half of colsum[n0 + li]: rows m = lh, lh + 2, ... in ascending order
    for (int m0 = 0; m0 < p.M; m0 += 2 * U) {
        float av[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = m0 + 2 * u + lh;
            av[u] = (m < p.M && a_ok) ? ap[(size_t)m * p.lda] : 0.f;
            bv[u] = (m < p.M && b_ok) ? bp[(size_t)m * p.ldb] : 0.f;
            if (p.a_mode == 2) {
                const float d = (m < p.M && a_ok) ? ap2[(size_t)m * p.lda] : 0.f;
                av[u] *= d * d * d;
            }
            if (p.b_sq) bv[u] *= bv[u];
            asum += av[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
    }
    if (p.colsum && tk == 0) {  // fixed order: even rows + odd rows
        const float other = __shfl_xor(asum, 32);
        if (lh == 0 && a_ok) p.colsum[n0 + li] = asum + other;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = n0 + (r & 3) + 8 * (r >> 2) + 4 * lh, col = k0 + li;
        if (row < p.N && col < p.K) p.C[(size_t)row * p.ldc + col] = acc[r] * p.scale;
    }
}

__global__ void __launch_bounds__(256) skinny_tn_kernel(const SkinnyParams p) { skinny_tn_body(p, blockIdx.x); }

// ---- the modulation bank: every ModulatedConv2d's EqualLinear of one generator pass (stylegan2_common_layers.py:311-313 —
// StyledConv / ToRGB layers, all fed by the same w) as ONE launch forward and TWO backward.  Host-side segment table in the kernel
// arguments; each segment is one layer's [n, K] weight with its own output / gradient pointers (no concatenated copies).
constexpr int BANK_MAX = 40;
struct BankSeg {
    const float* w;     // [n][K], row stride K
    const float* bias;  // [n] or NULL
    float* s;           // forward: [M][n]
    const float* gs;    // backward: [M][n]
    float* gw;          // backward: [n][K] or NULL
    float* gbias;       // backward: [n] or NULL
    int n;
    int blk_nt, blk_tn;  // first workgroup of this segment in the NT / TN launch
    int chunk;           // first 8-wide chunk of this segment on the concatenated reduction axis of the NN launch
};
struct BankParams {
    const float* x;  // [M][K] (ldx)
    float* gx;       // backward: [M][gx_pad] (ldgx), columns K..gx_pad zero
    int M, K, ldx, ldgx, gx_pad, nseg, nchunks;
    float scale;
    BankSeg seg[BANK_MAX];
};

__device__ __forceinline__ int bank_find(const BankParams& bp, int v, int BankSeg::*first) {
    int l = 0;
    while (l + 1 < bp.nseg && v >= bp.seg[l + 1].*first) ++l;
    return l;
}

__global__ void __launch_bounds__(64 * SK_WAVES) linear_bank_nt_kernel(const BankParams bp) {
    __shared__ float red[SK_WAVES][16][64];
    const int l = bank_find(bp, blockIdx.x, &BankSeg::blk_nt);
    const BankSeg& g = bp.seg[l];
    SkinnyParams p{bp.x, g.w, g.s, g.bias, bp.M, g.n, bp.K, bp.ldx, bp.K, g.n, g.n, bp.scale, 0, 1.f, 1.f};
    skinny_nt_body(p, blockIdx.x - g.blk_nt, blockIdx.y, red);
}

__global__ void __launch_bounds__(256) linear_bank_tn_kernel(const BankParams bp) {
    const int l = bank_find(bp, blockIdx.x, &BankSeg::blk_tn);
    const BankSeg& g = bp.seg[l];
    SkinnyParams p{g.gs, bp.x, g.gw, nullptr, bp.M, g.n, bp.K, g.n, bp.ldx, bp.K, bp.K, bp.scale, 0, 1.f, 1.f};
    p.colsum = g.gbias;
    skinny_tn_body(p, blockIdx.x - g.blk_tn);
}

// gx[m][k] = scale * sum over all segments and their n of gs_l[m][n] * w_l[n][k]: skinny_nn over the concatenated reduction axis
// (chunks of 8 in segment order, split over the 16 waves exactly as in skinny_nn_kernel: the summation order is fixed)
__global__ void __launch_bounds__(64 * SK_WAVES) linear_bank_nn_kernel(const BankParams bp) {
    __shared__ float red[SK_WAVES][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int k0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const bool a_ok = m0 + li < bp.M, b_ok = k0 + li < bp.K;
    const int arow = a_ok ? m0 + li : 0, bcol = b_ok ? k0 + li : 0;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int U = 4;
    int l = 0;
    for (int c0 = wave * U; c0 < bp.nchunks; c0 += SK_WAVES * U) {
        f32x4 av[U];
        float bv[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u;
            const bool ok = c < bp.nchunks;
            while (l + 1 < bp.nseg && c >= bp.seg[l + 1].chunk) ++l;
            const BankSeg& g = bp.seg[l];
            const int n = 8 * (c - g.chunk) + 4 * lh;  // n % 8 == 0 per segment: a chunk never straddles two layers
            av[u] = (ok && a_ok) ? *reinterpret_cast<const f32x4*>(g.gs + (size_t)arow * g.n + n) : (f32x4)(0.f);
#pragma unroll
            for (int t = 0; t < 4; ++t) bv[u][t] = (ok && b_ok) ? g.w[(size_t)(n + t) * bp.K + bcol] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][t], bv[u][t], acc, 0, 0, 0);
    }
    SkinnyParams p{};
    p.C = bp.gx; p.ldc = bp.ldgx; p.scale = bp.scale; p.pad_value = 0.f;
    reduce_store(p, acc, red, wave, lane, m0, k0, bp.M, bp.K, bp.gx_pad, false);
}

// wsq[co][ci] = sum_taps W[co][ci][t]^2 (the weight factor of the demodulation, :318) and its gradient back into the weight:
// gW[co][ci][t] = 2 * W[co][ci][t] * g_wsq[co][ci].  W / gW: contiguous [Cout][Cin][taps].
__global__ void __launch_bounds__(256) weight_sq_sum_kernel(const float* __restrict__ w, float* __restrict__ wsq, long n, int taps) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f;
    for (int t = 0; t < taps; ++t) {
        const float v = w[i * taps + t];
        acc += v * v;
    }
    wsq[i] = acc;
}

__global__ void __launch_bounds__(256) demod_wgrad_kernel(const float* __restrict__ w, const float* __restrict__ g_wsq, float* __restrict__ gw,
                                                          long n, int taps) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * taps) return;
    gw[i] = 2.f * w[i] * g_wsq[i / taps];
}

}  // namespace

extern "C" {

int gif_weight_sq_sum_f32(const float* w, float* wsq, int cout, int cin, int taps, gif_stream_t stream) {
    GIF_REQUIRE(w && wsq && cout > 0 && cin > 0 && taps > 0, "weight_sq_sum: bad arguments");
    const long n = (long)cout * cin;
    weight_sq_sum_kernel<<<gif::cdiv(n, 256), 256, 0, gif::as_stream(stream)>>>(w, wsq, n, taps);
    return gif::check_launch("weight_sq_sum");
}

int gif_demod_wgrad_f32(const float* w, const float* g_wsq, float* gw, int cout, int cin, int taps, gif_stream_t stream) {
    GIF_REQUIRE(w && g_wsq && gw && cout > 0 && cin > 0 && taps > 0, "demod_wgrad: bad arguments");
    const long n = (long)cout * cin;
    demod_wgrad_kernel<<<gif::cdiv(n * taps, 256), 256, 0, gif::as_stream(stream)>>>(w, g_wsq, gw, n, taps);
    return gif::check_launch("demod_wgrad");
}

// d[b][co] = rsqrt(scale2 * sum_ci s[b][ci]^2 * wsq[co][ci] + eps); columns cout..cout_pad of d are written as 1
int gif_style_demod_f32(const float* s, const float* wsq, float* d, int B, int cout, int cin, int lds, int ldw, int ldd, int cout_pad,
                        float scale2, float eps, gif_stream_t stream) {
    GIF_REQUIRE(s && wsq && d && B >= 0 && cout > 0 && cin > 0, "style_demod: bad arguments");
    GIF_REQUIRE(cin % 4 == 0 && lds % 4 == 0 && ldw % 4 == 0 && ((uintptr_t)s & 15) == 0 && ((uintptr_t)wsq & 15) == 0,
                "style_demod: cin and the row strides must be multiples of 4 floats, operands 16-byte aligned");
    GIF_REQUIRE(cout_pad >= cout && ldd >= cout_pad, "style_demod: cout_pad / ldd too small");
    if (B == 0) return 0;
    SkinnyParams p{s, wsq, d, nullptr, B, cout, cin, lds, ldw, ldd, cout_pad, scale2, 0, 1.f, 1.f};
    p.a_mode = 1; p.epi = 1; p.eps = eps; p.pad_value = 1.f;
    skinny_nt_kernel<<<dim3(gif::cdiv(cout_pad, 32), gif::cdiv(B, 32)), 64 * SK_WAVES, 0, gif::as_stream(stream)>>>(p);
    return gif::check_launch("style_demod");
}

// Backward of the demodulation w.r.t. the modulation: with g_acc = gd * (-scale2 / 2) * d^3 (d = (scale2 * acc + eps)^-1/2),
//   gs_total[b][ci] = gs_in[b][ci] + 2 * s[b][ci] * sum_co g_acc[b][co] * wsq[co][ci]      (gs_in may be NULL; columns cin..cin_pad zero)
int gif_style_demod_bwd_s_f32(const float* gd, const float* d, const float* wsq, const float* s, const float* gs_in, float* gs_total,
                              int B, int cout, int cin, int ldg, int ldw, int lds, int cin_pad, float scale2, gif_stream_t stream) {
    GIF_REQUIRE(gd && d && wsq && s && gs_total && B >= 0 && cout > 0 && cin > 0, "style_demod_bwd_s: bad arguments");
    GIF_REQUIRE(ldg % 4 == 0 && ldg >= (cout + 3) / 4 * 4 && ((uintptr_t)gd & 15) == 0 && ((uintptr_t)d & 15) == 0,
                "style_demod_bwd_s: gd / d rows must be 16-byte aligned multiples of 4 floats covering cout");
    GIF_REQUIRE(cin_pad >= cin && lds >= cin_pad, "style_demod_bwd_s: cin_pad / lds too small");
    if (B == 0) return 0;
    SkinnyParams p{gd, wsq, gs_total, nullptr, B, cout, cin, ldg, ldw, lds, cin_pad, -0.5f * scale2, 0, 1.f, 1.f};
    p.A2 = d; p.a_mode = 2; p.epi = 2; p.E1 = gs_in; p.E2 = s; p.lde = lds;
    skinny_nn_kernel<<<dim3(gif::cdiv(cin_pad, 32), gif::cdiv(B, 32)), 64 * SK_WAVES, 0, gif::as_stream(stream)>>>(p);
    return gif::check_launch("style_demod_bwd_s");
}

// ... and w.r.t. the weight factor: g_wsq[co][ci] = sum_b g_acc[b][co] * s[b][ci]^2
int gif_style_demod_bwd_w_f32(const float* gd, const float* d, const float* s, float* g_wsq, int B, int cout, int cin, int ldg, int lds,
                              float scale2, gif_stream_t stream) {
    GIF_REQUIRE(gd && d && s && g_wsq && B >= 0 && cout > 0 && cin > 0, "style_demod_bwd_w: bad arguments");
    SkinnyParams p{gd, s, g_wsq, nullptr, B, cout, cin, ldg, lds, cin, cin, -0.5f * scale2, 0, 1.f, 1.f};
    p.A2 = d; p.a_mode = 2; p.b_sq = 1;
    const int tiles = gif::cdiv(cout, 32) * gif::cdiv(cin, 32);
    skinny_tn_kernel<<<gif::cdiv(tiles, 4), 256, 0, gif::as_stream(stream)>>>(p);
    return gif::check_launch("style_demod_bwd_w");
}

int gif_linear_nt_f32(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                      int cpad, float scale, int act, float slope, float gain, gif_stream_t stream) {
    GIF_REQUIRE(A && B && C && M >= 0 && N > 0 && K > 0, "linear_nt: bad arguments");
    GIF_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0,
                "linear_nt: K and the row strides must be multiples of 4 floats, operands 16-byte aligned (K=%d lda=%d ldb=%d)", K, lda, ldb);
    GIF_REQUIRE(cpad >= N && ldc >= cpad, "linear_nt: cpad %d / ldc %d too small for N=%d", cpad, ldc, N);
    if (M == 0) return 0;
    SkinnyParams p{A, B, C, bias, M, N, K, lda, ldb, ldc, cpad, scale, act, slope, gain};
    skinny_nt_kernel<<<dim3(gif::cdiv(cpad, 32), gif::cdiv(M, 32)), 64 * SK_WAVES, 0, gif::as_stream(stream)>>>(p);
    return gif::check_launch("linear_nt");
}

int gif_linear_nn_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int cpad,
                      float scale, gif_stream_t stream) {
    GIF_REQUIRE(A && B && C && M >= 0 && N > 0 && K > 0, "linear_nn: bad arguments");
    GIF_REQUIRE(lda % 4 == 0 && lda >= (N + 3) / 4 * 4 && ((uintptr_t)A & 15) == 0,
                "linear_nn: lda must be a multiple of 4 floats covering N rounded up to 4, A 16-byte aligned (N=%d lda=%d)", N, lda);
    GIF_REQUIRE(cpad >= K && ldc >= cpad, "linear_nn: cpad %d / ldc %d too small for K=%d", cpad, ldc, K);
    if (M == 0) return 0;
    SkinnyParams p{A, B, C, nullptr, M, N, K, lda, ldb, ldc, cpad, scale, 0, 1.f, 1.f};
    skinny_nn_kernel<<<dim3(gif::cdiv(cpad, 32), gif::cdiv(M, 32)), 64 * SK_WAVES, 0, gif::as_stream(stream)>>>(p);
    return gif::check_launch("linear_nn");
}

int gif_linear_tn_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, float scale,
                      gif_stream_t stream) {
    GIF_REQUIRE(A && B && C && M >= 0 && N > 0 && K > 0 && ldc >= K, "linear_tn: bad arguments");
    SkinnyParams p{A, B, C, nullptr, M, N, K, lda, ldb, ldc, K, scale, 0, 1.f, 1.f};
    const int tiles = gif::cdiv(N, 32) * gif::cdiv(K, 32);
    skinny_tn_kernel<<<gif::cdiv(tiles, 4), 256, 0, gif::as_stream(stream)>>>(p);
    return gif::check_launch("linear_tn");
}

static int bank_fill(BankParams& bp, const gif_linear_bank_seg* segs, int nseg, bool backward, const char* what) {
    GIF_REQUIRE(segs && nseg > 0 && nseg <= BANK_MAX, "%s: 1..%d segments per launch (got %d)", what, BANK_MAX, nseg);
    int nt = 0, tn = 0, ch = 0;
    for (int l = 0; l < nseg; ++l) {
        const gif_linear_bank_seg& g = segs[l];
        GIF_REQUIRE(g.w && g.n > 0 && g.n % 8 == 0 && ((uintptr_t)g.w & 15) == 0, "%s: segment %d needs a 16-byte aligned weight and n %% 8 == 0", what, l);
        GIF_REQUIRE(backward ? (g.gs != nullptr && ((uintptr_t)g.gs & 15) == 0) : g.s != nullptr, "%s: segment %d has no %s", what, l,
                    backward ? "16-byte aligned gs" : "output");
        bp.seg[l] = BankSeg{g.w, g.bias, g.s, g.gs, g.gw, g.gbias, g.n, nt, tn, ch};
        nt += gif::cdiv(g.n, 32);
        tn += gif::cdiv(gif::cdiv(g.n, 32) * gif::cdiv(bp.K, 32), 4);
        ch += g.n / 8;
    }
    bp.nseg = nseg; bp.nchunks = ch;
    return 0;
}

// s_l[M][n_l] = scale * x[M][K] @ w_l[n_l][K]^T + bias_l  for every segment, one launch
int gif_linear_bank_fwd_f32(const float* x, int M, int K, int ldx, const gif_linear_bank_seg* segs, int nseg, float scale,
                            gif_stream_t stream) {
    GIF_REQUIRE(x && M >= 0 && K > 0 && K % 4 == 0 && ldx % 4 == 0 && ldx >= K && ((uintptr_t)x & 15) == 0,
                "linear_bank_fwd: x must be 16-byte aligned with K and ldx multiples of 4 floats");
    BankParams bp{};
    bp.x = x; bp.M = M; bp.K = K; bp.ldx = ldx; bp.scale = scale;
    if (int rc = bank_fill(bp, segs, nseg, false, "linear_bank_fwd")) return rc;
    if (M == 0) return 0;
    const BankSeg& last = bp.seg[nseg - 1];
    linear_bank_nt_kernel<<<dim3(last.blk_nt + gif::cdiv(last.n, 32), gif::cdiv(M, 32)), 64 * SK_WAVES, 0, gif::as_stream(stream)>>>(bp);
    return gif::check_launch("linear_bank_fwd");
}

// gw_l[n_l][K] = scale * gs_l^T @ x,  gbias_l[n_l] = column sums of gs_l  (one launch; either pointer may be NULL in every segment
// together);  gx[M][gx_pad] = scale * sum_l gs_l @ w_l  (second launch; gx may be NULL)
int gif_linear_bank_bwd_f32(const float* x, int M, int K, int ldx, const gif_linear_bank_seg* segs, int nseg, float scale, float* gx,
                            int ldgx, int gx_pad, gif_stream_t stream) {
    GIF_REQUIRE(x && M >= 0 && K > 0 && ldx >= K, "linear_bank_bwd: bad arguments");
    GIF_REQUIRE(!gx || (gx_pad >= K && ldgx >= gx_pad), "linear_bank_bwd: gx_pad / ldgx too small");
    BankParams bp{};
    bp.x = x; bp.gx = gx; bp.M = M; bp.K = K; bp.ldx = ldx; bp.ldgx = ldgx; bp.gx_pad = gx_pad; bp.scale = scale;
    if (int rc = bank_fill(bp, segs, nseg, true, "linear_bank_bwd")) return rc;
    int want_w = 0;
    for (int l = 0; l < nseg; ++l) want_w += segs[l].gw != nullptr;
    GIF_REQUIRE(want_w == 0 || want_w == nseg, "linear_bank_bwd: gw must be given for all segments or for none");
    if (M == 0) return 0;
    const BankSeg& last = bp.seg[nseg - 1];
    if (want_w) {
        const int blocks = last.blk_tn + gif::cdiv(gif::cdiv(last.n, 32) * gif::cdiv(K, 32), 4);
        linear_bank_tn_kernel<<<blocks, 256, 0, gif::as_stream(stream)>>>(bp);
        if (int rc = gif::check_launch("linear_bank_bwd (weights)")) return rc;
    }
    if (gx) {
        linear_bank_nn_kernel<<<dim3(gif::cdiv(gx_pad, 32), gif::cdiv(M, 32)), 64 * SK_WAVES, 0, gif::as_stream(stream)>>>(bp);
        return gif::check_launch("linear_bank_bwd (input)");
    }
    return 0;
}
}
