// HBM-bound NHWC kernels of the G/D step for gfx950: upfirdn2d FIR resampling, fused bias + leaky-ReLU
// (forward / backward), channel and per-sample reductions, minibatch standard deviation, squared-norm.
// All of them move float4 (16 B) per lane along the contiguous channel axis; reductions finish inside a
// wavefront with shuffles (64 lanes) and cross waves through a few LDS words.
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4fma(float s, float4 a, float4 c) {
    return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w));
}
__device__ __forceinline__ float lrelu(float v, float slope, float gain) { return (v > 0.f ? v : v * slope) * gain; }

// ---------------------------------------------------------------------------------------------------------
// upfirdn2d (stylegan2_common_layers.py:42-72).  One lane = one output pixel x 4 channels.
//   y[oy,ox] = sum_{a,b} kf[a,b] * xp[oy*down + a, ox*down + b],   xp[u,v] = x[(u-pady0)/up, (v-padx0)/up]
//   when divisible and in range, else 0;  kf = k flipped when flip (the reference flips, :64).
// ---------------------------------------------------------------------------------------------------------
template <typename T>
struct UpfirdnParams {
    const T* x;
    const float* k;
    T* y;
    const float* bias;
    const T* residual;
    int B, Hi, Wi, C, Ho, Wo, up, down, padx0, pady0, KH, KW, flip, act;
    float slope, gain;
    // gradient-producer fusions (gif_conv_epilogue ABI 2): y *= mask_gain * (mask_src > 0 ? 1 : mask_slope) after the activation;
    // part_cs [gridDim.x][C] = per-workgroup column sums of the stored values (blur kernels only, C/4 a divisor of 256)
    const T* mask_src;
    float mask_slope, mask_gain;
    float* part_cs;
    unsigned* sat_flag;  // f16: "a store saturated" flag word (common.h store4_flag), else NULL
};

// FIR epilogue tail shared by the blur kernels: leaky-ReLU-backward mask of the tensor this gradient flows into + the running
// column sum of what is stored
template <typename T>
__device__ __forceinline__ void fir_mask_sum(const UpfirdnParams<T>& p, size_t o, float4& v, float4& cs) {
    if (p.mask_src) {
        const float4 m = gif::load4(p.mask_src + o);
        v.x *= p.mask_gain * (m.x > 0.f ? 1.f : p.mask_slope); v.y *= p.mask_gain * (m.y > 0.f ? 1.f : p.mask_slope);
        v.z *= p.mask_gain * (m.z > 0.f ? 1.f : p.mask_slope); v.w *= p.mask_gain * (m.w > 0.f ? 1.f : p.mask_slope);
    }
    if (p.part_cs) { cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w; }
}

// per-workgroup column sums: thread t owns channel quad t % C4 for the whole grid-stride loop (256 % C4 == 0), so the sums of
// the 256 / C4 threads per quad are added through LDS in a fixed order and written to row blockIdx.x of the partial buffer
__device__ __forceinline__ void fir_block_colsum(float* part_cs, float4 cs, int C4) {
    __shared__ float4 red[256];
    red[threadIdx.x] = cs;
    __syncthreads();
    if ((int)threadIdx.x < C4) {
        float4 a = red[threadIdx.x];
        for (int k = threadIdx.x + C4; k < 256; k += C4) a = f4add(a, red[k]);
        reinterpret_cast<float4*>(part_cs)[(size_t)blockIdx.x * C4 + threadIdx.x] = a;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) upfirdn2d_kernel(const UpfirdnParams<T> p) {
    __shared__ float kf[16];
    if (threadIdx.x < p.KH * p.KW) {
        int a = threadIdx.x / p.KW, b = threadIdx.x % p.KW;
        int sa = p.flip ? p.KH - 1 - a : a, sb = p.flip ? p.KW - 1 - b : b;
        kf[threadIdx.x] = p.k[sa * p.KW + sb];
    }
    __syncthreads();
    const int C4 = p.C >> 2;
    const long total = (long)p.B * p.Ho * p.Wo * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int c4 = (int)(idx % C4);
        long pix = idx / C4;
        int ox = (int)(pix % p.Wo);
        long t = pix / p.Wo;
        int oy = (int)(t % p.Ho);
        int b = (int)(t / p.Ho);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const T* xb = p.x + (size_t)b * p.Hi * p.Wi * p.C + c4 * 4;
        for (int a = 0; a < p.KH; ++a) {
            int u = oy * p.down + a - p.pady0;
            if (u < 0) continue;
            int iy = u / p.up;
            if (iy * p.up != u || iy >= p.Hi) continue;
            for (int bb = 0; bb < p.KW; ++bb) {
                int v = ox * p.down + bb - p.padx0;
                if (v < 0) continue;
                int ix = v / p.up;
                if (ix * p.up != v || ix >= p.Wi) continue;
                float4 xv = gif::load4(xb + ((size_t)iy * p.Wi + ix) * p.C);
                acc = f4fma(kf[a * p.KW + bb], xv, acc);
            }
        }
        size_t o = (size_t)pix * p.C + c4 * 4;
        if (p.residual) acc = f4add(acc, gif::load4(p.residual + o));
        if (p.bias) acc = f4add(acc, *reinterpret_cast<const float4*>(p.bias + c4 * 4));
        if (p.act) {
            acc.x = lrelu(acc.x, p.slope, p.gain); acc.y = lrelu(acc.y, p.slope, p.gain);
            acc.z = lrelu(acc.z, p.slope, p.gain); acc.w = lrelu(acc.w, p.slope, p.gain);
        }
        gif::store4_flag(p.y + o, acc, p.sat_flag);
    }
}

// 4x4 FIR with decimation by 2 (up = 1, down = 2; the ResBlock skip path: blur + 1x1 stride-2 conv reads every second
// blurred pixel only) and with zero insertion by 2 (up = 2, down = 1; its adjoint, and the ToRGB skip up-sampling):
// fully unrolled, no divisions; the up = 2 kernel touches only the 2x2 taps whose parity matches the output pixel.
template <typename T, int UP, int DOWN>
__global__ void __launch_bounds__(256) fir4x4_resample_kernel(const UpfirdnParams<T> p) {
    static_assert((UP == 1 && DOWN == 2) || (UP == 2 && DOWN == 1), "decimate-by-2 or interpolate-by-2");
    __shared__ float kfs[16];
    if (threadIdx.x < 16) {
        int a = threadIdx.x >> 2, b = threadIdx.x & 3;
        kfs[threadIdx.x] = p.k[p.flip ? (3 - a) * 4 + (3 - b) : a * 4 + b];
    }
    __syncthreads();
    float kf[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) kf[i] = kfs[i];
    // grid: x over (ox, c4) of one output row, y over (b, oy): the row decomposition is block-uniform (scalar ALU) and the
    // per-lane index math is two 32-bit operations (64-bit div/mod per output made this kernel ALU- instead of HBM-bound)
    const unsigned C4 = (unsigned)p.C >> 2;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned row = blockIdx.y;
    const int b = (int)(row / (unsigned)p.Ho), oy = (int)(row - (unsigned)b * (unsigned)p.Ho);
    const unsigned row_items = (unsigned)p.Wo * C4;
    for (unsigned it = blockIdx.x * blockDim.x + threadIdx.x; it < row_items; it += gridDim.x * blockDim.x) {
        const int ox = (int)(it / C4);
        const int c4 = (int)(it - (unsigned)ox * C4);
        const long pix = ((long)b * p.Ho + oy) * p.Wo + ox;
        const T* xb = p.x + (size_t)b * p.Hi * p.Wi * p.C + c4 * 4;
        float4 acc = zero4;
        if (DOWN == 2) {
            const int iy0 = oy * 2 - p.pady0, ix0 = ox * 2 - p.padx0;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int iy = iy0 + a;
                const bool rok = (unsigned)iy < (unsigned)p.Hi;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const int ix = ix0 + bb;
                    const bool ok = rok && (unsigned)ix < (unsigned)p.Wi;
                    float4 xv = gif::load4(xb + (ok ? ((size_t)iy * p.Wi + ix) * p.C : 0));
                    if (!ok) xv = zero4;
                    acc = f4fma(kf[a * 4 + bb], xv, acc);
                }
            }
        } else {
            // u = oy + a - pady0 must be even: a = a0, a0 + 2 with a0 = (oy + pady0) & 1 ; input row (u >> 1)
            const int a0 = (oy + p.pady0) & 1, b0 = (ox + p.padx0) & 1;
#pragma unroll
            for (int ia = 0; ia < 2; ++ia) {
                const int a = a0 + 2 * ia;
                const int u = oy + a - p.pady0;
                const int iy = u >> 1;
                const bool rok = u >= 0 && iy < p.Hi;
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    const int bb = b0 + 2 * ib;
                    const int v = ox + bb - p.padx0;
                    const int ix = v >> 1;
                    const bool ok = rok && v >= 0 && ix < p.Wi;
                    float4 xv = gif::load4(xb + (ok ? ((size_t)iy * p.Wi + ix) * p.C : 0));
                    if (!ok) xv = zero4;
                    acc = f4fma(kf[a * 4 + bb], xv, acc);
                }
            }
        }
        size_t o = (size_t)pix * p.C + c4 * 4;
        if (p.residual) acc = f4add(acc, gif::load4(p.residual + o));
        if (p.bias) acc = f4add(acc, *reinterpret_cast<const float4*>(p.bias + c4 * 4));
        if (p.act) {
            acc.x = lrelu(acc.x, p.slope, p.gain); acc.y = lrelu(acc.y, p.slope, p.gain);
            acc.z = lrelu(acc.z, p.slope, p.gain); acc.w = lrelu(acc.w, p.slope, p.gain);
        }
        gif::store4_flag(p.y + o, acc, p.sat_flag);
    }
}

// Block form of the up = 2 kernel above (round 6).  One output pixel per lane loads 4 input pixels, 64 B for its 16: the launch moved 4 x its
// OUTPUT through L1 / L2 and ran at 2.6 TB/s of in + out where the blur reaches 4.9 (profiles/r6_fir_block_probe.txt: 1.02 -> 0.49 ms for
// 64 x 128 channels at 128^2 -> 256^2, 5.5 TB/s; f16 0.96 -> 0.42).
// up = 2: the four output pixels (one of each parity phase) that read the SAME 2 x 2 input pixels are one lane's work — 4 loads per 4
// outputs.  Output rows oy with (oy + pady0) odd open a block: oy_s = 2 m - 1 + (pady0 & 1), input rows r = (oy_s + 1 - pady0) / 2 and r + 1;
// the block's first row takes kernel rows 1, 3, its second 0, 2 (columns alike).  Every output's sum runs over the same products in the same
// order as fir4x4_resample_kernel's: bit-identical results.
template <typename T>
__global__ void __launch_bounds__(256) fir4x4_up2_block_kernel(const UpfirdnParams<T> p) {
    __shared__ float kfs[16];
    if (threadIdx.x < 16) {
        int a = threadIdx.x >> 2, b = threadIdx.x & 3;
        kfs[threadIdx.x] = p.k[p.flip ? (3 - a) * 4 + (3 - b) : a * 4 + b];
    }
    __syncthreads();
    float kf[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) kf[i] = kfs[i];
    const unsigned C4 = (unsigned)p.C >> 2;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int qy = p.pady0 & 1, qx = p.padx0 & 1;
    const unsigned nby = (unsigned)(p.Ho + 2 - qy) >> 1, nbx = (unsigned)(p.Wo + 2 - qx) >> 1;
    const unsigned row = blockIdx.y;
    const int b = (int)(row / nby), m = (int)(row - (unsigned)b * nby);
    const int oy_s = 2 * m - 1 + qy;
    const int r = (oy_s + 1 - p.pady0) >> 1;  // (exact: the numerator is even)
    float4 bias4 = zero4;
    const unsigned row_items = nbx * C4;
    for (unsigned it = blockIdx.x * blockDim.x + threadIdx.x; it < row_items; it += gridDim.x * blockDim.x) {
        const int n = (int)(it / C4);
        const int c4 = (int)(it - (unsigned)n * C4);
        const int ox_s = 2 * n - 1 + qx;
        const int c = (ox_s + 1 - p.padx0) >> 1;
        const T* xb = p.x + (size_t)b * p.Hi * p.Wi * p.C + c4 * 4;
        float4 xv[2][2];
#pragma unroll
        for (int ia = 0; ia < 2; ++ia)
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
                const int iy = r + ia, ix = c + ib;
                const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                xv[ia][ib] = gif::load4(xb + (ok ? ((size_t)iy * p.Wi + ix) * p.C : 0));
                if (!ok) xv[ia][ib] = zero4;
            }
        if (p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + c4 * 4);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int oy = oy_s + dy;
            if ((unsigned)oy >= (unsigned)p.Ho) continue;
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int ox = ox_s + dx;
                if ((unsigned)ox >= (unsigned)p.Wo) continue;
                const int a0 = 1 - dy, b0 = 1 - dx;
                float4 acc = zero4;
#pragma unroll
                for (int ia = 0; ia < 2; ++ia)
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) acc = f4fma(kf[(a0 + 2 * ia) * 4 + b0 + 2 * ib], xv[ia][ib], acc);
                const size_t o = ((((size_t)b * p.Ho + oy) * p.Wo + ox) * C4 + c4) * 4;
                if (p.residual) acc = f4add(acc, gif::load4(p.residual + o));
                if (p.bias) acc = f4add(acc, bias4);
                if (p.act) {
                    acc.x = lrelu(acc.x, p.slope, p.gain); acc.y = lrelu(acc.y, p.slope, p.gain);
                    acc.z = lrelu(acc.z, p.slope, p.gain); acc.w = lrelu(acc.w, p.slope, p.gain);
                }
                gif::store4_flag(p.y + o, acc, p.sat_flag);
            }
        }
    }
}

// (down = 2 stays on the one-pixel-per-lane kernel: its 16 loads per output overlap in L1 — 4.8 TB/s of in + out; a 2 x 4-pixel block form with
// 7.5 loads per output measured 13-18 % SLOWER, profiles/r6_fir_block_probe.txt.)

// Fast path for the blur (up = down = 1, 4x4 FIR): one lane produces a TY x TX patch of output pixels for 4
// channels, so every input float4 is loaded once per patch ((TY+3)*(TX+3) loads for TY*TX outputs: 4.4 loads
// per output instead of 16).  Lanes are consecutive along the channel axis => fully coalesced 16-B accesses.
template <typename T, int TY, int TX>
__global__ void __launch_bounds__(256) blur4x4_tiled_kernel(const UpfirdnParams<T> p) {
    __shared__ float kfs[16];
    if (threadIdx.x < 16) {
        int a = threadIdx.x >> 2, b = threadIdx.x & 3;
        kfs[threadIdx.x] = p.k[p.flip ? (3 - a) * 4 + (3 - b) : a * 4 + b];
    }
    __syncthreads();
    float kf[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) kf[i] = kfs[i];
    const int C4 = p.C >> 2;
    const int nyb = (p.Ho + TY - 1) / TY, nxb = (p.Wo + TX - 1) / TX;
    const long total = (long)p.B * nyb * nxb * C4;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int c4 = (int)(idx % C4);
        long t = idx / C4;
        int xb = (int)(t % nxb);
        t /= nxb;
        int yb = (int)(t % nyb);
        int b = (int)(t / nyb);
        const int oy0 = yb * TY, ox0 = xb * TX;
        const T* xb_ = p.x + (size_t)b * p.Hi * p.Wi * p.C + c4 * 4;
        float4 acc[TY][TX];
#pragma unroll
        for (int i = 0; i < TY; ++i)
#pragma unroll
            for (int j = 0; j < TX; ++j) acc[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < TY + 3; ++r) {
            const int iy = oy0 + r - p.pady0;
            const bool rok = (unsigned)iy < (unsigned)p.Hi;
#pragma unroll
            for (int c = 0; c < TX + 3; ++c) {
                const int ix = ox0 + c - p.padx0;
                const bool ok = rok && (unsigned)ix < (unsigned)p.Wi;
                float4 v = gif::load4(xb_ + (ok ? ((size_t)iy * p.Wi + ix) * p.C : 0));
                if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int ty = 0; ty < TY; ++ty) {
                    const int a = r - ty;
                    if (a < 0 || a > 3) continue;
#pragma unroll
                    for (int tx = 0; tx < TX; ++tx) {
                        const int bb = c - tx;
                        if (bb < 0 || bb > 3) continue;
                        acc[ty][tx] = f4fma(kf[a * 4 + bb], v, acc[ty][tx]);
                    }
                }
            }
        }
#pragma unroll
        for (int ty = 0; ty < TY; ++ty) {
            const int oy = oy0 + ty;
            if (oy >= p.Ho) continue;
#pragma unroll
            for (int tx = 0; tx < TX; ++tx) {
                const int ox = ox0 + tx;
                if (ox >= p.Wo) continue;
                size_t o = (((size_t)b * p.Ho + oy) * p.Wo + ox) * p.C + c4 * 4;
                float4 v = acc[ty][tx];
                if (p.residual) v = f4add(v, gif::load4(p.residual + o));
                if (p.bias) v = f4add(v, *reinterpret_cast<const float4*>(p.bias + c4 * 4));
                if (p.act) {
                    v.x = lrelu(v.x, p.slope, p.gain); v.y = lrelu(v.y, p.slope, p.gain);
                    v.z = lrelu(v.z, p.slope, p.gain); v.w = lrelu(v.w, p.slope, p.gain);
                }
                fir_mask_sum(p, o, v, cs);
                gif::store4_flag(p.y + o, v, p.sat_flag);
            }
        }
    }
    if (p.part_cs) fir_block_colsum(p.part_cs, cs, C4);
}

// Sliding-window variant for tall images: one lane walks down TYL output rows of a TX-wide column strip (4 channels),
// loading every input row of the strip ONCE (TX+3 float4) and scattering it into a ring of 4 partially accumulated
// output rows: (TYL+3)(TX+3)/(TYL*TX) = 2.1 loads per output instead of 4.4, same tap order => bit-identical sums.
template <typename T, int TYL, int TX>
__global__ void __launch_bounds__(256) blur4x4_rows_kernel(const UpfirdnParams<T> p) {
    __shared__ float kfs[16];
    if (threadIdx.x < 16) {
        int a = threadIdx.x >> 2, b = threadIdx.x & 3;
        kfs[threadIdx.x] = p.k[p.flip ? (3 - a) * 4 + (3 - b) : a * 4 + b];
    }
    __syncthreads();
    float kf[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) kf[i] = kfs[i];
    const int C4 = p.C >> 2;
    const int nyb = (p.Ho + TYL - 1) / TYL, nxb = (p.Wo + TX - 1) / TX;
    const long total = (long)p.B * nyb * nxb * C4;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 bias4 = zero4;
    float4 cs = zero4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int c4 = (int)(idx % C4);
        long t = idx / C4;
        int xb = (int)(t % nxb);
        t /= nxb;
        int yb = (int)(t % nyb);
        int b = (int)(t / nyb);
        const int oy0 = yb * TYL, ox0 = xb * TX;
        const T* xb_ = p.x + (size_t)b * p.Hi * p.Wi * p.C + c4 * 4;
        if (p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + c4 * 4);
        float4 acc[4][TX];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TX; ++j) acc[i][j] = zero4;
        static_assert((TYL + 3 + 3) / 4 * 4 >= TYL + 3, "row loop covers the halo");
        for (int r4 = 0; r4 < TYL + 3; r4 += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r4 + q;  // input row oy0 + r - pady0 feeds output rows oy0 + r - a, a = 0..3
                const int iy = oy0 + r - p.pady0;
                const bool rok = r < TYL + 3 && (unsigned)iy < (unsigned)p.Hi;
                float4 v[TX + 3];
#pragma unroll
                for (int c = 0; c < TX + 3; ++c) {
                    const int ix = ox0 + c - p.padx0;
                    const bool ok = rok && (unsigned)ix < (unsigned)p.Wi;
                    v[c] = gif::load4(xb_ + (ok ? ((size_t)iy * p.Wi + ix) * p.C : 0));
                    if (!ok) v[c] = zero4;
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int slot = (q - a) & 3;  // ring slot of output row r - a
#pragma unroll
                    for (int tx = 0; tx < TX; ++tx)
#pragma unroll
                        for (int bb = 0; bb < 4; ++bb) acc[slot][tx] = f4fma(kf[a * 4 + bb], v[tx + bb], acc[slot][tx]);
                }
                // output row r - 3 is complete
                const int done = (q - 3) & 3;
                const int oy = oy0 + r - 3;
                if (r >= 3 && r - 3 < TYL && oy < p.Ho) {
#pragma unroll
                    for (int tx = 0; tx < TX; ++tx) {
                        const int ox = ox0 + tx;
                        if (ox >= p.Wo) continue;
                        size_t o = (((size_t)b * p.Ho + oy) * p.Wo + ox) * p.C + c4 * 4;
                        float4 vv = acc[done][tx];
                        if (p.residual) vv = f4add(vv, gif::load4(p.residual + o));
                        if (p.bias) vv = f4add(vv, bias4);
                        if (p.act) {
                            vv.x = lrelu(vv.x, p.slope, p.gain); vv.y = lrelu(vv.y, p.slope, p.gain);
                            vv.z = lrelu(vv.z, p.slope, p.gain); vv.w = lrelu(vv.w, p.slope, p.gain);
                        }
                        fir_mask_sum(p, o, vv, cs);
                        gif::store4_flag(p.y + o, vv, p.sat_flag);
                    }
                }
#pragma unroll
                for (int tx = 0; tx < TX; ++tx) acc[done][tx] = zero4;
            }
        }
    }
    if (p.part_cs) fir_block_colsum(p.part_cs, cs, C4);
}

// ---------------------------------------------------------------------------------------------------------
// bias + leaky ReLU (FusedLeakyReLU.forward, stylegan2_common_layers.py:32-39)
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) bias_act_kernel(const T* __restrict__ x, const float* __restrict__ bias,
                                                       const T* __restrict__ res, T* __restrict__ y,
                                                       long n4, int C4, float slope, float gain) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = gif::load4(x + 4 * i);
        if (res) v = f4add(v, gif::load4(res + 4 * i));
        if (bias) v = f4add(v, reinterpret_cast<const float4*>(bias)[i % C4]);
        v.x = lrelu(v.x, slope, gain); v.y = lrelu(v.y, slope, gain);
        v.z = lrelu(v.z, slope, gain); v.w = lrelu(v.w, slope, gain);
        gif::store4(y + 4 * i, v);
    }
}

// Column-sum building block: block (bx) owns rows [bx*rows_per_block, ...), thread owns one float4 column
// and one of R row lanes; cross-lane reduction over the R row lanes goes through LDS.
// MODE 0: v = x ; MODE 1 (bias_act backward): v = gy * gain * (y > 0 ? 1 : slope), also stored to gx.
// MODE 2 (mul): v = a*b, optional scaled output s[b,c]*a.
// MODE 3: v = a * (inv_act(b) - res - bias[c]) with inv_act the inverse of gain*lrelu(., slope) (fused-epilogue modconv).
template <int MODE, typename T>
__global__ void __launch_bounds__(256)
colsum_stage1(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out_ew,
              const float* __restrict__ scale, float* __restrict__ partial, long nrows, int C4, long rows_per_block,
              float slope, float gain, int want_sum, const T* __restrict__ res = nullptr,
              const float* __restrict__ bias = nullptr, unsigned* sat_flag = nullptr) {
    __shared__ float4 red[256];
    const int R = 256 / C4;  // row lanes (C4 <= 256)
    const int col = threadIdx.x % C4, rl = threadIdx.x / C4;
    const bool active = rl < R;
    // blockIdx.y = sample (MODE 2) — rows are relative to the sample
    const long base = (long)blockIdx.y * nrows;
    long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > nrows) r1 = nrows;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
    if (MODE == 2 && scale && active) sc = reinterpret_cast<const float4*>(scale)[(long)blockIdx.y * C4 + col];
    float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == 3 && bias && active) bs = reinterpret_cast<const float4*>(bias)[col];
    const float ig = 1.f / gain, igs = 1.f / (gain * slope);
    if (active) {
        for (long r = r0 + rl; r < r1; r += R) {
            long i = (base + r) * C4 + col;
            float4 v = gif::load4(a + 4 * i);
            if (MODE == 1) {
                float4 yv = gif::load4(b + 4 * i);
                v.x *= gain * (yv.x > 0.f ? 1.f : slope); v.y *= gain * (yv.y > 0.f ? 1.f : slope);
                v.z *= gain * (yv.z > 0.f ? 1.f : slope); v.w *= gain * (yv.w > 0.f ? 1.f : slope);
                gif::store4_flag(out_ew + 4 * i, v, sat_flag);
            } else if (MODE == 2) {
                float4 bv = gif::load4(b + 4 * i);
                if (out_ew) gif::store4_flag(out_ew + 4 * i, make_float4(sc.x * v.x, sc.y * v.y, sc.z * v.z, sc.w * v.w), sat_flag);
                v.x *= bv.x; v.y *= bv.y; v.z *= bv.z; v.w *= bv.w;
            } else if (MODE == 3) {
                float4 yv = gif::load4(b + 4 * i);
                float4 rv = res ? gif::load4(res + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
                v.x *= yv.x * (yv.x > 0.f ? ig : igs) - rv.x - bs.x; v.y *= yv.y * (yv.y > 0.f ? ig : igs) - rv.y - bs.y;
                v.z *= yv.z * (yv.z > 0.f ? ig : igs) - rv.z - bs.z; v.w *= yv.w * (yv.w > 0.f ? ig : igs) - rv.w - bs.w;
            }
            acc = f4add(acc, v);
        }
    }
    if (!want_sum) return;
    red[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0) {
        for (int k = 1; k < R; ++k) acc = f4add(acc, red[k * C4 + col]);
        reinterpret_cast<float4*>(partial)[((long)blockIdx.y * gridDim.x + blockIdx.x) * C4 + col] = acc;
    }
}

// out[y][c] = sum_k partial[y][k][c] ; block = 64 channels x 4 partial-row lanes (short dependent chains)
__global__ void __launch_bounds__(256) colsum_stage2(const float* __restrict__ partial, float* __restrict__ out, int nblk,
                                                     int C) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float acc = 0.f;
    if (c < C) {
        // 8 independent partial sums per lane (fixed assignment row -> slot, fixed final order): 8 loads in flight instead of one
        // dependent L2 round trip per row — the launch is pure latency (r4: 215 launches, 3.3 ms per iteration at 1024^2)
        const float* pp = partial + (size_t)blockIdx.y * nblk * C + c;
        float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int k = rl;
        for (; k + 28 < nblk; k += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) part[u] += pp[(size_t)(k + 4 * u) * C];
        }
        for (int u = 0; k < nblk; k += 4, ++u) part[u] += pp[(size_t)k * C];
        acc = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
    }
    red[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && c < C) out[(size_t)blockIdx.y * C + c] = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
}

inline int colsum_blocks(long nrows, int C4) {
    int R = 256 / C4;
    long want = (nrows + (long)R * 16 - 1) / ((long)R * 16);  // >= 16 rows per row-lane
    if (want > 512) want = 512;
    if (want < 1) want = 1;
    return (int)want;
}

// ---------------------------------------------------------------------------------------------------------
// minibatch stddev (stg2_discriminator.py:59-65)
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mbstd_stat_kernel(const float* __restrict__ x, float* __restrict__ stat, int M,
                                                         int G, long n /*H*W*C*/) {
    __shared__ float red[4];
    const int m = blockIdx.x;
    float acc = 0.f;
    for (long e = threadIdx.x; e < n; e += blockDim.x) {
        float mean = 0.f;
        for (int g = 0; g < G; ++g) mean += x[((long)g * M + m) * n + e];
        mean /= (float)G;
        float var = 0.f;
        for (int g = 0; g < G; ++g) {
            float d = x[((long)g * M + m) * n + e] - mean;
            var += d * d;
        }
        acc += sqrtf(var / (float)G + 1e-8f);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) stat[m] = (red[0] + red[1] + red[2] + red[3]) / (float)n;
}

__global__ void __launch_bounds__(256) mbstd_write_kernel(const float* __restrict__ x, const float* __restrict__ stat,
                                                          float* __restrict__ y, int B, int HW, int C, int Cy, int M) {
    const int Cy4 = Cy >> 2;
    long total = (long)B * HW * Cy4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int c4 = (int)(idx % Cy4);
        long pix = idx / Cy4;
        int b = (int)(pix / HW);
        int c = c4 * 4;
        float4 v;
        if (c + 3 < C) {
            v = *reinterpret_cast<const float4*>(x + pix * C + c);
        } else {
            float s = stat[b % M];
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = (c + j < C) ? x[pix * C + c + j] : (c + j == C ? s : 0.f);
            v = make_float4(e[0], e[1], e[2], e[3]);
        }
        *reinterpret_cast<float4*>(y + pix * Cy + c) = v;
    }
}

// gx[b,e] = gy[b,e(:C)] + gstat[m]/n * (x[b,e]-mean)/(G*sd),  gstat[m] = sum over group members & pixels of gy[..., C]
__global__ void __launch_bounds__(256) mbstd_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                        float* __restrict__ gx, int M, int G, int HW, int C, int Cy) {
    __shared__ float red[4];
    __shared__ float gs;
    const int m = blockIdx.x;
    const long n = (long)HW * C;
    float acc = 0.f;
    for (int e = threadIdx.x; e < G * HW; e += blockDim.x) {
        int g = e / HW, hw = e - g * HW;
        acc += gy[(((long)g * M + m) * HW + hw) * Cy + C];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) gs = (red[0] + red[1] + red[2] + red[3]) / (float)n;
    __syncthreads();
    const float gstat = gs;
    for (long e = threadIdx.x; e < n; e += blockDim.x) {
        int hw = (int)(e / C), c = (int)(e - (long)hw * C);
        float xv[8];
        float mean = 0.f;
        for (int g = 0; g < G; ++g) {
            xv[g] = x[((long)g * M + m) * n + e];
            mean += xv[g];
        }
        mean /= (float)G;
        float var = 0.f;
        for (int g = 0; g < G; ++g) var += (xv[g] - mean) * (xv[g] - mean);
        float sd = sqrtf(var / (float)G + 1e-8f);
        float k = gstat / ((float)G * sd);
        for (int g = 0; g < G; ++g) {
            long b = (long)g * M + m;
            gx[b * n + e] = gy[(b * HW + hw) * Cy + c] + k * (xv[g] - mean);
        }
    }
}

// out[b] = sum g[b,:]^2  (grad_penalty_loss, losses.py:87-99)
__global__ void __launch_bounds__(1024) sqnorm_kernel(const float* __restrict__ g, float* __restrict__ out, long n) {
    __shared__ float red[16];
    const float* gb = g + (size_t)blockIdx.x * n;
    float acc = 0.f;
    long n4 = n >> 2;
    for (long i = threadIdx.x; i < n4; i += blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(gb)[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) acc += gb[i] * gb[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int k = 0; k < 16; ++k) s += red[k];
        out[blockIdx.x] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Condition pyramid level (stg2_generator.py:309-314): F.interpolate(cond, (S,S), 'bilinear', align_corners=False) for
// an integer down-scale factor f = R/S.  For even f the two source taps per axis are f*d + f/2 - 1 and + f/2 with
// weight 0.5 each; f == 1 is a copy.  Same operation order as ATen's kernel: 0.5*(0.5*a+0.5*b) + 0.5*(0.5*c+0.5*d).
// Backward scatters 0.25*g to the 4 taps (each source pixel receives at most one contribution => plain stores).
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bilinear_down_kernel(const float* __restrict__ x, float* __restrict__ y, int B,
                                                            int R, int S, int C4) {
    const int f = R / S, o0 = f / 2 - 1;
    long total = (long)B * S * S * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int c4 = (int)(idx % C4);
        long pix = idx / C4;
        int ox = (int)(pix % S);
        long t = pix / S;
        int oy = (int)(t % S), b = (int)(t / S);
        const float4* xb = reinterpret_cast<const float4*>(x) + (size_t)b * R * R * C4 + c4;
        float4 r;
        if (f == 1) {
            r = xb[((size_t)oy * R + ox) * C4];
        } else {
            int y0 = oy * f + o0, x0 = ox * f + o0;
            float4 a = xb[((size_t)y0 * R + x0) * C4], bq = xb[((size_t)y0 * R + x0 + 1) * C4];
            float4 c = xb[((size_t)(y0 + 1) * R + x0) * C4], d = xb[((size_t)(y0 + 1) * R + x0 + 1) * C4];
            r.x = 0.5f * (0.5f * a.x + 0.5f * bq.x) + 0.5f * (0.5f * c.x + 0.5f * d.x);
            r.y = 0.5f * (0.5f * a.y + 0.5f * bq.y) + 0.5f * (0.5f * c.y + 0.5f * d.y);
            r.z = 0.5f * (0.5f * a.z + 0.5f * bq.z) + 0.5f * (0.5f * c.z + 0.5f * d.z);
            r.w = 0.5f * (0.5f * a.w + 0.5f * bq.w) + 0.5f * (0.5f * c.w + 0.5f * d.w);
        }
        reinterpret_cast<float4*>(y)[idx] = r;
    }
}

__global__ void __launch_bounds__(256) bilinear_down_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int B,
                                                                int R, int S, int C4) {
    const int f = R / S, o0 = f / 2 - 1;
    long total = (long)B * R * R * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int c4 = (int)(idx % C4);
        long pix = idx / C4;
        int ix = (int)(pix % R);
        long t = pix / R;
        int iy = (int)(t % R), b = (int)(t / R);
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f == 1) {
            r = reinterpret_cast<const float4*>(gy)[idx];
        } else {
            int ry = iy % f - o0, rx = ix % f - o0;  // tap position inside the f x f cell: taps are 0 and 1
            if ((ry == 0 || ry == 1) && (rx == 0 || rx == 1)) {
                float4 g = reinterpret_cast<const float4*>(gy)[(((size_t)b * S + iy / f) * S + ix / f) * C4 + c4];
                r = make_float4(0.25f * g.x, 0.25f * g.y, 0.25f * g.z, 0.25f * g.w);
            }
        }
        reinterpret_cast<float4*>(gx)[idx] = r;
    }
}

inline int ew_grid(long n) {
    long b = (n + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

template <typename T>
int upfirdn2d_impl(const T* x, const float* k, T* y, int B, int Hi, int Wi, int C, int Ho, int Wo, int up, int down, int padx0,
                   int pady0, int KH, int KW, int flip, const gif_conv_epilogue* e, gif_stream_t stream) {
    GIF_REQUIRE(x && k && y, "upfirdn2d: null pointer");
    GIF_REQUIRE(B >= 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 4 == 0, "upfirdn2d: bad dims (C=%d)", C);
    GIF_REQUIRE(up >= 1 && down >= 1 && KH >= 1 && KW >= 1 && KH * KW <= 16, "upfirdn2d: bad up/down/kernel");
    if (B == 0) return 0;
    UpfirdnParams<T> p{};
    p.x = x; p.k = k; p.y = y;
    p.bias = e ? e->bias : nullptr;
    p.residual = e ? static_cast<const T*>(e->residual) : nullptr;
    p.act = e ? e->act : 0;
    p.slope = e ? e->slope : 0.f;
    p.gain = e ? e->gain : 1.f;
    p.B = B; p.Hi = Hi; p.Wi = Wi; p.C = C; p.Ho = Ho; p.Wo = Wo; p.up = up; p.down = down;
    p.padx0 = padx0; p.pady0 = pady0; p.KH = KH; p.KW = KW; p.flip = flip;
    p.sat_flag = sizeof(T) == 2 ? gif::f16_sat_flag() : nullptr;
    // gradient-producer fusions (blur kernels): leaky-ReLU-backward mask + bias-gradient column sums in the epilogue
    const bool blur = up == 1 && down == 1 && KH == 4 && KW == 4;
    const bool fused = e && (e->mask_src || e->colsum || e->dot || e->dot_src);
    float* colsum_out = nullptr;
    if (fused) {
        GIF_REQUIRE(!e->dot && !e->dot_src, "upfirdn2d: the dot fusion is a convolution epilogue only");
        GIF_REQUIRE(blur, "upfirdn2d: mask / colsum fusions need the 4x4 blur form (up = down = 1)");
        GIF_REQUIRE(!e->colsum || (e->red_ws && (C & (C - 1)) == 0 && C <= 1024), "upfirdn2d: colsum needs red_ws and a power-of-two C <= 1024");
        p.mask_src = static_cast<const T*>(e->mask_src);
        p.mask_slope = e->mask_slope; p.mask_gain = e->mask_gain;
        colsum_out = e->colsum;
        p.part_cs = colsum_out ? e->red_ws : nullptr;
    }
    hipStream_t s = gif::as_stream(stream);
    if (blur && (long)B * ((Ho + 15) / 16) * ((Wo + 3) / 4) * (C / 4) >= 256L * 256 * 2) {
        // enough columns strips to fill the chip with 16-row sliding windows
        long total = (long)B * ((Ho + 15) / 16) * ((Wo + 3) / 4) * (C / 4);
        const int grid = ew_grid(total);
        blur4x4_rows_kernel<T, 16, 4><<<grid, 256, 0, s>>>(p);
        if (colsum_out) return gif::reduce_partials(p.part_cs, colsum_out, 1, grid, C, p.part_cs + (size_t)grid * C, s);
        return gif::check_launch("upfirdn2d(blur rows)");
    }
    if (blur) {
        constexpr int TY = 2, TX = 4;
        long total = (long)B * ((Ho + TY - 1) / TY) * ((Wo + TX - 1) / TX) * (C / 4);
        const int grid = ew_grid(total);
        blur4x4_tiled_kernel<T, TY, TX><<<grid, 256, 0, s>>>(p);
        if (colsum_out) return gif::reduce_partials(p.part_cs, colsum_out, 1, grid, C, p.part_cs + (size_t)grid * C, s);
        return gif::check_launch("upfirdn2d(blur)");
    }
    long total = (long)B * Ho * Wo * (C / 4);
    if (KH == 4 && KW == 4 && (long)B * Ho <= 65535 && ((up == 1 && down == 2) || (up == 2 && down == 1))) {
        // GIF_FIR_BLOCK=0: the one-pixel-per-lane kernel for up = 2 as well (A/B; the block kernel gives the same bits)
        static const int block_on = gif::knob("GIF_FIR_BLOCK") ? atoi(gif::knob("GIF_FIR_BLOCK")) != 0 : 1;
        if (block_on && up == 2) {
            const long nby = (Ho + 2 - (pady0 & 1)) / 2, nbx = (Wo + 2 - (padx0 & 1)) / 2;
            long gx = (nbx * (C / 4) + 255) / 256;
            if (gx > 64) gx = 64;
            const dim3 grid((unsigned)gx, (unsigned)((long)B * nby));
            fir4x4_up2_block_kernel<T><<<grid, 256, 0, gif::as_stream(stream)>>>(p);
            return gif::check_launch("upfirdn2d(up 2, block)");
        }
        const long row_items = (long)Wo * (C / 4);
        long gx = (row_items + 255) / 256;
        if (gx > 64) gx = 64;
        const dim3 grid((unsigned)gx, (unsigned)((long)B * Ho));
        if (down == 2) fir4x4_resample_kernel<T, 1, 2><<<grid, 256, 0, gif::as_stream(stream)>>>(p);
        else fir4x4_resample_kernel<T, 2, 1><<<grid, 256, 0, gif::as_stream(stream)>>>(p);
        return gif::check_launch("upfirdn2d(resample by 2)");
    }
    upfirdn2d_kernel<T><<<ew_grid(total), 256, 0, gif::as_stream(stream)>>>(p);
    return gif::check_launch("upfirdn2d");
}

template <typename T>
int bias_act_impl(const T* x, const float* bias, const T* residual, T* y, int64_t npix, int C, float slope, float gain,
                  gif_stream_t stream) {
    GIF_REQUIRE(x && y && npix >= 0 && C > 0 && C % 4 == 0, "bias_act: bad arguments (C=%d)", C);
    if (npix == 0) return 0;
    long n4 = npix * (C / 4);
    bias_act_kernel<T><<<ew_grid(n4), 256, 0, gif::as_stream(stream)>>>(x, bias, residual, y, n4, C / 4, slope, gain);
    return gif::check_launch("bias_act");
}

template <typename T>
int bias_act_bwd_impl(const T* gy, const T* y, T* gx, float* gbias, float* partial, int64_t npix, int C, float slope, float gain,
                      gif_stream_t stream) {
    GIF_REQUIRE(gy && y && gx && npix >= 0 && C > 0 && C % 4 == 0 && C <= 1024, "bias_act_bwd: bad arguments (C=%d)", C);
    GIF_REQUIRE(!gbias || partial, "bias_act_bwd: gbias needs a partial buffer");
    if (npix == 0) return 0;
    hipStream_t s = gif::as_stream(stream);
    int C4 = C / 4;
    int nblk = colsum_blocks(npix, C4);
    long rpb = (npix + nblk - 1) / nblk;
    colsum_stage1<1, T><<<dim3(nblk, 1), 256, 0, s>>>(gy, y, gx, nullptr, partial, npix, C4, rpb, slope, gain, gbias != nullptr, nullptr, nullptr,
                                                       sizeof(T) == 2 ? gif::f16_sat_flag() : nullptr);
    if (gbias) colsum_stage2<<<dim3(gif::cdiv(C, 64), 1), 256, 0, s>>>(partial, gbias, nblk, C);
    return gif::check_launch("bias_act_bwd");
}

template <typename T>
int colsum_impl(const T* x, float* out, float* partial, int64_t npix, int C, gif_stream_t stream) {
    GIF_REQUIRE(x && out && partial && npix >= 0 && C > 0 && C % 4 == 0 && C <= 1024, "colsum: bad arguments (C=%d)", C);
    hipStream_t s = gif::as_stream(stream);
    int C4 = C / 4;
    int nblk = colsum_blocks(npix, C4);
    long rpb = (npix + nblk - 1) / nblk;
    colsum_stage1<0, T><<<dim3(nblk, 1), 256, 0, s>>>(x, nullptr, nullptr, nullptr, partial, npix, C4, rpb, 0.f, 1.f, 1);
    colsum_stage2<<<dim3(gif::cdiv(C, 64), 1), 256, 0, s>>>(partial, out, nblk, C);
    return gif::check_launch("colsum");
}

}  // namespace

namespace gif {

int reduce_partials(const float* partial, float* out, int Y, int nblk, int C, float* tmp, hipStream_t s) {
    if (Y <= 0 || nblk <= 0) return 0;
    if (Y == 1 && nblk > 512) {  // one long reduction: 64 groups first (two blocks of 64 channels alone would crawl through it)
        const int S = 64, per = (nblk + S - 1) / S;
        // groups of `per` rows; the last group may be short: pad by reducing [k*per, min((k+1)*per, nblk)) — stage 2 takes a row
        // count per group, so run the full groups and the tail separately
        const int full = nblk / per, tail = nblk - full * per;
        colsum_stage2<<<dim3(cdiv(C, 64), full), 256, 0, s>>>(partial, tmp, per, C);
        if (tail) colsum_stage2<<<dim3(cdiv(C, 64), 1), 256, 0, s>>>(partial + (size_t)full * per * C, tmp + (size_t)full * C, tail, C);
        colsum_stage2<<<dim3(cdiv(C, 64), 1), 256, 0, s>>>(tmp, out, full + (tail ? 1 : 0), C);
    } else if (Y > 1 && nblk >= 1024 && nblk % 16 == 0 && (long)Y * 16 <= 4096) {
        // per-sample sums over thousands of rows (the halo kernels' one row per 16 x 16 patch at 1024^2): Y blocks alone would
        // crawl; 16 groups per sample first (groups of one sample are consecutive rows of tmp)
        colsum_stage2<<<dim3(cdiv(C, 64), Y * 16), 256, 0, s>>>(partial, tmp, nblk / 16, C);
        colsum_stage2<<<dim3(cdiv(C, 64), Y), 256, 0, s>>>(tmp, out, 16, C);
    } else {
        colsum_stage2<<<dim3(cdiv(C, 64), Y), 256, 0, s>>>(partial, out, nblk, C);
    }
    return check_launch("reduce_partials");
}

}  // namespace gif

namespace {

inline int mul_reduce_chunks(int64_t HW) { return colsum_blocks(HW, 32) > 64 ? 64 : colsum_blocks(HW, 32); }

template <typename T>
int mul_reduce_impl(const T* a, const T* b, const float* scale, T* scaled, float* out, float* partial, int B, int64_t HW,
                    int C, gif_stream_t stream) {
    GIF_REQUIRE(a && b && out && partial && B >= 0 && HW > 0 && C > 0 && C % 4 == 0 && C <= 1024,
                "mul_reduce: bad arguments (C=%d)", C);
    GIF_REQUIRE(!scaled || scale, "mul_reduce: scaled output needs scale");
    if (B == 0) return 0;
    hipStream_t s = gif::as_stream(stream);
    int nchunk = mul_reduce_chunks(HW);
    long rpb = (HW + nchunk - 1) / nchunk;
    colsum_stage1<2, T><<<dim3(nchunk, B), 256, 0, s>>>(a, b, scaled, scale, partial, HW, C / 4, rpb, 0.f, 1.f, 1, nullptr, nullptr,
                                                         sizeof(T) == 2 ? gif::f16_sat_flag() : nullptr);
    colsum_stage2<<<dim3(gif::cdiv(C, 64), B), 256, 0, s>>>(partial, out, nchunk, C);
    return gif::check_launch("mul_reduce");
}

template <typename T>
int act_inv_mul_reduce_impl(const T* g, const T* y, const T* residual, const float* bias, float* out, float* partial, int B,
                            int64_t HW, int C, float slope, float gain, gif_stream_t stream) {
    GIF_REQUIRE(g && y && out && partial && B >= 0 && HW > 0 && C > 0 && C % 4 == 0 && C <= 1024,
                "act_inv_mul_reduce: bad arguments (C=%d)", C);
    GIF_REQUIRE(slope > 0.f && gain > 0.f, "act_inv_mul_reduce: activation must be invertible (slope, gain > 0)");
    if (B == 0) return 0;
    hipStream_t s = gif::as_stream(stream);
    int nchunk = mul_reduce_chunks(HW);
    long rpb = (HW + nchunk - 1) / nchunk;
    colsum_stage1<3, T><<<dim3(nchunk, B), 256, 0, s>>>(g, y, static_cast<T*>(nullptr), nullptr, partial, HW, C / 4, rpb, slope, gain,
                                                         1, residual, bias);
    colsum_stage2<<<dim3(gif::cdiv(C, 64), B), 256, 0, s>>>(partial, out, nchunk, C);
    return gif::check_launch("act_inv_mul_reduce");
}

}  // namespace

namespace {

// ------------------------------------------------------------------------------------------------ input assembly (round 4)
// Discriminator.forward (stg2_discriminator.py:48-53) concatenates image and condition along the channel axis; the NHWC
// kernels want that tensor channel-padded (9 -> 12 / 16) and, in f16 mode, as halfs.  torch.cat + F.pad + .to() + .contiguous()
// were 4 full-resolution passes (3.3 ms per iteration at 1024^2, batch 8: profiles/r4_aten_crumbs_f16_1024.txt); this is one:
// one lane per pixel gathers the channels of up to two fp32 sources with arbitrary strides (NCHW: coalesced over the lanes per
// channel) and writes the pixel's padded channel vector (consecutive lanes write consecutive 16 .. 64-byte pieces).
struct PackSrc {
    const float* p;
    int C, c_off;
    long sb, sc, sy, sx;  // element strides
};

template <typename T>
__global__ void __launch_bounds__(256) pack_nhwc_kernel(PackSrc s0, PackSrc s1, T* __restrict__ dst, int H, int W, int Cp, long npix,
                                                        unsigned* sat_flag) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < npix; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W);
        const long r = idx / W;
        const int y = (int)(r % H);
        const long b = r / H;
        const float* b0 = s0.p + b * s0.sb + y * s0.sy + x * s0.sx;
        const float* b1 = s1.p ? s1.p + b * s1.sb + y * s1.sy + x * s1.sx : nullptr;
        T* out = dst + idx * Cp;
        for (int c = 0; c < Cp; c += 4) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ch = c + k;
                float t = 0.f;
                if (ch >= s0.c_off && ch < s0.c_off + s0.C) t = b0[(ch - s0.c_off) * s0.sc];
                else if (b1 && ch >= s1.c_off && ch < s1.c_off + s1.C) t = b1[(ch - s1.c_off) * s1.sc];
                v[k] = t;
            }
            // (PackNhwcFn is also the backward of UnpackNhwcFn: in R1's double backward this store carries f16 gradients, so a clamp or a
            // non-finite value must reach the loss scaler like every other f16 gradient store — advisor finding, round 4)
            gif::store4_flag(out + c, make_float4(v[0], v[1], v[2], v[3]), sat_flag);
        }
    }
}

// the adjoint w.r.t. one source: channels [c_off, c_off + C) of an NHWC tensor -> fp32 [B,H,W,C] (channels_last, unpadded)
template <typename T>
__global__ void __launch_bounds__(256) unpack_nhwc_kernel(const T* __restrict__ src, float* __restrict__ dst, int Cp, int c_off, int C,
                                                          long npix) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < npix; idx += (long)gridDim.x * blockDim.x) {
        const T* in = src + idx * Cp + c_off;
        float* out = dst + idx * C;
        for (int k = 0; k < C; ++k) out[k] = (float)in[k];
    }
}

template <typename T>
int pack_nhwc_impl(const float* src0, int C0, int off0, const int64_t* st0, const float* src1, int C1, int off1, const int64_t* st1,
                   T* dst, int B, int H, int W, int Cp, gif_stream_t stream) {
    GIF_REQUIRE(src0 && dst && st0 && B >= 0 && H > 0 && W > 0 && Cp > 0 && Cp % 4 == 0, "pack_nhwc: bad arguments (Cp=%d)", Cp);
    GIF_REQUIRE(C0 > 0 && off0 >= 0 && off0 + C0 <= Cp, "pack_nhwc: source 0 channels [%d, %d) outside [0, %d)", off0, off0 + C0, Cp);
    GIF_REQUIRE(!src1 || (st1 && C1 > 0 && off1 >= off0 + C0 && off1 + C1 <= Cp), "pack_nhwc: source 1 must follow source 0 inside Cp");
    if (B == 0) return 0;
    PackSrc s0{src0, C0, off0, st0[0], st0[1], st0[2], st0[3]};
    PackSrc s1{src1, src1 ? C1 : 0, off1, src1 ? st1[0] : 0, src1 ? st1[1] : 0, src1 ? st1[2] : 0, src1 ? st1[3] : 0};
    const long npix = (long)B * H * W;
    pack_nhwc_kernel<T><<<ew_grid(npix), 256, 0, gif::as_stream(stream)>>>(s0, s1, dst, H, W, Cp, npix, sizeof(T) == 2 ? gif::f16_sat_flag() : nullptr);
    return gif::check_launch("pack_nhwc");
}

template <typename T>
int unpack_nhwc_impl(const T* src, float* dst, int B, int H, int W, int Cp, int c_off, int C, gif_stream_t stream) {
    GIF_REQUIRE(src && dst && B >= 0 && H > 0 && W > 0 && Cp > 0 && C > 0 && c_off >= 0 && c_off + C <= Cp, "unpack_nhwc: bad arguments");
    if (B == 0) return 0;
    const long npix = (long)B * H * W;
    unpack_nhwc_kernel<T><<<ew_grid(npix), 256, 0, gif::as_stream(stream)>>>(src, dst, Cp, c_off, C, npix);
    return gif::check_launch("unpack_nhwc");
}

}  // namespace

extern "C" {

int gif_upfirdn2d_f32(const float* x, const float* k, float* y, int B, int Hi, int Wi, int C, int Ho, int Wo, int up,
                      int down, int padx0, int pady0, int KH, int KW, int flip, const gif_conv_epilogue* e,
                      gif_stream_t stream) {
    return upfirdn2d_impl<float>(x, k, y, B, Hi, Wi, C, Ho, Wo, up, down, padx0, pady0, KH, KW, flip, e, stream);
}
int gif_upfirdn2d_f16(const void* x, const float* k, void* y, int B, int Hi, int Wi, int C, int Ho, int Wo, int up,
                      int down, int padx0, int pady0, int KH, int KW, int flip, const gif_conv_epilogue* e,
                      gif_stream_t stream) {
    return upfirdn2d_impl<gif::f16>(static_cast<const gif::f16*>(x), k, static_cast<gif::f16*>(y), B, Hi, Wi, C, Ho, Wo, up, down,
                                    padx0, pady0, KH, KW, flip, e, stream);
}

int gif_bias_act_f32(const float* x, const float* bias, const float* residual, float* y, int64_t npix, int C,
                     float slope, float gain, gif_stream_t stream) {
    return bias_act_impl<float>(x, bias, residual, y, npix, C, slope, gain, stream);
}
int gif_bias_act_f16(const void* x, const float* bias, const void* residual, void* y, int64_t npix, int C, float slope,
                     float gain, gif_stream_t stream) {
    return bias_act_impl<gif::f16>(static_cast<const gif::f16*>(x), bias, static_cast<const gif::f16*>(residual),
                                   static_cast<gif::f16*>(y), npix, C, slope, gain, stream);
}

int64_t gif_conv_epilogue_ws_floats(int64_t out_rows, int cout) {
    if (out_rows <= 0 || cout <= 0) return 0;
    return gif::epilogue_ws_rows(out_rows) * (int64_t)((cout + 3) / 4 * 4);
}

int64_t gif_colsum_partial_floats(int64_t npix, int C) {
    if (C <= 0 || C % 4 || C > 1024) return 0;
    return (int64_t)colsum_blocks(npix, C / 4) * C;
}

int gif_bias_act_bwd_f32(const float* gy, const float* y, float* gx, float* gbias, float* partial, int64_t npix, int C,
                         float slope, float gain, gif_stream_t stream) {
    return bias_act_bwd_impl<float>(gy, y, gx, gbias, partial, npix, C, slope, gain, stream);
}
int gif_bias_act_bwd_f16(const void* gy, const void* y, void* gx, float* gbias, float* partial, int64_t npix, int C,
                         float slope, float gain, gif_stream_t stream) {
    return bias_act_bwd_impl<gif::f16>(static_cast<const gif::f16*>(gy), static_cast<const gif::f16*>(y), static_cast<gif::f16*>(gx),
                                       gbias, partial, npix, C, slope, gain, stream);
}

int gif_colsum_f32(const float* x, float* out, float* partial, int64_t npix, int C, gif_stream_t stream) {
    return colsum_impl<float>(x, out, partial, npix, C, stream);
}
int gif_colsum_f16(const void* x, float* out, float* partial, int64_t npix, int C, gif_stream_t stream) {
    return colsum_impl<gif::f16>(static_cast<const gif::f16*>(x), out, partial, npix, C, stream);
}

int gif_mul_reduce_chunks(int64_t HW) { return mul_reduce_chunks(HW); }

int gif_mul_reduce_f32(const float* a, const float* b, const float* scale, float* scaled, float* out, float* partial,
                       int B, int64_t HW, int C, gif_stream_t stream) {
    return mul_reduce_impl<float>(a, b, scale, scaled, out, partial, B, HW, C, stream);
}
int gif_mul_reduce_f16(const void* a, const void* b, const float* scale, void* scaled, float* out, float* partial, int B,
                       int64_t HW, int C, gif_stream_t stream) {
    return mul_reduce_impl<gif::f16>(static_cast<const gif::f16*>(a), static_cast<const gif::f16*>(b), scale,
                                     static_cast<gif::f16*>(scaled), out, partial, B, HW, C, stream);
}

int gif_act_inv_mul_reduce_f32(const float* g, const float* y, const float* residual, const float* bias, float* out,
                               float* partial, int B, int64_t HW, int C, float slope, float gain, gif_stream_t stream) {
    return act_inv_mul_reduce_impl<float>(g, y, residual, bias, out, partial, B, HW, C, slope, gain, stream);
}
int gif_act_inv_mul_reduce_f16(const void* g, const void* y, const void* residual, const float* bias, float* out,
                               float* partial, int B, int64_t HW, int C, float slope, float gain, gif_stream_t stream) {
    return act_inv_mul_reduce_impl<gif::f16>(static_cast<const gif::f16*>(g), static_cast<const gif::f16*>(y),
                                             static_cast<const gif::f16*>(residual), bias, out, partial, B, HW, C, slope, gain,
                                             stream);
}

int gif_pack_nhwc_f32(const float* src0, int C0, int off0, const int64_t* strides0, const float* src1, int C1, int off1,
                      const int64_t* strides1, float* dst, int B, int H, int W, int Cp, gif_stream_t stream) {
    return pack_nhwc_impl<float>(src0, C0, off0, strides0, src1, C1, off1, strides1, dst, B, H, W, Cp, stream);
}
int gif_pack_nhwc_f16(const float* src0, int C0, int off0, const int64_t* strides0, const float* src1, int C1, int off1,
                      const int64_t* strides1, void* dst, int B, int H, int W, int Cp, gif_stream_t stream) {
    return pack_nhwc_impl<gif::f16>(src0, C0, off0, strides0, src1, C1, off1, strides1, static_cast<gif::f16*>(dst), B, H, W, Cp, stream);
}
int gif_unpack_nhwc_f32(const float* src, float* dst, int B, int H, int W, int Cp, int c_off, int C, gif_stream_t stream) {
    return unpack_nhwc_impl<float>(src, dst, B, H, W, Cp, c_off, C, stream);
}
int gif_unpack_nhwc_f16(const void* src, float* dst, int B, int H, int W, int Cp, int c_off, int C, gif_stream_t stream) {
    return unpack_nhwc_impl<gif::f16>(static_cast<const gif::f16*>(src), dst, B, H, W, Cp, c_off, C, stream);
}

int gif_bilinear_down_f32(const float* x, float* y, int B, int R, int S, int C, int backward, gif_stream_t stream) {
    GIF_REQUIRE(x && y && B >= 0 && R > 0 && S > 0 && C > 0 && C % 4 == 0, "bilinear_down: bad arguments");
    GIF_REQUIRE(R % S == 0 && (R == S || (R / S) % 2 == 0), "bilinear_down: R/S must be 1 or an even integer (got %d/%d)", R, S);
    if (B == 0) return 0;
    hipStream_t s = gif::as_stream(stream);
    if (!backward)
        bilinear_down_kernel<<<ew_grid((long)B * S * S * (C / 4)), 256, 0, s>>>(x, y, B, R, S, C / 4);
    else  // x = grad of the level [B,S,S,C], y = grad of the full-resolution condition [B,R,R,C]
        bilinear_down_bwd_kernel<<<ew_grid((long)B * R * R * (C / 4)), 256, 0, s>>>(x, y, B, R, S, C / 4);
    return gif::check_launch("bilinear_down");
}

int gif_mbstd_fwd_f32(const float* x, float* y, float* stat, int B, int H, int W, int C, int Cy, int G,
                      gif_stream_t stream) {
    GIF_REQUIRE(x && y && stat && B > 0 && G >= 1 && G <= 8 && B % G == 0, "mbstd: batch %d not divisible by group %d", B, G);
    GIF_REQUIRE(C > 0 && Cy >= C + 1 && Cy % 4 == 0, "mbstd: bad channel counts C=%d Cy=%d", C, Cy);
    hipStream_t s = gif::as_stream(stream);
    int M = B / G;
    mbstd_stat_kernel<<<M, 256, 0, s>>>(x, stat, M, G, (long)H * W * C);
    long total = (long)B * H * W * (Cy / 4);
    mbstd_write_kernel<<<ew_grid(total), 256, 0, s>>>(x, stat, y, B, H * W, C, Cy, M);
    return gif::check_launch("mbstd_fwd");
}

int gif_mbstd_bwd_f32(const float* x, const float* gy, float* gx, int B, int H, int W, int C, int Cy, int G,
                      gif_stream_t stream) {
    GIF_REQUIRE(x && gy && gx && B > 0 && G >= 1 && G <= 8 && B % G == 0, "mbstd_bwd: bad batch/group");
    GIF_REQUIRE(C > 0 && Cy >= C + 1, "mbstd_bwd: bad channel counts");
    int M = B / G;
    mbstd_bwd_kernel<<<M, 256, 0, gif::as_stream(stream)>>>(x, gy, gx, M, G, H * W, C, Cy);
    return gif::check_launch("mbstd_bwd");
}

int gif_sqnorm_per_sample_f32(const float* g, float* out, int B, int64_t n, gif_stream_t stream) {
    GIF_REQUIRE(g && out && B >= 0 && n > 0, "sqnorm: bad arguments");
    if (B == 0) return 0;
    sqnorm_kernel<<<B, 1024, 0, gif::as_stream(stream)>>>(g, out, n);
    return gif::check_launch("sqnorm");
}
}

// ---------------------------------------------------------------------------------------------------------
// texture-interpolation loss core (InterpolatedTextureLoss.pairwise_texture_loss, loss_functions/losses.py:147-160, with
// the common-visibility masking of its call site :171-174 folded in):
//   loss = mean_{c,h,w} sigmoid(((a - b) * ma * mb)^2) * f          a, b [C,H,W]; ma, mb [H,W] u8 or null; f [H,W]
// Two-stage deterministic reduction; the backward is one pointwise pass (gb = -ga).
// ---------------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void __launch_bounds__(256) tex_pair_loss_stage1(const float* __restrict__ a, const float* __restrict__ b,
                                                            const uint8_t* __restrict__ ma, const uint8_t* __restrict__ mb,
                                                            const float* __restrict__ f, float* __restrict__ partial,
                                                            long n, long HW) {
    __shared__ float red[256];
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long hw = i % HW;
        float d = a[i] - b[i];
        if (ma && !ma[hw]) d = 0.f;
        if (mb && !mb[hw]) d = 0.f;
        acc += sigmoidf_(d * d) * f[hw];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void __launch_bounds__(256) tex_pair_loss_stage2(const float* __restrict__ partial, float* __restrict__ loss,
                                                            int nblk, float inv_n) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = red[0] * inv_n;
}

__global__ void __launch_bounds__(256) tex_pair_loss_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                const uint8_t* __restrict__ ma, const uint8_t* __restrict__ mb,
                                                                const float* __restrict__ f, const float* __restrict__ gloss,
                                                                float* __restrict__ ga, long n, long HW, float inv_n) {
    const float g = *gloss * inv_n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long hw = i % HW;
        float d = a[i] - b[i];
        bool vis = true;
        if (ma && !ma[hw]) vis = false;
        if (mb && !mb[hw]) vis = false;
        float out = 0.f;
        if (vis) {
            const float s = sigmoidf_(d * d);
            out = g * f[hw] * s * (1.f - s) * 2.f * d;
        }
        ga[i] = out;
    }
}

inline int tex_pair_blocks(long n) {
    long b = (n + 255) / 256;
    return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int gif_texture_pair_loss_partials(int64_t n) { return tex_pair_blocks(n); }

int gif_texture_pair_loss_f32(const float* a, const float* b, const uint8_t* ma, const uint8_t* mb, const float* f,
                              float* partial, float* loss, int C, int64_t HW, gif_stream_t stream) {
    GIF_REQUIRE(a && b && f && partial && loss && C > 0 && HW > 0, "texture_pair_loss: bad arguments");
    const long n = (long)C * HW;
    const int nblk = tex_pair_blocks(n);
    hipStream_t s = gif::as_stream(stream);
    tex_pair_loss_stage1<<<nblk, 256, 0, s>>>(a, b, ma, mb, f, partial, n, HW);
    tex_pair_loss_stage2<<<1, 256, 0, s>>>(partial, loss, nblk, 1.0f / (float)n);
    return gif::check_launch("texture_pair_loss");
}

int gif_texture_pair_loss_bwd_f32(const float* a, const float* b, const uint8_t* ma, const uint8_t* mb, const float* f,
                                  const float* gloss, float* ga, int C, int64_t HW, gif_stream_t stream) {
    GIF_REQUIRE(a && b && f && gloss && ga && C > 0 && HW > 0, "texture_pair_loss_bwd: bad arguments");
    const long n = (long)C * HW;
    tex_pair_loss_bwd_kernel<<<tex_pair_blocks(n), 256, 0, gif::as_stream(stream)>>>(a, b, ma, mb, f, gloss, ga, n, HW,
                                                                                      1.0f / (float)n);
    return gif::check_launch("texture_pair_loss_bwd");
}
}
