// Image resize on NCHW fp32 batches — replaces fast_image_reshape() (dataset_loaders.py:26-34 =
// torch.nn.functional.interpolate(mode='bicubic' | 'bilinear', align_corners=False, no antialias)) of the input /
// visualisation pipeline so that resizing stays on the device.
//   src = scale * (dst + 0.5) - 0.5, scale = in / out
//   bilinear: src clamped at 0, taps (i0, min(i0 + 1, n - 1)), weights (1 - l, l)
//   bicubic : 4 taps i0 - 1 .. i0 + 2 with indices clamped to [0, n - 1], cubic convolution weights A = -0.75
// One lane per output pixel of one (b, c) plane; rows first, then columns (the order ATen's kernel uses).
// Backward: scatter of the same weights with atomicAdd (fp32; accumulation order is not deterministic).
#include "common.h"

namespace {

__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

struct Taps {
    int idx[4];
    float w[4];
    int n;
};

__device__ __forceinline__ Taps make_taps(int dst, int nin, float scale, int mode) {
    Taps t;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (mode == 0) {  // bilinear
        if (src < 0.f) src = 0.f;
        int i0 = (int)src;
        if (i0 > nin - 1) i0 = nin - 1;
        int i1 = i0 + (i0 < nin - 1 ? 1 : 0);
        float l = src - (float)i0;
        t.idx[0] = i0; t.idx[1] = i1;
        t.w[0] = 1.f - l; t.w[1] = l;
        t.n = 2;
    } else {  // bicubic
        const float A = -0.75f;
        float fl = floorf(src);
        int i0 = (int)fl;
        float x = src - fl;
        t.w[0] = cubic2(x + 1.f, A);
        t.w[1] = cubic1(x, A);
        t.w[2] = cubic1(1.f - x, A);
        t.w[3] = cubic2(2.f - x, A);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int i = i0 - 1 + k;
            t.idx[k] = i < 0 ? 0 : (i > nin - 1 ? nin - 1 : i);
        }
        t.n = 4;
    }
    return t;
}

__global__ void __launch_bounds__(256) resize_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long planes,
                                                         int Hi, int Wi, int Ho, int Wo, float sh, float sw, int mode) {
#pragma clang fp contract(off)
    const long total = planes * Ho * Wo;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % Wo);
        const long r = idx / Wo;
        const int oy = (int)(r % Ho);
        const long pl = r / Ho;
        const float* xp = x + pl * Hi * Wi;
        const Taps ty = make_taps(oy, Hi, sh, mode), tx = make_taps(ox, Wi, sw, mode);
        float acc = 0.f;
        for (int a = 0; a < ty.n; ++a) {
            const float* row = xp + (long)ty.idx[a] * Wi;
            float h = 0.f;
            for (int b = 0; b < tx.n; ++b) h += tx.w[b] * row[tx.idx[b]];
            acc += ty.w[a] * h;
        }
        y[idx] = acc;
    }
}

__global__ void __launch_bounds__(256) resize_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, long planes,
                                                         int Hi, int Wi, int Ho, int Wo, float sh, float sw, int mode) {
    const long total = planes * Ho * Wo;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % Wo);
        const long r = idx / Wo;
        const int oy = (int)(r % Ho);
        const long pl = r / Ho;
        float* gp = gx + pl * Hi * Wi;
        const Taps ty = make_taps(oy, Hi, sh, mode), tx = make_taps(ox, Wi, sw, mode);
        const float g = gy[idx];
        for (int a = 0; a < ty.n; ++a)
            for (int b = 0; b < tx.n; ++b) atomicAdd(gp + (long)ty.idx[a] * Wi + tx.idx[b], g * ty.w[a] * tx.w[b]);
    }
}

}  // namespace

extern "C" {

// x [planes, Hi, Wi] -> y [planes, Ho, Wo]  (planes = B*C of an NCHW tensor); mode 0 = bilinear, 1 = bicubic
int gif_resize_f32(const float* x, float* y, int64_t planes, int Hi, int Wi, int Ho, int Wo, int mode, gif_stream_t stream) {
    GIF_REQUIRE(x && y && planes >= 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "resize: bad arguments");
    GIF_REQUIRE(mode == 0 || mode == 1, "resize: mode must be 0 (bilinear) or 1 (bicubic)");
    if (planes == 0) return 0;
    long total = planes * Ho * Wo;
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    resize_fwd_kernel<<<(unsigned)blocks, 256, 0, gif::as_stream(stream)>>>(x, y, planes, Hi, Wi, Ho, Wo, (float)Hi / Ho,
                                                                             (float)Wi / Wo, mode);
    return gif::check_launch("resize");
}

// gx [planes, Hi, Wi] (zeroed inside) += adjoint of the resize applied to gy [planes, Ho, Wo]
int gif_resize_bwd_f32(const float* gy, float* gx, int64_t planes, int Hi, int Wi, int Ho, int Wo, int mode,
                       gif_stream_t stream) {
    GIF_REQUIRE(gy && gx && planes >= 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "resize_bwd: bad arguments");
    GIF_REQUIRE(mode == 0 || mode == 1, "resize_bwd: mode must be 0 (bilinear) or 1 (bicubic)");
    if (planes == 0) return 0;
    hipStream_t s = gif::as_stream(stream);
    if (hipMemsetAsync(gx, 0, (size_t)planes * Hi * Wi * sizeof(float), s) != hipSuccess) return gif::check_launch("resize_bwd memset");
    long total = planes * Ho * Wo;
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    resize_bwd_kernel<<<(unsigned)blocks, 256, 0, s>>>(gy, gx, planes, Hi, Wi, Ho, Wo, (float)Hi / Ho, (float)Wi / Wo, mode);
    return gif::check_launch("resize_bwd");
}
}
