// Z-buffer triangle rasteriser with barycentric attribute interpolation for gfx950.
//
// Replaces the reference's only native code,
//   my_utils/standard_rasterize_cuda/standard_rasterize_cuda_kernel.cu:111-233 (+ host :237-320),
// with a different, race-free formulation:
//   1. init   : key[p] = (ordered_bits(depth_in[p]) << 32) | 0xFFFFFFFF
//   2. faces  : one lane per (image, face); every covered pixel does ONE 64-bit atomicMin of
//               (ordered_bits(zp) << 32) | face  — depth test and winner selection in a single atomic,
//               so the reference's second launch (:252-269, a race work-around) is not needed and
//               exact-depth ties deterministically go to the lowest face index;
//   3. resolve: one lane per pixel re-evaluates the winning face at that pixel (same fp32 operation
//               order => same bits) and writes depth / face index / barycentrics or colours.
// Arithmetic follows the reference operation by operation with FP contraction off, so results are
// bit-identical to oracle/rasterize_ref.c.
#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr uint32_t kNoFace = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t ordered_bits(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

struct Face {
    float x0, y0, z0, x1, y1, z1, x2, y2, z2;
};

__device__ __forceinline__ Face load_face(const float* __restrict__ p) {
    Face f;
    f.x0 = p[0]; f.y0 = p[1]; f.z0 = p[2];
    f.x1 = p[3]; f.y1 = p[4]; f.z1 = p[5];
    f.x2 = p[6]; f.y2 = p[7]; f.z2 = p[8];
    return f;
}

// check_face_frontside, .cu:31-34
__device__ __forceinline__ bool front_facing(const Face& f) {
    return (f.y2 - f.y0) * (f.x1 - f.x0) < (f.y1 - f.y0) * (f.x2 - f.x0);
}

// barycentric_weight, .cu:78-109 (dot-product form; degenerate => inverDeno = 0)
struct BaryCtx {
    float v0x, v0y, v1x, v1y, dot00, dot01, dot11, inv;
};
__device__ __forceinline__ BaryCtx bary_setup(const Face& f) {
    BaryCtx c;
    c.v0x = f.x2 - f.x0; c.v0y = f.y2 - f.y0;
    c.v1x = f.x1 - f.x0; c.v1y = f.y1 - f.y0;
    c.dot00 = c.v0x * c.v0x + c.v0y * c.v0y;
    c.dot01 = c.v0x * c.v1x + c.v0y * c.v1y;
    c.dot11 = c.v1x * c.v1x + c.v1y * c.v1y;
    float den = c.dot00 * c.dot11 - c.dot01 * c.dot01;
    c.inv = (den == 0.0f) ? 0.0f : 1.0f / den;
    return c;
}
__device__ __forceinline__ void bary_at(const Face& f, const BaryCtx& c, float px, float py, float* w) {
    float v2x = px - f.x0, v2y = py - f.y0;
    float dot02 = c.v0x * v2x + c.v0y * v2y;
    float dot12 = c.v1x * v2x + c.v1y * v2y;
    float u = (c.dot11 * dot02 - c.dot01 * dot12) * c.inv;
    float v = (c.dot00 * dot12 - c.dot01 * dot02) * c.inv;
    w[0] = 1.0f - u - v;
    w[1] = v;
    w[2] = u;
}
__device__ __forceinline__ bool inside(const float* w) { return w[2] >= 0 && w[1] >= 0 && w[0] > 0; }  // .cu:144
__device__ __forceinline__ float persp_depth(const Face& f, const float* w) {                           // .cu:148
    return 1.0f / (w[0] / f.z0 + w[1] / f.z1 + w[2] / f.z2);
}

__global__ void raster_init_keys(const float* __restrict__ depth, unsigned long long* __restrict__ key, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) key[i] = ((unsigned long long)ordered_bits(depth[i]) << 32) | kNoFace;
}

__global__ void __launch_bounds__(256)
raster_faces(const float* __restrict__ fv, unsigned long long* __restrict__ key, int B, int F, int H, int W) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * F) return;
    int b = (int)(i / F);
    uint32_t fidx = (uint32_t)(i - (long)b * F);
    Face f = load_face(fv + i * 9);
    if (!front_facing(f)) return;
    // bbox, .cu:133-136
    int x_min = max((int)ceilf(fminf(f.x0, fminf(f.x1, f.x2))), 0);
    int x_max = min((int)floorf(fmaxf(f.x0, fmaxf(f.x1, f.x2))), W - 1);
    int y_min = max((int)ceilf(fminf(f.y0, fminf(f.y1, f.y2))), 0);
    int y_max = min((int)floorf(fmaxf(f.y0, fmaxf(f.y1, f.y2))), H - 1);
    BaryCtx c = bary_setup(f);
    unsigned long long* kb = key + (long)b * H * W;
    for (int y = y_min; y <= y_max; ++y) {
        for (int x = x_min; x <= x_max; ++x) {
            float w[3];
            bary_at(f, c, (float)x, (float)y, w);
            if (inside(w)) {
                float zp = persp_depth(f, w);
                if (zp == zp) {  // NaN never wins (fminf in the reference's atomicMin, .cu:8-18)
                    unsigned long long k = ((unsigned long long)ordered_bits(zp) << 32) | fidx;
                    atomicMin(kb + (long)y * W + x, k);
                }
            }
        }
    }
}

template <bool COLORS>
__global__ void __launch_bounds__(256)
raster_resolve(const float* __restrict__ fv, const float* __restrict__ fc,
               const unsigned long long* __restrict__ key, float* __restrict__ depth,
               int32_t* __restrict__ tri, float* __restrict__ out3, int B, int F, int H, int W) {
    long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long hw = (long)H * W;
    if (p >= (long)B * hw) return;
    unsigned long long k = key[p];
    uint32_t fidx = (uint32_t)(k & 0xFFFFFFFFu);
    if (fidx == kNoFace) return;  // pixel keeps the caller's depth / tri / payload
    int b = (int)(p / hw);
    int rem = (int)(p - (long)b * hw);
    int y = rem / W, x = rem - y * W;
    long fi = (long)b * F + fidx;
    Face f = load_face(fv + fi * 9);
    BaryCtx c = bary_setup(f);
    float w[3];
    bary_at(f, c, (float)x, (float)y, w);
    depth[p] = from_ordered_bits((uint32_t)(k >> 32));
    tri[p] = (int32_t)fidx;
    if (COLORS) {
        const float* cl = fc + fi * 9;  // [3 verts][3 channels], .cu:189-194
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) out3[p * 3 + ch] = w[0] * cl[ch] + w[1] * cl[3 + ch] + w[2] * cl[6 + ch];
    } else {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) out3[p * 3 + ch] = w[ch];
    }
}

int run(const float* fv, const float* fc, float* depth, int32_t* tri, float* out3, int B, int F, int H, int W,
        void* workspace, gif_stream_t stream) {
    GIF_REQUIRE(B >= 0 && F >= 0 && H > 0 && W > 0, "rasterize: bad dims B=%d F=%d H=%d W=%d", B, F, H, W);
    long npix = (long)B * H * W;
    if (npix == 0 || F == 0) return 0;
    GIF_REQUIRE(fv && depth && tri && out3 && workspace, "rasterize: null pointer");
    GIF_REQUIRE(((uintptr_t)workspace & 7) == 0, "rasterize: workspace must be 8-byte aligned");
    hipStream_t s = gif::as_stream(stream);
    auto* key = reinterpret_cast<unsigned long long*>(workspace);
    raster_init_keys<<<gif::cdiv(npix, 256), 256, 0, s>>>(depth, key, npix);
    raster_faces<<<gif::cdiv((long)B * F, 256), 256, 0, s>>>(fv, key, B, F, H, W);
    if (fc)
        raster_resolve<true><<<gif::cdiv(npix, 256), 256, 0, s>>>(fv, fc, key, depth, tri, out3, B, F, H, W);
    else
        raster_resolve<false><<<gif::cdiv(npix, 256), 256, 0, s>>>(fv, nullptr, key, depth, tri, out3, B, F, H, W);
    return gif::check_launch("rasterize");
}

}  // namespace

extern "C" {

int64_t gif_rasterize_workspace_bytes(int B, int H, int W) { return (int64_t)B * H * W * 8; }

int gif_rasterize_f32(const float* face_vertices, float* depth, int32_t* tri, float* bary, int B, int F, int H,
                      int W, void* workspace, gif_stream_t stream) {
    return run(face_vertices, nullptr, depth, tri, bary, B, F, H, W, workspace, stream);
}

int gif_rasterize_colors_f32(const float* face_vertices, const float* face_colors, float* depth, int32_t* tri,
                             float* images, int B, int F, int H, int W, void* workspace, gif_stream_t stream) {
    GIF_REQUIRE(face_colors || (long)B * F == 0, "rasterize_colors: null face_colors");
    return run(face_vertices, face_colors, depth, tri, images, B, F, H, W, workspace, stream);
}
}
