// BASELINE PROBE (tools/raster_bench.py only, never loaded by gif_amd): the round-2 rasteriser — one lane per (image, face)
// over the face's whole bounding box — kept to measure the round-3 kernel (gif_amd/csrc/rasterize.hip) against.
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -Igif_amd/csrc -Iinclude tools/probes/rasterize_v1.hip -o tools/probes/libraster_v1.so
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

namespace gif {
void set_error(const char*, ...) {}
}  // namespace gif

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

template <typename T>
struct Face {
    T x0, y0, z0, x1, y1, z1, x2, y2, z2;
};

template <typename T>
__device__ __forceinline__ Face<T> load_face(const T* __restrict__ p) {
    Face<T> f;
    f.x0 = p[0]; f.y0 = p[1]; f.z0 = p[2];
    f.x1 = p[3]; f.y1 = p[4]; f.z1 = p[5];
    f.x2 = p[6]; f.y2 = p[7]; f.z2 = p[8];
    return f;
}

// check_face_frontside, .cu:31-34
template <typename T>
__device__ __forceinline__ bool front_facing(const Face<T>& f) {
    return (f.y2 - f.y0) * (f.x1 - f.x0) < (f.y1 - f.y0) * (f.x2 - f.x0);
}

// barycentric_weight, .cu:78-109 (dot-product form; degenerate => inverDeno = 0)
template <typename T>
struct BaryCtx {
    T v0x, v0y, v1x, v1y, dot00, dot01, dot11, inv;
};
template <typename T>
__device__ __forceinline__ BaryCtx<T> bary_setup(const Face<T>& f) {
    BaryCtx<T> c;
    c.v0x = f.x2 - f.x0; c.v0y = f.y2 - f.y0;
    c.v1x = f.x1 - f.x0; c.v1y = f.y1 - f.y0;
    c.dot00 = c.v0x * c.v0x + c.v0y * c.v0y;
    c.dot01 = c.v0x * c.v1x + c.v0y * c.v1y;
    c.dot11 = c.v1x * c.v1x + c.v1y * c.v1y;
    T den = c.dot00 * c.dot11 - c.dot01 * c.dot01;
    c.inv = (den == T(0)) ? T(0) : T(1) / den;
    return c;
}
template <typename T>
__device__ __forceinline__ void bary_at(const Face<T>& f, const BaryCtx<T>& c, T px, T py, T* w) {
    T v2x = px - f.x0, v2y = py - f.y0;
    T dot02 = c.v0x * v2x + c.v0y * v2y;
    T dot12 = c.v1x * v2x + c.v1y * v2y;
    T u = (c.dot11 * dot02 - c.dot01 * dot12) * c.inv;
    T v = (c.dot00 * dot12 - c.dot01 * dot02) * c.inv;
    w[0] = T(1) - u - v;
    w[1] = v;
    w[2] = u;
}
template <typename T>
__device__ __forceinline__ bool inside(const T* w) { return w[2] >= 0 && w[1] >= 0 && w[0] > 0; }  // .cu:144
template <typename T>
__device__ __forceinline__ T persp_depth(const Face<T>& f, const T* w) {                           // .cu:148
    return T(1) / (w[0] / f.z0 + w[1] / f.z1 + w[2] / f.z2);
}

// bbox of a face clamped to the image, .cu:133-136
template <typename T>
__device__ __forceinline__ void face_bbox(const Face<T>& f, int H, int W, int& x_min, int& x_max, int& y_min, int& y_max) {
    x_min = max((int)ceil(fmin(f.x0, fmin(f.x1, f.x2))), 0);
    x_max = min((int)floor(fmax(f.x0, fmax(f.x1, f.x2))), W - 1);
    y_min = max((int)ceil(fmin(f.y0, fmin(f.y1, f.y2))), 0);
    y_max = min((int)floor(fmax(f.y0, fmax(f.y1, f.y2))), H - 1);
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
    Face<float> f = load_face(fv + i * 9);
    if (!front_facing(f)) return;
    int x_min, x_max, y_min, y_max;
    face_bbox(f, H, W, x_min, x_max, y_min, y_max);
    BaryCtx<float> c = bary_setup(f);
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

// ---- float64 variant (the reference dispatches AT_DISPATCH_FLOATING_TYPES, .cu:252,295).  A 64-bit depth and a face index
// do not fit one 64-bit atomic, so the winner is found in two passes over the faces: (A) 64-bit atomicMin of the ordered depth
// bits, (B) among the faces whose depth at the pixel EQUALS that minimum (recomputed: same arithmetic, same bits) a 32-bit
// atomicMin of the face index — the same deterministic "lowest face index wins an exact tie" rule as the float path.
__device__ __forceinline__ unsigned long long ordered_bits64(double d) {
    unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double from_ordered_bits64(unsigned long long u) {
    return __longlong_as_double((long long)((u & 0x8000000000000000ull) ? (u & 0x7FFFFFFFFFFFFFFFull) : ~u));
}

__global__ void raster_init_keys64(const double* __restrict__ depth, unsigned long long* __restrict__ zkey,
                                   uint32_t* __restrict__ fkey, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        zkey[i] = ordered_bits64(depth[i]);
        fkey[i] = kNoFace;
    }
}

template <int PASS>
__global__ void __launch_bounds__(256)
raster_faces64(const double* __restrict__ fv, unsigned long long* __restrict__ zkey, uint32_t* __restrict__ fkey, int B, int F,
               int H, int W) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * F) return;
    int b = (int)(i / F);
    uint32_t fidx = (uint32_t)(i - (long)b * F);
    Face<double> f = load_face(fv + i * 9);
    if (!front_facing(f)) return;
    int x_min, x_max, y_min, y_max;
    face_bbox(f, H, W, x_min, x_max, y_min, y_max);
    BaryCtx<double> c = bary_setup(f);
    const long base = (long)b * H * W;
    for (int y = y_min; y <= y_max; ++y) {
        for (int x = x_min; x <= x_max; ++x) {
            double w[3];
            bary_at(f, c, (double)x, (double)y, w);
            if (inside(w)) {
                double zp = persp_depth(f, w);
                if (zp == zp) {
                    const long p = base + (long)y * W + x;
                    const unsigned long long k = ordered_bits64(zp);
                    if (PASS == 0) atomicMin(zkey + p, k);
                    else if (zkey[p] == k) atomicMin(fkey + p, fidx);
                }
            }
        }
    }
}

template <typename T, bool COLORS>
__global__ void __launch_bounds__(256)
raster_resolve(const T* __restrict__ fv, const T* __restrict__ fc, const unsigned long long* __restrict__ key,
               const uint32_t* __restrict__ fkey, T* __restrict__ depth, int32_t* __restrict__ tri, T* __restrict__ out3, int B,
               int F, int H, int W) {
    long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long hw = (long)H * W;
    if (p >= (long)B * hw) return;
    unsigned long long k = key[p];
    uint32_t fidx = sizeof(T) == 4 ? (uint32_t)(k & 0xFFFFFFFFu) : fkey[p];
    if (fidx == kNoFace) return;  // pixel keeps the caller's depth / tri / payload
    int b = (int)(p / hw);
    int rem = (int)(p - (long)b * hw);
    int y = rem / W, x = rem - y * W;
    long fi = (long)b * F + fidx;
    Face<T> f = load_face(fv + fi * 9);
    BaryCtx<T> c = bary_setup(f);
    T w[3];
    bary_at(f, c, (T)x, (T)y, w);
    if (sizeof(T) == 4) depth[p] = (T)from_ordered_bits((uint32_t)(k >> 32));
    else depth[p] = (T)from_ordered_bits64(k);
    tri[p] = (int32_t)fidx;
    if (COLORS) {
        const T* cl = fc + fi * 9;  // [3 verts][3 channels], .cu:189-194
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
        raster_resolve<float, true><<<gif::cdiv(npix, 256), 256, 0, s>>>(fv, fc, key, nullptr, depth, tri, out3, B, F, H, W);
    else
        raster_resolve<float, false><<<gif::cdiv(npix, 256), 256, 0, s>>>(fv, nullptr, key, nullptr, depth, tri, out3, B, F, H, W);
    return gif::check_launch("rasterize");
}

int run64(const double* fv, const double* fc, double* depth, int32_t* tri, double* out3, int B, int F, int H, int W,
          void* workspace, gif_stream_t stream) {
    GIF_REQUIRE(B >= 0 && F >= 0 && H > 0 && W > 0, "rasterize_f64: bad dims B=%d F=%d H=%d W=%d", B, F, H, W);
    long npix = (long)B * H * W;
    if (npix == 0 || F == 0) return 0;
    GIF_REQUIRE(fv && depth && tri && out3 && workspace, "rasterize_f64: null pointer");
    GIF_REQUIRE(((uintptr_t)workspace & 7) == 0, "rasterize_f64: workspace must be 8-byte aligned");
    hipStream_t s = gif::as_stream(stream);
    auto* zkey = reinterpret_cast<unsigned long long*>(workspace);
    auto* fkey = reinterpret_cast<uint32_t*>(zkey + npix);
    const int fb = gif::cdiv((long)B * F, 256), pb = gif::cdiv(npix, 256);
    raster_init_keys64<<<pb, 256, 0, s>>>(depth, zkey, fkey, npix);
    raster_faces64<0><<<fb, 256, 0, s>>>(fv, zkey, fkey, B, F, H, W);
    raster_faces64<1><<<fb, 256, 0, s>>>(fv, zkey, fkey, B, F, H, W);
    if (fc) raster_resolve<double, true><<<pb, 256, 0, s>>>(fv, fc, zkey, fkey, depth, tri, out3, B, F, H, W);
    else raster_resolve<double, false><<<pb, 256, 0, s>>>(fv, nullptr, zkey, fkey, depth, tri, out3, B, F, H, W);
    return gif::check_launch("rasterize_f64");
}

}  // namespace

extern "C" {

int64_t v1_gif_rasterize_workspace_bytes(int B, int H, int W) { return (int64_t)B * H * W * 8; }

int v1_gif_rasterize_f32(const float* face_vertices, float* depth, int32_t* tri, float* bary, int B, int F, int H,
                      int W, void* workspace, gif_stream_t stream) {
    return run(face_vertices, nullptr, depth, tri, bary, B, F, H, W, workspace, stream);
}

int v1_gif_rasterize_colors_f32(const float* face_vertices, const float* face_colors, float* depth, int32_t* tri,
                             float* images, int B, int F, int H, int W, void* workspace, gif_stream_t stream) {
    GIF_REQUIRE(face_colors || (long)B * F == 0, "rasterize_colors: null face_colors");
    return run(face_vertices, face_colors, depth, tri, images, B, F, H, W, workspace, stream);
}

int64_t v1_gif_rasterize_workspace_bytes_f64(int B, int H, int W) { return (int64_t)B * H * W * 12; }

int v1_gif_rasterize_f64(const double* face_vertices, double* depth, int32_t* tri, double* bary, int B, int F, int H, int W,
                      void* workspace, gif_stream_t stream) {
    return run64(face_vertices, nullptr, depth, tri, bary, B, F, H, W, workspace, stream);
}

int v1_gif_rasterize_colors_f64(const double* face_vertices, const double* face_colors, double* depth, int32_t* tri,
                             double* images, int B, int F, int H, int W, void* workspace, gif_stream_t stream) {
    GIF_REQUIRE(face_colors || (long)B * F == 0, "rasterize_colors_f64: null face_colors");
    return run64(face_vertices, face_colors, depth, tri, images, B, F, H, W, workspace, stream);
}
}
