// Z-buffer triangle rasteriser with barycentric attribute interpolation for gfx950.
//
// Replaces the reference's only native code,
//   my_utils/standard_rasterize_cuda/standard_rasterize_cuda_kernel.cu:111-233 (+ host :237-320),
// with a different, race-free formulation built around the CU's LDS (160 KB): the image is cut into 64x64-pixel tiles whose
// z-buffers live in LDS; depth test and winner selection are ONE 64-bit LDS atomic-min of (ordered_bits(zp) << 32) | face per
// covered pixel, so the reference's second launch (:252-269, a race work-around) is not needed, no per-pixel atomic ever
// reaches HBM, and exact-depth ties deterministically go to the lowest face index.  See raster_bin / raster_tiles below.  (The
// reference runs one thread per face over its whole bounding box, :111-167: one large triangle serialises a lane while
// its 63 neighbours idle, and every pixel test is a global atomic.)
// Arithmetic follows the reference operation by operation with FP contraction off, so results are
// bit-identical to oracle/rasterize_ref.c.
#include <atomic>
#include <mutex>
#include <unordered_set>

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

// ---------------------------------------------------------------------------------------------------------------------
// Kernel 1, raster_bin: one lane per (image, face): front-facing test + clamped bounding box, then the face index is
// appended to the list of every 64x64-pixel tile its box overlaps.  A 256-face workgroup first counts per tile in LDS, takes
// ONE global atomicAdd per touched tile for the whole group, and scatters with LDS-ranked offsets — ~7 k global atomics for
// 32 x 13 776 faces instead of one per face.  (List order is not deterministic; the rasterised result is: see raster_tiles.)
//   workspace: count[B][T] (zeroed by the caller's memset node) | list[B][T][F]   (T = tiles per image)
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kBinThreads = 256;
constexpr int kMaxLdsTiles = 1024;  // images up to 2048 x 2048 count in LDS; larger ones fall back to one global atomic per entry

template <typename T>
__global__ void __launch_bounds__(kBinThreads)
raster_bin(const T* __restrict__ fv, uint32_t* __restrict__ count, uint32_t* __restrict__ list, int F, int H, int W, int tiles_x,
           int tiles_y) {
    __shared__ uint32_t cnt[kMaxLdsTiles], base[kMaxLdsTiles];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int nt = tiles_x * tiles_y;
    const bool lds = nt <= kMaxLdsTiles;
    const int fi = blockIdx.x * kBinThreads + tid;
    int tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;
    if (fi < F) {
        const Face<T> f = load_face(fv + ((long)b * F + fi) * 9);
        if (front_facing(f)) {
            int x_min, x_max, y_min, y_max;
            face_bbox(f, H, W, x_min, x_max, y_min, y_max);
            if (x_min <= x_max && y_min <= y_max) {
                tx0 = x_min >> 6; tx1 = x_max >> 6; ty0 = y_min >> 6; ty1 = y_max >> 6;
            }
        }
    }
    uint32_t* cb = count + (long)b * nt;
    uint32_t* lb = list + (long)b * nt * F;
    if (!lds) {
        for (int ty = ty0; ty <= ty1; ++ty)
            for (int tx = tx0; tx <= tx1; ++tx) {
                const int t = ty * tiles_x + tx;
                lb[(long)t * F + atomicAdd(cb + t, 1u)] = (uint32_t)fi;
            }
        return;
    }
    for (int t = tid; t < nt; t += kBinThreads) cnt[t] = 0;
    __syncthreads();
    for (int ty = ty0; ty <= ty1; ++ty)
        for (int tx = tx0; tx <= tx1; ++tx) atomicAdd(&cnt[ty * tiles_x + tx], 1u);
    __syncthreads();
    for (int t = tid; t < nt; t += kBinThreads) {
        const uint32_t c = cnt[t];
        base[t] = c ? atomicAdd(cb + t, c) : 0u;
        cnt[t] = 0;
    }
    __syncthreads();
    for (int ty = ty0; ty <= ty1; ++ty)
        for (int tx = tx0; tx <= tx1; ++tx) {
            const int t = ty * tiles_x + tx;
            lb[(long)t * F + base[t] + atomicAdd(&cnt[t], 1u)] = (uint32_t)fi;
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// Kernel 2, raster_tiles: one 512- or 1024-thread workgroup per (image, 64x64-pixel tile); the tile's z-buffer lives in LDS.
//   seed   : key[p] = (ordered_bits(depth_in[p]) << 32) | 0xFFFFFFFF from the caller's depth buffer
//   shade  : the tile's face list, one face per lane, box clipped to the tile: <= 16 px => the lane walks it; larger => the
//            wave walks it together, 8x8 pixels per step (face broadcast by shuffles).  Every covered pixel does ONE 64-bit
//            LDS atomic-min of (ordered_bits(zp) << 32) | face.  Per-pixel arithmetic is identical in both classes and to
//            the oracle, and min() is order independent: the result does not depend on the list order.
//   resolve: per pixel of the tile, re-evaluate the winner (same fp operation order => same bits), write depth / face
//            index / barycentrics or colours; untouched pixels keep the caller's contents.
// No global atomic per pixel, no key buffer in HBM, no second launch; a screen-filling triangle costs every tile 64 wave-steps.
// float64 (the reference dispatches AT_DISPATCH_FLOATING_TYPES, .cu:252,295): a 64-bit depth and a face index do not fit
// one 64-bit atomic, so shade runs twice: (PASS 0) atomic-min of the ordered depth bits, (PASS 1) among the faces
// whose depth at the pixel EQUALS that minimum (recomputed: same arithmetic, same bits) a 32-bit atomic-min of the face
// index — the same deterministic "lowest face index wins an exact tie" rule as the float path.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kTile = 64;          // pixels per tile edge
constexpr int kTilePix = kTile * kTile;
constexpr int kSmallArea = 16;     // clipped boxes up to this many pixels are walked by their own lane

__device__ __forceinline__ unsigned long long ordered_bits64(double d) {
    unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double from_ordered_bits64(unsigned long long u) {
    return __longlong_as_double((long long)((u & 0x8000000000000000ull) ? (u & 0x7FFFFFFFFFFFFFFFull) : ~u));
}

template <typename T>
struct TileKeys;
template <>
struct TileKeys<float> {
    unsigned long long* key;  // LDS [kTilePix]
    __device__ __forceinline__ void hit(int p, float zp, uint32_t fidx, int) const {
        atomicMin(key + p, ((unsigned long long)ordered_bits(zp) << 32) | fidx);
    }
};
template <>
struct TileKeys<double> {
    unsigned long long* key;  // LDS [kTilePix] ordered depth bits
    uint32_t* fkey;           // LDS [kTilePix]
    __device__ __forceinline__ void hit(int p, double zp, uint32_t fidx, int pass) const {
        const unsigned long long k = ordered_bits64(zp);
        if (pass == 0) atomicMin(key + p, k);
        else if (key[p] == k) atomicMin(fkey + p, fidx);
    }
};

// one pixel (global x, y; tile origin tx0, ty0) of one face
template <typename T>
__device__ __forceinline__ void shade(const Face<T>& f, const BaryCtx<T>& c, int x, int y, int tx0, int ty0, uint32_t fidx,
                                      const TileKeys<T>& keys, int pass) {
    T w[3];
    bary_at(f, c, (T)x, (T)y, w);
    if (inside(w)) {
        T zp = persp_depth(f, w);
        if (zp == zp) keys.hit((y - ty0) * kTile + (x - tx0), zp, fidx, pass);  // NaN never wins (fminf in the reference's atomicMin, .cu:8-18)
    }
}

template <typename T>
__device__ __forceinline__ Face<T> shfl_face(const Face<T>& f, int src) {
    Face<T> g;
    g.x0 = __shfl(f.x0, src, 64); g.y0 = __shfl(f.y0, src, 64); g.z0 = __shfl(f.z0, src, 64);
    g.x1 = __shfl(f.x1, src, 64); g.y1 = __shfl(f.y1, src, 64); g.z1 = __shfl(f.z1, src, 64);
    g.x2 = __shfl(f.x2, src, 64); g.y2 = __shfl(f.y2, src, 64); g.z2 = __shfl(f.z2, src, 64);
    return g;
}

// Measured (profiles/r3_raster_knobs.txt, body.obj x 32): 512 threads with the listed faces dealt round-robin over the waves is
// the best or within 10 % of the best of {512, 1024 threads} x {contiguous, round-robin} on all four workloads of
// tools/raster_bench.py (256^2: 36 vs 33 us; 512^2: 160 vs 194-289 us; screen-filling faces: 57 vs 73-130 us).
constexpr int kTileThreads = 512;
template <typename T, bool COLORS, int kThreads>
__global__ void __launch_bounds__(kThreads)
raster_tiles(const T* __restrict__ fv, const T* __restrict__ fc, uint32_t* __restrict__ count, const uint32_t* __restrict__ list,
             T* __restrict__ depth, int32_t* __restrict__ tri, T* __restrict__ out3, int F, int H, int W, int tiles_x, int tiles_y) {
    constexpr bool F64 = sizeof(T) == 8;
    __shared__ unsigned long long key[kTilePix];
    __shared__ uint32_t fkey[F64 ? kTilePix : 1];
    __shared__ int n_sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x % (tiles_x * tiles_y), b = blockIdx.x / (tiles_x * tiles_y);
    const int tx0 = (tile % tiles_x) * kTile, ty0 = (tile / tiles_x) * kTile;
    const int tx1 = min(tx0 + kTile, W) - 1, ty1 = min(ty0 + kTile, H) - 1;
    const long img = (long)b * H * W;
    TileKeys<T> keys;
    keys.key = key;
    if constexpr (F64) keys.fkey = fkey;

    // the tile's list length; the counter is handed back ZERO, so a caller that keeps its workspace needs no memset per call
    if (tid == 0) {
        n_sh = (int)count[blockIdx.x];  // blockIdx.x == b * tiles + tile
        count[blockIdx.x] = 0;
    }
    // first round's face of this lane, fetched speculatively (list entry -> face data: two dependent memory latencies that
    // would otherwise start only after the seeding barrier).  Entries beyond the list length are stale or uninitialised:
    // the index is clamped to a valid face and the lane ignores the data if j >= n.
    constexpr int NW = kThreads / 64;
    const uint32_t* cand = list + (long)blockIdx.x * F;
    const T* fvb = fv + (long)b * F * 9;
    const int j_first = lane * NW + wave;
    const uint32_t pre_idx = j_first < F ? min(cand[j_first], (uint32_t)(F - 1)) : 0u;
    const Face<T> pre_face = load_face(fvb + (long)pre_idx * 9);
    // ---- seed the tile's keys from the caller's depth buffer (all loads of a thread in flight together)
    constexpr int PPT = kTilePix / kThreads;  // pixels per thread
    {
        T dv[PPT];
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int p = tid + i * kThreads;
            const int x = tx0 + (p & (kTile - 1)), y = ty0 + (p >> 6);
            dv[i] = (x <= tx1 && y <= ty1) ? depth[img + (long)y * W + x] : T(0);
        }
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int p = tid + i * kThreads;
            if constexpr (F64) {
                key[p] = ordered_bits64(dv[i]);
                fkey[p] = kNoFace;
            } else {
                key[p] = ((unsigned long long)ordered_bits(dv[i]) << 32) | kNoFace;
            }
        }
    }
    __syncthreads();

    const int n = n_sh;
    const int lx = lane & 7, ly = lane >> 3;
    for (int pass = 0; pass < (F64 ? 2 : 1); ++pass) {
        {
            // ---- shade: one listed face per lane, dealt round-robin over the WAVES (a short list of 100 faces keeps every wave
            // busy with a few lanes instead of two waves with all of them: a wave walks its larger faces one after the other)
            for (int j0 = 0; j0 < n; j0 += kThreads) {
                const int j = j0 + lane * NW + wave;
                uint32_t fidx = 0;
                Face<T> f{};
                int x_min = 0, x_max = -1, y_min = 0, y_max = -1, area = 0;
                if (j < n) {
                    if (j0 == 0) {
                        fidx = pre_idx;
                        f = pre_face;
                    } else {
                        fidx = cand[j];
                        f = load_face(fvb + (long)fidx * 9);
                    }
                    face_bbox(f, H, W, x_min, x_max, y_min, y_max);
                    x_min = max(x_min, tx0); x_max = min(x_max, tx1);
                    y_min = max(y_min, ty0); y_max = min(y_max, ty1);
                    area = (x_max >= x_min && y_max >= y_min) ? (x_max - x_min + 1) * (y_max - y_min + 1) : 0;
                }
                if (area > 0 && area <= kSmallArea) {
                    const BaryCtx<T> c = bary_setup(f);
                    for (int y = y_min; y <= y_max; ++y)
                        for (int x = x_min; x <= x_max; ++x) shade(f, c, x, y, tx0, ty0, fidx, keys, pass);
                }
                unsigned long long todo = __ballot(area > kSmallArea);
                while (todo) {
                    const int src = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const Face<T> g = shfl_face(f, src);
                    const int gx0 = __shfl(x_min, src, 64), gx1 = __shfl(x_max, src, 64);
                    const int gy0 = __shfl(y_min, src, 64), gy1 = __shfl(y_max, src, 64);
                    const uint32_t gf = (uint32_t)__shfl((int)fidx, src, 64);
                    const BaryCtx<T> c = bary_setup(g);
                    for (int y0 = gy0; y0 <= gy1; y0 += 8)
                        for (int x0 = gx0; x0 <= gx1; x0 += 8) {
                            const int x = x0 + lx, y = y0 + ly;
                            if (x <= gx1 && y <= gy1) shade(g, c, x, y, tx0, ty0, gf, keys, pass);
                        }
                }
            }
            __syncthreads();  // float64: every depth key is final before the face keys are taken
        }
    }

    // ---- resolve: the winners' face data of a thread's pixels is fetched in one batch (one memory latency, not PPT of them)
    uint32_t fid[PPT];
    unsigned long long kk[PPT];
    Face<T> ff[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int p = tid + i * kThreads;
        const int x = tx0 + (p & (kTile - 1)), y = ty0 + (p >> 6);
        kk[i] = key[p];
        if constexpr (F64) fid[i] = fkey[p];
        else fid[i] = (uint32_t)(kk[i] & 0xFFFFFFFFu);
        if (x > tx1 || y > ty1) fid[i] = kNoFace;  // pixel keeps the caller's depth / tri / payload
    }
#pragma unroll
    for (int i = 0; i < PPT; ++i)
        if (fid[i] != kNoFace) ff[i] = load_face(fvb + (long)fid[i] * 9);
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        if (fid[i] == kNoFace) continue;
        const int p = tid + i * kThreads;
        const int x = tx0 + (p & (kTile - 1)), y = ty0 + (p >> 6);
        const long gp = img + (long)y * W + x;
        const BaryCtx<T> c = bary_setup(ff[i]);
        T w[3];
        bary_at(ff[i], c, (T)x, (T)y, w);
        if constexpr (F64) depth[gp] = from_ordered_bits64(kk[i]);
        else depth[gp] = from_ordered_bits((uint32_t)(kk[i] >> 32));
        tri[gp] = (int32_t)fid[i];
        if (COLORS) {
            const T* cl = fc + ((long)b * F + fid[i]) * 9;  // [3 verts][3 channels], .cu:189-194
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) out3[gp * 3 + ch] = w[0] * cl[ch] + w[1] * cl[3 + ch] + w[2] * cl[6 + ch];
        } else {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) out3[gp * 3 + ch] = w[ch];
        }
    }
}

inline long pad2(long n) { return (n + 1) / 2 * 2; }  // keeps the list 8-byte aligned behind the counters

// Workspaces whose tile counters the caller guarantees to be zero on entry (raster_tiles hands them back zero): no memset node
// per call for THESE pointers only (ABI 3: a property of the workspace, not of the process — another caller of the C API that
// passes an unzeroed workspace, as the default contract allows, is not affected).  A failed call drops the registration.
std::mutex g_clean_mu;
std::unordered_set<const void*> g_clean_ws;

bool workspace_is_clean(const void* ws) {
    std::lock_guard<std::mutex> lk(g_clean_mu);
    return g_clean_ws.count(ws) != 0;
}
void workspace_forget(const void* ws) {
    std::lock_guard<std::mutex> lk(g_clean_mu);
    g_clean_ws.erase(ws);
}

template <typename T>
int run(const T* fv, const T* fc, T* depth, int32_t* tri, T* out3, int B, int F, int H, int W, void* workspace,
        gif_stream_t stream, const char* who) {
    GIF_REQUIRE(B >= 0 && F >= 0 && H > 0 && W > 0, "%s: bad dims B=%d F=%d H=%d W=%d", who, B, F, H, W);
    if ((long)B * H * W == 0 || F == 0) return 0;
    GIF_REQUIRE(fv && depth && tri && out3 && workspace, "%s: null pointer", who);
    GIF_REQUIRE(((uintptr_t)workspace & 7) == 0, "%s: workspace must be 8-byte aligned", who);
    const int tiles_x = (W + kTile - 1) / kTile, tiles_y = (H + kTile - 1) / kTile;
    const long nt = (long)tiles_x * tiles_y;
    GIF_REQUIRE(B <= 65535 && B * nt < (1L << 31) && B * nt * F < (1L << 40), "%s: too many images / tiles / faces", who);
    hipStream_t s = gif::as_stream(stream);
    uint32_t* count = reinterpret_cast<uint32_t*>(workspace);
    uint32_t* list = count + pad2(B * nt);
    if (!workspace_is_clean(workspace)) {
        hipError_t me = hipMemsetAsync(count, 0, (size_t)B * nt * sizeof(uint32_t), s);
        if (me != hipSuccess) { gif::set_error("%s memset: %s", who, hipGetErrorString(me)); return (int)me; }
    }
    raster_bin<T><<<dim3((unsigned)gif::cdiv(F, kBinThreads), (unsigned)B), kBinThreads, 0, s>>>(fv, count, list, F, H, W, tiles_x, tiles_y);
    const dim3 grid((unsigned)(B * nt));
    if (fc) raster_tiles<T, true, kTileThreads><<<grid, kTileThreads, 0, s>>>(fv, fc, count, list, depth, tri, out3, F, H, W, tiles_x, tiles_y);
    else raster_tiles<T, false, kTileThreads><<<grid, kTileThreads, 0, s>>>(fv, nullptr, count, list, depth, tri, out3, F, H, W, tiles_x, tiles_y);
    const int rc = gif::check_launch(who);
    if (rc != 0) workspace_forget(workspace);  // the counters may be dirty: the next call through this pointer memsets again
    return rc;
}

}  // namespace

extern "C" {

int gif_rasterize_assume_clean_workspace(const void* workspace, int on) {
    GIF_REQUIRE(workspace, "rasterize_assume_clean_workspace: null workspace");
    std::lock_guard<std::mutex> lk(g_clean_mu);
    if (on) g_clean_ws.insert(workspace);
    else g_clean_ws.erase(workspace);
    return 0;
}

// per (image, 64x64 tile): one counter + a face list that can hold every face; the z-buffer itself never leaves LDS
int64_t gif_rasterize_workspace_bytes(int B, int F, int H, int W) {
    if (B <= 0 || F <= 0 || H <= 0 || W <= 0) return 8;
    const int64_t nt = (int64_t)((W + kTile - 1) / kTile) * ((H + kTile - 1) / kTile);
    return (pad2(B * nt) + B * nt * F) * 4;
}

int gif_rasterize_f32(const float* face_vertices, float* depth, int32_t* tri, float* bary, int B, int F, int H,
                      int W, void* workspace, gif_stream_t stream) {
    return run<float>(face_vertices, nullptr, depth, tri, bary, B, F, H, W, workspace, stream, "rasterize");
}

int gif_rasterize_colors_f32(const float* face_vertices, const float* face_colors, float* depth, int32_t* tri,
                             float* images, int B, int F, int H, int W, void* workspace, gif_stream_t stream) {
    GIF_REQUIRE(face_colors || (long)B * F == 0, "rasterize_colors: null face_colors");
    return run<float>(face_vertices, face_colors, depth, tri, images, B, F, H, W, workspace, stream, "rasterize_colors");
}

int gif_rasterize_f64(const double* face_vertices, double* depth, int32_t* tri, double* bary, int B, int F, int H, int W,
                      void* workspace, gif_stream_t stream) {
    return run<double>(face_vertices, nullptr, depth, tri, bary, B, F, H, W, workspace, stream, "rasterize_f64");
}

int gif_rasterize_colors_f64(const double* face_vertices, const double* face_colors, double* depth, int32_t* tri,
                             double* images, int B, int F, int H, int W, void* workspace, gif_stream_t stream) {
    GIF_REQUIRE(face_colors || (long)B * F == 0, "rasterize_colors_f64: null face_colors");
    return run<double>(face_vertices, face_colors, depth, tri, images, B, F, H, W, workspace, stream, "rasterize_colors_f64");
}
}
