// Per-vertex normals of a triangle mesh for gfx950 — first kernel of the condition-render pipeline around the
// rasteriser.  Replaces model/mesh_and_3d_helpers.py:5-37 (vertex_normals): the reference
// scatters three cross products per face with index_add_ (atomics on a GPU: summation order is a race).  Here the
// fixed topology is turned once into a vertex -> (face, corner) CSR list ordered exactly like the reference's three
// sequential index_add_ passes (corner 1, corner 2, corner 0; faces ascending), and every vertex GATHERS its
// contributions in that order: deterministic, atomic-free, and the same summation order as the CPU reference.
#include "common.h"

#pragma clang fp contract(off)

namespace {

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {  // torch.cross component order
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// entry = face * 4 + corner ; off[v] .. off[v+1] are the entries of vertex v
__global__ void __launch_bounds__(256) vertex_normals_kernel(const float* __restrict__ verts,
                                                             const int32_t* __restrict__ faces,
                                                             const int32_t* __restrict__ off,
                                                             const int32_t* __restrict__ ent, float* __restrict__ out,
                                                             int B, int V) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * V) return;
    int b = (int)(i / V), v = (int)(i - (long)b * V);
    const float* vb = verts + (size_t)b * V * 3;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    for (int e = off[v]; e < off[v + 1]; ++e) {
        int f = ent[e] >> 2, c = ent[e] & 3;
        V3 p0 = ld3(vb + 3 * faces[3 * f + 0]), p1 = ld3(vb + 3 * faces[3 * f + 1]), p2 = ld3(vb + 3 * faces[3 * f + 2]);
        V3 n;
        if (c == 1) n = cross(sub(p2, p1), sub(p0, p1));       // mesh_and_3d_helpers.py:27-28
        else if (c == 2) n = cross(sub(p0, p2), sub(p1, p2));  // :29-30
        else n = cross(sub(p1, p0), sub(p2, p0));              // :31-32
        nx += n.x; ny += n.y; nz += n.z;
    }
    // F.normalize(normals, eps=1e-6, dim=1): x / max(||x||_2, eps)   (:34)
    float len = sqrtf(nx * nx + ny * ny + nz * nz);
    float d = fmaxf(len, 1e-6f);
    out[i * 3 + 0] = nx / d;
    out[i * 3 + 1] = ny / d;
    out[i * 3 + 2] = nz / d;
}

}  // namespace

extern "C" int gif_vertex_normals_f32(const float* verts, const int32_t* faces, const int32_t* csr_off,
                                      const int32_t* csr_ent, float* normals, int B, int V, int F, gif_stream_t stream) {
    GIF_REQUIRE(B >= 0 && V > 0 && F >= 0, "vertex_normals: bad dims");
    if (B == 0) return 0;
    GIF_REQUIRE(verts && faces && csr_off && csr_ent && normals, "vertex_normals: null pointer");
    vertex_normals_kernel<<<gif::cdiv((long)B * V, 256), 256, 0, gif::as_stream(stream)>>>(verts, faces, csr_off, csr_ent,
                                                                                            normals, B, V);
    return gif::check_launch("vertex_normals");
}

// ---------------------------------------------------------------------------------------------------------------
// "Texture stealing" (SURVEY §8(f) row 2): FlameTextureSpace.compute_texture_map, model/stg2_generator.py:378-421.
// The reference builds a [B,T,T,2] sampling grid (zeros except at the valid UV texels, :402-404) and calls
// F.grid_sample (bilinear, zero padding, align_corners=False — torch 1.7 default) plus a second barycentric gather for
// the normal-z visibility mask (:409-417).  Here one kernel does, per (sample, texel): barycentric 3-D point of the
// texel's face -> orthographic projection (batch_orth_proj, y flipped :399-400) -> bilinear fetch of the source image ->
// texture texel, and the mask from the interpolated normal.  Texels that are not valid sample grid position (0,0), i.e.
// the image centre, exactly like the reference's zero-initialised grid.
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct TexParams {
    const float* img;      // [B,C,H,W]  (NCHW, the generator's output layout)
    const float* verts;    // [B,V,3]
    const float* normals;  // [B,V,3]
    const float* cam;      // [B,3]  (scale, tx, ty)
    const int32_t* map;    // [T*T] index into the valid-texel list or -1
    const int32_t* faces;  // [N,3] vertex ids of the texel's face
    const float* bc;       // [N,3] barycentric coordinates
    float* tex;            // [B,C,T,T]
    uint8_t* mask;         // [B,1,T,T]
    float* gimg;           // backward: [B,C,H,W] (zero-initialised)
    const float* gtex;     // backward: [B,C,T,T]
    int B, C, H, W, V, T;
};

struct Bilin {
    int x0, y0;
    float w00, w01, w10, w11;  // weights of (y0,x0) (y0,x1) (y1,x0) (y1,x1), already zeroed when out of the image
};

__device__ __forceinline__ Bilin bilinear_setup(float gx, float gy, int H, int W) {
    // grid_sample, align_corners=False: pixel = ((g + 1) * size - 1) / 2
    float ix = ((gx + 1.f) * W - 1.f) * 0.5f, iy = ((gy + 1.f) * H - 1.f) * 0.5f;
    float fx = floorf(ix), fy = floorf(iy);
    Bilin b;
    b.x0 = (int)fx; b.y0 = (int)fy;
    float ax = ix - fx, ay = iy - fy;
    bool x0ok = b.x0 >= 0 && b.x0 < W, x1ok = b.x0 + 1 >= 0 && b.x0 + 1 < W;
    bool y0ok = b.y0 >= 0 && b.y0 < H, y1ok = b.y0 + 1 >= 0 && b.y0 + 1 < H;
    b.w00 = (x0ok && y0ok) ? (1.f - ax) * (1.f - ay) : 0.f;
    b.w01 = (x1ok && y0ok) ? ax * (1.f - ay) : 0.f;
    b.w10 = (x0ok && y1ok) ? (1.f - ax) * ay : 0.f;
    b.w11 = (x1ok && y1ok) ? ax * ay : 0.f;
    return b;
}

__device__ __forceinline__ void texel_grid(const TexParams& p, int b, int n, float* gx, float* gy, float* nz) {
    if (n < 0) { *gx = 0.f; *gy = 0.f; *nz = 0.f; return; }
    const float* vb = p.verts + (size_t)b * p.V * 3;
    const float* nb = p.normals + (size_t)b * p.V * 3;
    int f0 = p.faces[3 * n], f1 = p.faces[3 * n + 1], f2 = p.faces[3 * n + 2];
    float b0 = p.bc[3 * n], b1 = p.bc[3 * n + 1], b2 = p.bc[3 * n + 2];
    float px = vb[3 * f0] * b0 + vb[3 * f1] * b1 + vb[3 * f2] * b2;          // :386-389
    float py = vb[3 * f0 + 1] * b0 + vb[3 * f1 + 1] * b1 + vb[3 * f2 + 1] * b2;
    float s = p.cam[3 * b], tx = p.cam[3 * b + 1], ty = p.cam[3 * b + 2];
    *gx = s * (px + tx);                                                       // batch_orth_proj :399
    *gy = -(s * (py + ty));                                                    // :400
    *nz = nb[3 * f0 + 2] * b0 + nb[3 * f1 + 2] * b1 + nb[3 * f2 + 2] * b2;    // :409-412
}

__global__ void __launch_bounds__(256) texture_map_kernel(const TexParams p) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int TT = p.T * p.T;
    if (i >= (long)p.B * TT) return;
    int b = (int)(i / TT), t = (int)(i - (long)b * TT);
    int n = p.map[t];
    float gx, gy, nz;
    texel_grid(p, b, n, &gx, &gy, &nz);
    Bilin bl = bilinear_setup(gx, gy, p.H, p.W);
    for (int c = 0; c < p.C; ++c) {
        const float* im = p.img + ((size_t)b * p.C + c) * p.H * p.W;
        float v = 0.f;
        if (bl.w00 != 0.f) v += bl.w00 * im[(size_t)bl.y0 * p.W + bl.x0];
        if (bl.w01 != 0.f) v += bl.w01 * im[(size_t)bl.y0 * p.W + bl.x0 + 1];
        if (bl.w10 != 0.f) v += bl.w10 * im[(size_t)(bl.y0 + 1) * p.W + bl.x0];
        if (bl.w11 != 0.f) v += bl.w11 * im[(size_t)(bl.y0 + 1) * p.W + bl.x0 + 1];
        p.tex[((size_t)b * p.C + c) * TT + t] = v;
    }
    p.mask[(size_t)b * TT + t] = (n >= 0 && nz < 0.f) ? 1 : 0;                 // :413-417
}

// d(texture)/d(image): scatter the 4 bilinear weights.  Atomics only touch the image gradient; the (many) invalid texels
// all hit the image centre, so their contributions are pre-summed per workgroup before the 4 atomics.
__global__ void __launch_bounds__(256) texture_map_bwd_kernel(const TexParams p) {
    __shared__ float red[4];
    const int TT = p.T * p.T;
    const int b = blockIdx.y, c = blockIdx.z;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    float g = 0.f;
    int n = -1;
    if (t < TT) {
        g = p.gtex[((size_t)b * p.C + c) * TT + t];
        n = p.map[t];
    }
    float* gim = p.gimg + ((size_t)b * p.C + c) * p.H * p.W;
    float inval = 0.f;
    if (t < TT && n >= 0) {
        float gx, gy, nz;
        texel_grid(p, b, n, &gx, &gy, &nz);
        Bilin bl = bilinear_setup(gx, gy, p.H, p.W);
        if (bl.w00 != 0.f) atomicAdd(gim + (size_t)bl.y0 * p.W + bl.x0, bl.w00 * g);
        if (bl.w01 != 0.f) atomicAdd(gim + (size_t)bl.y0 * p.W + bl.x0 + 1, bl.w01 * g);
        if (bl.w10 != 0.f) atomicAdd(gim + (size_t)(bl.y0 + 1) * p.W + bl.x0, bl.w10 * g);
        if (bl.w11 != 0.f) atomicAdd(gim + (size_t)(bl.y0 + 1) * p.W + bl.x0 + 1, bl.w11 * g);
    } else if (t < TT) {
        inval = g;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) inval += __shfl_xor(inval, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = inval;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = red[0] + red[1] + red[2] + red[3];
        if (s != 0.f) {
            Bilin bl = bilinear_setup(0.f, 0.f, p.H, p.W);
            if (bl.w00 != 0.f) atomicAdd(gim + (size_t)bl.y0 * p.W + bl.x0, bl.w00 * s);
            if (bl.w01 != 0.f) atomicAdd(gim + (size_t)bl.y0 * p.W + bl.x0 + 1, bl.w01 * s);
            if (bl.w10 != 0.f) atomicAdd(gim + (size_t)(bl.y0 + 1) * p.W + bl.x0, bl.w10 * s);
            if (bl.w11 != 0.f) atomicAdd(gim + (size_t)(bl.y0 + 1) * p.W + bl.x0 + 1, bl.w11 * s);
        }
    }
}

}  // namespace

extern "C" int gif_texture_map_f32(const float* img, const float* verts, const float* normals, const float* cam,
                                   const int32_t* texel_map, const int32_t* texel_faces, const float* texel_bc, float* tex,
                                   uint8_t* mask, int B, int C, int H, int W, int V, int T, gif_stream_t stream) {
    GIF_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0 && V > 0 && T > 0, "texture_map: bad dims");
    if (B == 0) return 0;
    GIF_REQUIRE(img && verts && normals && cam && texel_map && texel_faces && texel_bc && tex && mask, "texture_map: null pointer");
    TexParams p{};
    p.img = img; p.verts = verts; p.normals = normals; p.cam = cam; p.map = texel_map; p.faces = texel_faces; p.bc = texel_bc;
    p.tex = tex; p.mask = mask; p.B = B; p.C = C; p.H = H; p.W = W; p.V = V; p.T = T;
    texture_map_kernel<<<gif::cdiv((long)B * T * T, 256), 256, 0, gif::as_stream(stream)>>>(p);
    return gif::check_launch("texture_map");
}

extern "C" int gif_texture_map_bwd_f32(const float* gtex, const float* verts, const float* normals, const float* cam,
                                       const int32_t* texel_map, const int32_t* texel_faces, const float* texel_bc,
                                       float* gimg, int B, int C, int H, int W, int V, int T, gif_stream_t stream) {
    GIF_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0 && V > 0 && T > 0, "texture_map_bwd: bad dims");
    if (B == 0) return 0;
    GIF_REQUIRE(gtex && verts && normals && cam && texel_map && texel_faces && texel_bc && gimg, "texture_map_bwd: null pointer");
    hipStream_t s = gif::as_stream(stream);
    hipError_t me = hipMemsetAsync(gimg, 0, (size_t)B * C * H * W * sizeof(float), s);
    if (me != hipSuccess) { gif::set_error("texture_map_bwd memset: %s", hipGetErrorString(me)); return (int)me; }
    TexParams p{};
    p.gtex = gtex; p.verts = verts; p.normals = normals; p.cam = cam; p.map = texel_map; p.faces = texel_faces; p.bc = texel_bc;
    p.gimg = gimg; p.B = B; p.C = C; p.H = H; p.W = W; p.V = V; p.T = T;
    texture_map_bwd_kernel<<<dim3(gif::cdiv((long)T * T, 256), B, C), 256, 0, s>>>(p);
    return gif::check_launch("texture_map_bwd");
}
