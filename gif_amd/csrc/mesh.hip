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
