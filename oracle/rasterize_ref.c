/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not shipped, not measured as product.
 *
 * Scalar CPU restatement of GIF's bundled z-buffer triangle rasteriser
 *   /root/reference/my_utils/standard_rasterize_cuda/standard_rasterize_cuda_kernel.cu
 * so that the HIP kernel (gif_amd/csrc/rasterize.hip) can be checked bit-for-bit.
 * Parity status: PINNED — reproduces both golden OBJ fixtures of the reference
 * (data/obj/body_vis.obj, body_vis_z.obj; see tests/test_oracle_rasterize.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).  FP contraction
 * is disabled so every product/sum is rounded exactly once, like the HIP build.
 *
 * Semantics restated (reference line numbers refer to the .cu file above):
 *   front-face test     :31-34   (y2-y0)(x1-x0) < (y1-y0)(x2-x0)
 *   barycentric weights :78-109  dot-product form, inverDeno=0 when degenerate, w=(1-u-v, v, u)
 *   bbox                :133-136 ceil(min)/floor(max) clamped to [0,w-1]x[0,h-1], pixel centre = integer
 *   inside test + depth :144-148 bw2>=0 && bw1>=0 && bw0>0 ; zp = 1/(bw0/z0+bw1/z1+bw2/z2)
 *   winner write        :149-160 min depth wins; the winner's face index and bw (or colours) are stored
 * The reference resolves the race between triangles with an atomicMin and a second identical
 * launch (:252-269); sequentially that is "strictly smaller depth wins, first face wins a tie",
 * which is one of the outcomes the reference can produce and the one the HIP kernel fixes
 * deterministically (lowest face index among exact-depth ties).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* barycentric_weight(), .cu:78-109 */
static inline void bary_w(float *w, float px, float py, float x0, float y0, float x1, float y1,
                          float x2, float y2) {
    float v0x = x2 - x0, v0y = y2 - y0;
    float v1x = x1 - x0, v1y = y1 - y0;
    float v2x = px - x0, v2y = py - y0;
    float dot00 = v0x * v0x + v0y * v0y;
    float dot01 = v0x * v1x + v0y * v1y;
    float dot02 = v0x * v2x + v0y * v2y;
    float dot11 = v1x * v1x + v1y * v1y;
    float dot12 = v1x * v2x + v1y * v2y;
    float den = dot00 * dot11 - dot01 * dot01;
    float inv = (den == 0.0f) ? 0.0f : 1.0f / den;
    float u = (dot11 * dot02 - dot01 * dot12) * inv;
    float v = (dot00 * dot12 - dot01 * dot02) * inv;
    w[0] = 1.0f - u - v;
    w[1] = v;
    w[2] = u;
}

/* One face over its bbox.  colors==NULL: write barycentrics (forward_rasterize_cuda_kernel :111-167);
 * else write interpolated attributes (forward_rasterize_colors_cuda_kernel :170-233). */
static void raster_face(const float *face, const float *color, float *depth, int32_t *tri,
                        float *out3, int32_t *owner, int fidx, int h, int w) {
    float x0 = face[0], y0 = face[1], z0 = face[2];
    float x1 = face[3], y1 = face[4], z1 = face[5];
    float x2 = face[6], y2 = face[7], z2 = face[8];
    /* check_face_frontside(), .cu:31-34 */
    int front = (y2 - y0) * (x1 - x0) < (y1 - y0) * (x2 - x0);
    if (!front) return;
    int x_min = imax((int)ceilf(fminf(x0, fminf(x1, x2))), 0);
    int x_max = imin((int)floorf(fmaxf(x0, fmaxf(x1, x2))), w - 1);
    int y_min = imax((int)ceilf(fminf(y0, fminf(y1, y2))), 0);
    int y_max = imin((int)floorf(fmaxf(y0, fmaxf(y1, y2))), h - 1);
    for (int y = y_min; y <= y_max; ++y) {
        for (int x = x_min; x <= x_max; ++x) {
            float bw[3];
            bary_w(bw, (float)x, (float)y, x0, y0, x1, y1, x2, y2);
            if (bw[2] >= 0 && bw[1] >= 0 && bw[0] > 0) {
                float zp = 1.0f / (bw[0] / z0 + bw[1] / z1 + bw[2] / z2);
                int pix = y * w + x;
                /* atomicMin + "== zp" re-check of the reference, sequentially.  owner<0 means the
                 * pixel still holds the caller's initial depth: an exact tie with it is a win
                 * (reference: depth_buffer == zp after the atomicMin).  NaN zp never wins. */
                if (zp < depth[pix] || (zp == depth[pix] && owner[pix] < 0)) {
                    depth[pix] = zp;
                    owner[pix] = fidx;
                    tri[pix] = fidx;
                    if (color) {
                        for (int k = 0; k < 3; ++k)
                            out3[pix * 3 + k] =
                                bw[0] * color[0 + k] + bw[1] * color[3 + k] + bw[2] * color[6 + k];
                    } else {
                        for (int k = 0; k < 3; ++k) out3[pix * 3 + k] = bw[k];
                    }
                }
            }
        }
    }
}

/* face_vertices [B,F,3,3]; depth [B,H,W]; tri [B,H,W]; bary [B,H,W,3] — all caller-initialised,
 * updated in place (standard_rasterize_cuda.cpp:26-40). */
void oracle_rasterize(const float *face_vertices, float *depth, int32_t *tri, float *bary, int B,
                      int F, int H, int W) {
    int32_t *owner = (int32_t *)malloc(sizeof(int32_t) * (size_t)B * H * W);
    memset(owner, 0xff, sizeof(int32_t) * (size_t)B * H * W);
    for (int b = 0; b < B; ++b)
        for (int f = 0; f < F; ++f)
            raster_face(face_vertices + ((long)b * F + f) * 9, 0, depth + (long)b * H * W,
                        tri + (long)b * H * W, bary + (long)b * H * W * 3,
                        owner + (long)b * H * W, f, H, W);
    free(owner);
}

/* standard_rasterize_colors (standard_rasterize_cuda.cpp:59-75) */
void oracle_rasterize_colors(const float *face_vertices, const float *face_colors, float *depth,
                             int32_t *tri, float *images, int B, int F, int H, int W) {
    int32_t *owner = (int32_t *)malloc(sizeof(int32_t) * (size_t)B * H * W);
    memset(owner, 0xff, sizeof(int32_t) * (size_t)B * H * W);
    for (int b = 0; b < B; ++b)
        for (int f = 0; f < F; ++f)
            raster_face(face_vertices + ((long)b * F + f) * 9, face_colors + ((long)b * F + f) * 9,
                        depth + (long)b * H * W, tri + (long)b * H * W,
                        images + (long)b * H * W * 3, owner + (long)b * H * W, f, H, W);
    free(owner);
}
