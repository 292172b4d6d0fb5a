/* ORACLE — TEST INFRASTRUCTURE ONLY.  Body of oracle/rasterize_ref.c, instantiated once per floating type
 * (REAL = float: the pinned restatement; REAL = double: the same algorithm in double precision — the reference dispatches
 * AT_DISPATCH_FLOATING_TYPES, standard_rasterize_cuda_kernel.cu:252,295).  Macros: REAL, NAME(x), CEIL/FLOOR/FMIN/FMAX. */
/* barycentric_weight(), .cu:78-109 */
static inline void NAME(bary_w)(REAL *w, REAL px, REAL py, REAL x0, REAL y0, REAL x1, REAL y1,
                          REAL x2, REAL y2) {
    REAL v0x = x2 - x0, v0y = y2 - y0;
    REAL v1x = x1 - x0, v1y = y1 - y0;
    REAL v2x = px - x0, v2y = py - y0;
    REAL dot00 = v0x * v0x + v0y * v0y;
    REAL dot01 = v0x * v1x + v0y * v1y;
    REAL dot02 = v0x * v2x + v0y * v2y;
    REAL dot11 = v1x * v1x + v1y * v1y;
    REAL dot12 = v1x * v2x + v1y * v2y;
    REAL den = dot00 * dot11 - dot01 * dot01;
    REAL inv = (den == (REAL)0) ? (REAL)0 : (REAL)1 / den;
    REAL u = (dot11 * dot02 - dot01 * dot12) * inv;
    REAL v = (dot00 * dot12 - dot01 * dot02) * inv;
    w[0] = (REAL)1 - u - v;
    w[1] = v;
    w[2] = u;
}

/* One face over its bbox.  colors==NULL: write barycentrics (forward_rasterize_cuda_kernel :111-167);
 * else write interpolated attributes (forward_rasterize_colors_cuda_kernel :170-233). */
static void NAME(raster_face)(const REAL *face, const REAL *color, REAL *depth, int32_t *tri,
                        REAL *out3, int32_t *owner, int fidx, int h, int w) {
    REAL x0 = face[0], y0 = face[1], z0 = face[2];
    REAL x1 = face[3], y1 = face[4], z1 = face[5];
    REAL x2 = face[6], y2 = face[7], z2 = face[8];
    /* check_face_frontside(), .cu:31-34 */
    int front = (y2 - y0) * (x1 - x0) < (y1 - y0) * (x2 - x0);
    if (!front) return;
    int x_min = imax((int)CEIL(FMIN(x0, FMIN(x1, x2))), 0);
    int x_max = imin((int)FLOOR(FMAX(x0, FMAX(x1, x2))), w - 1);
    int y_min = imax((int)CEIL(FMIN(y0, FMIN(y1, y2))), 0);
    int y_max = imin((int)FLOOR(FMAX(y0, FMAX(y1, y2))), h - 1);
    for (int y = y_min; y <= y_max; ++y) {
        for (int x = x_min; x <= x_max; ++x) {
            REAL bw[3];
            NAME(bary_w)(bw, (REAL)x, (REAL)y, x0, y0, x1, y1, x2, y2);
            if (bw[2] >= 0 && bw[1] >= 0 && bw[0] > 0) {
                REAL zp = (REAL)1 / (bw[0] / z0 + bw[1] / z1 + bw[2] / z2);
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
void NAME(oracle_rasterize)(const REAL *face_vertices, REAL *depth, int32_t *tri, REAL *bary, int B,
                      int F, int H, int W) {
    int32_t *owner = (int32_t *)malloc(sizeof(int32_t) * (size_t)B * H * W);
    memset(owner, 0xff, sizeof(int32_t) * (size_t)B * H * W);
    for (int b = 0; b < B; ++b)
        for (int f = 0; f < F; ++f)
            NAME(raster_face)(face_vertices + ((long)b * F + f) * 9, 0, depth + (long)b * H * W,
                        tri + (long)b * H * W, bary + (long)b * H * W * 3,
                        owner + (long)b * H * W, f, H, W);
    free(owner);
}

/* standard_rasterize_colors (standard_rasterize_cuda.cpp:59-75) */
void NAME(oracle_rasterize_colors)(const REAL *face_vertices, const REAL *face_colors, REAL *depth,
                             int32_t *tri, REAL *images, int B, int F, int H, int W) {
    int32_t *owner = (int32_t *)malloc(sizeof(int32_t) * (size_t)B * H * W);
    memset(owner, 0xff, sizeof(int32_t) * (size_t)B * H * W);
    for (int b = 0; b < B; ++b)
        for (int f = 0; f < F; ++f)
            NAME(raster_face)(face_vertices + ((long)b * F + f) * 9, face_colors + ((long)b * F + f) * 9,
                        depth + (long)b * H * W, tri + (long)b * H * W,
                        images + (long)b * H * W * 3, owner + (long)b * H * W, f, H, W);
    free(owner);
}
