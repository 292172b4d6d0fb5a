/*
 * gif_hip.h — C ABI of libgif_hip.so: the MI355X (gfx950) kernels behind GIF's StyleGAN2
 * generator/discriminator hot path and its FLAME mesh rasteriser.
 *
 * The reference (ParthaEth/GIF) has exactly one native boundary — the pybind module
 * `standard_rasterize_cuda` (my_utils/standard_rasterize_cuda/standard_rasterize_cuda.cpp:79-82) —
 * and otherwise composes its layers from ATen calls (F.conv2d / F.conv_transpose2d / F.pad ...) inside
 * model/stylegan2_common_layers.py.  Every entry point below names the reference construct it
 * replaces.  All pointers are DEVICE pointers (fp32 / int32), all tensors are dense; activations are
 * NHWC ("channels last"), channel counts are multiples of 4 (callers zero-pad 3/6/9/513-channel
 * tensors).  `stream` is a hipStream_t passed as void*.  Every function returns 0 on success,
 * a positive hipError_t on a runtime failure, or a negative GIF_E* code on a bad argument;
 * gif_last_error() returns a static description of the last failure on the calling thread.
 * Nothing here synchronises the device or allocates memory: scratch is passed in by the caller.
 */
#ifndef GIF_HIP_H
#define GIF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GIF_EINVAL (-1)  /* bad argument (shape / alignment / null pointer) */
#define GIF_ENOSUP (-2)  /* configuration not supported by this build */

typedef void* gif_stream_t;

const char* gif_last_error(void);
int gif_abi_version(void);

/* f16-activation path (BASELINE configs[4]): every f16 activation store saturates at +-65504, which would hide an overflowing
 * activation GRADIENT from a dynamic loss scaler (the fp32 weight gradients computed from clamped values stay finite).  The
 * stores that can carry gradients (convolution / FIR epilogues, the leaky-ReLU backward, the modulation-gradient pass) raise a
 * per-device flag word when they clamp or see a non-finite value: clear it before backward(), OR it into the scaler's found_inf
 * scalar afterwards (both on the stream, no host synchronisation).  Only launches issued BETWEEN the two calls check their
 * stores (process-wide window): forward passes pay nothing. */
int gif_f16_overflow_clear(gif_stream_t stream);
int gif_f16_overflow_or_into(float* found_inf, gif_stream_t stream);
/* ABI 3: open (1) / close (0) the window without touching the flag word, so that a host can keep SEVERAL gradient passes of one
 * optimiser step inside it (the inner autograd.grad of R1 / the path-length regulariser, then backward()) and leave the forward
 * passes in between outside: clear once per step, watch(1) .. watch(0) around every gradient pass, or_into right after the
 * last one (loss_functions/losses.py:87-124 are the reference's regularisers). */
int gif_f16_overflow_watch(int on);

/* How the fp32 convolution contractions reach the matrix cores (process-wide; fp32 tensors in, fp32 tensors out either way):
 *   NATIVE : v_mfma_f32_32x32x2_f32 on the fp32 operands (157 TFLOP/s peak).
 *   BF16X3 : every fp32 operand element a is split into three bf16 terms a = hi + mid + lo (round-to-nearest at each level,
 *            8 + 8 + 8 mantissa bits, fp32's exponent range; the sum reproduces a to <= 2^-24 |a|), and a*b is accumulated in
 *            fp32 from the six products hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi on v_mfma_f32_32x32x16_bf16 (2.5 PFLOP/s
 *            peak => 417 TFLOP/s of fp32-equivalent work).  Each bf16*bf16 product is exact in fp32; the three dropped cross
 *            terms are <= 2^-23 |a*b| (typically 2^-25: below one fp32 product rounding).  Measured error against an fp64
 *            convolution: equal to or below the native fp32 MFMA path (tools/probes/x3_probe.py).
 *   F16X2  : (ABI 4) every operand element is scaled by a power of two 2^e that belongs to its ROW of the GEMM (output pixel /
 *            output channel: it factors out of the dot product) and split into two f16 terms x*2^e = hi + lo (11 + 11 significand
 *            bits); a*b is accumulated in fp32 from the THREE products hi*hi, hi*lo, lo*hi on v_mfma_f32_32x32x16_f16 (2.5 PFLOP/s
 *            peak => 833 TFLOP/s of fp32-equivalent work) and the result is multiplied by 2^-(e_a + e_b).  The dropped lo*lo is
 *            <= 2^-22 |a*b|.  Weight rows get their exponent from the packing kernel; activation rows carry a RUNNING exponent
 *            inside the kernel (a new row maximum is scaled into [2^13, 2^14); when a row outgrows its exponent its fp32
 *            accumulators are multiplied by the exact power of two).  Precision contract: elements carry 22 bits, or an
 *            absolute floor of 2^-38 of their row maximum; the floors cost at most 2^(m - 38) of a dot product's largest
 *            16-element group product, m = min over the K groups of the two operands' summed spreads (log2 row maximum / group
 *            maximum).  One operand with every non-zero group inside its window (2^14 activations / 2^16 weights) bounds m
 *            by that window (floor <= 2^-22; native fp32 MFMA: 2^-24 of the same quantity).  A launch in which BOTH operands
 *            have a group outside their windows raises a device-side gate and the op is recomputed by the BF16X3 kernels in
 *            the same stream ("guarded fallback": the bf16x3 launch that follows every f16x2 launch returns at once unless the
 *            gate is raised; no host synchronisation; gif_h2_fallback_stats counts the OPS that fell back — a transposed
 *            convolution's phases and a bulk + remainder launch pair count once).  Eager launches take their gate from a ring
 *            of 65536 words under a growing generation: nothing is reset, any number of streams.  Launches recorded by a
 *            stream capture (hipGraph) get a private word that a memset node clears on every replay — safe to replay, but
 *            each captured guarded launch keeps one of 65536 words per device for the life of the process (GIF_ENOSUP when
 *            they are used up: re-use graphs instead of re-capturing them).  Operands are 22-bit: results that cancel to
 *            < 2^-20 of their terms show it (2.1-2.5 x the native kernel's error there).  An Inf operand element comes out
 *            as NaN (the low term of an infinite high term is Inf - Inf), where NATIVE / BF16X3 propagate the Inf.
 *            Measured error against an fp64 convolution (tests/test_gpu_f16x2.py): 0.6-0.8 x the native fp32 MFMA path on
 *            normal data (16 exact products are summed before the accumulator rounds), held to <= 1.5 x on every adversarial
 *            operand set but exact cancellation (bound 4).
 *            Layers the f16x2 kernels do not take (tap-dense thin layers, Winograd GEMMs not built for it) run BF16X3.
 *            Default: the GIF_FP32_MFMA environment variable ("native" / "bf16x3" / "f16x2"), else F16X2. */
#define GIF_FP32_MFMA_NATIVE 0
#define GIF_FP32_MFMA_BF16X3 1
#define GIF_FP32_MFMA_F16X2 2
int gif_set_fp32_mfma_mode(int mode);
int gif_get_fp32_mfma_mode(void);
/* The bf16x3 entry points (fp32 activations in and out, exactly like their _f32 namesakes; stylegan2_common_layers.py:330-345).
 * The mode above is only the default a host should honour; the kernels are selected by the entry point.
 * Weights are pre-split once per optimiser step: gif_pack_weight_f32x3 writes wp3[tap][3][RP][CP] bf16 (hi, mid, lo planes;
 * RP/CP from gif_conv2d_pack_dims_x3; 6 bytes per padded element).  Activations are split inside the kernels after the LDS
 * read.  Layers with fewer than 24 input channels stay on the native kernels (gif_conv2d_x3_eligible() == 0). */
int gif_conv2d_x3_eligible(int cout, int cin);
int gif_conv2d_pack_dims_x3(int cout, int cin, int* RP, int* CP); /* like gif_conv2d_pack_dims; CP a multiple of 32 */
int gif_pack_weight_f32x3(const float* w, void* wp3, int R, int C, int KH, int KW, int RP, int CP, int64_t sr, int64_t sc,
                          int64_t sky, int64_t skx, float scale, gif_stream_t stream);
/* gif_conv2d_fwd_f32x3 / gif_conv2d_bwd_data_f32x3: declared next to their _f32 namesakes below */
/* The f16x2 entry points (ABI 4).  gif_pack_weight_f32h2 writes wp2 = [RP int32: the rows' exponents][RP int32: row flags the
 * packing sets when a 16-channel group of the row falls out of the precision window][tap][2][RP][CP] f16 (hi, lo planes of
 * scale * w * 2^e_row; RP/CP from gif_conv2d_pack_dims_x3; gif_pack_weight_f32h2_bytes of device memory).  The convolution entry
 * points take BOTH packings of the same weights: wp2 for the f16x2 kernels and wp3 (gif_pack_weight_f32x3) for the guarded
 * fallback; wp3 == NULL runs unguarded.  Eligibility as for bf16x3 (gif_conv2d_x3_eligible; no tap-dense variant). */
int64_t gif_pack_weight_f32h2_bytes(int KH, int KW, int RP, int CP);
int gif_pack_weight_f32h2(const float* w, void* wp2, int R, int C, int KH, int KW, int RP, int CP, int64_t sr, int64_t sc,
                          int64_t sky, int64_t skx, float scale, gif_stream_t stream);
/* wp2 and wp3 (= gif_pack_weight_f32x3's output) of the same weights in one launch */
int gif_pack_weight_f32h2x3(const float* w, void* wp2, void* wp3, int R, int C, int KH, int KW, int RP, int CP, int64_t sr, int64_t sc,
                            int64_t sky, int64_t skx, float scale, gif_stream_t stream);
/* the tap-dense K order (below) for the f16x2 kernels: both packings in one launch */
int64_t gif_pack_weight_f32h2_tapdense_bytes(int cin_act, int KH, int KW, int RP);
int gif_pack_weight_f32h2x3_tapdense(const float* w, void* wp2, void* wp3, int R, int C, int cin_act, int KH, int KW, int RP, int64_t sr,
                                     int64_t sc, int64_t sky, int64_t skx, float scale, gif_stream_t stream);
/* out2[0] = guarded OPS that took the bf16x3 fallback on the current device since the last reset, out2[1] = 0 (reserved);
 * synchronises the device */
int gif_h2_fallback_stats(uint64_t* out2, int reset);
/* Tap-dense K order for 3x3 layers with 8 <= cin_act < 32 contraction channels (the condition-noise convs 6->12->24 and the 24->C
 * layers that inject their result, stylegan2_common_layers.py:217-246): K runs over (tap, channel) without padding every tap to a
 * 32-float chunk — 9 taps of 24 channels take 7 K steps instead of 9, of 12 channels 4, of 8 channels 3.
 * gif_conv2d_x3_tapdense_steps() = number of steps (0: mode not applicable); gif_pack_weight_f32x3_tapdense writes
 * wp3[steps][3][RP][32] bf16 (RP from gif_conv2d_pack_dims_x3); the _tapdense convolution entry points (below) consume it.
 * Not for strided data gradients (their output phases use tap subsets) and not with per-sample input scales. */
int gif_conv2d_x3_tapdense_steps(int cin_act, int KH, int KW);
int gif_pack_weight_f32x3_tapdense(const float* w, void* wp3, int R, int C, int cin_act, int KH, int KW, int RP, int64_t sr, int64_t sc,
                                   int64_t sky, int64_t skx, float scale, gif_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Mesh rasteriser — replaces standard_rasterize_cuda.standard_rasterize / standard_rasterize_colors
 * (standard_rasterize_cuda.cpp:26-40, :59-75; kernels standard_rasterize_cuda_kernel.cu:111-233).
 * face_vertices [B,F,3,3] (x,y in pixel units, z>0), depth [B,H,W], tri [B,H,W] int32,
 * bary / images [B,H,W,3]: caller-allocated, caller-initialised, updated IN PLACE exactly like the
 * reference (depth=min, tri=face index of the min, bary/colour of that face).  Exact-depth ties are
 * resolved deterministically to the lowest face index (the reference leaves them to a race).
 * Two kernels (+ a memset of the tile counters): faces are binned into per-(image, 64x64-pixel tile) lists (front-facing test +
 * clamped bounding box; LDS-aggregated counting, one global atomic per workgroup and touched tile), then one workgroup per tile
 * keeps the tile's z-buffer in LDS from the seeding by the caller's depth buffer to the final write of depth / face /
 * attributes (LDS 64-bit atomic-min per covered pixel; no per-pixel global atomic, no key buffer in HBM).
 * `workspace`: gif_rasterize_workspace_bytes(B, F, H, W) bytes, 8-byte aligned (float32 and float64 entry points alike):
 * a counter and an F-entry list per tile — sized for the worst case, touched only where faces land (288 GB of HBM).
 * ---------------------------------------------------------------------------------------------- */
int64_t gif_rasterize_workspace_bytes(int B, int F, int H, int W);
/* The tile counters at the head of the workspace (4 bytes per image and tile) are zeroed by a memset node at the start of every
 * call and handed back zero by the tile kernel.  A host that keeps a zero-initialised workspace per (stream, problem size)
 * and never shares it between concurrent calls may switch the memset off FOR THAT WORKSPACE POINTER (ABI 3; ABI 2 had a
 * process-wide switch): on != 0 registers it, 0 forgets it (do that before the memory is freed); a call that fails forgets
 * it as well.  Calls through any other pointer keep the memset. */
int gif_rasterize_assume_clean_workspace(const void* workspace, int on);
int gif_rasterize_f32(const float* face_vertices, float* depth, int32_t* tri, float* bary, int B, int F,
                      int H, int W, void* workspace, gif_stream_t stream);
int gif_rasterize_colors_f32(const float* face_vertices, const float* face_colors, float* depth,
                             int32_t* tri, float* images, int B, int F, int H, int W, void* workspace,
                             gif_stream_t stream);

/* float64 variants: the reference dispatches both floating types (AT_DISPATCH_FLOATING_TYPES,
 * standard_rasterize_cuda_kernel.cu:252,295).  Same contract with double buffers and the same workspace (the tile keeps a
 * uint64 depth key + a uint32 face key per pixel in LDS and walks the faces twice).  NOTE: the reference's own
 * double path funnels the depth through fminf (.cu:19-29: the CAS loop of atomicMin(double*) calls fminf), i.e. it stores
 * FLOAT-rounded depths and then almost never finds `depth == zp`, so it leaves the face / barycentric buffers unwritten;
 * this implementation computes what that code intends: true double-precision minimum depth and its face. */
int gif_rasterize_f64(const double* face_vertices, double* depth, int32_t* tri, double* bary, int B, int F,
                      int H, int W, void* workspace, gif_stream_t stream);
int gif_rasterize_colors_f64(const double* face_vertices, const double* face_colors, double* depth,
                             int32_t* tri, double* images, int B, int F, int H, int W, void* workspace,
                             gif_stream_t stream);

/* Per-vertex normals — replaces vertex_normals() model/mesh_and_3d_helpers.py:5-37 (condition-render pipeline,
 * SURVEY §8(f) row 1).  verts [B,V,3]; faces [F,3] int32 (topology shared by the batch); csr_off [V+1] / csr_ent [3F]:
 * vertex -> entries (face*4 + corner), ordered corner 1, corner 2, corner 0 with faces ascending (the order of the
 * reference's three index_add_ passes); normals [B,V,3] = normalize(sum of face cross products, eps 1e-6). */
int gif_vertex_normals_f32(const float* verts, const int32_t* faces, const int32_t* csr_off, const int32_t* csr_ent,
                           float* normals, int B, int V, int F, gif_stream_t stream);

/* Texture stealing — replaces FlameTextureSpace.compute_texture_map (model/stg2_generator.py:378-421):
 * per (sample, UV texel) barycentric 3-D point -> orthographic projection (y flipped) -> bilinear fetch of the source image
 * (grid_sample, zero padding, align_corners=False), and the normal-z visibility mask.  img [B,C,H,W] NCHW; verts/normals
 * [B,V,3]; cam [B,3] = (scale, tx, ty); texel_map [T*T] = index into the valid-texel lists or -1 (those texels sample the
 * image centre, like the reference's zero grid); texel_faces [N,3] int32; texel_bc [N,3]; tex [B,C,T,T]; mask [B,1,T,T] u8.
 * The backward accumulates d tex / d img into gimg [B,C,H,W] (zeroed inside). */
int gif_texture_map_f32(const float* img, const float* verts, const float* normals, const float* cam,
                        const int32_t* texel_map, const int32_t* texel_faces, const float* texel_bc, float* tex,
                        uint8_t* mask, int B, int C, int H, int W, int V, int T, gif_stream_t stream);
int gif_texture_map_bwd_f32(const float* gtex, const float* verts, const float* normals, const float* cam,
                            const int32_t* texel_map, const int32_t* texel_faces, const float* texel_bc, float* gimg,
                            int B, int C, int H, int W, int V, int T, gif_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Convolution family (fp32 MFMA implicit GEMM) — replaces the F.conv2d / F.conv_transpose2d calls of
 *   ModulatedConv2d.forward   stylegan2_common_layers.py:307-349 (groups=batch trick -> in/out scales)
 *   EqualConv2d.forward       stylegan2_common_layers.py:175-184
 *   NoiseInjection.noise_conv stylegan2_common_layers.py:405-414
 * and their autograd (dgrad / wgrad).
 *
 * Geometry is always described through the underlying FORWARD convolution
 *   small[b,oy,ox,o] = sum_{ky,kx,i} W[o,i,ky,kx] * big[b, oy*stride+ky-pad, ox*stride+kx-pad, i]
 * "big" = [B,Hb,Wb,Cb] is the conv-input side, "small" = [B,Hs,Ws,Cs] the conv-output side.
 *   gif_conv2d_fwd_f32      : big  -> small   (conv2d)
 *   gif_conv2d_bwd_data_f32 : small -> big    (conv_transpose2d == dgrad), stride-2 runs as 4 dense phases
 *   gif_conv2d_wgrad_f32    : (small, big) -> per-split partial dW
 * Packed weights `wp` are [KH*KW][RP][CP] fp32 with rows = the op's OUTPUT channels, cols = the op's
 * INPUT channels, zero padded to the tile sizes reported by gif_conv2d_pack_dims(); tap index is always
 * the forward conv's ky*KW+kx.  Optional per-sample scales implement weight (de)modulation without
 * materialising per-sample weights:  y = act(out_scale[b,co] * conv(in_scale[b,ci] * x) + residual + bias).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t B;
    int32_t Hb, Wb, Cb; /* conv-input side  */
    int32_t Hs, Ws, Cs; /* conv-output side */
    int32_t KH, KW;     /* <= 3 x 3 */
    int32_t stride;     /* 1 or 2 */
    int32_t pad;
} gif_conv_geom;

typedef struct {
    const float* in_scale;  /* [B, Cin]  or NULL */
    const float* out_scale; /* [B, Cout] or NULL */
    const float* bias;      /* [Cout]    or NULL */
    const void* residual;   /* same shape AND element type as the output, or NULL (added before the activation) */
    int32_t act;            /* 0: identity, 1: gain * leaky_relu(., slope) */
    float slope, gain;
    /* ---- ABI 2: gradient-producer fusions (all optional: NULL / 0 = off).  When the op computes a GRADIENT w.r.t. a tensor t
     * that the forward pass produced with a fused leaky ReLU and / or consumed under a per-sample modulation, the passes that
     * used to follow — FusedLeakyReLU's backward (stylegan2_common_layers.py:22-39: mask, bias-gradient column sum) and the
     * modulation gradient of ModulatedConv2d (:311-320: sum_hw g * x) — run in this op's epilogue instead:
     *   v = contraction;  dot[b,c] += v * dot_src;  v = out_scale * v + residual + bias;  v = act(v);
     *   v *= mask_gain * (mask_src > 0 ? 1 : mask_slope);  store v;  colsum[c] += v
     * mask_src / dot_src: tensors of the OUTPUT's shape and element type (they may be the same tensor: it is then read once).
     * dot [B,Cout] and colsum [Cout] are written (not accumulated), deterministically (per-tile partial sums in red_ws, fixed
     * order reduction, no atomics).  red_ws: gif_conv_epilogue_ws_floats(B*Ho*Wo, Cout) floats, needed when dot or colsum is
     * set.  dot needs every sample's output pixels to be a multiple of 256 (a tile never straddles two samples) and a
     * single-phase op (not the stride-2 data gradient); the FIR kernels support mask_src / colsum only (power-of-two C). */
    const void* mask_src;
    float mask_slope, mask_gain;
    const void* dot_src;
    float* dot;
    float* colsum;
    float* red_ws;
    /* f16 entry points only: store the output as fp32 [B,Ho,Wo,Cout] instead of f16 (ToRGB in f16-activation mode: the running RGB
     * image reaches |v| ~ 8, where half precision resolves 3.9e-3 — the skip-connection sum is kept in fp32, it is 3 channels) */
    int32_t out_f32;
} gif_conv_epilogue;
int64_t gif_conv_epilogue_ws_floats(int64_t out_rows, int cout);

/* rows/cols padding (RP, CP) of the packed weight for an op with `cout` output and `cin` input channels */
int gif_conv2d_pack_dims(int cout, int cin, int* RP, int* CP);
/* wp[t][r][c] = scale * w[r*sr + c*sc + ky*sky + kx*skx]  (element strides; zero for r>=R or c>=C) */
int gif_pack_weight_f32(const float* w, float* wp, int R, int C, int KH, int KW, int RP, int CP, int64_t sr,
                        int64_t sc, int64_t sky, int64_t skx, float scale, gif_stream_t stream);
int gif_conv2d_fwd_f32(const float* big, const float* wp, float* small, const gif_conv_geom* g,
                       const gif_conv_epilogue* e, gif_stream_t stream);
int gif_conv2d_bwd_data_f32(const float* small, const float* wp, float* big, const gif_conv_geom* g,
                            const gif_conv_epilogue* e, gif_stream_t stream);
int gif_conv2d_fwd_f32x3(const float* big, const void* wp3, float* small, const gif_conv_geom* g, const gif_conv_epilogue* e,
                         gif_stream_t stream);
int gif_conv2d_bwd_data_f32x3(const float* small, const void* wp3, float* big, const gif_conv_geom* g,
                              const gif_conv_epilogue* e, gif_stream_t stream);
int gif_conv2d_fwd_f32h2(const float* big, const void* wp2, const void* wp3, float* small, const gif_conv_geom* g,
                         const gif_conv_epilogue* e, gif_stream_t stream);
int gif_conv2d_bwd_data_f32h2(const float* small, const void* wp2, const void* wp3, float* big, const gif_conv_geom* g,
                              const gif_conv_epilogue* e, gif_stream_t stream);
int gif_conv2d_fwd_f32h2_tapdense(const float* big, const void* wp2, const void* wp3, float* small, const gif_conv_geom* g,
                                  const gif_conv_epilogue* e, gif_stream_t stream);
int gif_conv2d_bwd_data_f32h2_tapdense(const float* small, const void* wp2, const void* wp3, float* big, const gif_conv_geom* g,
                                       const gif_conv_epilogue* e, gif_stream_t stream);
int gif_conv2d_fwd_f32x3_tapdense(const float* big, const void* wp3, float* small, const gif_conv_geom* g,
                                  const gif_conv_epilogue* e, gif_stream_t stream);
int gif_conv2d_bwd_data_f32x3_tapdense(const float* small, const void* wp3, float* big, const gif_conv_geom* g,
                                       const gif_conv_epilogue* e, gif_stream_t stream);
/* Partial weight gradients: ws[nsplit][KH*KW][RP][CP] with rows = small-side channels (o), cols = big-side
 * channels (i); (RP,CP) = gif_conv2d_wgrad_dims(Cs, Cb).  nsplit from gif_conv2d_wgrad_splits().
 * small_scale [B,Cs] / big_scale [B,Cb] (or NULL) are applied to the operands on load. */
int gif_conv2d_wgrad_dims(int Cs, int Cb, int* RP, int* CP);
int gif_conv2d_wgrad_splits(const gif_conv_geom* g);
int gif_conv2d_wgrad_f32(const float* small, const float* big, float* ws, const float* small_scale,
                         const float* big_scale, const gif_conv_geom* g, int nsplit, gif_stream_t stream);
/* same contract (operands, workspace, splits) on the bf16x3 kernels; layers with a <= 32-channel side run the native kernel */
int gif_conv2d_wgrad_f32x3(const float* small, const float* big, float* ws, const float* small_scale,
                           const float* big_scale, const gif_conv_geom* g, int nsplit, gif_stream_t stream);
/* same contract on the f16x2 kernels (ABI 4; both operands carry a running per-channel exponent) with the guarded bf16x3 fallback */
int gif_conv2d_wgrad_f32h2(const float* small, const float* big, float* ws, const float* small_scale,
                           const float* big_scale, const gif_conv_geom* g, int nsplit, gif_stream_t stream);
/* dw[r*sr + c*sc + ky*sky + kx*skx] = scale * sum_s ws[s][t][r][c]   (inverse of gif_pack_weight_f32) */
int gif_unpack_wgrad_f32(const float* ws, float* dw, int nsplit, int R, int C, int KH, int KW, int RP, int CP,
                         int64_t sr, int64_t sc, int64_t sky, int64_t skx, float scale, gif_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Winograd F(2x2,3x3) path for the SAME stride-1 / pad-1 3x3 convolutions (ModulatedConv2d
 * stylegan2_common_layers.py:343-347 without up/down-sampling, EqualConv2d :176, and their data
 * gradients): 16 instead of 36 multiplies per 2x2 output tile, output transform + epilogue fused.
 *   gif_winograd_pack_dims : padded dims of the transformed weight U [16][RP][CP]
 *   gif_winograd_weight_f32: U = G g G^T from a strided canonical weight view (rows = op output
 *                            channels, cols = op input channels); flip!=0 rotates the taps by 180 degrees
 *                            (data gradient); `scale` = the equalised-lr factor
 *   gif_conv3x3_winograd_f32: y [B,H,W,Co] = act(out_scale*conv3x3(in_scale*x [B,H,W,C]) + residual + bias);
 *                            V = scratch of gif_winograd_workspace_floats(B,H,W,C) floats (16 planes of the
 *                            transformed input, tile and channel dims padded to the GEMM blocks); H, W even;
 *                            C, Co multiples of 4.
 *                            Epilogue fields as in gif_conv2d_fwd_f32.
 * ---------------------------------------------------------------------------------------------- */
int gif_winograd_pack_dims(int cout, int cin, int* RP, int* CP);
int64_t gif_winograd_workspace_floats(int B, int H, int W, int C);
int gif_winograd_weight_f32(const float* w, float* U, int R, int C, int RP, int CP, int64_t sr, int64_t sc,
                            int64_t sky, int64_t skx, int flip, float scale, gif_stream_t stream);
int gif_conv3x3_winograd_f32(const float* x, const float* U, float* y, float* V, int B, int H, int W, int C,
                             int Co, const gif_conv_epilogue* e, gif_stream_t stream);
/* bf16x3 variants of the forward / data-gradient path (same contract; the transforms stay fp32): U3 is the transformed weight
 * split into its three bf16 terms, [16][3][RP][CP] with (RP, CP) = gif_winograd_pack_dims_x3 (RP a multiple of 128). */
int gif_winograd_pack_dims_x3(int cout, int cin, int* RP, int* CP);
int gif_winograd_weight_f32x3(const float* w, void* U3, int R, int C, int RP, int CP, int64_t sr, int64_t sc, int64_t sky,
                              int64_t skx, int flip, float scale, gif_stream_t stream);
/* f16x2 (ABI 4): U2 = [RP exponents][RP flags][16][2][RP][CP] f16 (gif_winograd_weight_f32h2_bytes), pack dims as for the bf16x3
 * GEMM; gif_conv3x3_winograd_f32h2 runs the f16x2 GEMM and then its guarded bf16x3 twin on U3 (NULL: unguarded). */
int64_t gif_winograd_weight_f32h2_bytes(int RP, int CP);
int gif_winograd_weight_f32h2(const float* w, void* U2, void* U3 /* optional: the bf16x3 transform too */, int R, int C, int RP, int CP,
                              int64_t sr, int64_t sc, int64_t sky, int64_t skx, int flip, float scale, gif_stream_t stream);
int gif_conv3x3_winograd_f32h2(const float* x, const void* U2, const void* U3, float* y, float* V, int B, int H, int W, int C, int Co,
                               const gif_conv_epilogue* e, gif_stream_t stream);
int gif_conv3x3_winograd_f32x3(const float* x, const void* U3, float* y, float* V, int B, int H, int W, int C, int Co,
                               const gif_conv_epilogue* e, gif_stream_t stream);
/* Weight gradient of the same convolution via Winograd F(3x3,2x2) (replaces autograd's wgrad of the F.conv2d calls
 * above): x [B,H,W,Cb] = conv input, gy [B,H,W,Cs] = output gradient, optional per-sample scales as in
 * gif_conv2d_wgrad_f32.  V / Mg = scratch of gif_winograd_workspace_floats(B,H,W,Cb / Cs) floats; ws = per-split partial
 * sums [nsplit][16][RP][CP] with (RP, CP) = gif_conv2d_wgrad_dims(pad32(Cs), pad32(Cb)); nsplit from
 * gif_conv3x3_winograd_wgrad_splits.  x may be NULL when V still holds the transform of the same x (and the same
 * big_scale) from gif_conv3x3_winograd_f32's forward pass.  gif_winograd_unpack_wgrad_f32 reduces the splits, applies the output transform and
 * scatters scale * dW into the strided canonical [R=Cs.., C=Cb.., 3, 3] view (deterministic: no atomics). */
int gif_conv3x3_winograd_wgrad_splits(int B, int H, int W, int Cs, int Cb);
int gif_conv3x3_winograd_wgrad_f32(const float* x, const float* gy, float* V, float* Mg, float* ws,
                                   const float* small_scale, const float* big_scale, int B, int H, int W, int Cs,
                                   int Cb, int nsplit, gif_stream_t stream);
/* same contract with the 16 plane GEMMs on the bf16x3 kernel (the transforms stay fp32) */
int gif_conv3x3_winograd_wgrad_f32x3(const float* x, const float* gy, float* V, float* Mg, float* ws,
                                     const float* small_scale, const float* big_scale, int B, int H, int W, int Cs,
                                     int Cb, int nsplit, gif_stream_t stream);
int gif_conv3x3_winograd_wgrad_f32h2(const float* x, const float* gy, float* V, float* Mg, float* ws,
                                     const float* small_scale, const float* big_scale, int B, int H, int W, int Cs,
                                     int Cb, int nsplit, gif_stream_t stream);  /* ABI 4: plane GEMMs on the f16x2 kernel + guarded fallback */
int gif_winograd_unpack_wgrad_f32(const float* ws, float* dw, int nsplit, int R, int C, int RP, int CP, int64_t sr,
                                  int64_t sc, int64_t sky, int64_t skx, float scale, gif_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * upfirdn2d — replaces upfirdn2d() stylegan2_common_layers.py:42-72 (Blur :136-152, Upsample :94-112,
 * Downsample :115-133): zero-insert by `up`, pad (negative = crop), correlate with the FIR `k` flipped
 * when flip!=0 (the reference flips => true convolution), keep every `down`-th sample.
 * x [B,Hi,Wi,C] -> y [B,Ho,Wo,C]; k is a DEVICE pointer to KH*KW taps.
 * Optional fused epilogue y = act(fir + residual + bias) (same struct as the conv; scales ignored).
 * ---------------------------------------------------------------------------------------------- */
int gif_upfirdn2d_f32(const float* x, const float* k, float* y, int B, int Hi, int Wi, int C, int Ho, int Wo,
                      int up, int down, int padx0, int pady0, int KH, int KW, int flip,
                      const gif_conv_epilogue* e, gif_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused bias + leaky-ReLU — replaces FusedLeakyReLU.forward stylegan2_common_layers.py:32-39
 * (y = gain * lrelu(x + residual + bias[c], slope)); x,y [npix, C].
 * Backward: gx = gy * gain * (y > 0 ? 1 : slope); gbias[c] = sum_pix gx  (gbias may be NULL).
 * `partial` is scratch of gif_colsum_partial_floats(npix, C) floats (only needed when gbias != NULL).
 * ---------------------------------------------------------------------------------------------- */
int gif_bias_act_f32(const float* x, const float* bias, const float* residual, float* y, int64_t npix, int C,
                     float slope, float gain, gif_stream_t stream);
int64_t gif_colsum_partial_floats(int64_t npix, int C);
int gif_bias_act_bwd_f32(const float* gy, const float* y, float* gx, float* gbias, float* partial,
                         int64_t npix, int C, float slope, float gain, gif_stream_t stream);
/* out[c] = sum_n x[n,c] */
int gif_colsum_f32(const float* x, float* out, float* partial, int64_t npix, int C, gif_stream_t stream);

/* out[b,c] = sum_hw a[b,hw,c]*b[b,hw,c] ; if scaled != NULL also scaled[b,hw,c] = scale[b,c]*a[b,hw,c]
 * (gradients of the modulation / demodulation scales of ModulatedConv2d).  partial: B*nchunk*C floats,
 * nchunk = gif_mul_reduce_chunks(HW). */
int gif_mul_reduce_chunks(int64_t HW);
int gif_mul_reduce_f32(const float* a, const float* b, const float* scale, float* scaled, float* out,
                       float* partial, int B, int64_t HW, int C, gif_stream_t stream);

/* ABI 3.  Input assembly of the NHWC kernels in ONE pass — replaces torch.cat((image, condition), 1) of Discriminator.forward
 * (stg2_discriminator.py:48-53) + channel padding + dtype / layout conversion: dst [B,H,W,Cp] (fp32 or f16) takes channels
 * [off0, off0 + C0) from src0 and, if src1 != NULL, [off1, off1 + C1) from src1 (fp32, logical [B,C,H,W] with arbitrary element
 * strides {batch, channel, row, column} — NCHW, channels-last and sliced views alike); every other channel is zero.
 * gif_unpack_nhwc_* is the adjoint w.r.t. one source: channels [c_off, c_off + C) of src [B,H,W,Cp] -> fp32 dst [B,H,W,C]. */
int gif_pack_nhwc_f32(const float* src0, int C0, int off0, const int64_t* strides0, const float* src1, int C1, int off1,
                      const int64_t* strides1, float* dst, int B, int H, int W, int Cp, gif_stream_t stream);
int gif_pack_nhwc_f16(const float* src0, int C0, int off0, const int64_t* strides0, const float* src1, int C1, int off1,
                      const int64_t* strides1, void* dst, int B, int H, int W, int Cp, gif_stream_t stream);
int gif_unpack_nhwc_f32(const float* src, float* dst, int B, int H, int W, int Cp, int c_off, int C, gif_stream_t stream);
int gif_unpack_nhwc_f16(const void* src, float* dst, int B, int H, int W, int Cp, int c_off, int C, gif_stream_t stream);

/* Condition pyramid level — replaces F.interpolate(cond, (S,S), 'bilinear', align_corners=False) of
 * StyledGenerator.forward (stg2_generator.py:309-314) for the integer ratios the model uses (R/S == 1 or even).
 * backward == 0: x [B,R,R,C] -> y [B,S,S,C];  backward != 0: x = grad [B,S,S,C] -> y = grad [B,R,R,C]. */
int gif_bilinear_down_f32(const float* x, float* y, int B, int R, int S, int C, int backward, gif_stream_t stream);

/* out[b,c] = sum_hw g * (act^-1(y) - residual - bias[c]) with act = gain*leaky_relu(., slope): the gradient of the
 * demodulation scale d[b,c] (times d) when bias/noise/activation are fused into the modulated conv's epilogue, i.e.
 * y = act(d*z + residual + bias) => d*z = act^-1(y) - residual - bias.  residual / bias may be NULL. */
int gif_act_inv_mul_reduce_f32(const float* g, const float* y, const float* residual, const float* bias, float* out,
                               float* partial, int B, int64_t HW, int C, float slope, float gain, gif_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Style path of ModulatedConv2d (stylegan2_common_layers.py:311-320): s = modulation(style) is gif_linear_nt_f32; the
 * demodulation d[b,co] = rsqrt(scale2 * sum_ci s[b,ci]^2 * wsq[co,ci] + eps), wsq[co,ci] = sum_taps W[co,ci,.]^2, and its backward
 * run as skinny GEMMs with the squares / rsqrt / chain-rule factors in their operand loads and epilogues (the reference
 * materialises the per-sample weight tensor [B,Cout,Cin,k,k] for this).  With g_acc = gd * (-scale2/2) * d^3:
 *   gif_style_demod_bwd_s_f32: gs_total = gs_in + 2 * s * (g_acc @ wsq)   (gs_in: the modulation gradient from the conv, or NULL)
 *   gif_style_demod_bwd_w_f32: g_wsq = g_acc^T @ s^2;   gif_demod_wgrad_f32: gW[co,ci,t] = 2 * W[co,ci,t] * g_wsq[co,ci]
 * Row strides in floats (multiples of 4, 16-byte aligned rows); padded columns: d -> 1, gs_total -> 0.
 * ---------------------------------------------------------------------------------------------- */
int gif_weight_sq_sum_f32(const float* w, float* wsq, int cout, int cin, int taps, gif_stream_t stream);
int gif_style_demod_f32(const float* s, const float* wsq, float* d, int B, int cout, int cin, int lds, int ldw, int ldd,
                        int cout_pad, float scale2, float eps, gif_stream_t stream);
int gif_style_demod_bwd_s_f32(const float* gd, const float* d, const float* wsq, const float* s, const float* gs_in,
                              float* gs_total, int B, int cout, int cin, int ldg, int ldw, int lds, int cin_pad, float scale2,
                              gif_stream_t stream);
int gif_style_demod_bwd_w_f32(const float* gd, const float* d, const float* s, float* g_wsq, int B, int cout, int cin, int ldg,
                              int lds, float scale2, gif_stream_t stream);
int gif_demod_wgrad_f32(const float* w, const float* g_wsq, float* gw, int cout, int cin, int taps, gif_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Minibatch standard deviation — replaces stg2_discriminator.py:59-65.
 * x [B,H,W,C] -> y [B,H,W,Cy] (Cy >= C+1): y[..., :C] = x, y[..., C] = stat[b % M], rest 0, with
 * M = B/G, stat[m] = mean_{c,h,w} sqrt(var_{g}(x[g*M+m]) + 1e-8) (biased variance over the G members).
 * stat [M] is also returned for the backward.
 * ---------------------------------------------------------------------------------------------- */
int gif_mbstd_fwd_f32(const float* x, float* y, float* stat, int B, int H, int W, int C, int Cy, int G,
                      gif_stream_t stream);
int gif_mbstd_bwd_f32(const float* x, const float* gy, float* gx, int B, int H, int W, int C, int Cy, int G,
                      gif_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * R1 / path-length reductions — replaces the norm in grad_penalty_loss loss_functions/losses.py:87-99
 * and PathLengthRegularizor :102-124:  out[b] = sum_{chw} g[b,...]^2   (n = elements per sample).
 * ---------------------------------------------------------------------------------------------- */
int gif_sqnorm_per_sample_f32(const float* g, float* out, int B, int64_t n, gif_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Texture-interpolation loss core — replaces InterpolatedTextureLoss.pairwise_texture_loss (loss_functions/losses.py:
 * 147-160) together with the common-visibility masking of its call site (:171-174):
 *   loss = mean_{c,h,w} sigmoid(((a - b) * ma * mb)^2) * f
 * a, b [C,H,W] fp32 textures; ma, mb [H,W] uint8 visibility masks or NULL; f [H,W] fp32 face-region mask;
 * partial = gif_texture_pair_loss_partials(C*HW) floats of scratch; loss / gloss = DEVICE scalars (no host sync).
 * The backward writes ga = d loss / d a * gloss (d loss / d b = -ga).  Deterministic two-stage reduction.
 * ---------------------------------------------------------------------------------------------- */
int gif_texture_pair_loss_partials(int64_t n);
int gif_texture_pair_loss_f32(const float* a, const float* b, const uint8_t* ma, const uint8_t* mb, const float* f,
                              float* partial, float* loss, int C, int64_t HW, gif_stream_t stream);
int gif_texture_pair_loss_bwd_f32(const float* a, const float* b, const uint8_t* ma, const uint8_t* mb, const float* f,
                                  const float* gloss, float* ga, int C, int64_t HW, gif_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Image resize — replaces fast_image_reshape() dataset_loaders.py:26-34 (F.interpolate, mode 'bilinear' or 'bicubic',
 * align_corners=False, no antialias) of the input / visualisation pipeline (generate_random_samples.py:190-191).
 * x [planes,Hi,Wi] -> y [planes,Ho,Wo] with planes = B*C of an NCHW fp32 tensor; mode 0 bilinear, 1 bicubic (A=-0.75,
 * border-clamped taps).  The backward scatters the same weights with fp32 atomics into gx (zeroed inside).
 * ---------------------------------------------------------------------------------------------- */
int gif_resize_f32(const float* x, float* y, int64_t planes, int Hi, int Wi, int Ho, int Wo, int mode,
                   gif_stream_t stream);
int gif_resize_bwd_f32(const float* gy, float* gx, int64_t planes, int Hi, int Wi, int Ho, int Wo, int mode,
                       gif_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Kernel timing for bench.py's roofline line: when enabled, every conv launch is bracketed by HIP
 * events on its own stream; gif_prof_read() synchronises those events and returns accumulated
 * milliseconds / work / launch count per kernel family:
 *   0 direct conv fwd+dgrad, LDS-DMA kernel (Cin >= 32; FLOPs)   5 same, register-staged kernel (Cin < 32)
 *   1 direct wgrad (FLOPs)
 *   2 Winograd GEMM fwd+dgrad                3 Winograd wgrad GEMM
 *     (2, 3: ALGORITHMIC direct-convolution FLOPs; the kernels execute 16/36 of them)
 *   4 Winograd input / gradient transforms (HBM bytes: tensor read once + transformed planes written once)
 *   5 direct conv on the register-staged kernel (Cin < 32), 6 f16 conv fwd / dgrad, 7 f16 weight gradient
 *   8 .. 11 the bf16x3 counterparts of 0, 1, 2, 3; 12 bf16x3 direct conv in the tap-dense K order (3x3, 8..28 contraction channels)
 * ---------------------------------------------------------------------------------------------- */
int gif_prof_enable(int on);
int gif_prof_read(int family, double* ms, double* flops, int64_t* launches);

/* ------------------------------------------------------------------------------------------------
 * f16-activation variants (BASELINE.json configs[4]: "fp16 activations with fp32 demodulation"; no counterpart in the
 * reference, which is fp32 only).  Activations, residuals and packed weights are IEEE half in HBM (void* here: C has no half
 * type); everything else keeps the fp32 contract of the function it mirrors: master weights, per-sample modulation /
 * demodulation scales, biases, FIR taps, all reductions, the weight-gradient workspace and dW are fp32, MFMA accumulates in
 * fp32 (v_mfma_f32_32x32x16_f16), epilogues run in fp32 and stores saturate at +-65504.  Channel counts must be multiples
 * of 8 (one 16-byte LDS-DMA chunk = 8 halfs).  The Winograd path is fp32 only.
 * ---------------------------------------------------------------------------------------------- */
int gif_conv2d_pack_dims_f16(int cout, int cin, int* RP, int* CP);
int gif_pack_weight_f16(const float* w, void* wp, int R, int C, int KH, int KW, int RP, int CP, int64_t sr, int64_t sc,
                        int64_t sky, int64_t skx, float scale, gif_stream_t stream);
/* ABI 3.  Thin high-resolution f16 layers (<= 64 contraction and <= 64 output channels, unit-stride gathers: stride-1 forward
 * convolutions, every data gradient incl. the output-parity phases of a transposed convolution — the 512^2 / 1024^2 blocks of
 * BASELINE configs[4], model/stg2_generator.py:159-209 at step 7 / 8) run the "halo" kernel inside gif_conv2d_fwd_f16 /
 * gif_conv2d_bwd_data_f16: a workgroup stages the input patch of a 16 x 16-pixel output patch plus its halo in LDS once and forms
 * the taps by shifted LDS reads (the gather kernel fetches every input pixel once per tap); per-sample modulation multiplies
 * the weight fragments in registers.  Same contracts, same epilogue.  GIF_F16_HALO=0 (read per launch) keeps the gather kernel.
 * The query answers for a FORWARD convolution with these activation channel counts on an Hs x Ws output grid. */
int gif_conv2d_f16_halo_eligible(int cin, int cout, int KH, int KW, int stride, int Hs, int Ws);
/* A/B switch of the f16 halo kernels (default on; the environment variable GIF_F16_HALO=0 sets the initial value, read once) */
int gif_conv2d_f16_halo_enable(int on);
int gif_conv2d_fwd_f16(const void* big, const void* wp, void* small, const gif_conv_geom* g, const gif_conv_epilogue* e,
                       gif_stream_t stream);
int gif_conv2d_bwd_data_f16(const void* small, const void* wp, void* big, const gif_conv_geom* g,
                            const gif_conv_epilogue* e, gif_stream_t stream);
int gif_conv2d_wgrad_dims_f16(int Cs, int Cb, int* RP, int* CP);
int gif_conv2d_wgrad_splits_f16(const gif_conv_geom* g);
int gif_conv2d_wgrad_f16(const void* small, const void* big, float* ws, const float* small_scale, const float* big_scale,
                         const gif_conv_geom* g, int nsplit, gif_stream_t stream);
int gif_upfirdn2d_f16(const void* x, const float* k, void* y, int B, int Hi, int Wi, int C, int Ho, int Wo, int up, int down,
                      int padx0, int pady0, int KH, int KW, int flip, const gif_conv_epilogue* e, gif_stream_t stream);
int gif_bias_act_f16(const void* x, const float* bias, const void* residual, void* y, int64_t npix, int C, float slope,
                     float gain, gif_stream_t stream);
int gif_bias_act_bwd_f16(const void* gy, const void* y, void* gx, float* gbias, float* partial, int64_t npix, int C,
                         float slope, float gain, gif_stream_t stream);
int gif_colsum_f16(const void* x, float* out, float* partial, int64_t npix, int C, gif_stream_t stream);
int gif_mul_reduce_f16(const void* a, const void* b, const float* scale, void* scaled, float* out, float* partial, int B,
                       int64_t HW, int C, gif_stream_t stream);
int gif_act_inv_mul_reduce_f16(const void* g, const void* y, const void* residual, const float* bias, float* out,
                               float* partial, int B, int64_t HW, int C, float slope, float gain, gif_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Skinny fp32 GEMMs of the linear layers — replace F.linear and its autograd in EqualLinear.forward
 * (stylegan2_common_layers.py:218-232: mapping network, modulation linears, discriminator head) when the row count is the
 * batch size.  Row-major operands with explicit row strides (floats); deterministic (fixed-order reduction).
 *   gif_linear_nt_f32: C[M][0:N] = act(scale * A[M][K] . B[N][K]^T + bias[N]), columns [N, cpad) written as zero
 *   gif_linear_nn_f32: C[M][0:K] = scale * A[M][N] . B[N][K]                  , columns [K, cpad) written as zero   (d/dx)
 *   gif_linear_tn_f32: C[N][K]   = scale * A[M][N]^T . B[M][K]                                                      (d/dW)
 * act: 0 identity, 1 gain * leaky_relu(., slope).  NT needs K, lda, ldb multiples of 4 and 16-byte aligned A, B.
 * ---------------------------------------------------------------------------------------------- */
int gif_linear_nt_f32(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int lda, int ldb,
                      int ldc, int cpad, float scale, int act, float slope, float gain, gif_stream_t stream);
int gif_linear_nn_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int cpad,
                      float scale, gif_stream_t stream);
int gif_linear_tn_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, float scale,
                      gif_stream_t stream);

/* The modulation bank: the EqualLinear of every ModulatedConv2d of one generator pass (stylegan2_common_layers.py:285-286 declares
 * it, :311-313 applies it; StyledConv :434-462 and ToRGB :465-490 hand every layer the same w at `len(style) < 2`,
 * stylegan2_common_layers.py Generator.forward) as one launch forward and two backward instead of one / two per layer.
 * `segs` is a HOST array (copied into the kernel arguments, at most 40 segments per call); each segment is one layer's weight
 * [n][K] (contiguous rows, n % 8 == 0) with its own output and gradient pointers — nothing is concatenated.
 *   fwd: s_l[M][n_l]  = scale * x[M][K] . w_l^T + bias_l                    (bias may be NULL)
 *   bwd: gw_l[n_l][K] = scale * gs_l^T . x,  gbias_l[n_l] = sum_m gs_l[m][:]  (gw for all segments or none; gbias may be NULL)
 *        gx[M][0:K]   = scale * sum_l gs_l . w_l, columns [K, gx_pad) zero   (gx may be NULL)
 * Deterministic: fixed summation order (segments in table order on the reduction axis of gx).  (ABI 3) */
typedef struct gif_linear_bank_seg {
    const float* w;
    const float* bias;
    float* s;
    const float* gs;
    float* gw;
    float* gbias;
    int n;
    int reserved;
} gif_linear_bank_seg;
int gif_linear_bank_fwd_f32(const float* x, int M, int K, int ldx, const gif_linear_bank_seg* segs, int nseg, float scale,
                            gif_stream_t stream);
int gif_linear_bank_bwd_f32(const float* x, int M, int K, int ldx, const gif_linear_bank_seg* segs, int nseg, float scale, float* gx,
                            int ldgx, int gx_pad, gif_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused Adam (+ EMA generator) over the flat gradient bucket — replaces torch.optim.Adam.step() (train.py:173, :243;
 * Adam(lr, betas=(0, 0.99**r)): no weight decay, no amsgrad) and generic_utils.accumulate (my_utils/generic_utils.py:63-76)
 * by one launch.  grad / exp_avg / exp_avg_sq are flat buffers sharing one offset table; parameters (and the EMA copies,
 * `ema` may be NULL per chunk) are addressed through a DEVICE array of chunks, each at most gif_adam_chunk_floats() elements.
 *   m += (g - m)*(1 - beta1);  v = v*beta2 + (1 - beta2)*g*g;  p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
 *   ema = ema*ema_decay + (1 - ema_decay)*p      (has_ema != 0)
 * bias_correction1/2 = 1 - beta^step are computed by the caller (host, double).
 * Loss scaling (f16 activation path): inv_grad_scale / found_inf are optional DEVICE scalars — every gradient is multiplied
 * by *inv_grad_scale, and when *found_inf != 0 the launch changes nothing (an overflowed step is skipped without a host sync).
 * ---------------------------------------------------------------------------------------------- */
typedef struct gif_adam_chunk {
    float* param;        /* first element of this chunk in the parameter tensor */
    float* ema;          /* same element of the EMA copy, or NULL */
    int64_t flat_offset; /* offset of the chunk in the flat grad / exp_avg / exp_avg_sq buffers */
    int32_t n;           /* elements in this chunk (<= gif_adam_chunk_floats()) */
    int32_t reserved;
} gif_adam_chunk;
int gif_adam_chunk_floats(void);
int gif_adam_ema_step_f32(const gif_adam_chunk* chunks, int nchunks, const float* grad_flat, float* exp_avg_flat,
                          float* exp_avg_sq_flat, float lr, float beta1, float beta2, float eps, double bias_correction1,
                          double bias_correction2, float ema_decay, int has_ema, const float* inv_grad_scale,
                          const float* found_inf, const float* dev_step, gif_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GIF_HIP_H */
