// fp32 MFMA implicit-GEMM "gather" convolution for gfx950 (NHWC).
//
// One kernel serves every dense contraction of the G/D step that maps input pixels to output pixels:
//   * conv2d forward, stride 1/2           (F.conv2d in ModulatedConv2d / EqualConv2d / NoiseInjection,
//                                           stylegan2_common_layers.py:345, :339, :176, :405-414)
//   * conv_transpose2d == data gradient    (F.conv_transpose2d stylegan2_common_layers.py:330 and autograd's dgrad);
//                                           stride 2 is run as 4 output-parity phases, each a DENSE gather over
//                                           the taps of that parity, so no MFMA work is spent on inserted zeros.
// GEMM view: M = B*Hp*Wp output pixels of the phase sub-grid, N = Cout, K = taps * Cin.
//   A[m,k] = in_scale[b,ci] * x[b, oy'*is + dy_t, ox'*is + dx_t, ci]   (NHWC: ci contiguous => float4 loads)
//   B[k,n] = wp[t][n][ci]                                              (packed, ci contiguous)
// The reference's per-sample weights (groups=batch) are never materialised: modulation is the in_scale on A,
// demodulation the out_scale in the epilogue (conv(x*s, W)*d, algebraically identical).
//
// Tiling: 256 threads = 4 waves; v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF peak).  Lane (i = lane&31, h = lane>>5)
// feeds A[i][k] / B[k][i] with k = 8*kk + 4*h + t for MFMA t of group kk: the same K permutation on both operands, so
// one ds_read_b128 per 32-row tile yields the operands of 4 MFMAs.  Two staging schemes:
//   * conv_gather_mfma_glds (every layer with Cin >= 32): LDS-DMA (global_load_lds_dwordx4), unpadded 128-byte LDS rows
//     with an XOR chunk swizzle on the DMA source address, per-sample input scales from an LDS table — see the comment
//     block above that kernel; tiles 128x128, 256x32 (Cout <= 32) and 64x64 (low-resolution layers).
//   * conv_gather_mfma (Cin < 32, BK = 8, and fallback): global -> VGPR -> LDS with a +4-float row pad (odd number of
//     16-B slots => conflict-free ds_read_b128), double buffered, loads of step s+1 issued before the MFMAs of step s.
// Both share conv_epilogue(): accumulators -> LDS -> coalesced float4 rows with out_scale / residual / bias / lrelu.
// The LDS-DMA kernel has two more instantiations of the same staging / addressing / epilogue: T = f16 activations
// (v_mfma_f32_32x32x16_f16) and X3 = fp32 tensors on the bf16 matrix cores ("bf16x3": exact three-way bf16 split of both operands,
// six v_mfma_f32_32x32x16_bf16 products per fp32 product, fp32 accumulation — the default mode, see glds_body and DESIGN.md 3a).
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "common.h"

#ifdef GIF_X3_TIMING_PROBE
// tools/probes/x3_sync_probe.sh: cycles the waves of the bf16x3 direct kernel spend at the mid-stage sync (own DMA wait, barrier), cycles in the K loop, waves
// [4] / [5]: cycles from kernel entry to the K loop (index tables, ring fill, first split) / from the K loop's end to the kernel's end (epilogue)
// [6] scale-back + guard, [7] accumulators -> LDS incl. barriers, [8] entry -> first DMA issue (index tables), [9] first issue -> ring filled
__device__ unsigned long long g_x3_probe[12];
extern "C" int gif_debug_x3_probe_read(unsigned long long* out8, int reset) {
    hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_x3_probe), 96) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_x3_probe), z, 96) != hipSuccess) return -1;
    }
    return 0;
}
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: plain vector loads/stores, never memcpy


// Activation / packed-weight pointers are typeless: fp32 (the reference dtype) or f16 (BASELINE config 5), chosen by the
// kernel's element-type template parameter.  Per-sample scales and the bias are always fp32.
struct GatherParams {
    const void* x;
    const void* wp;
    void* y;
    const float* in_scale;
    const float* out_scale;
    const float* bias;
    const void* residual;
    int B, Hi, Wi, Ci;  // input tensor
    int Ho, Wo, Co;     // output tensor
    int Hp, Wp;         // output sub-grid of this launch
    int os, ooy, oox;   // output pixel = (oy'*os + ooy, ox'*os + oox)
    int is;             // input  pixel = (oy'*is + dy[t], ox'*is + dx[t])
    // taps of this launch form a (nky x nkx) grid, generated arithmetically (no per-step table loads):
    //   tap (a,b): dy = dy0 + a*ddy, dx = dx0 + b*ddx, weight slice = (ky0 + a*kstep)*KW + (kx0 + b*kstep)
    int ntaps, nky, nkx, dy0, ddy, dx0, ddx, ky0, kx0, kstep, KW;
    int RP, CP;  // packed weight rows (>= Co, multiple of BN) / cols (>= Ci, multiple of BK)
    int act;
    float slope, gain;
    int M;  // B*Hp*Wp
    int tiles_m, tiles_n;
    int stab_nb, stab_stride;  // LDS table of per-sample input scales (LDS-DMA kernel): samples per tile, row stride
    const void* zero;          // 16 zero bytes in HBM: source of the LDS-DMA lanes that fall outside the tensor
    int m_begin;               // first GEMM row of this launch (a launch may cover only rows [m_begin, M))
    int x3;                    // fp32 only: 1 = wp is the pre-split bf16x3 packing, run the bf16 matrix-core kernels; 2 = f16x2 packing (planes only)
    const int* wexp;           // f16x2: exponents of the packed weight rows (the packing's header)
    unsigned* gate;            // f16x2: word the kernel raises (atomicMax with gate_gen) when a K group leaves the precision window;
    unsigned gate_gen;         //        bf16x3 launch with a gate: the guarded fallback — it runs iff *gate == gate_gen
    unsigned* h2_stats;        // [0] += 1 by a fallback launch that runs
    // gradient-producer fusions (gif_conv_epilogue ABI 2, see gif_hip.h): mask of the leaky ReLU this gradient flows into, and
    // per-tile partial sums (row part_row0 + tile_m of part_cs / part_dot, [rows][Co]) of the stored values / of contraction * dot_src
    const void* mask_src;
    const void* dot_src;
    float mask_slope, mask_gain;
    float* part_cs;
    float* part_dot;
    int part_row0;
    int part_cap;              // partial-sum rows the workspace holds per buffer (stores beyond it are dropped, the host reports them)
    int no_split;              // keep the launch in ONE tile size (per-sample partial sums need uniform rows)
    unsigned* sat_flag;        // f16: "a store saturated" flag word of the device (common.h store4_flag), else NULL
    int out_f32;               // f16 kernels: the output tensor is fp32
    int pair;                  // f16 kernels, Ci <= 32 (CP == 32): one 64-half K chunk = the channels of TWO taps (see glds_body)
    int dense;                 // bf16x3 kernels, 8 <= Ci < 32, 3x3: 16-byte chunks per tap (Ci / 4) of the tap-dense K order, else 0
    int t2_tx, t2_ty;          // halo kernel: 16 x 16-pixel patches per row / column of the output sub-grid
    int halo_dbg;              // halo kernel: ablation bits of the probe (GIF_HALO_DBG; results are wrong when set)
};

// partial-sum rows handed out to the launches of one op (bulk + remainder launches, transposed-conv phases), and the tile height
// of the last launch (host side, per calling thread)
thread_local int t_part_rows = 0;
thread_local int t_last_bm = 0;

// XCD-aware, bijective block remap (cdna guide T1): consecutive logical tiles share an XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    int xcd = bid % nx, idx = bid / nx;
    int q = nwg / nx, r = nwg % nx;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Epilogue shared by both kernels.  The accumulator tile is transposed through LDS (the staging buffers are free by
// then) so that global I/O is row-wise float4: residual / out_scale / bias loads and the output store are 16 B per lane
// and fully coalesced along the channel axis.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5).
// T2D (the halo kernel): the BM = 256 tile rows are a 16 x 16 patch of output pixels of ONE sample (row = 16 * ly + lx, patch
// origin (t2_oy0, t2_ox0) of sample t2_b in the launch's output sub-grid) instead of BM consecutive GEMM rows; m0 = tile index * BM
// only numbers the partial-sum rows.  LDS_FLOATS: floats of LDS the transposition may use (default: the staging buffers).
template <int BM, int BN, int LD, int MT, int NT, typename T = float, int THREADS = 256, bool T2D = false, int LDS_FLOATS = 0, int PFMAX = 4>
__device__ __forceinline__ void conv_epilogue(const GatherParams& p, f32x16 (&acc)[MT][NT], float* smem, int m0, int n0,
                                              int wm0, int wn0, int tid, int li, int lh, int HWp, int t2_b = 0, int t2_oy0 = 0,
                                              int t2_ox0 = 0) {
    const T* const res = static_cast<const T*>(p.residual);
    T* const yout = static_cast<T*>(p.y);
    constexpr int LDC = BN + 4;
    // the tile goes through LDS in EPI_CHUNKS row chunks so that it fits into the staging buffers' footprint.  Behind the chunk sits a row
    // table (ROWTAB floats per row): the row's 64-bit output offset (or ~0: no such row) and its sample index, computed ONCE per row by one
    // lane while the accumulators are written — the two integer divisions and the 64-bit multiply chain per row and LANE that the row loop
    // carried before were ~50 of its ~85 VALU instructions, and the loop is 16 % of a 256 x 128 tile's life at 128 channels
    // (profiles/r6_x3_life_probe.txt).  (Staging the three-stage ring's whole tile in ONE chunk — it fits — measured the same.)
    constexpr int ROWTAB = 3;
    constexpr int STAGE_FLOATS = LDS_FLOATS ? LDS_FLOATS : 2 * (BM + BN) * LD;
    constexpr int EPI_CHUNKS = (BM * (LDC + ROWTAB) <= STAGE_FLOATS) ? 1 : (BM / 2 * (LDC + ROWTAB) <= STAGE_FLOATS) ? 2
                               : (BM / 4 * (LDC + ROWTAB) <= STAGE_FLOATS) ? 4 : 8;
    constexpr int CR = BM / EPI_CHUNKS;  // rows per chunk
    static_assert(CR % 32 == 0 && CR * (LDC + ROWTAB) <= STAGE_FLOATS && CR <= THREADS, "epilogue chunk must fit the staging LDS");
    float* Cs = smem;  // [CR][LDC]
    unsigned long long* const roff = reinterpret_cast<unsigned long long*>(Cs + CR * LDC);  // [CR]
    int* const rsmp = reinterpret_cast<int*>(roff + CR);                                    // [CR]
    constexpr unsigned long long NO_ROW = ~0ull;
    constexpr int C4_ROW = BN / 4;           // float4 per tile row
    constexpr int EROWS = THREADS / C4_ROW;  // tile rows per pass
    constexpr int E_IT = CR / EROWS;
    static_assert(CR % EROWS == 0, "epilogue row mapping");
    const int e_row0 = tid / C4_ROW, e_c = (tid % C4_ROW) * 4;
    const int n = n0 + e_c;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && n < p.Co) bias4 = *reinterpret_cast<const float4*>(p.bias + n);
    const T* const msk = static_cast<const T*>(p.mask_src);
    const T* const dsrc = static_cast<const T*>(p.dot_src);
    const bool fused = msk || dsrc || p.part_cs || p.part_dot;  // workgroup-uniform
#if defined(GIF_EPI_PROBE) && GIF_EPI_PROBE == 3  // timing probe (tools/probes/epilogue_probe.sh; results are WRONG): NO epilogue — every lane
    // stores the sum of its accumulators, one dword (keeps the K loop alive): the bound for any epilogue rewrite
    {
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
        if (m0 + wm0 < p.M) yout[(size_t)(m0 + wm0) * p.Co + n0 + wn0 + (tid & 63)] = (T)sacc;
        return;
    }
#endif
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), ds = make_float4(0.f, 0.f, 0.f, 0.f);
    // demodulation factors of this lane's four columns: a tile rarely spans more than two samples, so the first sample's and its successor's
    // are loaded once (the row loop loaded them per row and waited for each: +11k cycles per tile on the generator's modulated convs)
    int osb = 0;
    // (ext-vector values, not float4 structs: hipcc kept the structs in stack slots and selected between their ADDRESSES per row)
    f32x4 osc0 = {1.f, 1.f, 1.f, 1.f}, osc1 = osc0;
    constexpr bool OSC = sizeof(T) == 4;  // (f16 kernels keep the per-row load: 8 more VGPRs cost them a wave per SIMD)
    if (OSC && p.out_scale && n < p.Co) {
        osb = T2D ? t2_b : (m0 < p.M ? m0 : p.M - 1) / HWp;  // workgroup-uniform
        osc0 = *reinterpret_cast<const f32x4*>(p.out_scale + (size_t)osb * p.Co + n);
        osc1 = *reinterpret_cast<const f32x4*>(p.out_scale + (size_t)(osb + 1 < p.B ? osb + 1 : osb) * p.Co + n);
    }
#pragma unroll
    for (int c = 0; c < EPI_CHUNKS; ++c) {
#ifdef GIF_X3_TIMING_PROBE
        const long long probe_c0 = clock64();
#endif
        __syncthreads();  // staging buffers (c == 0) / previous chunk fully consumed
        if (tid < CR) {  // row table of this chunk
            int b, oy, ox;
            bool okr;
            if constexpr (T2D) {
                const int r2 = c * CR + tid;
                b = t2_b; oy = t2_oy0 + (r2 >> 4); ox = t2_ox0 + (r2 & 15);
                okr = oy < p.Hp && ox < p.Wp;  // else: the patch overhangs the sub-grid
            } else {
                const int m = m0 + c * CR + tid;
                okr = m < p.M;
                const int mm = okr ? m : 0;
                b = mm / HWp;
                const int rr = mm - b * HWp;
                oy = rr / p.Wp; ox = rr - oy * p.Wp;
            }
            roff[tid] = okr ? (unsigned long long)((((size_t)b * p.Ho + (oy * p.os + p.ooy)) * p.Wo + (ox * p.os + p.oox)) * p.Co) : NO_ROW;
            rsmp[tid] = b;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int rbase = wm0 + i * 32;           // wave-uniform
            if (rbase / CR != c) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int row = rbase - c * CR + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    Cs[row * LDC + wn0 + j * 32 + li] = acc[i][j][r];
                }
        }
        __syncthreads();
#ifdef GIF_X3_TIMING_PROBE
        if ((tid & 63) == 0) atomicAdd(&g_x3_probe[7], (unsigned long long)(clock64() - probe_c0));
#endif
        if (n < p.Co) {
            // two copies of the row loop: the plain one carries none of the gradient-producer work (measured on the f16 step, whose
            // MFMA phase is 8x shorter: the extra branches and the running sums cost 2 % of the whole step when they ran always)
            // The plain row loop, and the gradient-producer one.  In the latter the rows go in batches of PF: FIRST the mask / dot
            // source loads of the whole batch, THEN the arithmetic and the stores — written as one loop, each row's load sits
            // behind the previous row's store in program order (the compiler cannot prove that y does not alias the sources) and
            // every row pays a full memory round trip on its own: the fused data gradient of 128 -> 128 channels at 256^2 took
            // 0.39 ms longer than the plain one for an 85 us read (round 3: "unexplained").  Only these loads are batched and PF is
            // 4: batching the residual / scale loads of the plain loop as well, 8 rows deep, cost 50-110 VGPRs per kernel and made
            // the f16 step 15 % SLOWER (occupancy; r4l).
            auto rows = [&](auto fused_tag) __attribute__((always_inline)) {
                constexpr bool FUSED = decltype(fused_tag)::value;
                // (the plain loop batches its LDS reads the same way — row table and tile values of PF rows first: one at a time, every row
                // paid two LDS round trips in sequence; its residual loads stay per row, see above)
                constexpr int PF = (FUSED || sizeof(T) == 4) ? (E_IT < PFMAX ? E_IT : PFMAX) : 1;  // (f16 kernels: occupancy-bound, see above)
                static_assert(E_IT % PF == 0, "epilogue batches");
                const bool two_src = FUSED && msk && dsrc && msk != dsrc;
                const f32x4 os0 = osc0, os1 = osc1;  // copies: a select between the by-reference captures is a select of stack addresses
#pragma unroll 4
                for (int it0 = 0; it0 < E_IT; it0 += PF) {
                    size_t off[PF];
                    int bs[PF];
                    bool ok[PF];
                    float4 xa[PF], xb[PF], vv[PF];
#pragma unroll
                    for (int k = 0; k < PF; ++k) {
                        const int row = e_row0 + (it0 + k) * EROWS;
                        const unsigned long long o = roff[row];
                        if constexpr (sizeof(T) == 4) vv[k] = *reinterpret_cast<const float4*>(Cs + row * LDC + e_c);
                        ok[k] = o != NO_ROW;
                        bs[k] = rsmp[row];
                        off[k] = (size_t)o + n;
                        if constexpr (FUSED) {
                            xa[k] = xb[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (ok[k]) {
                                if (dsrc) xa[k] = gif::load4(dsrc + off[k]);          // dot source (also the mask source if they coincide)
                                else if (msk) xa[k] = gif::load4(msk + off[k]);        // mask source only
                                if (two_src) xb[k] = gif::load4(msk + off[k]);         // distinct mask source
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < PF; ++k) {
                        if (!ok[k]) continue;
                        float4 v = vv[k];
                        if constexpr (sizeof(T) != 4) v = *reinterpret_cast<const float4*>(Cs + (e_row0 + (it0 + k) * EROWS) * LDC + e_c);
                        if (FUSED && dsrc) {  // modulation gradient: sum_pixels contraction * x
                            ds.x += v.x * xa[k].x; ds.y += v.y * xa[k].y; ds.z += v.z * xa[k].z; ds.w += v.w * xa[k].w;
                        }
                        if (p.out_scale) {
                            f32x4 d = bs[k] == osb ? os0 : os1;
                            if (!OSC || __builtin_expect(bs[k] > osb + 1, 0)) d = *reinterpret_cast<const f32x4*>(p.out_scale + (size_t)bs[k] * p.Co + n);
                            const float dx = d[0], dy = d[1], dz = d[2], dw = d[3];
                            v.x *= dx; v.y *= dy; v.z *= dz; v.w *= dw;
                        }
                        if (res) {
                            float4 rv = gif::load4(res + off[k]);
                            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                        }
                        v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
                        if (p.act) {
                            v.x = (v.x > 0.f ? v.x : v.x * p.slope) * p.gain; v.y = (v.y > 0.f ? v.y : v.y * p.slope) * p.gain;
                            v.z = (v.z > 0.f ? v.z : v.z * p.slope) * p.gain; v.w = (v.w > 0.f ? v.w : v.w * p.slope) * p.gain;
                        }
                        if (FUSED) {
                            if (msk) {  // backward of the leaky ReLU that produced the tensor this gradient belongs to
                                const float4 xs = two_src ? xb[k] : xa[k];
                                v.x *= p.mask_gain * (xs.x > 0.f ? 1.f : p.mask_slope); v.y *= p.mask_gain * (xs.y > 0.f ? 1.f : p.mask_slope);
                                v.z *= p.mask_gain * (xs.z > 0.f ? 1.f : p.mask_slope); v.w *= p.mask_gain * (xs.w > 0.f ? 1.f : p.mask_slope);
                            }
                            cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
                        }
#ifdef GIF_NOSTORE_PROBE  // timing probe (tools/probes/epilogue_probe.sh; results are WRONG): the row loop without its global stores
                        if (p.M < 0)
#endif
                        {
                            if (sizeof(T) == 2 && p.out_f32) gif::store4(static_cast<float*>(p.y) + off[k], v);
                            else gif::store4_flag(yout + off[k], v, p.sat_flag);
                        }
                    }
                }
            };
            if (fused) rows(std::true_type{});
            else rows(std::false_type{});
        }
    }
    if (p.part_cs || p.part_dot) {  // workgroup-uniform: per-tile partial sums, fixed order
        __syncthreads();
        float4* red = reinterpret_cast<float4*>(smem);  // [2][THREADS]
        red[tid] = cs;
        red[THREADS + tid] = ds;
        __syncthreads();
        if (tid < C4_ROW && n < p.Co) {
            float4 a = red[tid], b = red[THREADS + tid];
#pragma unroll 4  // (a full unroll of EROWS = 32 hoists 62 float4 loads: 248 VGPRs)
            for (int k = 1; k < EROWS; ++k) {
                const float4 a2 = red[k * C4_ROW + tid], b2 = red[THREADS + k * C4_ROW + tid];
                a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
                b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
            }
            const int prow_i = p.part_row0 + (m0 - p.m_begin) / BM;
            const size_t prow = (size_t)prow_i * p.Co + n;
            if (prow_i < p.part_cap) {  // never past the workspace: FusedSums::finish reports the overflow (tiles have >= 64 rows, so it cannot happen today)
                if (p.part_cs) *reinterpret_cast<float4*>(p.part_cs + prow) = a;
                if (p.part_dot) *reinterpret_cast<float4*>(p.part_dot + prow) = b;
            }
        }
    }
}

// Staging state of one thread: registers only (every index below is a compile-time constant after unrolling).
template <int A_IT, int B_IT>
struct Stage {
    f32x4 a[A_IT];
    f32x4 s[A_IT];
    f32x4 b[B_IT];
    unsigned mask;
};

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N>
__global__ void __launch_bounds__(256) conv_gather_mfma(const GatherParams p) {
    constexpr int THREADS = 256;
    constexpr int LD = BK + 4;  // padded LDS row (floats)
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int F4_ROW = BK / 4;                 // float4 per tile row
    constexpr int ROWS_PER_IT = THREADS / F4_ROW;  // tile rows covered by one pass of the 256 threads
    constexpr int A_IT = BM / ROWS_PER_IT;
    constexpr int B_IT = (BN + ROWS_PER_IT - 1) / ROWS_PER_IT;
    constexpr bool B_FULL = (BN % ROWS_PER_IT) == 0;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    static_assert(BM % ROWS_PER_IT == 0, "A tile / thread mapping");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                // [2][BM][LD]
    float* Bs = smem + 2 * BM * LD;  // [2][BN][LD]
    const float* const px = static_cast<const float*>(p.x);    // this kernel is fp32 only
    const float* const pw = static_cast<const float*>(p.wp);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;

    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
    const int m0 = p.m_begin + tm * BM, n0 = tn * BN;
    const int HWp = p.Hp * p.Wp;

    const int t_row = tid / F4_ROW, t_c4 = (tid % F4_ROW) * 4;
    const bool b_row_ok = B_FULL || (t_row + (B_IT - 1) * ROWS_PER_IT) < BN;

    // ---- per-thread A rows (fixed over the K loop).  32-bit element offsets (host checks numel < 2^31):
    // a_base = offset of (b, iy0, ix0, t_c4); a tap / K-chunk only adds the wave-uniform (dy*Wi+dx)*Ci + kc.
    int a_iy0[A_IT], a_ix0[A_IT], a_base[A_IT], a_soff[A_IT];
    unsigned row_ok = 0;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        int m = m0 + t_row + it * ROWS_PER_IT;
        bool ok = m < p.M;
        int mm = ok ? m : 0;
        int b = mm / HWp;
        int r = mm - b * HWp;
        int oy = r / p.Wp, ox = r - oy * p.Wp;
        a_iy0[it] = oy * p.is;
        a_ix0[it] = ox * p.is;
        a_base[it] = ((b * p.Hi + a_iy0[it]) * p.Wi + a_ix0[it]) * p.Ci + t_c4;
        a_soff[it] = b * p.Ci + t_c4;
        row_ok |= (ok ? 1u : 0u) << it;
    }

    Stage<A_IT, B_IT> st;
    const int ksteps_c = p.CP / BK;
    const int nsteps = p.ntaps * ksteps_c;
    const bool has_scale = p.in_scale != nullptr;
    int ld_a = 0, ld_b = 0, ld_kc = 0;  // (tap row, tap col, K-chunk) of the next load_global call

    // Branch-free global loads: out-of-range lanes read a valid dummy address and are zeroed when staged to LDS;
    // the modulation multiply is deferred to the LDS store so no load result is consumed before the MFMAs.
    auto load_global = [&]() __attribute__((always_inline)) {
        const int dy = p.dy0 + ld_a * p.ddy, dx = p.dx0 + ld_b * p.ddx;
        const int widx = (p.ky0 + ld_a * p.kstep) * p.KW + p.kx0 + ld_b * p.kstep;
        const int kc = ld_kc;
        const float* wt = pw + ((size_t)widx * p.RP + n0 + t_row) * p.CP + kc + t_c4;
        const int tap_off = (dy * p.Wi + dx) * p.Ci + kc;
        const bool ch_ok = kc + t_c4 < p.Ci;
        unsigned mask = 0;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            bool ok = ((row_ok >> it) & 1u) && ch_ok && (unsigned)(a_iy0[it] + dy) < (unsigned)p.Hi &&
                      (unsigned)(a_ix0[it] + dx) < (unsigned)p.Wi;
            int off = ok ? a_base[it] + tap_off : 0;
            st.a[it] = *reinterpret_cast<const f32x4*>(px + off);
            if (has_scale) st.s[it] = *reinterpret_cast<const f32x4*>(p.in_scale + (ok ? a_soff[it] + kc : 0));
            mask |= (ok ? 1u : 0u) << it;
        }
        st.mask = mask;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            bool ok = (it < B_IT - 1) || b_row_ok;
            st.b[it] = *reinterpret_cast<const f32x4*>(ok ? wt + (size_t)it * ROWS_PER_IT * p.CP : pw);
        }
        ld_kc += BK;
        if (ld_kc >= p.CP) {
            ld_kc = 0;
            if (++ld_b == p.nkx) { ld_b = 0; ++ld_a; }
        }
    };
    auto store_lds = [&](int buf) __attribute__((always_inline)) {
        float* Ab = As + buf * BM * LD + t_row * LD + t_c4;
        float* Bb = Bs + buf * BN * LD + t_row * LD + t_c4;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            f32x4 v = st.a[it];
            if (has_scale) v *= st.s[it];
            if (!((st.mask >> it) & 1u)) v = (f32x4)(0.f);
            *reinterpret_cast<f32x4*>(Ab + it * ROWS_PER_IT * LD) = v;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            if ((it < B_IT - 1) || b_row_ok) *reinterpret_cast<f32x4*>(Bb + it * ROWS_PER_IT * LD) = st.b[it];
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int buf) __attribute__((always_inline)) {
        const float* Ab = As + buf * BM * LD + (wm0 + li) * LD + lh * 4;
        const float* Bb = Bs + buf * BN * LD + (wn0 + li) * LD + lh * 4;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            float4 av[MT], bv[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) av[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LD + kk * 8);
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LD + kk * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    float a = t == 0 ? av[i].x : t == 1 ? av[i].y : t == 2 ? av[i].z : av[i].w;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        float b = t == 0 ? bv[j].x : t == 1 ? bv[j].y : t == 2 ? bv[j].z : bv[j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
    };

    load_global();
    store_lds(0);
    __syncthreads();
    int cur = 0;
    for (int step = 0; step + 1 < nsteps; ++step) {
        load_global();  // step+1: in flight during the MFMAs below
        __builtin_amdgcn_sched_barrier(0);  // keep every global load ahead of the MFMA block (hipcc sinks them otherwise)
        compute(cur);
        __builtin_amdgcn_sched_barrier(0);
        store_lds(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    compute(cur);

    conv_epilogue<BM, BN, LD, MT, NT>(p, acc, smem, m0, n0, wm0, wn0, tid, li, lh, HWp);
}

// ------------------------------------------------------------------------------------------------------------------
// LDS-DMA variant for BK = 32 without input modulation (all discriminator convs / dgrads, the generator's
// condition-noise convs).  Ablation on MI355X (profiles/r1_conv_ablation.md): MFMA loop alone 151 TF, + global loads
// 145 TF, + VGPR->LDS store pass 122 TF — the ds_write burst (and the staging VGPRs) is what costs, not the loads.
// Here every tile row segment goes global -> LDS directly (global_load_lds_dwordx4): no staging registers, no
// ds_write, no zero-fill selects.  The DMA destination is lane-linear (wave-uniform base + lane*16 B), so LDS rows are
// the unpadded 128-B K-chunks; bank conflicts of the ds_read_b128 operand reads are removed by an XOR swizzle of the
// 16-B chunk index with (row>>1)&7, applied on the SOURCE address of the DMA and on the read address (guide rule 21).
// Out-of-image / out-of-range lanes read a 16-byte zero page instead of being masked.
// SCALE: input modulation (ModulatedConv2d's s[b,ci], or d[b,co] in its dgrad) through an LDS table, see below.
// ------------------------------------------------------------------------------------------------------------------
// T = float: BK = 32 floats per LDS row, v_mfma_f32_32x32x2_f32, one ds_read_b128 feeds 4 MFMAs (k = 8*kk + 4*h + t).
// T = f16  : BK = 64 halfs per LDS row — the SAME 128-byte rows, 16-byte chunks and XOR swizzle — v_mfma_f32_32x32x16_f16:
//            one ds_read_b128 = 8 halfs = the whole operand of one MFMA (lane half h supplies k = 16*kk + 8*h .. +7),
//            fp32 accumulators, fp32 epilogue (demodulation, bias, residual, lrelu), saturating f16 store.  The per-sample
//            modulation table is f16 (x * s is what the f16 MFMA consumes anyway); demodulation stays fp32 in the epilogue.
// X3   : fp32 data, bf16 matrix cores ("bf16x3", see common.h split_pair).  A: the same fp32 rows are staged by the same
//        DMA; the 16-byte fragments of two consecutive k-groups (8 floats per lane) are split exactly into three bf16x8
//        vectors after the LDS read (and after the fp32 modulation multiply).  B: the weights arrive PRE-SPLIT
//        (gif_pack_weight_f32x3: [tap][3 terms][RP][CP] bf16), three [BN][32] bf16 tiles of 64-byte rows per stage whose
//        16-byte chunks are XOR-swizzled with (row >> 2) & 3, so a B operand is one ds_read_b128 and costs no VALU.
//        Six v_mfma_f32_32x32x16_bf16 per tile pair and 16 k-values.  On this chip VALU work does NOT hide under the MFMAs
//        of the same SIMD (measured: time ~ MFMA + VALU), so the split work per MFMA is what bounds the kernel.
// raw buffer descriptor (stride 0, num_records 2^32 - 1) + LDS-DMA through it: `buffer_load_dwordx4 v, s[rsrc], s_off offen lds`.
// (The builtins exist in the device pass only; the host pass of this translation unit sees empty stand-ins.)
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t make_buf_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xFFFFFFFFu, 0x00020000);
}
__device__ __forceinline__ void buf_load_lds16(buf_rsrc_t r, __attribute__((address_space(3))) void* lds, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds, 16, (int)voff, soff, 0, 0);
}
#else
struct buf_rsrc_t {};
__device__ __forceinline__ buf_rsrc_t make_buf_rsrc(const void*) { return {}; }
__device__ __forceinline__ void buf_load_lds16(buf_rsrc_t, __attribute__((address_space(3))) void*, unsigned, int) {}
#endif

// NST: stages of the LDS operand ring.  2 everywhere except the 8-wave f16x2 kernel (one workgroup per CU: 3 x 48 KB fit), whose stage
// is half as long as the bf16x3 one it replaced: with two stages the DMA of stage s + 1 is issued when stage s starts and a first-touch
// (HBM) row is not there a stage later (x3_sync_probe, f16x2: 3-15 % of the K loop waiting for the wave's own DMA on 3x3 layers, 38 %
// on 1x1); with three it has two stages to land.
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, bool SCALE, int BK, int X3 = 0, int NST = 2>
__device__ __forceinline__ void glds_body(const GatherParams& p, const int bid, const int nwg) {
    constexpr bool F16 = sizeof(T) == 2;
    static_assert(NST == 2 || (NST == 3 && X3 == 2), "three-stage ring: f16x2 schedule only");
    static_assert(!X3 || !F16, "the split path is an fp32 mode");
    constexpr int EPC = 16 / sizeof(T);  // elements per 16-byte chunk
    constexpr int LD = BK;             // unpadded LDS row (elements)
    constexpr int CH = BK / EPC;       // 16-byte chunks per row
    constexpr int RB = 256 / (BK * (int)sizeof(T));  // rows per 256-byte LDS bank row (2 for 128-byte rows)
    constexpr int THREADS = 64 * WAVES_M * WAVES_N;
    constexpr int NWAVES = WAVES_M * WAVES_N;
    constexpr int RPP = THREADS / CH;  // rows filled by one pass of the workgroup's lanes
    constexpr int RPW = 64 / CH;       // rows filled by one wave instruction (1 KiB)
    constexpr int KG = X3 ? CH / 4 : CH / 2;  // k-groups per step: two chunks (lane halves h = 0 / 1) each; X3: four
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int A_IT = BM / RPP, B_IT = (BN + RPP - 1) / RPP;
    static_assert((NWAVES == 4 || NWAVES == 8) && BM % RPP == 0 && (X3 || BN % RPP == 0 || BN < RPP), "tile config");
    static_assert(BK * sizeof(T) == 128, "only 128-byte rows are validated (a 64-byte-row variant measured 8-10 % slower)");
    // X3: pre-split weight tiles, [2][NPL][BN][32] 16-bit elements in 1-KiB DMA blocks of 16 rows; NPL = 3 bf16 planes (hi, mid, lo
    // of bf16x3) or 2 f16 planes (hi, lo of f16x2 — H2, common.h)
    constexpr bool H2 = X3 == 2;
    constexpr int NPL = H2 ? 2 : 3;
    constexpr int B3_BLK = NPL * BN / 16;                    // blocks per stage
    constexpr int B3_IT = (B3_BLK + NWAVES - 1) / NWAVES;    // blocks per wave and stage

    if constexpr (X3 == 1) {
        // guarded fallback of an f16x2 launch (common.h): nothing to do unless that launch (or the weight packing) raised the gate
        if (p.gate) {
            if (*p.gate < p.gate_gen) return;
            if (bid == 0 && threadIdx.x == 0 && p.h2_stats && p.m_begin == 0) atomicAdd(p.h2_stats, 1u);  // (not the remainder launch)
        }
    }
#ifdef GIF_DEPHASE_PROBE  // timing probe (tools/probes/dephase_probe.sh): the first round of workgroups starts in GIF_DEPHASE_PROBE phases, a K loop's
    // 1/PHASES apart on neighbouring CUs of an XCD — do the launch-wide bursts of ring fills and tile stores (every CU at once) cost time?
    if (X3 == 2 && bid < 256) {
        const int ph = (bid >> 3) % GIF_DEPHASE_PROBE;
        const long long d = (long long)(p.ntaps * (p.CP / BK)) * 4300 * ph / GIF_DEPHASE_PROBE, t0 = clock64();
        while (clock64() - t0 < d) __builtin_amdgcn_s_sleep(32);
    }
#endif
#ifdef GIF_X3_TIMING_PROBE
    const long long probe_entry = clock64();
    long long probe_loop_end = probe_entry;
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    T* As = reinterpret_cast<T*>(smem);  // [NST][BM][LD]
    T* Bs = As + NST * BM * LD;          // [NST][BN][LD]
    unsigned short* const B3 = reinterpret_cast<unsigned short*>(Bs);  // X3: [NST][NPL][BN][32] bf16 / f16
    const T* const px = static_cast<const T*>(p.x);
    const T* const pw = static_cast<const T*>(p.wp);
    const T* const pzero = static_cast<const T*>(p.zero);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
    const int tile = xcd_remap(bid, nwg);
    const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
    const int m0 = p.m_begin + tm * BM, n0 = tn * BN;
    const int HWp = p.Hp * p.Wp;
    // this lane fills LDS row tid/CH + RPP*it, physical 16-B chunk tid%CH, with the LOGICAL chunk (tid%CH)^f(row),
    // f(row) = (row / RB) % CH: the 16 rows of a ds_read_b128 lane group then hit 16 distinct 16-B slots
    const int t_row = tid / CH;
    // f16 layers with <= 32 input channels ("pair" mode, CP == 32): a 64-half K chunk would be half zero padding — half of the
    // MFMAs and half of the DMA pieces for nothing on the 1024^2 / 512^2 blocks.  Instead one chunk carries the 32 (padded)
    // channels of TWO consecutive taps: lanes whose logical 16-byte chunk lies in the upper half of the row fetch the second tap's
    // pixel (and its weight slice); 9 taps = 5 steps instead of 9.
    const bool pair = F16 && p.pair;
    const int lchunk = (tid % CH) ^ ((t_row / RB) % CH);
    const bool pair_hi = pair && lchunk >= CH / 2;
    // bf16x3 layers with 8 <= Ci < 32 ("tap-dense" mode, 3x3 only): a K chunk of 32 floats per tap would be 25 % (Ci = 24) to 75 %
    // (Ci = 8) zero padding.  Instead K runs densely over (tap, channel): 16-byte chunk q = 8 * step + lane chunk belongs to tap
    // q / (Ci/4), channels 4 * (q % (Ci/4)).. — each DMA lane fetches its own tap's pixel; the weights are packed in the same order
    // (gif_pack_weight_f32x3_tapdense).  9 taps of 24 channels = 7 steps instead of 9, of 12 channels = 4, of 8 channels = 3.
    const int dense_cpt = X3 ? p.dense : 0;
    const int src_c4 = dense_cpt ? 0 : (pair ? (lchunk & (CH / 2 - 1)) : lchunk) * EPC;
    const bool b_lane_ok = (BN % RPP == 0) || t_row < BN;

    int a_iy0[A_IT], a_ix0[A_IT], a_base[A_IT];
    unsigned row_ok = 0;
    // X3: the activation DMA is `buffer_load_dwordx4 ... offen lds` — per lane and row ONE byte offset (loop invariant) and ONE bit per
    // tap saying whether that tap's pixel exists; the tap's offset travels in the instruction's SGPR offset and a lane whose pixel is
    // padding sends an out-of-range offset (the buffer unit then writes zeros into LDS: tools/probes/buffer_lds_oob_probe.hip).  Per
    // step that is 3 VALU per DMA instruction instead of the ~10 of the bounds checks + 64-bit address + zero-page select below, and
    // the matrix pipe does not run while a SIMD issues VALU (profiles/r4_pmc_x3.md).  Needs < 2^30 input elements and <= 32 taps.
    constexpr bool BUF = X3 || F16;  // (the native fp32 kernel keeps 64-bit addresses: it is the entry point for tensors of any size)
    unsigned a_voff[A_IT], a_mask[A_IT];
    int min_off = 0;  // most negative tap offset (elements): folded into the buffer base so that the SGPR offsets are >= 0
    if constexpr (BUF) {
        const int dyl = p.dy0 + (p.nky - 1) * p.ddy, dxl = p.dx0 + (p.nkx - 1) * p.ddx;
        min_off = (min(p.dy0, dyl) * p.Wi + min(p.dx0, dxl)) * p.Ci;
    }
    // X3: the CH lanes of a row would each repeat the row's two integer divisions and tap tests, A_IT times: ONE lane per tile row does them
    // and parks (byte offset, tap mask) in the ring's last stage, which no DMA touches before the K loop's first barrier (7.7k of a 256 x 128
    // tile's 187k cycles went into these tables, profiles/r6_x3_life_probe.txt)
    unsigned* const rtab = reinterpret_cast<unsigned*>(As + (NST - 1) * BM * LD);  // [BM][2]
    if constexpr (X3 != 0) {
        static_assert(BM <= THREADS, "row table: one lane per tile row");
        if (tid < BM) {
            const int m = m0 + tid;
            const bool ok = m < p.M;
            const int mm = ok ? m : 0;
            const int b = mm / HWp;
            const int r = mm - b * HWp;
            const int oy = r / p.Wp, ox = r - oy * p.Wp;
            const int iy0 = oy * p.is, ix0 = ox * p.is;
            unsigned cm = 0, mk = 0;
            for (int tb = 0; tb < p.nkx; ++tb) cm |= ((unsigned)(ix0 + p.dx0 + tb * p.ddx) < (unsigned)p.Wi ? 1u : 0u) << tb;
            for (int ta = 0; ta < p.nky; ++ta)
                mk |= (ok && (unsigned)(iy0 + p.dy0 + ta * p.ddy) < (unsigned)p.Hi ? cm : 0u) << (ta * p.nkx);
            rtab[2 * tid] = (unsigned)(((b * p.Hi + iy0) * p.Wi + ix0) * p.Ci) * (unsigned)sizeof(T);
            rtab[2 * tid + 1] = mk;
        }
    }
#pragma unroll
    for (int it = 0; it < (X3 ? 0 : A_IT); ++it) {
        int m = m0 + t_row + it * RPP;
        bool ok = m < p.M;
        int mm = ok ? m : 0;
        int b = mm / HWp;
        int r = mm - b * HWp;
        int oy = r / p.Wp, ox = r - oy * p.Wp;
        if constexpr (BUF && !X3) {
            const int iy0 = oy * p.is, ix0 = ox * p.is;
            a_voff[it] = (unsigned)(((b * p.Hi + iy0) * p.Wi + ix0) * p.Ci + src_c4) * (unsigned)sizeof(T);
            unsigned mk = 0;
            for (int ta = 0; ta < p.nky; ++ta)
                for (int tb = 0; tb < p.nkx; ++tb) {
                    const bool v = ok && (unsigned)(iy0 + p.dy0 + ta * p.ddy) < (unsigned)p.Hi &&
                                   (unsigned)(ix0 + p.dx0 + tb * p.ddx) < (unsigned)p.Wi;
                    mk |= (v ? 1u : 0u) << (ta * p.nkx + tb);
                }
            a_mask[it] = mk;
        }
        if constexpr (!BUF || F16) {  // native fp32 kernel; f16: the tap-pair mode below (per-lane taps through the buffer unit measured 9 % slower)
            a_iy0[it] = oy * p.is;
            a_ix0[it] = ox * p.is;
            a_base[it] = ((b * p.Hi + a_iy0[it]) * p.Wi + a_ix0[it]) * p.Ci + src_c4;
            row_ok |= ((m < p.M) ? 1u : 0u) << it;
        }
    }
    // Modulated convs: the DMA cannot scale data in flight, so the per-sample input scales s[b, 0:Ci) of the (few)
    // samples this tile touches are parked in an LDS table and multiplied into the A fragments after the operand read
    // ("weight modulation" applied on the activation side; algebraically identical).  Filled BEFORE the first DMA so no
    // ordinary global load is outstanding while DMAs are in flight (hipcc would drain them with vmcnt(0)).
    T* Stab = X3 ? reinterpret_cast<T*>(B3 + NST * NPL * BN * 32) : As + 2 * (BM + BN) * LD;  // [stab_nb][stab_stride], row tails zeroed
    int s_row[MT];
    if (SCALE) {
        const int b_first = m0 / HWp;
        for (int e = tid; e < p.stab_nb * p.stab_stride; e += THREADS) {
            int bl = e / p.stab_stride, c = e - bl * p.stab_stride;
            int b = b_first + bl;
            Stab[e] = (T)((c < p.Ci && b < p.B) ? p.in_scale[(size_t)b * p.Ci + c] : 0.f);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int m = m0 + wm0 + i * 32 + li;
            if (m >= p.M) m = p.M - 1;
            s_row[i] = (m / HWp - b_first) * p.stab_stride + lh * EPC;
        }
    }
    if (SCALE || X3) __syncthreads();
    if constexpr (X3 != 0) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const uint2 e = *reinterpret_cast<const uint2*>(rtab + 2 * (t_row + it * RPP));
            a_voff[it] = e.x + (unsigned)src_c4 * (unsigned)sizeof(T);
            a_mask[it] = e.y;
        }
    }
    const int nsteps = dense_cpt ? (p.ntaps * dense_cpt + CH - 1) / CH : pair ? (p.ntaps + 1) / 2 : p.ntaps * (p.CP / BK);
    int ld_a = 0, ld_b = 0, ld_kc = 0;
    int cmp_kc = 0;  // K-chunk of the step being computed (for the scale lookup)
    // X3 weight DMA: block = wave + it * NWAVES covers rows (block % (BN/16)) * 16 + lane/4 of term block / (BN/16); this
    // lane's physical chunk lane%4 holds the logical chunk (lane%4) ^ ((lane/16)%4)  [(row>>2)&3 of a 16-aligned block]
    int b3_off[B3_IT];
    if constexpr (X3) {
#pragma unroll
        for (int it = 0; it < B3_IT; ++it) {
            const int blk = wave + it * NWAVES;
            const int term = blk / (BN / 16), r = (blk % (BN / 16)) * 16 + (lane >> 2);
            b3_off[it] = (term * p.RP + n0 + r) * p.CP + (((lane & 3) ^ ((lane >> 4) & 3)) << 3);
        }
    }
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    // X3: raw buffer descriptors (stride 0, num_records = 2^32 - 1: validity is the lane's own out-of-range offset)
    const buf_rsrc_t rs_a = make_buf_rsrc(px + (BUF ? min_off : 0));
    const buf_rsrc_t rs_b = make_buf_rsrc(p.wp);
    // f16 weight tile: this lane's byte offset inside a tap's [RP][CP] slice (rows it * RPP further down go through the SGPR offset)
    const unsigned b_voff = b_lane_ok ? (unsigned)((n0 + t_row) * p.CP + src_c4) * (unsigned)sizeof(T) : 0xFFFFFFFFu;

    // pair mode: taps (ld_a, ld_b) and its successor, chosen per lane
    auto issue_pair = [&](int buf) __attribute__((always_inline)) {
        int a1 = ld_a, b1 = ld_b + 1;
        if (b1 == p.nkx) { b1 = 0; ++a1; }
        const int ta = pair_hi ? a1 : ld_a, tb = pair_hi ? b1 : ld_b;
        const bool tap_ok = ta < p.nky && src_c4 < p.Ci;
        const int dy = p.dy0 + ta * p.ddy, dx = p.dx0 + tb * p.ddx;
        const int widx = (p.ky0 + ta * p.kstep) * p.KW + p.kx0 + tb * p.kstep;
        const int tap_off = (dy * p.Wi + dx) * p.Ci;
        T* Ad = As + buf * BM * LD + wave * RPW * LD;
        T* Bd = Bs + buf * BN * LD + wave * RPW * LD;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            bool ok = ((row_ok >> it) & 1u) && tap_ok && (unsigned)(a_iy0[it] + dy) < (unsigned)p.Hi &&
                      (unsigned)(a_ix0[it] + dx) < (unsigned)p.Wi;
            const T* g = ok ? px + (a_base[it] + tap_off) : pzero;
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Ad + it * RPP * LD), 16, 0, 0);
        }
        const T* wt = pw + ((size_t)widx * p.RP + n0 + t_row) * p.CP + src_c4;
        if (BN % RPP == 0 || wave * RPW < BN) {
#pragma unroll
            for (int it = 0; it < B_IT; ++it)
                __builtin_amdgcn_global_load_lds((gptr_t)((b_lane_ok && tap_ok) ? wt + (size_t)it * RPP * p.CP : pzero),
                                                 (lptr_t)(Bd + it * RPP * LD), 16, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (++ld_b == p.nkx) { ld_b = 0; ++ld_a; }
    };
    // (the dense order counts its K steps in ld_a, which the tap-grid order does not need there: with a counter of its own hipcc merged
    // "++ld_step" and "++ld_a" into ONE store through a selected address, which kept both on the stack — a scratch load + s_waitcnt
    // vmcnt(0) per stage and, the DMA's SGPR offsets no longer provably uniform, a waterfall loop around every buffer_load … lds)
    auto issue_dense = [&](int buf) __attribute__((always_inline)) {
        if constexpr (X3 != 0) {
            const int q = ld_a * CH + lchunk;
            const int t = (q * ((65536 + dense_cpt - 1) / (dense_cpt > 0 ? dense_cpt : 1))) >> 16;  // q / cpt (exact for q < 128)
            const int ch = (q - t * dense_cpt) * EPC;
            const int ta = (t * 11) >> 5, tb = t - 3 * ta;  // 3x3 tap grid: t / 3, t % 3 (t < 12)
            const int dy = p.dy0 + ta * p.ddy, dx = p.dx0 + tb * p.ddx;
            const unsigned tap_bytes = (unsigned)((dy * p.Wi + dx) * p.Ci + ch - min_off) * 4u;  // this LANE's tap (garbage past the last tap: masked)
            T* Ad = As + buf * BM * LD + wave * RPW * LD;
#ifdef GIF_KXSHARE_PROBE  // (timing probe, results wrong: activation pieces on every third K step only)
            if (ld_a % 3 == 0)
#endif
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const unsigned voff = (a_voff[it] + tap_bytes) | (__builtin_amdgcn_ubfe(a_mask[it], (unsigned)t, 1u) - 1u);
                buf_load_lds16(rs_a, (lptr_t)(Ad + it * RPP * LD), voff, 0);
            }
            const int so_b = ld_a * NPL * p.RP * p.CP * 2;
#pragma unroll
            for (int it = 0; it < B3_IT; ++it) {
                const int blk = wave + it * NWAVES;  // wave-uniform
                if (B3_BLK % NWAVES == 0 || blk < B3_BLK)
                    buf_load_lds16(rs_b, (lptr_t)(B3 + (buf * B3_BLK + blk) * 512), (unsigned)(b3_off[it] * 2), so_b);
            }
            ++ld_a;
        }
    };
    auto issue = [&](int buf) __attribute__((always_inline)) {
        if constexpr (F16 && !X3) {
            if (pair) { issue_pair(buf); return; }
        }
        if constexpr (X3 != 0) {
            if (dense_cpt) { issue_dense(buf); return; }
        }
        const int dy = p.dy0 + ld_a * p.ddy, dx = p.dx0 + ld_b * p.ddx;
        const int widx = (p.ky0 + ld_a * p.kstep) * p.KW + p.kx0 + ld_b * p.kstep;
        const int kc = ld_kc;
        const int tap_off = (dy * p.Wi + dx) * p.Ci + kc;
        const bool ch_ok = kc + src_c4 < p.Ci;
        T* Ad = As + buf * BM * LD + wave * RPW * LD;  // wave-uniform; lane l lands at +l*16 bytes
        T* Bd = Bs + buf * BN * LD + wave * RPW * LD;
        if constexpr (X3) {
            const unsigned t_cur = (unsigned)(ld_a * p.nkx + ld_b);            // wave-uniform
            const int so_a = (tap_off - min_off) * 4;                            // >= 0, wave-uniform
            const unsigned ch_or = ch_ok ? 0u : 0xFFFFFFFFu;
#ifdef GIF_KXSHARE_PROBE  // timing probe (tools/probes/kxshare_probe.sh; results are WRONG): the activation tile is staged for the first
            // tap of a kernel row only — what staging a row + halo ONCE for its three kx taps would issue
            if (ld_b == 0)
#endif
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const unsigned voff = a_voff[it] | (__builtin_amdgcn_ubfe(a_mask[it], t_cur, 1u) - 1u) | ch_or;
                buf_load_lds16(rs_a, (lptr_t)(Ad + it * RPP * LD), voff, so_a);
            }
            const int so_b = (widx * NPL * p.RP * p.CP + kc) * 2;
#pragma unroll
            for (int it = 0; it < B3_IT; ++it) {
                const int blk = wave + it * NWAVES;  // wave-uniform
                if (B3_BLK % NWAVES == 0 || blk < B3_BLK)
                    buf_load_lds16(rs_b, (lptr_t)(B3 + (buf * B3_BLK + blk) * 512), (unsigned)(b3_off[it] * 2), so_b);
            }
        } else if constexpr (F16) {
            const unsigned t_cur = (unsigned)(ld_a * p.nkx + ld_b);
            const int so_a = (tap_off - min_off) * (int)sizeof(T);
            const unsigned ch_or = ch_ok ? 0u : 0xFFFFFFFFu;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const unsigned voff = a_voff[it] | (__builtin_amdgcn_ubfe(a_mask[it], t_cur, 1u) - 1u) | ch_or;
                buf_load_lds16(rs_a, (lptr_t)(Ad + it * RPP * LD), voff, so_a);
            }
            if (BN % RPP == 0 || wave * RPW < BN) {  // wave-uniform: waves beyond the B tile issue nothing
                const int so_b = (widx * p.RP * p.CP + kc) * (int)sizeof(T);
#pragma unroll
                for (int it = 0; it < B_IT; ++it)
                    buf_load_lds16(rs_b, (lptr_t)(Bd + it * RPP * LD), b_voff, so_b + it * RPP * p.CP * (int)sizeof(T));
            }
        } else {
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                bool ok = ((row_ok >> it) & 1u) && ch_ok && (unsigned)(a_iy0[it] + dy) < (unsigned)p.Hi &&
                          (unsigned)(a_ix0[it] + dx) < (unsigned)p.Wi;
                const T* g = ok ? px + (a_base[it] + tap_off) : pzero;
                __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Ad + it * RPP * LD), 16, 0, 0);
            }
            const T* wt = pw + ((size_t)widx * p.RP + n0 + t_row) * p.CP + kc + src_c4;
            if (BN % RPP == 0 || wave * RPW < BN) {  // wave-uniform: waves beyond the B tile issue nothing
#pragma unroll
                for (int it = 0; it < B_IT; ++it)
                    __builtin_amdgcn_global_load_lds((gptr_t)(b_lane_ok ? wt + (size_t)it * RPP * p.CP : pzero),
                                                     (lptr_t)(Bd + it * RPP * LD), 16, 0, 0);
            }
        }
        ld_kc += BK;
        if (ld_kc >= p.CP) {
            ld_kc = 0;
            if (++ld_b == p.nkx) { ld_b = 0; ++ld_a; }
        }
    };

    // GIF_DMA_SPREAD (probe build, tools/probes/dma_spread_probe.sh): the stage's DMA pieces of the bf16x3 / f16x2 tap-grid order are issued
    // ONE AT A TIME between the MFMAs of the step's first k-group instead of back to back ahead of it (the microarchitecture guide prices a
    // piece at ~60 cycles among bare MFMAs and 100-185 inside a phase that already carries pieces and operand reads).  sp_begin fixes the
    // stage's wave-uniform offsets, sp_piece(k) issues piece k (A pieces first), sp_end advances the K counters.
    int sp_buf = 0, sp_so_a = 0, sp_so_b = 0;
    unsigned sp_t = 0, sp_chor = 0;
    bool sp_on = false;  // wave-uniform: this step has a stage to fetch
    [[maybe_unused]] auto sp_begin = [&](int buf) __attribute__((always_inline)) {
        const int dy = p.dy0 + ld_a * p.ddy, dx = p.dx0 + ld_b * p.ddx;
        const int widx = (p.ky0 + ld_a * p.kstep) * p.KW + p.kx0 + ld_b * p.kstep;
        sp_buf = buf;
        sp_t = (unsigned)(ld_a * p.nkx + ld_b);
        sp_so_a = ((dy * p.Wi + dx) * p.Ci + ld_kc - min_off) * 4;
        sp_chor = (ld_kc + src_c4 < p.Ci) ? 0u : 0xFFFFFFFFu;
        sp_so_b = (widx * NPL * p.RP * p.CP + ld_kc) * 2;
        ld_kc += BK;
        if (ld_kc >= p.CP) {
            ld_kc = 0;
            if (++ld_b == p.nkx) { ld_b = 0; ++ld_a; }
        }
    };
    [[maybe_unused]] auto sp_piece = [&](int k) __attribute__((always_inline)) {
        if constexpr (X3 != 0) {
            if (k < A_IT) {
                T* Ad = As + sp_buf * BM * LD + wave * RPW * LD;
                const unsigned voff = a_voff[k] | (__builtin_amdgcn_ubfe(a_mask[k], sp_t, 1u) - 1u) | sp_chor;
                buf_load_lds16(rs_a, (lptr_t)(Ad + k * RPP * LD), voff, sp_so_a);
            } else {
                const int it = k - A_IT;
                const int blk = wave + it * NWAVES;  // wave-uniform
                if (B3_BLK % NWAVES == 0 || blk < B3_BLK)
                    buf_load_lds16(rs_b, (lptr_t)(B3 + (sp_buf * B3_BLK + blk) * 512), (unsigned)(b3_off[it] * 2), sp_so_b);
            }
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fsw = (li / RB) % CH;  // f(row) of every row this lane reads (tile bases are multiples of 32)
    // operand fragments of one k-group: register double buffer, read one group ahead of its MFMAs (16 bytes per fragment:
    // 4 floats or 8 halfs)
    typedef typename std::conditional<F16, gif::f16x8_t, f32x4>::type frag_t;
    auto next_kc = [&](int kc) { return (kc + BK >= p.CP) ? 0 : kc + BK; };
    int cur = 0;
    if constexpr (X3) {
        // ---- bf16x3 / f16x2 schedule.  A k-group is 16 k-values: two 16-byte fragments per 32-row tile and lane.  Software
        // pipeline, per group: [LDS reads of the NEXT group's fp32 fragments] then the NPROD*MT*NT MFMAs of THIS group's
        // split operands with the split of the next group's fragments interleaved between them — the matrix pipe and the VALU
        // are separate, and the in-order wave only overlaps them when the instruction stream alternates (sched_barrier pins it).
        // bf16x3: 6 products, 11 VALU per pair of floats.  f16x2 (H2): 3 products, 6 VALU per pair + the running row scale:
        // per group and tile the maximum |a| of the lane's 8 floats (both lane halves exchange theirs: they feed the same row),
        // compared with the largest magnitude the row's current exponent can hold; in the rare case that a row outgrows it the
        // new exponent takes effect for the fragments split from here on and the row's accumulators are multiplied by the exact
        // power of two once the MFMAs of the group in flight have issued (`rescale`).
        gif::u32x4_t sa[2][NPL][MT], sb[2][NPL][NT];  // [slot][term][tile]: 8 packed bf16 / f16 each
        f32x4 ra[MT][2], rs[MT][2];                  // raw A fragments (and modulation scales) of the group being split
        const int b3_sw = (li >> 2) & 3;             // (row >> 2) & 3 of every weight row this lane reads
        // f16x2 row state (lane li and li + 32 hold the same values): exponent, its float 2^e, the largest |a| it can take, the
        // pending exponent change, and the guard's statistics (row maximum, smallest non-zero group maximum as bits - 1)
        int h_ex[MT], h_dl[MT];
        float h_sc[MT], h_lim[MT], h_max[MT];
        unsigned h_gmin[MT];
        bool h_need = false;  // wave-uniform: some row of this wave changed its exponent in the last `track`
        if constexpr (H2) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                h_ex[i] = 126; h_dl[i] = 0;
                h_sc[i] = gif::h2_pow2(126); h_lim[i] = gif::kH2Limit * gif::h2_pow2(-126);
                h_max[i] = 0.f; h_gmin[i] = 0xFFFFFFFFu;
            }
        }
        // LDS reads of group q of stage `buf`: raw fp32 A fragments (split later, piecewise) and the pre-split B operands
        auto read_raw = [&](int buf, int q, int kc, int slot) __attribute__((always_inline)) {
            const T* Ab = As + buf * BM * LD + (wm0 + li) * LD;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                // lane half h supplies k = 16 q + 8 h + 0..7 of the group (the MFMA's own operand layout, which is also the
                // order of the pre-split weights): fp32 chunks 4 q + 2 h and 4 q + 2 h + 1 of the 128-byte row
                const int lc = q * 4 + lh * 2 + u;
                const int c = (lc ^ fsw) * EPC;
#pragma unroll
                for (int i = 0; i < MT; ++i) ra[i][u] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LD + c);
                if (SCALE) {
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        rs[i][u] = *reinterpret_cast<const f32x4*>(Stab + s_row[i] - lh * EPC + kc + lc * EPC);
                }
            }
            const unsigned short* Bb = B3 + buf * NPL * BN * 32 + (wn0 + li) * 32 + (((q * 2 + lh) ^ b3_sw) << 3);
#pragma unroll
            for (int t = 0; t < NPL; ++t)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    sb[slot][t][j] = *reinterpret_cast<const gif::u32x4_t*>(Bb + (t * BN + j * 32) * 32);
        };
        // f16x2: modulation multiply, group maximum, exponent decision of the raw fragments just read
        auto track = [&]() __attribute__((always_inline)) {
            if constexpr (H2) {
                h_need = false;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    if (SCALE) { ra[i][0] *= rs[i][0]; ra[i][1] *= rs[i][1]; }
                    float m = fmaxf(fmaxf(fabsf(ra[i][0][0]), fabsf(ra[i][0][1])), fabsf(ra[i][0][2]));
                    m = fmaxf(fmaxf(m, fabsf(ra[i][0][3])), fabsf(ra[i][1][0]));
                    m = fmaxf(fmaxf(m, fabsf(ra[i][1][1])), fabsf(ra[i][1][2]));
                    m = fmaxf(m, fabsf(ra[i][1][3]));
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                    m = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));  // the row's 16 k-values of this group
                    h_max[i] = fmaxf(h_max[i], m);
                    h_gmin[i] = min(h_gmin[i], __float_as_uint(m) - 1u);          // an all-zero group (0 - 1 = 0xffffffff) never wins
                    h_dl[i] = 0;
                    if (__builtin_amdgcn_ballot_w64(m > h_lim[i]) != 0) {          // wave-uniform (scalar branch), rare after a row's first groups
                        // per-lane update by selects: no EXEC manipulation anywhere near the MFMA stream
                        const int ne = m > h_lim[i] ? gif::h2_exp_for(__float_as_uint(m), gif::kH2Target) : h_ex[i];
                        h_dl[i] = ne - h_ex[i];
                        h_ex[i] = ne;
                        h_sc[i] = gif::h2_pow2(ne);
                        h_lim[i] = ldexpf(gif::kH2Limit, -ne);
                        h_need = true;
                    }
                }
            }
        };
        // f16x2: rows whose exponent changed: acc *= 2^(e' - e), exact.  Accumulator register r of a 32x32 tile holds row
        // (r & 3) + 8 (r >> 2) + 4 lh, whose exponent change lives in lane `row`.
        auto rescale = [&]() __attribute__((always_inline)) {
            if constexpr (H2) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int d = __builtin_amdgcn_ds_bpermute(((r & 3) + 8 * (r >> 2) + 4 * lh) * 4, h_dl[i]);
#pragma unroll
                        for (int j = 0; j < NT; ++j) acc[i][j][r] = ldexpf(acc[i][j][r], d);
                    }
            }
        };
        // piece k of the split of one group: one pair of floats -> one packed dword of each term (bf16x3: 11 VALU, +2 modulated;
        // f16x2: 6)
        constexpr int NP = MT * 4;
        auto split_piece = [&](int slot, int k) __attribute__((always_inline)) {
            const int f = k / 4, e = k % 4;  // A tile, dword of the packed operand
            float a0 = ra[f][e / 2][(e % 2) * 2], a1 = ra[f][e / 2][(e % 2) * 2 + 1];
            if constexpr (H2) {
                unsigned h, l;
                // (two row tiles per lane: hipcc keeps the two scales in one register pair and selects the odd one with op_sel:[0,1] on
                // v_pk_mul_f32 / v_pk_fma_f32 — that form returned wrong products in the weight-gradient kernel whenever two waves
                // shared a SIMD, conv_wgrad.hip; the scalar form is two instructions longer)
                if constexpr (MT > 1) gif::split_pair_h2_scalar(a0, a1, h_sc[f], h, l);
                else gif::split_pair_h2(a0, a1, h_sc[f], h, l);
                sa[slot][0][f][e] = h; sa[slot][1][f][e] = l;
            } else {
                if (SCALE) { a0 *= rs[f][e / 2][(e % 2) * 2]; a1 *= rs[f][e / 2][(e % 2) * 2 + 1]; }
                unsigned h, m, l;
                gif::split_pair_scalar(a0, a1, h, m, l);
                sa[slot][0][f][e] = h; sa[slot][1][f][e] = m; sa[slot][2][f][e] = l;
            }
        };
        // The NPROD*MT*NT MFMAs of the group in `slot` (smallest terms first; bf16x3: lo*mid, mid*lo, lo*lo <= 2^-23 of the
        // product are not formed, f16x2: lo*lo <= 2^-22; term-major: consecutive MFMAs write different accumulators), with the
        // preparation of the next group (-> slot `nslot`, or nothing if nslot < 0) placed piecewise between them.
        // sched_barrier(0) after every MFMA and every piece: the compiler keeps exactly this interleave (sched_group_barrier's
        // VALU class also matches MFMAs).
        constexpr int NPROD = H2 ? 3 : 6;
        constexpr int NPROD_ = NPROD;
        constexpr int LEAD = H2 ? 4 : 6;  // MFMAs ahead of the first piece: they cover the LDS latency of the raw reads
        constexpr int SP_N = A_IT + B3_IT;                     // DMA pieces per stage and wave
        [[maybe_unused]] constexpr int SP_EVERY = (NPROD_ * MT * NT) / SP_N > 0 ? (NPROD_ * MT * NT) / SP_N : 1;  // one piece every SP_EVERY MFMAs
        auto group = [&](int slot, int nslot, auto dma_tag) __attribute__((always_inline)) {
            [[maybe_unused]] constexpr bool dma = decltype(dma_tag)::value;
            constexpr int TA6[6] = {2, 0, 1, 1, 0, 0}, TB6[6] = {0, 2, 1, 0, 1, 0};
            constexpr int TA3[3] = {1, 0, 0}, TB3[3] = {0, 1, 0};
            int n = 0, piece = H2 ? -1 : 0;  // piece -1: the tracking step of f16x2
            [[maybe_unused]] int spk = 0;
#pragma unroll
            for (int t = (H2 ? 0 : GIF_X3_FIRST_TERM); t < NPROD; ++t)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        if constexpr (H2)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gif::f16x8_t, sa[slot][TA3[t]][i]),
                                                                               __builtin_bit_cast(gif::f16x8_t, sb[slot][TB3[t]][j]),
                                                                               acc[i][j], 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gif::bf16x8_t, sa[slot][TA6[t]][i]),
                                                                                __builtin_bit_cast(gif::bf16x8_t, sb[slot][TB6[t]][j]),
                                                                                acc[i][j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        ++n;
#ifdef GIF_DMA_SPREAD
                        if (dma && spk < SP_N && n == 1 + spk * SP_EVERY) {
                            if (sp_on) sp_piece(spk);
                            ++spk;
                            __builtin_amdgcn_sched_barrier(0);
                        }
#endif
                        if (nslot >= 0 && n >= LEAD && piece < NP) {
                            if (piece < 0) track();
                            else split_piece(nslot, piece);
                            ++piece;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
#ifdef GIF_DMA_SPREAD
            if (dma) {
#pragma unroll
                for (; spk < SP_N; ++spk)
                    if (sp_on) sp_piece(spk);
            }
#endif
            if (nslot >= 0) {
#pragma unroll
                for (; piece < NP; ++piece) {
                    if (piece < 0) track();
                    else split_piece(nslot, piece);
                }
                if constexpr (H2) {
                    if (h_need) rescale();
                }
            }
        };
        static_assert(KG == 2, "bf16x3 / f16x2: 32 floats per K chunk");
        // DMA instructions per stage and wave (NST == 3: the newest stage stays in flight across the mid-stage sync)
        constexpr int DMA_PER_STAGE = A_IT + B3_IT;
        static_assert(NST == 2 || B3_BLK % NWAVES == 0, "three-stage ring: every wave issues the same number of pieces");
#ifdef GIF_X3_TIMING_PROBE
        const long long probe_i0 = clock64();
#endif
        issue(0);
        if constexpr (NST == 3) {
            if (nsteps > 1) {
                issue(1);
#ifndef GIF_NOFILLWAIT_PROBE  // timing probe (tools/probes/epilogue_probe.sh; results are WRONG): the first stage is read before it has landed — what
                // a first stage prefetched under the PREVIOUS tile's epilogue (a persistent workgroup) would save
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_STAGE) : "memory");
#endif
            }
        }
#ifdef GIF_NOFILLWAIT_PROBE
        if (NST == 3 && nsteps > 1) __builtin_amdgcn_s_barrier();
        else
#endif
        __syncthreads();
#ifdef GIF_X3_TIMING_PROBE
        if (lane == 0) {
            atomicAdd(&g_x3_probe[8], (unsigned long long)(probe_i0 - probe_entry));
            atomicAdd(&g_x3_probe[9], (unsigned long long)(clock64() - probe_i0));
        }
#endif
        read_raw(0, 0, 0, 0);
        track();  // (f16x2: first exponents; the accumulators are still zero)
#pragma unroll
        for (int k = 0; k < NP; ++k) split_piece(0, k);
#ifdef GIF_X3_TIMING_PROBE
        long long probe_sync = 0, probe_wait = 0;
        const long long probe_t0 = clock64();
#endif
        for (int step = 0; step + 1 < nsteps; ++step) {
#ifdef GIF_NO_DMA_PROBE  // timing probe (tools/probes/no_dma_probe.sh): the K loop without its LDS-DMA issue — results are WRONG
            if (step < 1) issue(cur ^ 1);
#else
#ifdef GIF_DMA_SPREAD
            constexpr bool spread = true;  // (probe build: tap-grid launches only; the tap-dense order is NOT handled)
#else
            constexpr bool spread = false;
#endif
            if constexpr (NST == 3) {
                // stage step + 2 into the buffer of stage step - 1 (its last operand read preceded the previous mid-stage barrier)
                sp_on = step + 2 < nsteps;
                if (sp_on) {
                    if (spread) sp_begin(cur == 0 ? 2 : cur - 1);
                    else issue(cur == 0 ? 2 : cur - 1);
                }
            } else {
                sp_on = true;
                if (spread) sp_begin(cur ^ 1);
                else issue(cur ^ 1);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g + 1 < KG; ++g) {
                read_raw(cur, g + 1, cmp_kc, (g + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (spread) {
                    if (g == 0) group(g & 1, (g + 1) & 1, std::true_type{});
                    else group(g & 1, (g + 1) & 1, std::false_type{});
                } else {
                    group(g & 1, (g + 1) & 1, std::false_type{});
                }
            }
#ifdef GIF_X3_TIMING_PROBE
            {   // time parked at the mid-stage sync, split into the wave's own DMA wait and the barrier
                const long long ta = clock64();
                if (NST == 3 && step + 2 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_STAGE) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const long long tb = clock64();
                if constexpr (NST == 3) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                } else {
                    __syncthreads();
                }
                probe_wait += tb - ta;
                probe_sync += clock64() - tb;
            }
#else
            if constexpr (NST == 3) {
                // stage step + 1 has landed (this wave's pieces; behind the barrier everybody's), stage step + 2 may stay in flight
                if (step + 2 < nsteps) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(DMA_PER_STAGE) : "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            } else {
                __syncthreads();
            }
#endif
            cmp_kc = next_kc(cmp_kc);
            if constexpr (NST == 3) cur = cur == 2 ? 0 : cur + 1;
            else cur ^= 1;
            read_raw(cur, 0, cmp_kc, 0);
            __builtin_amdgcn_sched_barrier(0);
            group((KG - 1) & 1, 0, std::false_type{});
        }
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            if (g + 1 < KG) {
                read_raw(cur, g + 1, cmp_kc, (g + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            group(g & 1, g + 1 < KG ? (g + 1) & 1 : -1, std::false_type{});
        }
#ifdef GIF_X3_TIMING_PROBE
        if (lane == 0) {
            atomicAdd(&g_x3_probe[0], (unsigned long long)probe_wait);
            atomicAdd(&g_x3_probe[1], (unsigned long long)probe_sync);
            atomicAdd(&g_x3_probe[2], (unsigned long long)(clock64() - probe_t0));
            atomicAdd(&g_x3_probe[3], 1ull);
            atomicAdd(&g_x3_probe[4], (unsigned long long)(probe_t0 - probe_entry));
        }
        probe_loop_end = clock64();
#endif
        if constexpr (H2) {
            // guard: a row one of whose 16-element K groups lies more than 2^kH2Window below the row maximum (the group's values no
            // longer carry 22 bits) ...
            bool wide = false;
#pragma unroll
            for (int i = 0; i < MT; ++i)
                wide |= (int)(__float_as_uint(h_max[i]) >> 23) - (int)((h_gmin[i] + 1u) >> 23) > gif::kH2Window;
            if (p.gate) {
                // ... AND one of the weight rows it meets was flagged by the packing: a narrow group only costs accuracy where the OTHER
                // operand is narrow too (common.h: the error floor is 2^(m - 38) of the dot product's largest group product, m = the
                // smallest sum of the two operands' group spreads; one in-window operand bounds m by its window)
                bool wflag = false;
#pragma unroll
                for (int j = 0; j < NT; ++j) wflag |= p.wexp[p.RP + n0 + wn0 + j * 32 + li] != 0;
                if (__builtin_amdgcn_ballot_w64(wide) != 0 && __builtin_amdgcn_ballot_w64(wflag) != 0 && lane == 0) atomicMax(p.gate, p.gate_gen);
            }
            // back to the operands' own scale: acc[row][col] *= 2^-(e_row + e_col)
            int wex[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) wex[j] = p.wexp[n0 + wn0 + j * 32 + li];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int er = __builtin_amdgcn_ds_bpermute(((r & 3) + 8 * (r >> 2) + 4 * lh) * 4, h_ex[i]);
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j][r] = ldexpf(acc[i][j][r], -(er + wex[j]));
                }
        }
    } else {
    frag_t av[2][MT], bv[2][NT];
    auto frag_read = [&](int buf, int kk, int slot, int kc) __attribute__((always_inline)) {
        const T* Ab = As + buf * BM * LD + (wm0 + li) * LD;
        const T* Bb = Bs + buf * BN * LD + (wn0 + li) * LD;
        const int c = ((kk * 2 + lh) ^ fsw) * EPC;
#pragma unroll
        for (int i = 0; i < MT; ++i) av[slot][i] = *reinterpret_cast<const frag_t*>(Ab + i * 32 * LD + c);
#pragma unroll
        for (int j = 0; j < NT; ++j) bv[slot][j] = *reinterpret_cast<const frag_t*>(Bb + j * 32 * LD + c);
        if (SCALE) {
#pragma unroll
            for (int i = 0; i < MT; ++i)  // pair mode: the chunk's upper half holds the second tap's channels 0..31 again
                av[slot][i] *= *reinterpret_cast<const frag_t*>(Stab + s_row[i] + (pair ? (kk * 2 * EPC) & (BK / 2 - 1) : kc + kk * 2 * EPC));
        }
    };
    auto mfma_group = [&](int slot) __attribute__((always_inline)) {
        if constexpr (F16) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[slot][i], bv[slot][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, MT + NT + (SCALE ? MT : 0), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[slot][i][t], bv[slot][j][t], acc[i][j], 0, 0, 0);
            // pin: the LDS reads of the NEXT group (issued just before this call) go out ahead of these MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, MT + NT + (SCALE ? MT : 0), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * MT * NT, 0);
        }
    };

    // Schedule of one step (KG k-groups g0..g{KG-1} of the current buffer):
    //   DMA(step+1) | g0 | ... | g{KG-2} | vmcnt(0)+barrier | read g0 of step+1 | g{KG-1}
    // The last group's fragments are in registers before the barrier, so every wave is done with the current buffer
    // when it arrives, and the first operand read of the next step runs under the MFMAs of the last group.
    issue(0);
    __syncthreads();  // the workgroup release waits for the outstanding LDS-DMA (vmcnt(0)) of every wave
    frag_read(0, 0, 0, 0);
    for (int step = 0; step + 1 < nsteps; ++step) {
        issue(cur ^ 1);  // DMA of step+1 runs under the MFMAs of this step
#pragma unroll
        for (int g = 0; g + 1 < KG; ++g) {
            frag_read(cur, g + 1, (g + 1) & 1, cmp_kc);
            mfma_group(g & 1);
        }
        __syncthreads();
        cmp_kc = next_kc(cmp_kc);
        cur ^= 1;
        frag_read(cur, 0, KG & 1, cmp_kc);
        mfma_group((KG - 1) & 1);
        if (KG & 1) {  // odd group count: realign the register slots (KG is even here, kept for safety)
#pragma unroll
            for (int i = 0; i < MT; ++i) av[0][i] = av[1][i];
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[0][j] = bv[1][j];
        }
    }
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        if (g + 1 < KG) frag_read(cur, g + 1, (g + 1) & 1, cmp_kc);
        mfma_group(g & 1);
    }
    }
#ifdef GIF_X3_TIMING_PROBE
    if (X3 && lane == 0) atomicAdd(&g_x3_probe[6], (unsigned long long)(clock64() - probe_loop_end));
#endif
    conv_epilogue<BM, BN, 32, MT, NT, T, THREADS>(p, acc, smem, m0, n0, wm0, wn0, tid, li, lh, HWp);
#ifdef GIF_X3_TIMING_PROBE
    if (X3 && lane == 0) atomicAdd(&g_x3_probe[5], (unsigned long long)(clock64() - probe_loop_end));
#endif
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, bool SCALE, int BK, int X3 = 0, int NST = 2>
// (f16x2 on 4 waves: min. 2 workgroups per CU = a 256-register budget, so that the accumulators stay in architectural VGPRs for
// the VALU rescale path — see conv_wgrad.hip)
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N, (X3 == 2 && WAVES_M * WAVES_N == 4) ? 2 : 1) conv_gather_mfma_glds(const GatherParams p) {
    glds_body<T, BM, BN, WAVES_M, WAVES_N, SCALE, BK, X3, NST>(p, (int)blockIdx.x, (int)gridDim.x);
}

// Up to 4 independent problems (the output-parity phases of a small transposed convolution) in ONE launch: each phase
// alone would fill a fraction of the chip (e.g. 104 workgroups), together they run side by side.
struct MultiParams {
    GatherParams ph[4];
    int wg_end[4];  // exclusive prefix sums of the phases' workgroup counts
    int nph;
};

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, bool SCALE, int BK, int X3 = 0, int NST = 2>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N, (X3 == 2 && WAVES_M * WAVES_N == 4) ? 2 : 1) conv_gather_mfma_glds_multi(const MultiParams mp) {
    const int b = (int)blockIdx.x;
    int k = 0;
    while (k + 1 < mp.nph && b >= mp.wg_end[k]) ++k;  // wave-uniform
    const int begin = k ? mp.wg_end[k - 1] : 0;
    glds_body<T, BM, BN, WAVES_M, WAVES_N, SCALE, BK, X3, NST>(mp.ph[k], b - begin, mp.wg_end[k] - begin);
}

// ------------------------------------------------------------------------------------------------------------------
// f16x2 "row" kernel for stride-1 3x3 layers with <= 32 output channels (round 6: the C -> 24 data gradients of the condition-noise convs,
// stylegan2_common_layers.py:405-414 backwards).  On a 256 x 32 tile the gather kernel stages 32 KB of activations per tap and K chunk for
// 0.5 MFLOP — 14.5 FLOP per staged byte, 82-85 TFLOP/s: it sits on its LDS-DMA (profiles/r6_kxshare_probe.txt: -33 % with a third of the
// activation pieces).  Here the M tile is 256 consecutive output pixels = whole image rows (or a 256-pixel piece of one): for one kernel
// ROW dy and one 32-channel K chunk the input pixels of those rows PLUS one halo pixel on either side are staged ONCE ([288][32] fp32 by
// LDS-DMA, same unpadded 128-byte rows and (row >> 1) & 7 chunk swizzle as glds_body: out-of-image pixels are out-of-range buffer offsets
// = zeros), and the three dx taps are three operand reads of the same buffer shifted by one row each — any 32 consecutive rows stay
// conflict free under that swizzle (16 lanes of a ds_read_b128 group = 16 distinct row indices mod 16).  A stage therefore carries 36 MFMAs
// per wave (3 taps x 2 k-groups x 2 row tiles x 3 products) behind ONE barrier where the gather kernel has 12.  The weights of a stage (3
// taps x 32 rows x 32 k, two f16 planes: 12 KB shared by all tiles) do not go through LDS at all: every lane keeps its operand fragments
// in registers, loaded one stage ahead straight from L2.  4 waves x (64 pixels x 32 channels), two workgroups per CU (2 x 72 KB of LDS).
// Running row exponents, accumulator rescale, guard and epilogue are glds_body's (the guarded twin is the gather kernel in bf16x3 on the
// same 256-row tiles, so fused column sums / dot products land in the same partial rows).
// ------------------------------------------------------------------------------------------------------------------
constexpr int kRowsThinA = 288;  // staged rows per stage: 256 + 2 halo pixels per image row of the tile (<= 8 rows), in whole 32-row passes

__global__ void __launch_bounds__(256, 2) conv3x3_rows_thin_h2(const GatherParams p) {
    constexpr int BM = 256, BN = 32, LD = 32, THREADS = 256, MT = 2, NT = 1, AROWS = kRowsThinA, A_IT = AROWS / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;  // [2][AROWS][LD]
    typedef __attribute__((address_space(3))) void* lptr_t;
    const float* const px = static_cast<const float*>(p.x);
    const unsigned short* const pw = static_cast<const unsigned short*>(p.wp);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm0 = wave * 64;
    const int tile = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int m0 = tile * BM;
    const int HWp = p.Hp * p.Wp;
    // the tile's image rows: `segw` output pixels each, staged as `seg` buffer rows (one halo pixel on either side)
    const int segw = p.Wp >= BM ? BM : p.Wp, seg = segw + 2, nseg = BM / segw;

    // ---- activation DMA: buffer row r = tid / 8 + 32 * it holds input pixel (b, oy + dy, ox0 + j - 1) of its segment, logical 16-byte chunk
    // (tid % 8) ^ f(r); the byte offset for dy = 0 is loop invariant, dy and the K chunk travel in the SGPR offset (base moved one image
    // row up so that it stays >= 0), one validity bit per dy
    unsigned a_voff[A_IT], a_mask[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int r = (tid >> 3) + 32 * it;
        const int sg = r / seg, j = r - sg * seg;
        const int ms = m0 + sg * segw;
        const bool in_tile = sg < nseg && ms < p.M;
        const int mm = in_tile ? ms : 0;
        const int b = mm / HWp, rr = mm - b * HWp;
        const int oy = rr / p.Wp, ox0 = rr - oy * p.Wp;
        const int ix = ox0 + j - 1;
        const bool okx = in_tile && (unsigned)ix < (unsigned)p.Wi;
        const int c4 = ((tid & 7) ^ ((r >> 1) & 7)) * 4;
        a_voff[it] = (unsigned)(((b * p.Hi + oy) * p.Wi + ix) * p.Ci + c4) * 4u;
        unsigned mk = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) mk |= ((okx && (unsigned)(oy + d - 1) < (unsigned)p.Hi) ? 1u : 0u) << d;
        a_mask[it] = mk;
    }
    const buf_rsrc_t rs_a = make_buf_rsrc(px - (size_t)p.Wi * p.Ci);

    const int kch = p.CP / 32, nst = 3 * kch;
    auto issue_a = [&](int buf, int st) __attribute__((always_inline)) {
        const int ta = st / kch, kc = (st - ta * kch) * 32;
        const int dy = p.dy0 + ta * p.ddy;  // -1, 0, 1 in the launch's order
        const int so = ((dy + 1) * p.Wi * p.Ci + kc) * 4;
        float* Ad = As + buf * AROWS * LD + wave * 8 * LD;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int r = (tid >> 3) + 32 * it;
            const int c4 = ((tid & 7) ^ ((r >> 1) & 7)) * 4;
            const unsigned bad = (__builtin_amdgcn_ubfe(a_mask[it], (unsigned)(dy + 1), 1u) - 1u) | ((kc + c4 < p.Ci) ? 0u : 0xFFFFFFFFu);
            buf_load_lds16(rs_a, (lptr_t)(Ad + it * 32 * LD), a_voff[it] | bad, so);
        }
    };
    // ---- weights: fragments of (tap column tb, k-group q, term t) of the stage, straight from the f16x2 packing [tap][2][RP][CP]
    gif::u32x4_t sb[2][3][2][2];
    auto issue_b = [&](auto set_tag, int st) __attribute__((always_inline)) {
        constexpr int set = decltype(set_tag)::value;
        const int ta = st / kch, kc = (st - ta * kch) * 32;
#pragma unroll
        for (int tb = 0; tb < 3; ++tb) {
            const int widx = (p.ky0 + ta * p.kstep) * p.KW + p.kx0 + tb * p.kstep;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    sb[set][tb][q][t] = *reinterpret_cast<const gif::u32x4_t*>(pw + ((size_t)(widx * 2 + t) * p.RP + li) * p.CP + kc + q * 16 + lh * 8);
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
    // running row exponents (glds_body, f16x2): lanes li and li + 32 feed the same row and agree
    int h_ex[MT], h_dl[MT];
    float h_sc[MT], h_lim[MT], h_max[MT];
    unsigned h_gmin[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        h_ex[i] = 126; h_dl[i] = 0;
        h_sc[i] = gif::h2_pow2(126); h_lim[i] = gif::kH2Limit * gif::h2_pow2(-126);
        h_max[i] = 0.f; h_gmin[i] = 0xFFFFFFFFu;
    }
    // buffer row of this lane's output pixel in row tile i, before the tap shift: pixel index in the tile + 2 halo rows per image row passed
    int a_row[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) a_row[i] = wm0 + i * 32 + li + 2 * ((wm0 + i * 32) / segw);

    auto compute = [&](int buf, auto set_tag) __attribute__((always_inline)) {
        constexpr int set = decltype(set_tag)::value;
        const float* Ab = As + buf * AROWS * LD;
#pragma unroll
        for (int tb = 0; tb < 3; ++tb) {
            const int shift = p.dx0 + tb * p.ddx + 1;  // 0, 1, 2 in the launch's order (wave-uniform)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                f32x4 ra[MT][2];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int brow = a_row[i] + shift;
                    const int f = (brow >> 1) & 7;
#pragma unroll
                    for (int u = 0; u < 2; ++u) ra[i][u] = *reinterpret_cast<const f32x4*>(Ab + brow * LD + (((q * 4 + lh * 2 + u) ^ f) << 2));
                }
                gif::u32x4_t sa[2][MT];
                bool need = false;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    float m = fmaxf(fmaxf(fabsf(ra[i][0][0]), fabsf(ra[i][0][1])), fabsf(ra[i][0][2]));
                    m = fmaxf(fmaxf(m, fabsf(ra[i][0][3])), fabsf(ra[i][1][0]));
                    m = fmaxf(fmaxf(m, fabsf(ra[i][1][1])), fabsf(ra[i][1][2]));
                    m = fmaxf(m, fabsf(ra[i][1][3]));
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                    m = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
                    h_max[i] = fmaxf(h_max[i], m);
                    h_gmin[i] = min(h_gmin[i], __float_as_uint(m) - 1u);
                    h_dl[i] = 0;
                    if (__builtin_amdgcn_ballot_w64(m > h_lim[i]) != 0) {  // wave-uniform, rare after a row's first groups
                        const int ne = m > h_lim[i] ? gif::h2_exp_for(__float_as_uint(m), gif::kH2Target) : h_ex[i];
                        h_dl[i] = ne - h_ex[i];
                        h_ex[i] = ne;
                        h_sc[i] = gif::h2_pow2(ne);
                        h_lim[i] = ldexpf(gif::kH2Limit, -ne);
                        need = true;
                    }
                }
                if (need) {  // rows whose exponent changed: acc *= 2^(e' - e), exact (every earlier product has been accumulated)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int d = __builtin_amdgcn_ds_bpermute(((r & 3) + 8 * (r >> 2) + 4 * lh) * 4, h_dl[i]);
                            acc[i][0][r] = ldexpf(acc[i][0][r], d);
                        }
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        unsigned h, l;
                        gif::split_pair_h2_scalar(ra[i][e / 2][(e % 2) * 2], ra[i][e / 2][(e % 2) * 2 + 1], h_sc[i], h, l);
                        sa[0][i][e] = h; sa[1][i][e] = l;
                    }
                constexpr int TA3[3] = {1, 0, 0}, TB3[3] = {0, 1, 0};  // lo*hi, hi*lo, hi*hi
#pragma unroll
                for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gif::f16x8_t, sa[TA3[t3]][i]),
                                                                           __builtin_bit_cast(gif::f16x8_t, sb[set][tb][q][TB3[t3]]), acc[i][0], 0, 0, 0);
            }
        }
    };

    issue_a(0, 0);
    issue_b(std::integral_constant<int, 0>{}, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) as an instruction the compiler's own wait insertion sees (it tracks the weight loads)
    __syncthreads();
    for (int st = 0; st < nst; st += 2) {
        if (st + 1 < nst) {
            issue_a(1, st + 1);
            issue_b(std::integral_constant<int, 1>{}, st + 1);
        }
        compute(0, std::integral_constant<int, 0>{});
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) as an instruction the compiler's own wait insertion sees (it tracks the weight loads)
        __syncthreads();
        if (st + 1 < nst) {
            if (st + 2 < nst) {
                issue_a(0, st + 2);
                issue_b(std::integral_constant<int, 0>{}, st + 2);
            }
            compute(1, std::integral_constant<int, 1>{});
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) as an instruction the compiler's own wait insertion sees (it tracks the weight loads)
            __syncthreads();
        }
    }

    // guard (common.h): a narrow K group in one of this wave's rows meeting a flagged weight row raises the launch's gate
    {
        bool wide = false;
#pragma unroll
        for (int i = 0; i < MT; ++i) wide |= (int)(__float_as_uint(h_max[i]) >> 23) - (int)((h_gmin[i] + 1u) >> 23) > gif::kH2Window;
        if (p.gate) {
            const bool wflag = p.wexp[p.RP + li] != 0;
            if (__builtin_amdgcn_ballot_w64(wide) != 0 && __builtin_amdgcn_ballot_w64(wflag) != 0 && lane == 0) atomicMax(p.gate, p.gate_gen);
        }
        const int wex = p.wexp[li];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int er = __builtin_amdgcn_ds_bpermute(((r & 3) + 8 * (r >> 2) + 4 * lh) * 4, h_ex[i]);
                acc[i][0][r] = ldexpf(acc[i][0][r], -(er + wex));
            }
    }
    conv_epilogue<BM, BN, LD, MT, NT, float, THREADS, false, 2 * AROWS * LD>(p, acc, smem, m0, 0, wm0, 0, tid, li, lh, HWp);
}

// ------------------------------------------------------------------------------------------------------------------
// f16 "halo" kernel for the thin high-resolution layers (<= 64 contraction channels, <= 64 output channels: the 512^2 / 1024^2
// blocks of BASELINE configs[4], stg2_generator.py:159-209 at step = 7, 8; the condition-noise convs and ToRGB at every size).
// The gather kernel above fetches every input pixel once PER TAP through the vector-memory path (9 x 64 B per output pixel at
// 32 channels, one 1-KiB LDS-DMA piece per two MFMAs): those layers ran at 0.08 of the f16 peak, bound by DMA issue.  Here a
// workgroup owns a 16 x 16 patch of output pixels of one sample: the input patch + halo ((16 + nky - 1) x (16 + nkx - 1) pixels,
// all channels) is staged in LDS ONCE by LDS-DMA (out-of-image pixels and channel padding read the zero page), and the taps are
// formed by SHIFTED LDS reads — a tap moves every lane's pixel index by the same wave-uniform offset.  A pixel is CPP = CP / 8
// 16-byte chunks; chunk c of pixel q sits at physical chunk c ^ (q / (16 / CPP)) % CPP, so the 16 pixels of a ds_read_b128 lane
// group hit 16 distinct 16-byte slots whatever the tap shift (rule 21; applied on the SOURCE address of the DMA, whose LDS
// destination is lane-linear).  The launch's weight slices ([tap][BN][CP] halfs, <= 36 KB) are staged behind the patch by the
// same DMA pass, so the tap loop is pure ds_read + MFMA with no memory latency inside (a first version loaded the B fragments of
// the next tap from L2: every tap waited for them, 0.34 ms for 32 -> 32 channels at 1024^2 — r4 halo probe).  Modulation
// (ModulatedConv2d's s[b,ci], stylegan2_common_layers.py:311-320) multiplies the WEIGHT fragments — a patch belongs to one
// sample, so this is the reference's per-sample weight modulation, done per wavefront in registers — demodulation stays in the
// fp32 epilogue.  4 waves x (64 pixels x BN channels); v_mfma_f32_32x32x16_f16, fp32 accumulators; shared conv_epilogue (T2D).
// The launch is HBM-shaped: 32 -> 32 channels at 1024^2 moves 64 B in + 64 B out per pixel for 18 KFLOP.
// ------------------------------------------------------------------------------------------------------------------
template <int BN, int CP>
constexpr int halo_lds_floats() { return CP == 32 ? 5376 : 10496; }  // (18 * 18 * CPP chunks rounded up to 64) * 4 floats

template <int BN, int CP>
__global__ void __launch_bounds__(256, 3) conv_halo_f16(const GatherParams p) {
    typedef gif::f16 T;
    constexpr int TH = 16, TW = 16, BM = TH * TW;
    constexpr int CPP = CP / 8;                   // 16-byte chunks per pixel
    constexpr int CPP_SHIFT = CPP == 4 ? 2 : 3;
    constexpr int PPS_SHIFT = CPP == 4 ? 2 : 1;   // q >> PPS_SHIFT = q / (16 / CPP): the swizzle advances once per 256 bytes
    constexpr int SWM = CPP - 1;
    constexpr int KG = CP / 16, MT = 2, NT = BN / 32;
    constexpr int HALO_HALFS = halo_lds_floats<BN, CP>() * 2;
    static_assert((CP == 32 || CP == 64) && (BN == 32 || BN == 64), "halo kernel configurations");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    T* const Xs = reinterpret_cast<T*>(smem);  // [HALO_HALFS] patch buffer (later the epilogue's transposition buffer), then the weight slices
    T* const Wsm = Xs + HALO_HALFS;
    const T* const px = static_cast<const T*>(p.x);
    const T* const pw = static_cast<const T*>(p.wp);
    const T* const pzero = static_cast<const T*>(p.zero);
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int tile = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int tpi = p.t2_tx * p.t2_ty;
    const int b = tile / tpi, tr = tile - b * tpi;
    const int tyi = tr / p.t2_tx, txi = tr - tyi * p.t2_tx;
    const int dy_min = p.ddy > 0 ? p.dy0 : p.dy0 + (p.nky - 1) * p.ddy;
    const int dx_min = p.ddx > 0 ? p.dx0 : p.dx0 + (p.nkx - 1) * p.ddx;
    const int HWh = TW + p.nkx - 1;
    const int nchunks = (TH + p.nky - 1) * HWh * CPP;
    const bool has_scale = p.in_scale != nullptr;
    const int dbg = p.halo_dbg;  // ablation builds of the probe only (results are wrong): 1 no patch DMA, 2 no taps, 4 no stores

    // ---- stage the input patch + halo (one pass, LDS-DMA).  Lane's chunk e = 64 * wave + lane + 256 * it: its physical chunk e % CPP
    // is the same in every pass; its pixel index q advances by 256 / CPP, i.e. (dqy, dqx) in the halo grid (no division per pass).
    if (!(dbg & 1)) {
        constexpr int QSTEP = 256 / CPP;
        const int dqy = QSTEP / HWh, dqx = QSTEP - dqy * HWh;  // scalar
        int q = (wave * 64 + lane) >> CPP_SHIFT;
        int hy = q / HWh, hx = q - hy * HWh;
        const int cphys = lane & SWM;
        const T* const xb = px + (size_t)b * p.Hi * p.Wi * p.Ci;
        const int gy0 = tyi * TH + dy_min, gx0 = txi * TW + dx_min;
        for (int e0 = wave * 64; e0 < nchunks; e0 += 256) {  // wave-uniform
            const int c = cphys ^ ((q >> PPS_SHIFT) & SWM);
            const int gy = gy0 + hy, gx = gx0 + hx;
            const bool ok = e0 + lane < nchunks && (unsigned)gy < (unsigned)p.Hi && (unsigned)gx < (unsigned)p.Wi && c * 8 < p.Ci;
            const T* g = ok ? xb + ((gy * p.Wi + gx) * p.Ci + c * 8) : pzero;  // (32-bit offsets inside one sample: host check)
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Xs + e0 * 8), 16, 0, 0);
            q += QSTEP;
            hx += dqx; hy += dqy;
            if (hx >= HWh) { hx -= HWh; ++hy; }
        }
    }
    // ---- stage the launch's weight slices [tap][BN][CP] behind the patch (same swizzle: a row of CP halfs is a "pixel"; tap
    // slices start at multiples of 32 rows, so a row's swizzle term only depends on its channel n)
    {
        constexpr int WPT = BN * CPP;  // chunks per tap: 128 .. 512
        const int cphys = lane & SWM;
        if constexpr (WPT >= 256) {
            const int n = (wave * 64 + lane) >> CPP_SHIFT;
            const int c = cphys ^ ((n >> PPS_SHIFT) & SWM);
            int ta = 0, tb = 0;
            for (int t = 0; t < p.ntaps; ++t) {
                const int widx = (p.ky0 + ta * p.kstep) * p.KW + p.kx0 + tb * p.kstep;
                if (++tb == p.nkx) { tb = 0; ++ta; }
                const T* const wt = pw + (size_t)widx * p.RP * CP;  // scalar
#pragma unroll
                for (int k = 0; k < WPT / 256; ++k)  // rows n + k * 256 / CPP: the same swizzle term
                    __builtin_amdgcn_global_load_lds((gptr_t)(wt + (n + k * (256 / CPP)) * CP + c * 8),
                                                     (lptr_t)(Wsm + (t * WPT + k * 256 + wave * 64) * 8), 16, 0, 0);
            }
        } else {  // 128 chunks per tap = two pieces: waves 0, 1 take the even taps, waves 2, 3 the odd ones
            const int n = ((wave & 1) * 64 + lane) >> CPP_SHIFT;
            const int c = cphys ^ ((n >> PPS_SHIFT) & SWM);
            for (int t = wave >> 1; t < p.ntaps; t += 2) {  // wave-uniform
                const int a = t / p.nkx, bb = t - a * p.nkx;
                const int widx = (p.ky0 + a * p.kstep) * p.KW + p.kx0 + bb * p.kstep;
                __builtin_amdgcn_global_load_lds((gptr_t)(pw + ((size_t)widx * p.RP + n) * CP + c * 8),
                                                 (lptr_t)(Wsm + (t * WPT + (wave & 1) * 64) * 8), 16, 0, 0);
            }
        }
    }
    // ---- per-sample modulation of the contraction channels, as f16 multipliers of the weight fragments
    gif::f16x8_t sreg[KG];
    if (has_scale) {
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            const int ci = kg * 16 + lh * 8;
            float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
            if (ci < p.Ci) {
                s0 = *reinterpret_cast<const float4*>(p.in_scale + (size_t)b * p.Ci + ci);
                s1 = *reinterpret_cast<const float4*>(p.in_scale + (size_t)b * p.Ci + ci + 4);
            }
            sreg[kg][0] = (T)s0.x; sreg[kg][1] = (T)s0.y; sreg[kg][2] = (T)s0.z; sreg[kg][3] = (T)s0.w;
            sreg[kg][4] = (T)s1.x; sreg[kg][5] = (T)s1.y; sreg[kg][6] = (T)s1.z; sreg[kg][7] = (T)s1.w;
        }
    }
    // lane-constant byte offsets: B fragment (j, kg) of tap 0, the chunk term of A fragment kg, the lane's pixel in the halo grid
    int boff[NT][KG], ckg[KG], q0[MT];
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
        ckg[kg] = (kg * 2 + lh) << 4;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int r = j * 32 + li;
            boff[j][kg] = ((r << CPP_SHIFT) + ((kg * 2 + lh) ^ ((r >> PPS_SHIFT) & SWM))) << 4;
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) q0[i] = (4 * wave + 2 * i + (li >> 4)) * HWh + (li & 15);
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const char* const Xb = reinterpret_cast<const char*>(Xs);
    const char* Wb = reinterpret_cast<const char*>(Wsm);
    __syncthreads();  // the workgroup release waits for every wave's outstanding LDS-DMA (patch and weights)
    int ta = 0, tb = 0;  // tap (row, column) of the grid, advanced without divisions
    for (int t = 0; t < ((dbg & 2) ? 0 : p.ntaps); ++t) {
        const int off = (p.dy0 + ta * p.ddy - dy_min) * HWh + (p.dx0 + tb * p.ddx - dx_min);  // wave-uniform tap shift
        if (++tb == p.nkx) { tb = 0; ++ta; }
        gif::f16x8_t av[KG][MT], bw[KG][NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)  // B fragments: lane (li, lh) = output channel j * 32 + li, input channels 16 kg + 8 lh .. + 7
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) bw[kg][j] = *reinterpret_cast<const gif::f16x8_t*>(Wb + boff[j][kg]);
        Wb += BN * CP * 2;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int q = q0[i] + off;
            const int g16 = (q << (CPP_SHIFT + 4)) ^ ((q << (4 - PPS_SHIFT)) & (SWM << 4));
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) av[kg][i] = *reinterpret_cast<const gif::f16x8_t*>(Xb + (g16 ^ ckg[kg]));
        }
        if (has_scale) {
#pragma unroll
            for (int kg = 0; kg < KG; ++kg)
#pragma unroll
                for (int j = 0; j < NT; ++j) bw[kg][j] *= sreg[kg];
        }
#pragma unroll
        for (int kg = 0; kg < KG; ++kg)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[kg][i], bw[kg][j], acc[i][j], 0, 0, 0);
    }
    if (dbg & 4) return;
    conv_epilogue<BM, BN, 8, MT, NT, T, 256, true, halo_lds_floats<BN, CP>()>(p, acc, smem, tile * BM, 0, wave * 64, 0, tid, li, lh, 0, b, tyi * TH, txi * TW);
}

struct TileCfg {
    int BM, BN, BK;
};

// fp32: BK = 32 floats (LDS-DMA kernel) or 8 (register-staged kernel of the Cin < 32 layers).
// f16 : BK = 64 halfs, LDS-DMA kernel only (channel counts are multiples of 8 there: every 16-byte DMA chunk is 8 halfs).
template <typename T>
inline TileCfg pick_cfg(int cout, int cin) {
    TileCfg c;
    // f16 also has a 128x64 tile: the 64-channel layers of the 512^2 / 1024^2 blocks would waste half of a 128-wide N tile
    c.BN = cout <= 32 ? 32 : ((sizeof(T) == 2 && cout <= 64) ? 64 : 128);
    c.BK = sizeof(T) == 2 ? 64 : (cin < 32 ? 8 : 32);
    c.BM = c.BN == 32 ? 256 : 128;
    return c;
}

template <typename K>
int launch_kernel(K kern, GatherParams& p, int BM, int BN, int BK, hipStream_t s, gif::LdsAttr& attr) {
    p.tiles_m = gif::cdiv(p.M - p.m_begin, BM);
    p.tiles_n = p.RP / BN;
    size_t lds = (size_t)2 * (BM + BN) * (BK + 4) * sizeof(float);
    attr.ensure(reinterpret_cast<const void*>(kern), lds);
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n));
    p.part_row0 = t_part_rows;
    t_part_rows += p.tiles_m;
    t_last_bm = BM;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
    return 0;
}

template <int BM, int BN, int BK, int WMv, int WNv>
int launch_simple(GatherParams& p, hipStream_t s) {
    static gif::LdsAttr attr;
    return launch_kernel(conv_gather_mfma<BM, BN, BK, WMv, WNv>, p, BM, BN, BK, s, attr);
}

// LDS bytes of the per-sample scale table and its geometry (elements of T)
template <typename T, int BM>
inline size_t scale_table(GatherParams& p) {
    const int HWp = p.Hp * p.Wp;
    int nb = (BM - 1) / HWp + 2;  // samples a BM-row tile can touch
    if (nb > p.B) nb = p.B;
    p.stab_nb = nb;
    p.stab_stride = p.CP + 16 / (int)sizeof(T);  // + one 16-byte chunk: consecutive samples start 4 banks apart
    return (size_t)nb * p.stab_stride * sizeof(T);
}

// LDS bytes of the double-buffered operand tiles: 128-byte rows; X3: the weight tile is three 64-byte-row bf16 tiles
template <int BM, int BN, int X3, int NST = 2>
constexpr size_t stage_bytes() { return X3 ? (size_t)NST * BM * 128 + (size_t)NST * (X3 == 2 ? 2 : 3) * BN * 64 : (size_t)2 * (BM + BN) * 128; }

// GIF_H2_RING=2: two-stage operand ring in the 8-wave f16x2 kernels too (A/B)
inline bool h2_ring3() {
    static const int on = gif::knob("GIF_H2_RING") ? atoi(gif::knob("GIF_H2_RING")) != 2 : 1;
    return on != 0;
}

template <typename T, int BM, int BN, int WMv, int WNv, bool SCALE, int X3 = 0, int NST = 2>
int launch_glds_impl(GatherParams& p, hipStream_t s) {
    constexpr int BK = 128 / sizeof(T);  // 128-byte LDS rows
    static gif::LdsAttr attr;
    p.tiles_m = gif::cdiv(p.M - p.m_begin, BM);
    p.tiles_n = p.RP / BN;
    size_t lds = stage_bytes<BM, BN, X3, NST>();
    p.stab_nb = 0;
    p.stab_stride = 0;
    if (SCALE) lds += scale_table<T, BM>(p);
    if (lds > 160 * 1024) return -100;  // fp32 caller falls back to the register-staged kernel
    auto kern = conv_gather_mfma_glds<T, BM, BN, WMv, WNv, SCALE, BK, X3, NST>;
    attr.ensure(reinterpret_cast<const void*>(kern), lds);
    // passed as a kernel argument: a GOT load inside the K loop costs a scalar memory round trip + s_waitcnt per stage
    p.zero = gif::zero_page16();
    if (!p.zero) return -101;
    p.part_row0 = t_part_rows;
    t_part_rows += p.tiles_m;
    t_last_bm = BM;
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(64 * WMv * WNv), lds, s, p);
    return 0;
}

// all phases in one launch (64x64 tiles for the low-resolution layers, 256x128 / 8 waves for the big bf16x3 ones: one grid
// instead of up to eight launches with a partly filled last round each); returns -100 if the configuration does not fit
template <typename T, bool SCALE, int X3 = 0, int BM = 64, int BN = 64, int WMv = 2, int WNv = 2, int NST = 2>
int launch_glds_multi(GatherParams* ph, int nph, hipStream_t s) {
    constexpr int BK = 128 / sizeof(T);
    static gif::LdsAttr attr;
    const float* zero_page = gif::zero_page16();
    if (!zero_page) return -101;
    MultiParams mp{};
    size_t lds_max = 0;
    int total = 0, rows = 0;
    for (int i = 0; i < nph; ++i) {
        GatherParams& p = ph[i];
        p.tiles_m = gif::cdiv(p.M - p.m_begin, BM);
        p.tiles_n = p.RP / BN;
        size_t lds = stage_bytes<BM, BN, X3, NST>();
        p.stab_nb = 0;
        p.stab_stride = 0;
        if (SCALE) lds += scale_table<T, BM>(p);
        if (lds > lds_max) lds_max = lds;
        p.zero = zero_page;
        total += p.tiles_m * p.tiles_n;
        p.part_row0 = t_part_rows + rows;
        rows += p.tiles_m;
        mp.ph[i] = p;
        mp.wg_end[i] = total;
    }
    mp.nph = nph;
    if (lds_max > 160 * 1024) return -100;
    t_part_rows += rows;
    t_last_bm = BM;
    auto kern = conv_gather_mfma_glds_multi<T, BM, BN, WMv, WNv, SCALE, BK, X3, NST>;
    attr.ensure(reinterpret_cast<const void*>(kern), lds_max);
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(64 * WMv * WNv), lds_max, s, mp);
    return 0;
}

// conv3x3_rows_thin_h2: eligibility and launch (f16x2 launches of stride-1 3x3 layers with <= 32 output channels whose 256-row tiles are
// whole image rows or 256-pixel pieces of one; GIF_H2_ROWS_THIN=0: the gather kernel, A/B)
inline bool rows_thin_ok(const GatherParams& p) {
    static const int off = gif::knob("GIF_H2_ROWS_THIN") ? atoi(gif::knob("GIF_H2_ROWS_THIN")) == 0 : 0;
    const bool unit = (p.ddy == 1 || p.ddy == -1) && (p.ddx == 1 || p.ddx == -1) && p.dy0 + p.ddy == 0 && p.dx0 + p.ddx == 0;
    return !off && p.x3 == 2 && !p.dense && !p.in_scale && p.nky == 3 && p.nkx == 3 && unit && p.is == 1 && p.os == 1 && p.ooy == 0 &&
           p.oox == 0 && p.RP == 32 && p.CP % 32 == 0 && p.m_begin == 0 && p.M % 256 == 0 && p.Hp == p.Hi && p.Wp == p.Wi &&
           p.Ho == p.Hp && p.Wo == p.Wp && (p.Wp % 256 == 0 || (p.Wp >= 32 && 256 % p.Wp == 0)) &&
           ((long)p.B * p.Hi * p.Wi + p.Wi) * p.Ci * 4 < (1L << 32);
}
inline int launch_rows_thin(GatherParams& p, hipStream_t s) {
    static gif::LdsAttr attr;
    const size_t lds = (size_t)2 * kRowsThinA * 32 * sizeof(float);
    p.tiles_m = p.M / 256;
    p.tiles_n = 1;
    p.stab_nb = 0;
    p.stab_stride = 0;
    attr.ensure(reinterpret_cast<const void*>(conv3x3_rows_thin_h2), lds);
    p.zero = gif::zero_page16();
    if (!p.zero) return -101;
    p.part_row0 = t_part_rows;
    t_part_rows += p.tiles_m;
    t_last_bm = 256;
    hipLaunchKernelGGL(conv3x3_rows_thin_h2, dim3((unsigned)p.tiles_m), dim3(256), lds, s, p);
    return 0;
}

template <typename T, int BM, int BN, int WMv, int WNv>
int launch_glds(GatherParams& p, hipStream_t s) {
    if constexpr (sizeof(T) == 4) {
        if constexpr (WNv == 1 || (BM == 64 && BN == 64)) {  // the wave layouts the f16x2 kernels are built for
            if constexpr (WMv * WNv == 8) {
                if (p.x3 == 2 && h2_ring3()) {
                    const int rc = p.in_scale ? launch_glds_impl<T, BM, BN, WMv, WNv, true, 2, 3>(p, s)
                                              : launch_glds_impl<T, BM, BN, WMv, WNv, false, 2, 3>(p, s);
                    if (rc != -100) return rc;  // (-100: the scale table did not fit beside three stages)
                }
            }
            if (p.x3 == 2)
                return p.in_scale ? launch_glds_impl<T, BM, BN, WMv, WNv, true, 2>(p, s)
                                  : launch_glds_impl<T, BM, BN, WMv, WNv, false, 2>(p, s);
        }
        if (p.x3 == 2) return -102;
        if (p.x3)
            return p.in_scale ? launch_glds_impl<T, BM, BN, WMv, WNv, true, 1>(p, s)
                              : launch_glds_impl<T, BM, BN, WMv, WNv, false, 1>(p, s);
    }
    return p.in_scale ? launch_glds_impl<T, BM, BN, WMv, WNv, true>(p, s)
                      : launch_glds_impl<T, BM, BN, WMv, WNv, false>(p, s);
}

// 256x128 tile of the bf16x3 path: 8 waves as 8 (M) x 1 (N) — wave tile 32 x 128: ONE activation fragment to split per 24 MFMAs
// (the 4 x 2 layout splits two; GIF_X3_WAVES=42 selects it for A/B)
template <typename T>
int launch_big_x3(GatherParams& p, hipStream_t s) {
    static const int layout = gif::knob("GIF_X3_WAVES") ? atoi(gif::knob("GIF_X3_WAVES")) : 81;
    return (layout == 42 && p.x3 == 1) ? launch_glds<T, 256, 128, 4, 2>(p, s) : launch_glds<T, 256, 128, 8, 1>(p, s);
}

// 128x128 tile: 2 x 2 waves of 64 x 64; bf16x3: 4 x 1 waves of 32 x 128 (one activation fragment to split per 24 MFMAs)
template <typename T>
int launch_128(GatherParams& p, hipStream_t s) {
    if constexpr (sizeof(T) == 4) {
        static const int layout = gif::knob("GIF_X3_WAVES") ? atoi(gif::knob("GIF_X3_WAVES")) : 81;
        if (p.x3 == 2 || (p.x3 && layout != 42)) return launch_glds<T, 128, 128, 4, 1>(p, s);
    }
    return launch_glds<T, 128, 128, 2, 2>(p, s);
}

template <typename T>
int launch_multi(GatherParams* ph, int nph, bool scale, hipStream_t s) {
    if constexpr (sizeof(T) == 4) {
        if (ph[0].x3 == 2) return scale ? launch_glds_multi<T, true, 2>(ph, nph, s) : launch_glds_multi<T, false, 2>(ph, nph, s);
        if (ph[0].x3) return scale ? launch_glds_multi<T, true, 1>(ph, nph, s) : launch_glds_multi<T, false, 1>(ph, nph, s);
    }
    return scale ? launch_glds_multi<T, true>(ph, nph, s) : launch_glds_multi<T, false>(ph, nph, s);
}

// f16 halo kernel: launch configuration (tiles_n == 1: RP <= 64 is one N tile)
template <int BN, int CP>
int launch_halo_impl(GatherParams& p, hipStream_t s) {
    static gif::LdsAttr attr;
    p.t2_tx = gif::cdiv(p.Wp, 16);
    p.t2_ty = gif::cdiv(p.Hp, 16);
    p.tiles_m = p.B * p.t2_tx * p.t2_ty;
    p.tiles_n = 1;
    const size_t lds = (size_t)halo_lds_floats<BN, CP>() * sizeof(float) + (size_t)p.ntaps * BN * CP * 2;
    auto kern = conv_halo_f16<BN, CP>;
#ifdef GIF_HALO_PROBE  // ablation bits of tools/probes (results are wrong when set): never in the production library
    p.halo_dbg = gif::knob("GIF_HALO_DBG") ? atoi(gif::knob("GIF_HALO_DBG")) : 0;
#else
    p.halo_dbg = 0;
#endif
    attr.ensure(reinterpret_cast<const void*>(kern), lds);
    p.zero = gif::zero_page16();
    if (!p.zero) return -101;
    p.part_row0 = t_part_rows;
    t_part_rows += p.tiles_m;
    t_last_bm = 256;
    hipLaunchKernelGGL(kern, dim3((unsigned)p.tiles_m), dim3(256), lds, s, p);
    return 0;
}

// GIF_F16_HALO=0 / gif_conv2d_f16_halo_enable(0): the gather kernel everywhere (A/B knob; read once, tests use the setter)
inline std::atomic<int>& halo_switch() {
    static std::atomic<int> on{gif::knob("GIF_F16_HALO") ? (atoi(gif::knob("GIF_F16_HALO")) != 0) : 1};
    return on;
}

// bytes of the weight slices a halo launch stages in LDS: [ntaps][BN][CP] halfs
inline long halo_weight_bytes(const GatherParams& p) {
    const int bn = p.RP <= 32 ? 32 : 64, cp = p.CP <= 32 ? 32 : 64;
    return (long)p.ntaps * bn * cp * 2;
}

// Which f16 launches take the halo kernel: unit-stride gathers (forward stride 1, every data gradient incl. the output-parity
// phases of a transposed convolution) over a tap grid of <= 3 x 3 with <= 64 contraction and <= 64 output channels, on a
// sub-grid that fills at least one patch.  The modulation-gradient dot fusion needs whole patches (one partial row per patch,
// Hp * Wp / 256 of them per sample).  GIF_F16_HALO=0: A/B knob (the gather kernel).
inline bool halo_eligible(const GatherParams& p) {
    if (!halo_switch().load(std::memory_order_relaxed) || p.is != 1 || p.RP > 64 || p.CP > 64 || p.m_begin != 0) return false;
    if (p.nky < 1 || p.nky > 3 || p.nkx < 1 || p.nkx > 3 || (p.ddy != 1 && p.ddy != -1) || (p.ddx != 1 && p.ddx != -1)) return false;
    if (p.Hp < 16 || p.Wp < 16 || (long)p.B * gif::cdiv(p.Hp, 16) * gif::cdiv(p.Wp, 16) >= (1L << 23)) return false;
    if (p.part_dot && (p.Hp % 16 || p.Wp % 16)) return false;
    if (halo_weight_bytes(p) > 36864) return false;  // 9 taps of 64 x 64 channels (72 KB) stay on the gather kernel
    return true;
}

int launch_halo(GatherParams& p, hipStream_t s) {
    if (p.RP <= 32) return p.CP <= 32 ? launch_halo_impl<32, 32>(p, s) : launch_halo_impl<32, 64>(p, s);
    return p.CP <= 32 ? launch_halo_impl<64, 32>(p, s) : launch_halo_impl<64, 64>(p, s);
}

// GIF_CONV_VARIANT=1 forces the register-staged kernel everywhere (A/B benchmarking only)
int conv_variant() {
    static int v = -1;
    if (v < 0) {
        const char* e = gif::knob("GIF_CONV_VARIANT");
        v = e ? atoi(e) : 0;
    }
    return v;
}

template <typename T>
int launch(GatherParams& p, hipStream_t s) {
    constexpr bool F16 = sizeof(T) == 2;
    if (p.M <= 0 || p.ntaps <= 0) return 0;
    if ((long)p.B * p.Hi * p.Wi * p.Ci >= (1L << 31) || (long)p.B * p.Ho * p.Wo * p.Co >= (1L << 31)) {
        gif::set_error("conv: tensors of >= 2^31 elements are not supported (32-bit offsets)");
        return GIF_ENOSUP;
    }
    if ((p.x3 || F16) && ((long)p.B * p.Hi * p.Wi * p.Ci * (long)sizeof(T) > (1L << 32) - (1L << 26) || p.nky * p.nkx > 32)) {
        gif::set_error("conv (%s): the buffer-addressed activation DMA takes < 4 GiB of input and <= 32 taps (%ld elements, %d taps)%s",
                       F16 ? "f16" : "bf16x3", (long)p.B * p.Hi * p.Wi * p.Ci, p.nky * p.nkx, F16 ? "" : ": use the _f32 entry point");
        return GIF_ENOSUP;
    }
    TileCfg c = pick_cfg<T>(p.Co, p.Ci);
    if (p.x3) c.BK = 32;
    // LDS-DMA path: every layer whose K chunk is a 128-byte row (rows are 16-byte aligned in HBM: Ci % 4 == 0 for fp32,
    // Ci % 8 == 0 for f16)
    const bool only_glds = F16 || p.x3;  // no register-staged fallback for these operand formats
    const bool glds = only_glds || (c.BK == 32 && conv_variant() != 1);
    auto fail_f16 = [&](int rc) {
        if (rc != 0) gif::set_error("conv (%s): launch configuration does not fit (rc=%d)", F16 ? "f16" : "bf16x3", rc);
        return rc == 0 ? 0 : GIF_ENOSUP;
    };
    if (p.x3 && p.Ci < 24 && !p.dense) {
        gif::set_error("conv (bf16x3): needs >= 24 input channels (gif_conv2d_x3_eligible)");
        return GIF_ENOSUP;
    }
    if constexpr (F16) {
        if (halo_eligible(p) && launch_halo(p, s) == 0) return 0;
        if (c.BN == 64) return fail_f16(launch_glds<T, 128, 64, 2, 2>(p, s));
        // f16 MFMAs are 8x shorter than fp32 ones while an LDS-DMA piece costs the same to issue: on 128x128 tiles a wave issues
        // one 1-KiB piece per two MFMAs and the loop is bound by DMA issue + LDS traffic, not by the matrix pipe.  Layers with
        // >= 256 output channels and enough rows run 256x256 tiles on 8 waves (2 x 4, wave tile 128x64: one piece per four
        // MFMAs, 0.75 instead of 1 operand read per MFMA; 128 KB of LDS, one workgroup per CU): 512->512 at 64^2 780 -> 886
        // TFLOP/s, 256->256 at 128^2 710 -> 752.  GIF_F16_TILE256=0: A/B knob.
        static const int t256_off = gif::knob("GIF_F16_TILE256") ? atoi(gif::knob("GIF_F16_TILE256")) == 0 : 0;
        if (!t256_off && c.BN == 128 && p.RP % 256 == 0 && (long)gif::cdiv(p.M, 256) * (p.RP / 256) >= 512 &&
            launch_glds<T, 256, 256, 2, 4>(p, s) == 0)
            return 0;
    }
    if (c.BN == 128 && (F16 || c.BK == 32)) {
        // low-resolution layers (4x4 .. 16x16 at batch 32): a 128x128 grid would leave most CUs idle behind a
        // 144-step K loop; 64x64 tiles give 4x the workgroups (and 32 KB of LDS: 4 per CU) at a quarter of the latency
        const long tiles128 = (long)gif::cdiv(p.M, 128) * (p.RP / 128);
        if constexpr (!F16) {
            // low-resolution bf16x3 layers with at least one workgroup per CU: 128x64 tiles, 4 waves stacked along M (wave tile
            // 32x64: one activation fragment split per 12 MFMAs instead of per 6 on the 64x64 tile's 32x32 wave tiles) — 512->512 at
            // 16^2 143 -> 165 TFLOP/s, modulated 127 -> 157, stride-2 512->512 at 33^2 142 -> 170 (profiles/r3_dispatch_ab.txt)
            if (p.x3 && tiles128 < 384 && (long)gif::cdiv(p.M, 128) * (p.RP / 64) >= 256 &&
                launch_glds<T, 128, 64, 4, 1>(p, s) == 0)
                return 0;
        }
        if (glds && tiles128 < 384 && launch_glds<T, 64, 64, 2, 2>(p, s) == 0) return 0;
        if constexpr (!F16) {
            // bf16x3: the pre-split weight tile (48 KB) + the fp32 activation tile (32 KB) fill half a CU's LDS exactly, and a
            // modulated conv's scale table no longer fits beside them.  Big layers run 256x128 tiles on 8 waves instead: one
            // workgroup per CU (112 KB + table), still two waves per SIMD, a quarter less operand traffic per MFMA.
            static const int big_off = gif::knob("GIF_X3_BIG") ? atoi(gif::knob("GIF_X3_BIG")) == 0 : 0;
            const long tn = p.RP / 128, tiles256 = (long)gif::cdiv(p.M, 256) * tn;
            // tap-dense layers (K = 9 taps x 8..28 channels: 3..7 stages): one 8-wave workgroup per CU spends most of a tile in its
            // prologue and epilogue with nothing else resident; two 4-wave workgroups per CU on 128x128 tiles: 24 -> 256 at 128^2 129 ->
            // 133 TFLOP/s, 24 -> 512 at 64^2 130 -> 134, the family in the step 8.57 -> 8.19 ms (GIF_DENSE_TILE=256: A/B)
            static const int dense128 = gif::knob("GIF_DENSE_TILE") ? atoi(gif::knob("GIF_DENSE_TILE")) != 256 : 1;
            if (p.x3 && !big_off && tiles256 >= 512 && !(p.dense && dense128)) {
                const long slots = 256, full = tiles256 / slots, rem = tiles256 % slots;
                if (!p.no_split && rem > 0 && rem * 2 <= slots && slots % tn == 0) {  // nearly empty last round: remainder rows on 64x64 tiles
                    const int M = p.M;
                    const int m_bulk = (int)(full * slots / tn) * 256;
                    p.M = m_bulk;
                    if (launch_big_x3<T>(p, s) == 0) {
                        p.M = M;
                        p.m_begin = m_bulk;
                        int rc = launch_glds<T, 64, 64, 2, 2>(p, s);
                        p.m_begin = 0;
                        return rc;
                    }
                    p.M = M;
                } else if (launch_big_x3<T>(p, s) == 0) {
                    return 0;
                }
            }
        }
        if (glds && conv_variant() != 3) {
            // Tile quantisation: 512 workgroups of this kernel are resident (2 per CU), so T tiles cost ceil(T / 512) rounds.
            // The odd-sized phase grids of the transposed convolutions (129^2, 65^2, 33^2 pixels) give e.g. 4161 or 1092
            // tiles = 8.13 / 2.13 rounds: the nearly empty last round costs 10-30 %.  Split such launches: the full rounds
            // on 128x128 tiles, the remaining rows on 64x64 tiles (4x the workgroups, a quarter of the latency each).
            const long slots = 512, tn = p.RP / 128;
            const long full = tiles128 / slots, rem = tiles128 % slots;
            if (!p.no_split && full >= 1 && rem > 0 && rem * 2 <= slots && slots % tn == 0) {
                const int M = p.M;
                const int m_bulk = (int)(full * slots / tn) * 128;
                p.M = m_bulk;
                if (launch_128<T>(p, s) == 0) {
                    p.M = M;
                    p.m_begin = m_bulk;
                    int rc = launch_glds<T, 64, 64, 2, 2>(p, s);
                    p.m_begin = 0;
                    return rc;
                }
                p.M = M;
            }
        }
        if (glds) {
            int rc = launch_128<T>(p, s);
            if (rc == 0) return 0;
            if (only_glds) return fail_f16(rc);
        }
        if constexpr (!F16) return launch_simple<128, 128, 32, 2, 2>(p, s);
    }
    if constexpr (!F16) {
        if (rows_thin_ok(p)) return launch_rows_thin(p, s);  // f16x2, stride-1 3x3, <= 32 output channels: rows + halo staged once per kernel row
    }
    if (only_glds) return fail_f16(launch_glds<T, 256, 32, 4, 1>(p, s));
    if constexpr (!F16) {
        if (c.BN == 128 && c.BK == 8) return launch_simple<128, 128, 8, 2, 2>(p, s);
        if (c.BN == 32 && c.BK == 32) {
            if (glds && launch_glds<T, 256, 32, 4, 1>(p, s) == 0) return 0;
            return launch_simple<256, 32, 32, 4, 1>(p, s);
        }
        return launch_simple<256, 32, 8, 4, 1>(p, s);
    }
    return GIF_ENOSUP;
}

int check_geom(const gif_conv_geom* g, const char* who) {
    GIF_REQUIRE(g, "%s: null geometry", who);
    GIF_REQUIRE(g->B >= 0 && g->Hb > 0 && g->Wb > 0 && g->Hs > 0 && g->Ws > 0, "%s: bad spatial dims", who);
    GIF_REQUIRE(g->Cb > 0 && g->Cs > 0 && g->Cb % 4 == 0 && g->Cs % 4 == 0,
                "%s: channel counts must be positive multiples of 4 (Cb=%d Cs=%d)", who, g->Cb, g->Cs);
    GIF_REQUIRE(g->KH >= 1 && g->KH <= 3 && g->KW >= 1 && g->KW <= 3, "%s: kernel %dx%d unsupported", who, g->KH, g->KW);
    GIF_REQUIRE(g->stride == 1 || g->stride == 2, "%s: stride %d unsupported", who, g->stride);
    GIF_REQUIRE(g->pad >= 0, "%s: negative pad", who);
    GIF_REQUIRE((g->Hs - 1) * g->stride + g->KH - g->pad <= g->Hb + g->pad &&
                    (g->Ws - 1) * g->stride + g->KW - g->pad <= g->Wb + g->pad,
                "%s: small grid %dx%d does not fit big grid %dx%d (k=%d s=%d p=%d)", who, g->Hs, g->Ws, g->Hb,
                g->Wb, g->KH, g->stride, g->pad);
    return 0;
}

void fill_epilogue(GatherParams& p, const gif_conv_epilogue* e) {
    p.in_scale = e ? e->in_scale : nullptr;
    p.out_scale = e ? e->out_scale : nullptr;
    p.bias = e ? e->bias : nullptr;
    p.residual = e ? e->residual : nullptr;
    p.act = e ? e->act : 0;
    p.slope = e ? e->slope : 0.f;
    p.gain = e ? e->gain : 1.f;
    p.mask_src = e ? e->mask_src : nullptr;
    p.mask_slope = e ? e->mask_slope : 1.f;
    p.mask_gain = e ? e->mask_gain : 1.f;
    p.dot_src = e ? e->dot_src : nullptr;
    p.sat_flag = nullptr;
    p.out_f32 = e ? e->out_f32 : 0;
}

// tap-dense K order of the bf16x3 kernels (GatherParams::dense): 3x3 kernels over the full tap grid, contraction channels 8..28
int check_tapdense(const gif_conv_geom* g, const gif_conv_epilogue* e, int cin, bool transposed, const char* who) {
    GIF_REQUIRE(gif_conv2d_x3_tapdense_steps(cin, g->KH, g->KW) > 0, "%s: tap-dense mode needs a 3x3 kernel and 8 <= Cin < 32 (got %dx%d, %d)",
                who, g->KH, g->KW, cin);
    GIF_REQUIRE(!transposed || g->stride == 1, "%s: tap-dense mode: a strided data gradient runs tap subsets per output phase", who);
    GIF_REQUIRE(!(e && e->in_scale), "%s: tap-dense mode has no per-sample input scales", who);
    return 0;
}

// Gradient-producer fusions of one op: validate, carve the partial-sum buffers out of red_ws (before the launches) and reduce
// them in a fixed order (after).  out_rows = B * Ho * Wo of the op's output.
struct FusedSums {
    float* colsum = nullptr;
    float* dot = nullptr;
    float* part_cs = nullptr;
    float* part_dot = nullptr;
    float* tmp = nullptr;
    long cap = 0;  // partial rows available per buffer

    bool active() const { return colsum || dot; }

    int begin(GatherParams& p, const gif_conv_epilogue* e, long out_rows, int Co, int B, long hw, bool single_phase, const char* who) {
        t_part_rows = 0;
        if (!e || (!e->colsum && !e->dot)) return 0;
        GIF_REQUIRE(e->red_ws, "%s: colsum / dot need the red_ws workspace (gif_conv_epilogue_ws_floats)", who);
        GIF_REQUIRE(!e->dot || (e->dot_src && single_phase && hw % 256 == 0),
                    "%s: the dot fusion needs dot_src, a single-phase op and a multiple of 256 output pixels per sample (got %ld)", who, hw);
        colsum = e->colsum;
        dot = e->dot;
        cap = out_rows / 64 + 16;
        part_cs = e->red_ws;
        part_dot = e->red_ws + cap * Co;
        tmp = e->red_ws + 2 * cap * Co;
        p.part_cs = colsum ? part_cs : nullptr;
        p.part_dot = dot ? part_dot : nullptr;
        p.part_cap = (int)cap;
        p.no_split = dot ? 1 : 0;
        (void)B;
        return 0;
    }
    int finish(int Co, int B, long hw, hipStream_t s, const char* who) {
        if (!colsum && !dot) return 0;
        const int rows = t_part_rows;
        GIF_REQUIRE(rows > 0 && rows <= cap, "%s: partial-sum rows %d exceed the workspace (%ld)", who, rows, cap);
        if (colsum)
            if (int rc = gif::reduce_partials(part_cs, colsum, 1, rows, Co, tmp, s)) return rc;
        if (dot) {
            GIF_REQUIRE(t_last_bm > 0 && hw % t_last_bm == 0 && (long)rows * t_last_bm == (long)B * hw,
                        "%s: dot fusion: tiles of %d rows do not partition the %d samples of %ld pixels", who, t_last_bm, B, hw);
            if (int rc = gif::reduce_partials(part_dot, dot, B, (int)(hw / t_last_bm), Co, tmp, s)) return rc;
        }
        return 0;
    }
};


template <typename T>
void pack_dims(int cout, int cin, int* RP, int* CP, bool x3 = false) {
    TileCfg c = pick_cfg<T>(cout, cin);
    if (x3) c.BK = 32;  // the bf16x3 kernels only have 32-float K chunks: 24..31 input channels are zero-padded to one chunk
    *RP = (cout + c.BN - 1) / c.BN * c.BN;
    *CP = (cin + c.BK - 1) / c.BK * c.BK;
    if (sizeof(T) == 2 && cin <= 32) *CP = 32;  // "pair" mode of the f16 kernel: two taps per 64-half K chunk (GatherParams::pair)
}

// f16x2 launches: the packing is [header: RP row exponents + flag][planes]; the launch gets a fresh gate word (common.h)
inline int h2_operands(GatherParams& p, const void* wp2, const void* wp_fallback, hipStream_t s) {
    p.wexp = static_cast<const int*>(wp2);
    p.wp = static_cast<const char*>(wp2) + gif::h2_header_bytes(p.RP);
    p.gate = nullptr; p.gate_gen = 0; p.h2_stats = nullptr;
    if (wp_fallback) {
        const gif::H2Gate gt = gif::h2_next_gate(s);
        if (gt.err) return gt.err;
        p.gate = gt.word; p.gate_gen = gt.gen;
        p.h2_stats = gif::h2_stats_words();
    }
    return 0;
}
// the same parameters for the guarded bf16x3 launch
inline void h2_to_fallback(GatherParams& p, const void* wp3) {
    p.x3 = 1;
    p.wp = wp3;
    p.wexp = nullptr;
}

template <typename T>
int check_channels(const gif_conv_geom* g, const char* who) {
    if (sizeof(T) == 2)
        GIF_REQUIRE(g->Cb % 8 == 0 && g->Cs % 8 == 0, "%s: f16 channel counts must be multiples of 8 (Cb=%d Cs=%d)", who, g->Cb, g->Cs);
    return 0;
}

template <typename T>
int conv2d_fwd_impl(const void* big, const void* wp, void* small, const gif_conv_geom* g, const gif_conv_epilogue* e,
                    gif_stream_t stream, const char* who, int x3 = 0, bool dense = false, const void* wp_fallback = nullptr) {
    if (int rc = check_geom(g, who)) return rc;
    if (int rc = check_channels<T>(g, who)) return rc;
    if (g->B == 0) return 0;
    GIF_REQUIRE(big && wp && small, "%s: null pointer", who);
    GatherParams p{};
    p.x = big; p.wp = wp; p.y = small; p.x3 = x3;
    fill_epilogue(p, e);
    if (sizeof(T) == 2) p.sat_flag = gif::f16_sat_flag();
    p.B = g->B; p.Hi = g->Hb; p.Wi = g->Wb; p.Ci = g->Cb;
    p.Ho = g->Hs; p.Wo = g->Ws; p.Co = g->Cs;
    p.Hp = g->Hs; p.Wp = g->Ws; p.os = 1; p.ooy = 0; p.oox = 0; p.is = g->stride;
    p.nky = g->KH; p.nkx = g->KW; p.ntaps = g->KH * g->KW;
    p.dy0 = -g->pad; p.ddy = 1; p.dx0 = -g->pad; p.ddx = 1;
    p.ky0 = 0; p.kx0 = 0; p.kstep = 1; p.KW = g->KW;
    pack_dims<T>(p.Co, p.Ci, &p.RP, &p.CP, x3 != 0);
    p.pair = (sizeof(T) == 2 && p.CP == 32) ? 1 : 0;
    if (dense) {
        if (int rc = check_tapdense(g, e, g->Cb, false, who)) return rc;
        p.dense = g->Cb / 4;
    }
    if (x3 == 2)
        if (int rc = h2_operands(p, wp, wp_fallback, gif::as_stream(stream))) return rc;
    p.M = p.B * p.Hp * p.Wp;
    double flops = 2.0 * p.M * (double)p.Co * p.Ci * p.ntaps;
    const int fam = sizeof(T) == 2 ? 6 : dense ? (x3 == 2 ? 17 : 12) : x3 == 2 ? 13 : x3 ? 8 : (p.Ci >= 32 ? 0 : 5);
    FusedSums sums;
    if (int rc = sums.begin(p, e, p.M, p.Co, p.B, (long)p.Hp * p.Wp, true, who)) return rc;
    gif::ProfScope prof(fam, flops, gif::as_stream(stream), p.M, p.Co, p.Ci, p.ntaps * 10 + g->stride);
    if (int rc = launch<T>(p, gif::as_stream(stream))) return rc;
    if (x3 == 2 && p.gate && wp_fallback) {  // guarded fallback: the same op on the bf16x3 kernels, a no-op unless the gate was raised
        // FusedSums::finish reduces the per-tile partials through (t_part_rows, t_last_bm): normally those rows were written by the
        // f16x2 launch, so the twin must cut the op into the SAME tiles (advisor, round 5: nothing else enforces it)
        const int rows2 = t_part_rows, bm2 = t_last_bm;
        h2_to_fallback(p, wp_fallback);
        t_part_rows = 0;
        if (int rc = launch<T>(p, gif::as_stream(stream))) return rc;
        if (sums.active() && (t_part_rows != rows2 || t_last_bm != bm2)) {
            gif::set_error("%s: the guarded bf16x3 twin tiles the op differently from the f16x2 launch (%d x %d vs %d x %d rows): fused "
                           "column sums would mix partials", who, t_part_rows, t_last_bm, rows2, bm2);
            return GIF_ENOSUP;
        }
    }
    if (int rc = sums.finish(p.Co, p.B, (long)p.Hp * p.Wp, gif::as_stream(stream), who)) return rc;
    return gif::check_launch(who);
}

template <typename T>
int conv2d_bwd_data_impl(const void* small, const void* wp, void* big, const gif_conv_geom* g, const gif_conv_epilogue* e,
                         gif_stream_t stream, const char* who, int x3 = 0, bool dense = false, const void* wp_fallback = nullptr) {
    if (int rc = check_geom(g, who)) return rc;
    if (int rc = check_channels<T>(g, who)) return rc;
    if (g->B == 0) return 0;
    GIF_REQUIRE(small && wp && big, "%s: null pointer", who);
    hipStream_t s = gif::as_stream(stream);
    GatherParams base{};
    base.x = small; base.wp = wp; base.y = big; base.x3 = x3;
    fill_epilogue(base, e);
    if (sizeof(T) == 2) base.sat_flag = gif::f16_sat_flag();
    base.B = g->B; base.Hi = g->Hs; base.Wi = g->Ws; base.Ci = g->Cs;
    base.Ho = g->Hb; base.Wo = g->Wb; base.Co = g->Cb;
    pack_dims<T>(base.Co, base.Ci, &base.RP, &base.CP, x3 != 0);
    base.pair = (sizeof(T) == 2 && base.CP == 32) ? 1 : 0;
    if (x3 == 2)
        if (int rc = h2_operands(base, wp, wp_fallback, s)) return rc;
    if (dense) {
        if (int rc = check_tapdense(g, e, g->Cs, true, who)) return rc;
        base.dense = g->Cs / 4;
    }
    const int st = g->stride;
    FusedSums sums;
    if (int rc = sums.begin(base, e, (long)g->B * g->Hb * g->Wb, base.Co, g->B, (long)g->Hb * g->Wb, st == 1, who)) return rc;
    auto pmod = [st](int a) { return ((a % st) + st) % st; };
    // Build the (up to 4) output-parity phases; phases with no tap (e.g. 1x1 stride 2) are zero-filled.
    GatherParams ph[4];
    int nph = 0;
    bool need_zero = false;
    for (int py = 0; py < st; ++py)
        for (int px = 0; px < st; ++px) {
            GatherParams p = base;
            p.Hp = (g->Hb - py + st - 1) / st;
            p.Wp = (g->Wb - px + st - 1) / st;
            if (p.Hp <= 0 || p.Wp <= 0) continue;
            p.os = st; p.ooy = py; p.oox = px; p.is = 1;
            // taps with ky == (py+pad) mod st (and likewise kx): small pixel = big' + (py+pad-ky)/st
            p.ky0 = pmod(py + g->pad); p.kx0 = pmod(px + g->pad); p.kstep = st; p.KW = g->KW;
            p.nky = p.ky0 < g->KH ? (g->KH - p.ky0 + st - 1) / st : 0;
            p.nkx = p.kx0 < g->KW ? (g->KW - p.kx0 + st - 1) / st : 0;
            p.ntaps = p.nky * p.nkx;
            p.dy0 = (py + g->pad - p.ky0) / st; p.ddy = -1;
            p.dx0 = (px + g->pad - p.kx0) / st; p.ddx = -1;
            if (p.ntaps == 0) { need_zero = true; continue; }
            p.M = p.B * p.Hp * p.Wp;
            ph[nph++] = p;
        }
    if (need_zero) {
        GIF_REQUIRE(!(e && (e->bias || e->residual || e->act || e->mask_src || e->colsum || e->dot)), "%s: epilogue unsupported with empty phases", who);
        hipError_t me = hipMemsetAsync(big, 0, (size_t)g->B * g->Hb * g->Wb * g->Cb * sizeof(T), s);
        if (me != hipSuccess) { gif::set_error("%s memset: %s", who, hipGetErrorString(me)); return (int)me; }
    }
    // algorithmic FLOPs of a transposed conv: every small-side pixel scatters through every tap
    double flops = 2.0 * g->B * (double)g->Hs * g->Ws * g->KH * g->KW * (double)g->Cs * g->Cb;
    {
        const int fam = sizeof(T) == 2 ? 6 : dense ? (x3 == 2 ? 17 : 12) : x3 == 2 ? 13 : x3 ? 8 : (base.Ci >= 32 ? 0 : 5);
        gif::ProfScope prof(fam, flops, s, g->B * g->Hb * g->Wb, base.Co, base.Ci, -(g->KH * g->KW * 10 + g->stride));
        // small transposed convs: every phase alone would sit on the 64x64-tile path with a partly filled chip
        auto run_phases = [&]() -> int {
            bool merged = false;
            if (nph > 1 && conv_variant() == 0) {
                TileCfg c = pick_cfg<T>(base.Co, base.Ci);
                if (x3) c.BK = 32;
                bool small_all = c.BN == 128 && (sizeof(T) == 2 || c.BK == 32);
                for (int i = 0; i < nph && small_all; ++i)
                    small_all = (long)gif::cdiv(ph[i].M, 128) * (ph[i].RP / 128) < 384 &&
                                (long)ph[i].B * ph[i].Hi * ph[i].Wi * ph[i].Ci < (1L << 31) &&
                                (long)ph[i].B * ph[i].Ho * ph[i].Wo * ph[i].Co < (1L << 31);
                if (small_all)
                    merged = launch_multi<T>(ph, nph, base.in_scale != nullptr, s) == 0;
                if constexpr (sizeof(T) == 4) {
                    // big bf16x3 / f16x2 transposed convs: every phase would run 256x128 tiles on its own (bulk + remainder launch each)
                    static const int big_multi_off = gif::knob("GIF_X3_MULTI_BIG") ? atoi(gif::knob("GIF_X3_MULTI_BIG")) == 0 : 0;
                    bool big_all = x3 && !small_all && !big_multi_off && c.BN == 128;
                    for (int i = 0; i < nph && big_all; ++i)
                        big_all = (long)gif::cdiv(ph[i].M, 256) * (ph[i].RP / 128) >= 512 &&
                                  (long)ph[i].B * ph[i].Hi * ph[i].Wi * ph[i].Ci < (1L << 31) &&
                                  (long)ph[i].B * ph[i].Ho * ph[i].Wo * ph[i].Co < (1L << 31);
                    if (big_all)
                        merged = (ph[0].x3 == 2 ? (h2_ring3() ? (base.in_scale ? launch_glds_multi<T, true, 2, 256, 128, 8, 1, 3>(ph, nph, s)
                                                                               : launch_glds_multi<T, false, 2, 256, 128, 8, 1, 3>(ph, nph, s))
                                                              : (base.in_scale ? launch_glds_multi<T, true, 2, 256, 128, 8, 1>(ph, nph, s)
                                                                               : launch_glds_multi<T, false, 2, 256, 128, 8, 1>(ph, nph, s)))
                                                : (base.in_scale ? launch_glds_multi<T, true, 1, 256, 128, 8, 1>(ph, nph, s)
                                                                 : launch_glds_multi<T, false, 1, 256, 128, 8, 1>(ph, nph, s))) == 0;
                }
            }
            if (!merged)
                for (int i = 0; i < nph; ++i)
                    if (int rc = launch<T>(ph[i], s)) return rc;
            return 0;
        };
        if (int rc = run_phases()) return rc;
        if (x3 == 2 && base.gate && wp_fallback) {  // guarded fallback on the bf16x3 kernels (a no-op unless the gate was raised)
            const int rows2 = t_part_rows, bm2 = t_last_bm;
            for (int i = 0; i < nph; ++i) {
                h2_to_fallback(ph[i], wp_fallback);
                if (i) ph[i].h2_stats = nullptr;  // gif_h2_fallback_stats counts OPS: only the first phase's twin reports
            }
            t_part_rows = 0;
            if (int rc = run_phases()) return rc;
            if (sums.active() && (t_part_rows != rows2 || t_last_bm != bm2)) {  // (see conv2d_fwd_impl)
                gif::set_error("%s: the guarded bf16x3 twin tiles the op differently from the f16x2 launch (%d x %d vs %d x %d rows)", who,
                               t_part_rows, t_last_bm, rows2, bm2);
                return GIF_ENOSUP;
            }
        }
        if (int rc = sums.finish(base.Co, g->B, (long)g->Hb * g->Wb, s, who)) return rc;
    }
    return gif::check_launch(who);
}

}  // namespace

extern "C" {

int gif_conv2d_pack_dims(int cout, int cin, int* RP, int* CP) {
    GIF_REQUIRE(cout > 0 && cin > 0 && RP && CP, "pack_dims: bad arguments");
    pack_dims<float>(cout, cin, RP, CP);
    return 0;
}

int gif_conv2d_pack_dims_f16(int cout, int cin, int* RP, int* CP) {
    GIF_REQUIRE(cout > 0 && cin > 0 && RP && CP, "pack_dims_f16: bad arguments");
    pack_dims<gif::f16>(cout, cin, RP, CP);
    return 0;
}

int gif_conv2d_fwd_f32(const float* big, const float* wp, float* small, const gif_conv_geom* g,
                       const gif_conv_epilogue* e, gif_stream_t stream) {
    return conv2d_fwd_impl<float>(big, wp, small, g, e, stream, "conv2d_fwd");
}

int gif_conv2d_bwd_data_f32(const float* small, const float* wp, float* big, const gif_conv_geom* g,
                            const gif_conv_epilogue* e, gif_stream_t stream) {
    return conv2d_bwd_data_impl<float>(small, wp, big, g, e, stream, "conv2d_bwd_data");
}

// >= 24 input channels: a 24-channel layer wastes a quarter of its one 32-float K chunk and still beats the native kernel
int gif_conv2d_x3_eligible(int cout, int cin) { return cout > 0 && cin >= 24 && cin % 4 == 0 ? 1 : 0; }

int gif_conv2d_pack_dims_x3(int cout, int cin, int* RP, int* CP) {
    GIF_REQUIRE(cout > 0 && cin > 0 && RP && CP, "pack_dims_x3: bad arguments");
    pack_dims<float>(cout, cin, RP, CP, true);
    return 0;
}

int gif_conv2d_fwd_f32x3(const float* big, const void* wp3, float* small, const gif_conv_geom* g, const gif_conv_epilogue* e,
                         gif_stream_t stream) {
    return conv2d_fwd_impl<float>(big, wp3, small, g, e, stream, "conv2d_fwd_f32x3", 1);
}

int gif_conv2d_bwd_data_f32x3(const float* small, const void* wp3, float* big, const gif_conv_geom* g,
                              const gif_conv_epilogue* e, gif_stream_t stream) {
    return conv2d_bwd_data_impl<float>(small, wp3, big, g, e, stream, "conv2d_bwd_data_f32x3", 1);
}

int gif_conv2d_fwd_f32h2(const float* big, const void* wp2, const void* wp3, float* small, const gif_conv_geom* g,
                         const gif_conv_epilogue* e, gif_stream_t stream) {
    return conv2d_fwd_impl<float>(big, wp2, small, g, e, stream, "conv2d_fwd_f32h2", 2, false, wp3);
}

int gif_conv2d_bwd_data_f32h2(const float* small, const void* wp2, const void* wp3, float* big, const gif_conv_geom* g,
                              const gif_conv_epilogue* e, gif_stream_t stream) {
    return conv2d_bwd_data_impl<float>(small, wp2, big, g, e, stream, "conv2d_bwd_data_f32h2", 2, false, wp3);
}

/* tap-dense K order on the f16x2 kernels (packings from gif_pack_weight_f32h2x3_tapdense; wp3 == NULL: unguarded) */
int gif_conv2d_fwd_f32h2_tapdense(const float* big, const void* wp2, const void* wp3, float* small, const gif_conv_geom* g,
                                  const gif_conv_epilogue* e, gif_stream_t stream) {
    return conv2d_fwd_impl<float>(big, wp2, small, g, e, stream, "conv2d_fwd_f32h2_tapdense", 2, true, wp3);
}

int gif_conv2d_bwd_data_f32h2_tapdense(const float* small, const void* wp2, const void* wp3, float* big, const gif_conv_geom* g,
                                       const gif_conv_epilogue* e, gif_stream_t stream) {
    return conv2d_bwd_data_impl<float>(small, wp2, big, g, e, stream, "conv2d_bwd_data_f32h2_tapdense", 2, true, wp3);
}

int gif_conv2d_fwd_f32x3_tapdense(const float* big, const void* wp3, float* small, const gif_conv_geom* g,
                                  const gif_conv_epilogue* e, gif_stream_t stream) {
    return conv2d_fwd_impl<float>(big, wp3, small, g, e, stream, "conv2d_fwd_f32x3_tapdense", 1, true);
}

int gif_conv2d_bwd_data_f32x3_tapdense(const float* small, const void* wp3, float* big, const gif_conv_geom* g,
                                       const gif_conv_epilogue* e, gif_stream_t stream) {
    return conv2d_bwd_data_impl<float>(small, wp3, big, g, e, stream, "conv2d_bwd_data_f32x3_tapdense", 1, true);
}

int gif_conv2d_f16_halo_enable(int on) {
    halo_switch().store(on ? 1 : 0, std::memory_order_relaxed);
    return 0;
}

// would a FORWARD f16 convolution of this shape (activation channel counts, output grid Hs x Ws) run the halo kernel?
int gif_conv2d_f16_halo_eligible(int cin, int cout, int KH, int KW, int stride, int Hs, int Ws) {
    if (cin <= 0 || cout <= 0 || KH < 1 || KW < 1 || stride < 1 || Hs <= 0 || Ws <= 0) return 0;
    GatherParams p{};
    pack_dims<gif::f16>(cout, cin, &p.RP, &p.CP);
    p.is = stride; p.nky = KH; p.nkx = KW; p.ntaps = KH * KW; p.ddy = 1; p.ddx = 1; p.Hp = Hs; p.Wp = Ws; p.B = 1;
    return halo_eligible(p) ? 1 : 0;
}

int gif_conv2d_fwd_f16(const void* big, const void* wp, void* small, const gif_conv_geom* g, const gif_conv_epilogue* e,
                       gif_stream_t stream) {
    return conv2d_fwd_impl<gif::f16>(big, wp, small, g, e, stream, "conv2d_fwd_f16");
}

int gif_conv2d_bwd_data_f16(const void* small, const void* wp, void* big, const gif_conv_geom* g,
                            const gif_conv_epilogue* e, gif_stream_t stream) {
    return conv2d_bwd_data_impl<gif::f16>(small, wp, big, g, e, stream, "conv2d_bwd_data_f16");
}
}
