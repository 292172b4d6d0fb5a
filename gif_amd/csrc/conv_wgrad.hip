// fp32 MFMA weight-gradient kernel for gfx950 (NHWC) + weight pack / unpack.
//
// Replaces autograd's wgrad of the F.conv2d / F.conv_transpose2d calls in
// stylegan2_common_layers.py:176, :330, :339, :345, :405-414.
//   dW[t][o][i] = sum_{b,oy,ox} small[b,oy,ox,o]*ss[b,o] * big[b, oy*s+ky-pad, ox*s+kx-pad, i]*bs[b,i]
// GEMM view per tap t: M = Cs (o), N = Cb (i), K = B*Hs*Ws pixels.  Both operands are channel-contiguous in
// HBM, i.e. K is the SLOW axis: tiles are staged as [pixel][channel] and read back with lanes along the channel
// axis (ds_read_b32, 128 contiguous bytes per half wave => conflict free), which is exactly the A[i][k] / B[k][j]
// operand shape of v_mfma_f32_32x32x2_f32 with k = pixel.
// The pixel axis is split over gridDim.z; every split writes its own partial tile (deterministic: no atomics),
// and gif_unpack_wgrad_f32 reduces the splits while scattering into the canonical [O,I,KH,KW]-style tensor.
// The per-sample scales implement the wgrad of the modulated convolution (x*s and dy*d) on the fly.
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BKP_MAX = 32;  // host-side rounding unit of the pixel chunks (any BKP below divides it)

struct WgradParams {
    const void* sm;   // small side [B,Hs,Ws,Cs]   (fp32, or f16 for the T = f16 instantiation)
    const void* bg;   // big side   [B,Hb,Wb,Cb]
    float* ws;        // [nsplit][T][RP][CP]
    const float* ss;  // [B,Cs] or null
    const float* bs;  // [B,Cb] or null
    int B, Hs, Ws, Cs, Hb, Wb, Cb;
    int KW, stride, pad, T;
    int RP, CP;
    long Ntot;   // B*Hs*Ws
    long chunk;  // pixels per split (multiple of BKP)
    int tiles_q, tiles_pq;
    int stab_nb;  // LDS scale table (scaled LDS-DMA path): samples a pixel chunk can touch
    const void* zero;   // 16 zero bytes in HBM (source of out-of-range LDS-DMA lanes), passed as an argument
    // "planes" mode (Winograd wgrad): tap t has no spatial shift but its own operand planes sm + t*sm_plane, bg + t*bg_plane
    long sm_plane, bg_plane;
    // f16x2 (common.h "h2"): gate word the kernel raises when a K group leaves the precision window; bf16x3 launch WITH a gate:
    // the guarded fallback, runs iff *gate == gate_gen
    unsigned* gate;
    unsigned gate_gen;
    unsigned* h2_stats;
    // conv_wgrad_h2v2<.., TAPS>: the 128 tile columns hold `tpt` taps x Cb channels (thin big side), `tgroups` column tiles cover the T taps
    int tpt, tgroups;
};

// XCD-aware, bijective block remap: XCD k (= blockIdx % 8 by dispatch order) gets a contiguous range of logical ids.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    int xcd = bid % nx, idx = bid / nx;
    int q = nwg / nx, r = nwg % nx;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// GLDS = true (no per-sample scales): both operand tiles go global -> LDS directly (global_load_lds_dwordx4, lane-linear
// destination == the [pixel][channel] tile layout), no staging registers / ds_write; invalid lanes read a zero page.
// TAB = true (GLDS with per-sample scales, needs Hs*Ws % BKP == 0 so that a stage never straddles two samples): the
// scale vectors ss[b, r0:r0+BP) / bs[b, c0:c0+BQ) of the samples this chunk touches sit in an LDS table and are
// multiplied into the operand fragments (the DMA cannot scale data in flight).
// T = f16 (GLDS only): the same [pixel][channel] tiles in halfs (16-byte DMA chunk = 8 channels); v_mfma_f32_32x32x16_f16 wants
// 8 consecutive K (= pixels) of ONE channel per lane, i.e. a transposed operand, which is gathered with eight 16-bit LDS reads
// per fragment (lane = channel: the 32 lanes of a half wave read 64 contiguous bytes of one pixel row, conflict free).
// Accumulators and the partial-sum workspace stay fp32.
// X3 (T = float, GLDS): fp32 tiles, bf16 matrix cores ("bf16x3", common.h split_pair): like the f16 path every lane gathers 8
// consecutive pixels of its channel per operand tile (eight ds_read_b32, conflict free), then splits the 8 floats into the
// three bf16x8 terms; six v_mfma_f32_32x32x16_bf16 per tile pair and 16 pixels.  Both operands are activations, so neither can
// be pre-split; the split (36 VALU per fragment) overlaps with the MFMAs of the other resident waves.
// X3 = 2 ("f16x2", common.h): the same tiles, v_mfma_f32_32x32x16_f16 on two f16 terms per operand under per-CHANNEL power-of-two
// scales (a row of either operand is a channel, K runs over pixels): both operands carry a running exponent per lane, three products.
template <typename T, int BP, int BQ, int WAVES_P, int WAVES_Q, bool GLDS, int BKP, bool TAB = false, int X3 = 0>
// (f16x2, 4 waves: "at least 2 workgroups per CU" caps the budget at 256 registers, which makes hipcc keep the accumulators in
// architectural VGPRs — the rare rescale path multiplies them with VALU instructions; from AGPRs that costs 64 temporaries and the
// kernel would no longer fit two waves per SIMD)
__global__ void __launch_bounds__(64 * WAVES_P * WAVES_Q, (X3 == 2 && WAVES_P * WAVES_Q == 4) ? 2 : 1) conv_wgrad_mfma(const WgradParams p) {
    constexpr bool F16 = sizeof(T) == 2;
    if constexpr (X3 == 1) {
        if (p.gate) {  // guarded fallback of an f16x2 launch: nothing to do unless that launch raised the gate
            if (*p.gate < p.gate_gen) return;
            if (blockIdx.x == 0 && threadIdx.x == 0 && p.h2_stats) atomicAdd(p.h2_stats, 1u);
        }
    }
    static_assert(!X3 || (!F16 && GLDS && BKP % 16 == 0), "bf16x3: fp32 tiles through the LDS-DMA path, 16-pixel K steps");
    static_assert(X3 != 2 || BKP == 32, "f16x2: the software-pipelined 32-pixel stages only");
    constexpr int EPC = 16 / sizeof(T);  // elements per 16-byte chunk
    constexpr int THREADS = 64 * WAVES_P * WAVES_Q;
    constexpr int WPt = BP / WAVES_P, WQt = BQ / WAVES_Q;
    constexpr int MT = WPt / 32, NT = WQt / 32;
    constexpr int P_ROWS = THREADS / (BP / EPC), Q_ROWS = THREADS / (BQ / EPC);  // pixel rows per pass
    constexpr int P_IT = BKP / P_ROWS, Q_IT = BKP / Q_ROWS;
    static_assert(BKP % P_ROWS == 0 && BKP % Q_ROWS == 0 && P_IT >= 1 && Q_IT >= 1, "tile/thread mapping");
    static_assert(!F16 || (GLDS && BKP % 16 == 0), "f16: LDS-DMA staging, 16-pixel MFMA K steps");

    // ONE LDS object (two separate __shared__ arrays make hipcc drain the LDS-DMA with vmcnt(0) before every ds_read)
    extern __shared__ __attribute__((aligned(16))) float wg_smem[];
    typedef T PTile[BKP][BP];
    typedef T QTile[BKP][BQ];
    PTile* Ps = reinterpret_cast<PTile*>(wg_smem);                                        // [2][BKP][BP]
    QTile* Qs = reinterpret_cast<QTile*>(reinterpret_cast<T*>(wg_smem) + 2 * BKP * BP);   // [2][BKP][BQ]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wp0 = (wave / WAVES_Q) * WPt, wq0 = (wave % WAVES_Q) * WQt;
    // logical id = ((split * T) + tap) * tiles + tile: the 9 taps (and channel tiles) of one pixel chunk read the SAME
    // activations, so they are made consecutive and therefore co-resident on one XCD's L2.
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = lid % p.tiles_pq;
    const int t = (lid / p.tiles_pq) % p.T;
    const int split = lid / (p.tiles_pq * p.T);
    const int tq = tile % p.tiles_q, tp = tile / p.tiles_q;
    const int r0 = tp * BP, c0 = tq * BQ;
    const bool planes = p.sm_plane != 0;
    const int ky = planes ? 0 : t / p.KW, kx = planes ? 0 : t - ky * p.KW;
    const T* const smb = static_cast<const T*>(p.sm) + (size_t)t * p.sm_plane;
    const T* const bgb = static_cast<const T*>(p.bg) + (size_t)t * p.bg_plane;
    const T* const pzero = static_cast<const T*>(p.zero);
    const int n_begin = (int)((long)split * p.chunk);
    int n_end = n_begin + (int)p.chunk;
    if (n_end > (int)p.Ntot) n_end = (int)p.Ntot;
    const unsigned HWs = (unsigned)(p.Hs * p.Ws);

    const int p_row = tid / (BP / EPC), p_ch = r0 + (tid % (BP / EPC)) * EPC;
    const int q_row = tid / (BQ / EPC), q_ch = c0 + (tid % (BQ / EPC)) * EPC;
    const bool p_ch_ok = p_ch < p.Cs, q_ch_ok = q_ch < p.Cb;
    const bool has_ss = !TAB && p.ss != nullptr, has_bs = !TAB && p.bs != nullptr;
    float* Stab = reinterpret_cast<float*>(reinterpret_cast<T*>(wg_smem) + 2 * BKP * (BP + BQ));  // [stab_nb][BP + BQ] fp32
    int tab_row = 0, tab_rem = 0;                 // table row / pixel offset inside the sample of the stage being computed
    if (TAB) {
        const int b_first = n_begin / (int)HWs;
        for (int e = tid; e < p.stab_nb * (BP + BQ); e += THREADS) {
            int bl = e / (BP + BQ), c = e - bl * (BP + BQ);
            int b = b_first + bl;
            float v = 0.f;
            if (b < p.B) {
                if (c < BP) v = (r0 + c < p.Cs) ? (p.ss ? p.ss[(size_t)b * p.Cs + r0 + c] : 1.f) : 0.f;
                else v = (c0 + c - BP < p.Cb) ? (p.bs ? p.bs[(size_t)b * p.Cb + c0 + c - BP] : 1.f) : 0.f;
            }
            Stab[e] = v;
        }
        tab_rem = n_begin - b_first * (int)HWs;
        __syncthreads();
    }

    float4 p_reg[P_IT], ps_reg[P_IT], q_reg[Q_IT], qs_reg[Q_IT];
    unsigned p_mask = 0, q_mask = 0;

    // Per-thread pixel cursors, advanced by BKP pixels per stage with carry propagation (no divisions in the loop).
    // P side: linear pixel n (address n*Cs) and its sample index (for the per-sample scale).
    // Q side: (b, oy, ox) of the small-grid pixel; the big-grid pixel is (oy*stride+ky-pad, ox*stride+kx-pad).
    int p_n[P_IT], p_b[P_IT], p_rem[P_IT];
    int q_n[Q_IT], q_b[Q_IT], q_oy[Q_IT], q_ox[Q_IT];
#pragma unroll
    for (int it = 0; it < P_IT; ++it) {
        int n = n_begin + p_row + it * P_ROWS;
        p_n[it] = n;
        p_b[it] = (int)((unsigned)n / HWs);
        p_rem[it] = n - p_b[it] * (int)HWs;
    }
#pragma unroll
    for (int it = 0; it < Q_IT; ++it) {
        int n = n_begin + q_row + it * Q_ROWS;
        q_n[it] = n;
        unsigned b = (unsigned)n / HWs;
        unsigned r = (unsigned)n - b * HWs;
        unsigned oy = r / (unsigned)p.Ws;
        q_b[it] = (int)b; q_oy[it] = (int)oy; q_ox[it] = (int)(r - oy * (unsigned)p.Ws);
    }

    // Branch-free loads (invalid lanes read a dummy address and are zeroed at the LDS store); per-sample scales are
    // multiplied in at the LDS store, after the MFMAs of the current stage.
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto load_global = [&](int buf) __attribute__((always_inline)) {
        p_mask = 0;
        q_mask = 0;
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            bool ok = p_ch_ok && p_n[it] < n_end;
            if (GLDS) {
                // thread tid lands at byte offset 16*tid of pass `it` (== Ps[buf][p_row + it*P_ROWS][EPC*(tid % (BP/EPC))])
                const T* g = ok ? smb + (p_n[it] * p.Cs + p_ch) : pzero;
                __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(&Ps[buf][it * P_ROWS][0] + wave_u * (1024 / (int)sizeof(T))), 16, 0, 0);
                p_n[it] += BKP;
                continue;
            }
            if constexpr (!F16) p_reg[it] = *reinterpret_cast<const float4*>(smb + (ok ? p_n[it] * p.Cs + p_ch : 0));
            if (has_ss) ps_reg[it] = *reinterpret_cast<const float4*>(p.ss + (ok ? p_b[it] * p.Cs + p_ch : 0));
            p_mask |= (ok ? 1u : 0u) << it;
            p_n[it] += BKP;
            p_rem[it] += BKP;
            while (p_rem[it] >= (int)HWs) { p_rem[it] -= (int)HWs; ++p_b[it]; }
        }
#pragma unroll
        for (int it = 0; it < Q_IT; ++it) {
            int iy = q_oy[it] * p.stride + ky - p.pad, ix = q_ox[it] * p.stride + kx - p.pad;
            bool ok = q_ch_ok && q_n[it] < n_end && (unsigned)iy < (unsigned)p.Hb && (unsigned)ix < (unsigned)p.Wb;
            if (GLDS) {
                const T* g = ok ? bgb + (((q_b[it] * p.Hb + iy) * p.Wb + ix) * p.Cb + q_ch) : pzero;
                __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(&Qs[buf][it * Q_ROWS][0] + wave_u * (1024 / (int)sizeof(T))), 16, 0, 0);
            } else {
                if constexpr (!F16)
                    q_reg[it] = *reinterpret_cast<const float4*>(
                        bgb + (ok ? ((q_b[it] * p.Hb + iy) * p.Wb + ix) * p.Cb + q_ch : 0));
                if (has_bs) qs_reg[it] = *reinterpret_cast<const float4*>(p.bs + (ok ? q_b[it] * p.Cb + q_ch : 0));
                q_mask |= (ok ? 1u : 0u) << it;
            }
            q_n[it] += BKP;
            q_ox[it] += BKP;
            if (p.Ws >= BKP) {  // uniform: at most one row wrap per stage
                const bool wx = q_ox[it] >= p.Ws;
                q_ox[it] -= wx ? p.Ws : 0;
                q_oy[it] += wx ? 1 : 0;
                const bool wy = q_oy[it] >= p.Hs;
                q_oy[it] -= wy ? p.Hs : 0;
                q_b[it] += wy ? 1 : 0;
            } else {
                while (q_ox[it] >= p.Ws) { q_ox[it] -= p.Ws; ++q_oy[it]; }
                while (q_oy[it] >= p.Hs) { q_oy[it] -= p.Hs; ++q_b[it]; }
            }
        }
    };
    auto store_lds = [&](int buf) __attribute__((always_inline)) {
        if constexpr (GLDS) return;  // the DMA already wrote the tiles
        else {
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            float4 v = p_reg[it];
            if (has_ss) { v.x *= ps_reg[it].x; v.y *= ps_reg[it].y; v.z *= ps_reg[it].z; v.w *= ps_reg[it].w; }
            if (!((p_mask >> it) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&Ps[buf][p_row + it * P_ROWS][(tid % (BP / 4)) * 4]) = v;
        }
#pragma unroll
        for (int it = 0; it < Q_IT; ++it) {
            float4 v = q_reg[it];
            if (has_bs) { v.x *= qs_reg[it].x; v.y *= qs_reg[it].y; v.z *= qs_reg[it].z; v.w *= qs_reg[it].w; }
            if (!((q_mask >> it) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&Qs[buf][q_row + it * Q_ROWS][(tid % (BQ / 4)) * 4]) = v;
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

    auto compute = [&](int buf) __attribute__((always_inline)) {
        if constexpr (X3 == 1) {
            float psv[MT], qsv[NT];
            if (TAB) {
                const float* row = Stab + tab_row * (BP + BQ);
#pragma unroll
                for (int i = 0; i < MT; ++i) psv[i] = row[wp0 + i * 32 + li];
#pragma unroll
                for (int j = 0; j < NT; ++j) qsv[j] = row[BP + wq0 + j * 32 + li];
                tab_rem += BKP;
                if (tab_rem >= (int)HWs) { tab_rem -= (int)HWs; ++tab_row; }
            }
#pragma unroll
            for (int ks = 0; ks < BKP / 16; ++ks) {
                const int k0 = 16 * ks + 8 * lh;  // this lane half's 8 pixels of the K step
                float a[MT][8], b[NT][8];
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) a[i][e] = Ps[buf][k0 + e][wp0 + i * 32 + li];
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) b[j][e] = Qs[buf][k0 + e][wq0 + j * 32 + li];
                gif::u32x4_t sa[3][MT], sb[3][NT];
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x0 = a[i][2 * e], x1 = a[i][2 * e + 1];
                        if (TAB) { x0 *= psv[i]; x1 *= psv[i]; }
                        unsigned h, m, l;
                        gif::split_pair(x0, x1, h, m, l);
                        sa[0][i][e] = h; sa[1][i][e] = m; sa[2][i][e] = l;
                    }
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x0 = b[j][2 * e], x1 = b[j][2 * e + 1];
                        if (TAB) { x0 *= qsv[j]; x1 *= qsv[j]; }
                        unsigned h, m, l;
                        gif::split_pair(x0, x1, h, m, l);
                        sb[0][j][e] = h; sb[1][j][e] = m; sb[2][j][e] = l;
                    }
                // smallest terms first; mid*lo, lo*mid, lo*lo (<= 2^-23 of the product) are not formed
                constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int t6 = GIF_X3_FIRST_TERM; t6 < 6; ++t6)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gif::bf16x8_t, sa[TA[t6]][i]),
                                                                                __builtin_bit_cast(gif::bf16x8_t, sb[TB[t6]][j]),
                                                                                acc[i][j], 0, 0, 0);
            }
            return;
        } else if constexpr (F16) {
            float psv[MT], qsv[NT];
            if (TAB) {
                const float* row = Stab + tab_row * (BP + BQ);
#pragma unroll
                for (int i = 0; i < MT; ++i) psv[i] = row[wp0 + i * 32 + li];
#pragma unroll
                for (int j = 0; j < NT; ++j) qsv[j] = row[BP + wq0 + j * 32 + li];
                tab_rem += BKP;
                if (tab_rem >= (int)HWs) { tab_rem -= (int)HWs; ++tab_row; }
            }
#pragma unroll
            for (int ks = 0; ks < BKP / 16; ++ks) {
                gif::f16x8_t af[MT], bf[NT];
                const int k0 = 16 * ks + 8 * lh;  // this lane half's 8 pixels of the K step
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) af[i][e] = Ps[buf][k0 + e][wp0 + i * 32 + li];
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bf[j][e] = Qs[buf][k0 + e][wq0 + j * 32 + li];
                if (TAB) {
#pragma unroll
                    for (int i = 0; i < MT; ++i) af[i] *= (gif::f16)psv[i];
#pragma unroll
                    for (int j = 0; j < NT; ++j) bf[j] *= (gif::f16)qsv[j];
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
            return;
        } else {
        // operand fragments of k-step ks+1 are fetched from LDS before the MFMAs of k-step ks (register double buffer)
        float av[2][MT], bv[2][NT];
        float psv[MT], qsv[NT];
        if (TAB) {
            const float* row = Stab + tab_row * (BP + BQ);
#pragma unroll
            for (int i = 0; i < MT; ++i) psv[i] = row[wp0 + i * 32 + li];
#pragma unroll
            for (int j = 0; j < NT; ++j) qsv[j] = row[BP + wq0 + j * 32 + li];
            tab_rem += BKP;
            if (tab_rem >= (int)HWs) { tab_rem -= (int)HWs; ++tab_row; }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) av[0][i] = Ps[buf][lh][wp0 + i * 32 + li];
#pragma unroll
        for (int j = 0; j < NT; ++j) bv[0][j] = Qs[buf][lh][wq0 + j * 32 + li];
#pragma unroll
        for (int ks = 0; ks < BKP / 2; ++ks) {
            const int c = ks & 1, n = c ^ 1;
            if (ks + 1 < BKP / 2) {
#pragma unroll
                for (int i = 0; i < MT; ++i) av[n][i] = Ps[buf][2 * ks + 2 + lh][wp0 + i * 32 + li];
#pragma unroll
                for (int j = 0; j < NT; ++j) bv[n][j] = Qs[buf][2 * ks + 2 + lh][wq0 + j * 32 + li];
            }
            if (TAB) {
#pragma unroll
                for (int i = 0; i < MT; ++i) av[c][i] *= psv[i];
#pragma unroll
                for (int j = 0; j < NT; ++j) bv[c][j] *= qsv[j];
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][i], bv[c][j], acc[i][j], 0, 0, 0);
            // pin the schedule: the LDS reads of k-step ks+1 are issued BEFORE the MFMAs of k-step ks
            __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);  // DS reads
            __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);  // MFMA
        }
        }
    };

    if constexpr (X3 && BKP == 32) {
        // ---- bf16x3 / f16x2, software-pipelined (two 16-pixel groups per stage; same schedule as conv_gather_mfma_glds' X3 path):
        //   DMA(stage+1) | MFMAs(g0) + read & split g1 | barrier | MFMAs(g1) + read & split g0 of stage+1
        // The split of the NEXT group ((MT + NT) * 4 pieces: bf16x3 11 VALU each, +2 with per-sample scales; f16x2 6 VALU + the
        // running exponents) sits piecewise between the MFMAs of the current one; sched_barrier(0) pins the interleave.
        constexpr bool H2 = X3 == 2;
        constexpr int NPL = H2 ? 2 : 3;
        constexpr int NF = MT + NT;                   // operand fragments per group: P tiles, then Q tiles
        gif::u32x4_t sa[2][NPL][MT], sb[2][NPL][NT];  // [slot][term][tile]
        float ra[MT][8], rb[NT][8];                   // raw fragments of the group being split
        float psv[MT], qsv[NT];                       // per-sample scales of the stage being split (TAB)
        // f16x2 state per fragment (channel = lane li of the tile; lanes li and li + 32 agree): exponent, 2^e, the largest |v| it
        // holds, pending exponent change, the guard's statistics (channel maximum, smallest non-zero group maximum as bits - 1)
        int h_ex[NF], h_dl[NF];
        float h_sc[NF], h_lim[NF], h_max[NF];
        unsigned h_gmin[NF];
        bool h_need = false;
        if constexpr (H2) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                h_ex[f] = 126; h_dl[f] = 0;
                h_sc[f] = gif::h2_pow2(126); h_lim[f] = gif::kH2Limit * gif::h2_pow2(-126);
                h_max[f] = 0.f; h_gmin[f] = 0xFFFFFFFFu;
            }
        }
        auto scales = [&]() __attribute__((always_inline)) {
            if (TAB) {
                const float* row = Stab + tab_row * (BP + BQ);
#pragma unroll
                for (int i = 0; i < MT; ++i) psv[i] = row[wp0 + i * 32 + li];
#pragma unroll
                for (int j = 0; j < NT; ++j) qsv[j] = row[BP + wq0 + j * 32 + li];
                tab_rem += BKP;
                if (tab_rem >= (int)HWs) { tab_rem -= (int)HWs; ++tab_row; }
            }
        };
        auto read_raw = [&](int buf, int ks) __attribute__((always_inline)) {
            const int k0 = 16 * ks + 8 * lh;  // this lane half's 8 pixels of the group
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) ra[i][e] = Ps[buf][k0 + e][wp0 + i * 32 + li];
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) rb[j][e] = Qs[buf][k0 + e][wq0 + j * 32 + li];
        };
        // f16x2: per-sample scale, group maximum and exponent decision of fragment f of the raw group just read
        auto track = [&](int f) __attribute__((always_inline)) {
            if constexpr (H2) {
                float* v = f < MT ? ra[f] : rb[f - MT];
                if (TAB) {
                    const float sv = f < MT ? psv[f] : qsv[f - MT];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= sv;
                }
                float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fabsf(v[2]));
                m = fmaxf(fmaxf(m, fabsf(v[3])), fabsf(v[4]));
                m = fmaxf(fmaxf(m, fabsf(v[5])), fabsf(v[6]));
                m = fmaxf(m, fabsf(v[7]));
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                m = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));  // the channel's 16 pixels of this group
                h_max[f] = fmaxf(h_max[f], m);
                h_gmin[f] = min(h_gmin[f], __float_as_uint(m) - 1u);
                h_dl[f] = 0;
                if (__builtin_amdgcn_ballot_w64(m > h_lim[f]) != 0) {  // wave-uniform (scalar branch), rare after a channel's first groups
                    // per-lane update by selects: no EXEC manipulation anywhere near the MFMA stream
                    const int ne = m > h_lim[f] ? gif::h2_exp_for(__float_as_uint(m), gif::kH2Target) : h_ex[f];
                    h_dl[f] = ne - h_ex[f];
                    h_ex[f] = ne;
                    h_sc[f] = gif::h2_pow2(ne);
                    h_lim[f] = ldexpf(gif::kH2Limit, -ne);
                    h_need = true;
                }
            }
        };
        // f16x2: acc[i][j][r] *= 2^(change of its P row + change of its Q column); the row (r & 3) + 8 (r >> 2) + 4 lh of P tile i
        // lives in lane `row`, the column of Q tile j is this lane's own
        auto rescale = [&]() __attribute__((always_inline)) {
            if constexpr (H2) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int d = __builtin_amdgcn_ds_bpermute(((r & 3) + 8 * (r >> 2) + 4 * lh) * 4, h_dl[i]);
#pragma unroll
                        for (int j = 0; j < NT; ++j) acc[i][j][r] = ldexpf(acc[i][j][r], d + h_dl[MT + j]);
                    }
            }
        };
        constexpr int NPC = NF * 4;
        auto split_piece = [&](int slot, int k) __attribute__((always_inline)) {
            const int f = k / 4, e = k % 4;
            if constexpr (H2) {
                unsigned h, l;
                // scalar multiplies / subtracts: with several scales per lane hipcc keeps them in register pairs and picks the odd element
                // with `op_sel:[0,1]` on v_pk_mul_f32 / v_pk_fma_f32 — the kernel then dropped products in lanes 16-31 / 48-63 of the
                // second tile of each operand whenever two of its waves shared a SIMD (correct with one workgroup per CU, and with the
                // scale as a constant: tools/probes/h2_wgrad_debug.py); cause on the hardware side not established
                const float scv = h_sc[f];
                if (f < MT) {
                    gif::split_pair_h2_scalar(ra[f][2 * e], ra[f][2 * e + 1], scv, h, l);
                    sa[slot][0][f][e] = h; sa[slot][1][f][e] = l;
                } else {
                    gif::split_pair_h2_scalar(rb[f - MT][2 * e], rb[f - MT][2 * e + 1], scv, h, l);
                    sb[slot][0][f - MT][e] = h; sb[slot][1][f - MT][e] = l;
                }
            } else {
                unsigned h, m, l;
                if (f < MT) {
                    float x0 = ra[f][2 * e], x1 = ra[f][2 * e + 1];
                    if (TAB) { x0 *= psv[f]; x1 *= psv[f]; }
                    gif::split_pair_scalar(x0, x1, h, m, l);
                    sa[slot][0][f][e] = h; sa[slot][1][f][e] = m; sa[slot][2][f][e] = l;
                } else {
                    float x0 = rb[f - MT][2 * e], x1 = rb[f - MT][2 * e + 1];
                    if (TAB) { x0 *= qsv[f - MT]; x1 *= qsv[f - MT]; }
                    gif::split_pair_scalar(x0, x1, h, m, l);
                    sb[slot][0][f - MT][e] = h; sb[slot][1][f - MT][e] = m; sb[slot][2][f - MT][e] = l;
                }
            }
        };
        // preparation step `q` of the next group: f16x2 runs a fragment's tracking step right before its four split pieces
        constexpr int NSTEP = H2 ? NF * 5 : NPC;
        auto prep = [&](int nslot, int q) __attribute__((always_inline)) {
            if constexpr (H2) {
                if (q % 5 == 0) track(q / 5);
                else split_piece(nslot, (q / 5) * 4 + q % 5 - 1);
            } else {
                split_piece(nslot, q);
            }
        };
        constexpr int NPROD = H2 ? 3 : 6;
        constexpr int LEAD = H2 ? 3 : 6;  // MFMAs ahead of the first piece: they cover the latency of the 32 ds_read_b32 (issued as 20 ds_read2)
        auto group = [&](int slot, int nslot) __attribute__((always_inline)) {
            constexpr int TA6[6] = {2, 0, 1, 1, 0, 0}, TB6[6] = {0, 2, 1, 0, 1, 0};
            constexpr int TA3[3] = {1, 0, 0}, TB3[3] = {0, 1, 0};
            int n = 0, q = 0;
            if constexpr (H2) h_need = false;
#pragma unroll
            for (int t6 = (H2 ? 0 : GIF_X3_FIRST_TERM); t6 < NPROD; ++t6)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        if constexpr (H2)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gif::f16x8_t, sa[slot][TA3[t6]][i]),
                                                                               __builtin_bit_cast(gif::f16x8_t, sb[slot][TB3[t6]][j]),
                                                                               acc[i][j], 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gif::bf16x8_t, sa[slot][TA6[t6]][i]),
                                                                                __builtin_bit_cast(gif::bf16x8_t, sb[slot][TB6[t6]][j]),
                                                                                acc[i][j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        ++n;
                        // f16x2 has half the MFMAs to hide twice the steps under: two steps per MFMA slot
                        if (nslot >= 0 && n >= LEAD) {
#pragma unroll
                            for (int rep2 = 0; rep2 < (H2 ? 2 : 1); ++rep2)
                                if (q < NSTEP) prep(nslot, q++);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            if (nslot >= 0) {
#pragma unroll
                for (; q < NSTEP; ++q) prep(nslot, q);
                if constexpr (H2) {
                    if (h_need) rescale();
                }
            }
        };
        if (n_begin < n_end) {
            load_global(0);
            __syncthreads();
            scales();
            read_raw(0, 0);
#pragma unroll
            for (int q = 0; q < NSTEP; ++q) prep(0, q);  // (f16x2: first exponents; the accumulators are still zero)
            int cur = 0;
            for (int n0 = n_begin; n0 + BKP < n_end; n0 += BKP) {
                load_global(cur ^ 1);  // stage n0 + BKP: its buffer was last read before the previous stage's barrier
                __builtin_amdgcn_sched_barrier(0);
                read_raw(cur, 1);
                __builtin_amdgcn_sched_barrier(0);
                group(0, 1);
                __syncthreads();
                cur ^= 1;
                scales();
                read_raw(cur, 0);
                __builtin_amdgcn_sched_barrier(0);
                group(1, 0);
            }
            read_raw(cur, 1);
            __builtin_amdgcn_sched_barrier(0);
            group(0, 1);
            group(1, -1);
        }
        if constexpr (H2) {
            bool wide_p = false, wide_q = false;  // (a narrow group only costs accuracy where the other operand is narrow too: common.h)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const bool wd = (int)(__float_as_uint(h_max[f]) >> 23) - (int)((h_gmin[f] + 1u) >> 23) > gif::kH2Window;
                if (f < MT) wide_p |= wd;
                else wide_q |= wd;
            }
            if (p.gate && __builtin_amdgcn_ballot_w64(wide_p) != 0 && __builtin_amdgcn_ballot_w64(wide_q) != 0 && lane == 0)
                atomicMax(p.gate, p.gate_gen);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int er = __builtin_amdgcn_ds_bpermute(((r & 3) + 8 * (r >> 2) + 4 * lh) * 4, h_ex[i]);
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j][r] = ldexpf(acc[i][j][r], -(er + h_ex[MT + j]));
                }
        }
    } else if (n_begin < n_end) {
        load_global(0);
        store_lds(0);
        __syncthreads();
        int cur = 0;
        for (int n0 = n_begin; n0 + BKP < n_end; n0 += BKP) {
            load_global(cur ^ 1);  // stage n0 + BKP
            __builtin_amdgcn_sched_barrier(0);
            compute(cur);
            __builtin_amdgcn_sched_barrier(0);
            store_lds(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
        compute(cur);
    }

    float* out = p.ws + ((size_t)split * p.T + t) * p.RP * p.CP;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = r0 + wp0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                int col = c0 + wq0 + j * 32 + li;
                out[(size_t)row * p.CP + col] = acc[i][j][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------------------------
// f16x2 weight gradient, cooperative pre-split through LDS ("v2", round 5).  In conv_wgrad_mfma<..., 2> every wave gathers its operand
// fragments with eight ds_read_b32 each and splits them itself: every element is read and split by TWO waves (2 x 2 wave layout) and the
// loop is VALU-bound (136 VALU per 12 MFMAs).  Here a stage (32 pixels x 128 + 128 channels of fp32, landed by LDS-DMA as before) is
// converted ONCE: thread c of the 256 owns channel c (0..127: small side, 128..255: big side), reads its 32 pixels (conflict free:
// consecutive threads = consecutive channels), keeps the channel's running exponent PRIVATELY (no lane exchange), splits them into
// the two f16 terms and writes them as two 64-byte rows [channel][32 pixels] — the layout of the pre-split weights of the direct
// kernel (16-byte chunks XOR-swizzled, here with the f(row) below): an MFMA operand (8 consecutive pixels of one channel) is then ONE
// ds_read_b128 per term.  Per thread and stage: ~120 VALU + 24 LDS operations for the conversion, 16 ds_read_b128 + 24 MFMAs for the
// products (before: 272 VALU, 64 ds_read_b32).  Exponent changes travel through a small LDS array: the converting thread stores its
// channel's change, raises the stage's flag, and the MFMA waves multiply their accumulators (rows and columns) before the stage's
// products.  LDS: raw fp32 stage 32 KB (single buffer: the next stage's DMA is issued once the conversion has read it) + operand
// planes 32 KB => two workgroups per CU, whose phases interleave.  128 x 128 tiles, 4 waves as 2 x 2.
// TAPS (round 5, thin big side: the 24-channel condition-noise maps): a 128 x 32 tile per tap stages 16 KB of gy for every 4 KB of x and
// sits on the LDS-DMA fill rate (13 FLOP per staged byte: 80 TFLOP/s measured).  Here the 128 tile columns are `tpt` TAPS x Cb channels
// (5 x 24): gy is staged once per 5 taps, every column's thread gathers x at ITS tap's shift; 9 taps = 2 column tiles.
// BUF (round 6): the stage's LDS-DMA pieces go through buffer descriptors.  The small side's per-lane byte offset is loop invariant (the
// stage travels in the instruction's SGPR offset; rows behind the split's end lie outside the descriptor and land as zeros): no VALU at
// all per piece.  The big side keeps (ox, oy) per piece for the padding test and a running 32-bit byte offset that advances by constants
// (+32 pixels; + a constant at a row wrap; + another at a sample wrap): ~19 VALU per piece where the 64-bit form spent ~50 — three
// multiplies, three bounds checks, a zero-page select and the (ox, oy, b) wrap with both of its code paths; the ISA of round 5 showed
// ~400 instructions per stage and wave for the addressing next to ~180 for the conversion and 24 MFMAs.  Needs both operands within
// 3.5 GiB (checked on the host: larger launches keep the 64-bit form, BUF = false).
template <bool TAB, bool TAPS = false, bool BUF = false>
__global__ void __launch_bounds__(256, 2) conv_wgrad_h2v2(const WgradParams p) {
    constexpr int BP = 128, BQ = 128, BKP = 32, THREADS = 256, MT = 2, NT = 2, P_ROWS = 8, Q_ROWS = 8, P_IT = 4, Q_IT = 4;
    extern __shared__ __attribute__((aligned(16))) float wg_smem[];
    float (*Ps)[BP] = reinterpret_cast<float (*)[BP]>(wg_smem);                         // [32][128] fp32
    float (*Qs)[BQ] = reinterpret_cast<float (*)[BQ]>(wg_smem + BKP * BP);              // [32][128] fp32
    unsigned short* Op = reinterpret_cast<unsigned short*>(wg_smem + BKP * (BP + BQ));  // [2 terms][256 channels][32 pixels] f16
    int* dexp = reinterpret_cast<int*>(Op + 2 * 256 * 32);                              // [256] exponent change of the stage
    int* fexp = dexp + 256;                                                             // [256] final exponents
    int* flags = fexp + 256;                                                            // [0], [1]: a channel changed (stage parity); [2], [3]: narrow group on the P / Q side
    float* Stab = reinterpret_cast<float*>(flags + 8);                                  // TAB: [stab_nb][256]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wp0 = (wave >> 1) * 64, wq0 = (wave & 1) * 64;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = lid % p.tiles_pq;
    const int TG = TAPS ? p.tgroups : p.T;
    const int t = (lid / p.tiles_pq) % TG;  // TAPS: the tap GROUP
    const int split = lid / (p.tiles_pq * TG);
    const int tq = tile % p.tiles_q, tp = tile / p.tiles_q;
    const int r0 = tp * BP, c0 = TAPS ? 0 : tq * BQ;
    const bool planes = !TAPS && p.sm_plane != 0;
    // TAPS: this thread's four staged columns belong to tap t * tpt + q_tl (lanes past the last tap stage zeros)
    const int q_tl = TAPS ? ((tid % (BQ / 4)) * 4) / p.Cb : 0;
    const int q_tap = TAPS ? t * p.tpt + q_tl : t;
    const int ky = planes ? 0 : q_tap / p.KW, kx = planes ? 0 : q_tap - ky * p.KW;
    const float* const smb = static_cast<const float*>(p.sm) + (TAPS ? 0 : (size_t)t * p.sm_plane);
    const float* const bgb = static_cast<const float*>(p.bg) + (TAPS ? 0 : (size_t)t * p.bg_plane);
    const float* const pzero = static_cast<const float*>(p.zero);
    const int n_begin = (int)((long)split * p.chunk);
    int n_end = n_begin + (int)p.chunk;
    if (n_end > (int)p.Ntot) n_end = (int)p.Ntot;
    const unsigned HWs = (unsigned)(p.Hs * p.Ws);

    const int p_row = tid / (BP / 4), p_ch = r0 + (tid % (BP / 4)) * 4;
    const int q_row = tid / (BQ / 4), q_ch = TAPS ? (tid % (BQ / 4)) * 4 - q_tl * p.Cb : c0 + (tid % (BQ / 4)) * 4;
    const bool p_ch_ok = p_ch < p.Cs, q_ch_ok = TAPS ? (q_tl < p.tpt && q_tap < p.T) : q_ch < p.Cb;
    int tab_row = 0, tab_rem = 0;
    if (tid < 8) flags[tid] = 0;
    if (TAB) {
        const int b_first = n_begin / (int)HWs;
        for (int e = tid; e < p.stab_nb * (BP + BQ); e += THREADS) {
            int bl = e / (BP + BQ), c = e - bl * (BP + BQ);
            int b = b_first + bl;
            float v = 0.f;
            if (b < p.B) {
                if (c < BP) v = (r0 + c < p.Cs) ? (p.ss ? p.ss[(size_t)b * p.Cs + r0 + c] : 1.f) : 0.f;
                else v = (c0 + c - BP < p.Cb) ? (p.bs ? p.bs[(size_t)b * p.Cb + c0 + c - BP] : 1.f) : 0.f;
            }
            Stab[e] = v;
        }
        tab_rem = n_begin - b_first * (int)HWs;
    }
    int p_n[P_IT], q_n[Q_IT], q_b[Q_IT], q_oy[Q_IT], q_ox[Q_IT];
    unsigned p_vo[P_IT], q_bo[Q_IT];  // BUF: byte offsets (small side: loop invariant; big side: running)
#pragma unroll
    for (int it = 0; it < P_IT; ++it) {
        p_n[it] = n_begin + p_row + it * P_ROWS;
        p_vo[it] = p_ch_ok ? (unsigned)(p_n[it] * p.Cs + p_ch) * 4u : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int it = 0; it < Q_IT; ++it) {
        int n = n_begin + q_row + it * Q_ROWS;
        q_n[it] = n;
        unsigned b = (unsigned)n / HWs;
        unsigned r = (unsigned)n - b * HWs;
        unsigned oy = r / (unsigned)p.Ws;
        q_b[it] = (int)b; q_oy[it] = (int)oy; q_ox[it] = (int)(r - oy * (unsigned)p.Ws);
        q_bo[it] = (unsigned)((((int)b * p.Hb + q_oy[it] * p.stride + ky - p.pad) * p.Wb + q_ox[it] * p.stride + kx - p.pad) * p.Cb + q_ch) * 4u;
    }
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // BUF: descriptors and the constants of the running offsets (all wave-uniform).  Small side: rows >= n_end are outside; big side: the
    // whole tensor (a stage only reaches past n_end in the LAST split, where those rows belong to sample B: outside) — planes mode: the
    // plane up to n_end (its padding rows are uninitialised memory)
    const gif::buf_rsrc_t rs_p = gif::make_buf_rsrc(smb, (unsigned)n_end * (unsigned)p.Cs * 4u);
    const gif::buf_rsrc_t rs_q = gif::make_buf_rsrc(bgb, planes ? (unsigned)n_end * (unsigned)p.Cb * 4u
                                                                  : (unsigned)p.B * (unsigned)p.Hb * (unsigned)p.Wb * (unsigned)p.Cb * 4u);
    const unsigned q_step = (unsigned)(BKP * p.stride * p.Cb) * 4u;
    const unsigned q_drow = (unsigned)((p.Wb - p.Ws) * p.stride * p.Cb) * 4u;           // on top of the linear advance, mod 2^32
    const unsigned q_dsmp = (unsigned)((p.Hb - p.Hs * p.stride) * p.Wb * p.Cb) * 4u;
    const int q_cy = ky - p.pad, q_cx = kx - p.pad;
    int p_soff = 0;
    auto load_global = [&]() __attribute__((always_inline)) {
        if constexpr (BUF) {
#pragma unroll
            for (int it = 0; it < P_IT; ++it)
                gif::buf_load_lds16(rs_p, (lptr_t)(&Ps[it * P_ROWS][0] + wave_u * 256), p_vo[it], p_soff);
            p_soff += BKP * p.Cs * 4;
#pragma unroll
            for (int it = 0; it < Q_IT; ++it) {
                const int iy = q_oy[it] * p.stride + q_cy, ix = q_ox[it] * p.stride + q_cx;
                const bool ok = q_ch_ok && (unsigned)iy < (unsigned)p.Hb && (unsigned)ix < (unsigned)p.Wb;
                gif::buf_load_lds16(rs_q, (lptr_t)(&Qs[it * Q_ROWS][0] + wave_u * 256), ok ? q_bo[it] : 0xFFFFFFFFu, 0);
                q_bo[it] += q_step;
                q_ox[it] += BKP;
                if (p.Ws >= BKP) {  // uniform: at most one row wrap per stage
                    const bool wx = q_ox[it] >= p.Ws;
                    q_ox[it] -= wx ? p.Ws : 0;
                    q_oy[it] += wx ? 1 : 0;
                    q_bo[it] += wx ? q_drow : 0u;
                    const bool wy = q_oy[it] >= p.Hs;
                    q_oy[it] -= wy ? p.Hs : 0;
                    q_bo[it] += wy ? q_dsmp : 0u;
                } else {
                    while (q_ox[it] >= p.Ws) { q_ox[it] -= p.Ws; ++q_oy[it]; q_bo[it] += q_drow; }
                    while (q_oy[it] >= p.Hs) { q_oy[it] -= p.Hs; q_bo[it] += q_dsmp; }
                }
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            const bool ok = p_ch_ok && p_n[it] < n_end;
            const float* g = ok ? smb + (p_n[it] * p.Cs + p_ch) : pzero;
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(&Ps[it * P_ROWS][0] + wave_u * 256), 16, 0, 0);
            p_n[it] += BKP;
        }
#pragma unroll
        for (int it = 0; it < Q_IT; ++it) {
            const int iy = q_oy[it] * p.stride + ky - p.pad, ix = q_ox[it] * p.stride + kx - p.pad;
            const bool ok = q_ch_ok && q_n[it] < n_end && (unsigned)iy < (unsigned)p.Hb && (unsigned)ix < (unsigned)p.Wb;
            const float* g = ok ? bgb + (((q_b[it] * p.Hb + iy) * p.Wb + ix) * p.Cb + q_ch) : pzero;
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(&Qs[it * Q_ROWS][0] + wave_u * 256), 16, 0, 0);
            q_n[it] += BKP;
            q_ox[it] += BKP;
            if (p.Ws >= BKP) {  // uniform: at most one row wrap per stage
                const bool wx = q_ox[it] >= p.Ws;
                q_ox[it] -= wx ? p.Ws : 0;
                q_oy[it] += wx ? 1 : 0;
                const bool wy = q_oy[it] >= p.Hs;
                q_oy[it] -= wy ? p.Hs : 0;
                q_b[it] += wy ? 1 : 0;
            } else {
                while (q_ox[it] >= p.Ws) { q_ox[it] -= p.Ws; ++q_oy[it]; }
                while (q_oy[it] >= p.Hs) { q_oy[it] -= p.Hs; ++q_b[it]; }
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

    // ---- this thread's channel (conversion role): running exponent and the guard's statistics, all private
    const float* const rawcol = tid < 128 ? &Ps[0][tid] : &Qs[0][tid - 128];
    unsigned short* const oprow = Op + tid * 32;
    // 16-byte chunk swizzle of the operand planes: f(row) = bit 2 | (bit 1 ^ bit 3) << 1 of the row index.  ds_write_b128 is serviced in groups
    // of 8 CONTIGUOUS lanes on 32 banks (two 64-byte rows per bank window), ds_read_b128 in the four non-contiguous 16-lane groups of the
    // microarchitecture guide's LDS table on 64 banks; this f puts both on distinct 16-byte slots (the direct kernel's (row >> 2) & 3 is
    // conflict free for the reads only: the conversion's stores ran 2-way, 109 M conflict cycles per launch in the round's PMC pass)
    const int wsw = ((tid >> 2) & 1) | ((((tid >> 1) ^ (tid >> 3)) & 1) << 1);
    int h_ex = 126;
    float h_sc = gif::h2_pow2(126), h_lim = gif::kH2Limit * gif::h2_pow2(-126), h_max = 0.f;
    unsigned h_gmin = 0xFFFFFFFFu;
    auto convert = [&](int par) __attribute__((always_inline)) {
        float v[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) v[e] = rawcol[e * 128];
        if (TAB) {
            const float sv = Stab[tab_row * (BP + BQ) + tid];
#pragma unroll
            for (int e = 0; e < 32; ++e) v[e] *= sv;
            tab_rem += BKP;
            if (tab_rem >= (int)HWs) { tab_rem -= (int)HWs; ++tab_row; }
        }
        float m0 = 0.f, m1 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { m0 = fmaxf(m0, fabsf(v[e])); m1 = fmaxf(m1, fabsf(v[16 + e])); }
        const float m = fmaxf(m0, m1);
        h_max = fmaxf(h_max, m);
        h_gmin = min(h_gmin, min(__float_as_uint(m0) - 1u, __float_as_uint(m1) - 1u));
        int d = 0;
        if (m > h_lim) {  // this channel outgrew its exponent (private state: plain divergence; rare after the first stages)
            const int ne = gif::h2_exp_for(__float_as_uint(m), gif::kH2Target);
            d = ne - h_ex;
            h_ex = ne;
            h_sc = gif::h2_pow2(ne);
            h_lim = ldexpf(gif::kH2Limit, -ne);
            atomicOr(&flags[par], 1);
        }
        dexp[tid] = d;
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // 8 pixels = one 16-byte chunk of each term
            gif::u32x4_t hi4, lo4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                unsigned h, l;
                gif::split_pair_h2(v[8 * j + 2 * k], v[8 * j + 2 * k + 1], h_sc, h, l);
                hi4[k] = h; lo4[k] = l;
            }
            const int phys = (j ^ wsw) << 3;
            *reinterpret_cast<gif::u32x4_t*>(oprow + phys) = hi4;
            *reinterpret_cast<gif::u32x4_t*>(oprow + 256 * 32 + phys) = lo4;
        }
    };
    // ---- product role
    const int rsw = ((li >> 2) & 1) | ((((li >> 1) ^ (li >> 3)) & 1) << 1);  // f(row) of every operand row this lane reads (tile bases are multiples of 32)
    auto products = [&](int par) __attribute__((always_inline)) {
        if (flags[par] != 0) {  // workgroup-uniform: some channel of this stage changed its exponent: acc *= 2^(row change + column change)
            int dc[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) dc[j] = dexp[128 + wq0 + j * 32 + li];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = dexp[wp0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j][r] = ldexpf(acc[i][j][r], dr + dc[j]);
                }
        }
        if (tid == 0) flags[par ^ 1] = 0;  // the next stage's conversions raise it again (they start behind the next barrier)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            gif::u32x4_t sa[2][MT], sb[2][NT];
            const int ch = ((2 * g + lh) ^ rsw) << 3;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
                for (int i = 0; i < MT; ++i) sa[tt][i] = *reinterpret_cast<const gif::u32x4_t*>(Op + (tt * 256 + wp0 + i * 32 + li) * 32 + ch);
#pragma unroll
                for (int j = 0; j < NT; ++j) sb[tt][j] = *reinterpret_cast<const gif::u32x4_t*>(Op + (tt * 256 + 128 + wq0 + j * 32 + li) * 32 + ch);
            }
            constexpr int TA3[3] = {1, 0, 0}, TB3[3] = {0, 1, 0};  // lo*hi, hi*lo, hi*hi
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gif::f16x8_t, sa[TA3[t3]][i]),
                                                                           __builtin_bit_cast(gif::f16x8_t, sb[TB3[t3]][j]), acc[i][j], 0, 0, 0);
        }
    };

    if (n_begin < n_end) {
        load_global();
        int par = 0;
        for (int n0 = n_begin; n0 < n_end; n0 += BKP) {
            __syncthreads();  // the stage's fp32 tiles have landed (vmcnt(0) of every wave); the previous stage's operand reads are done
            convert(par);
#ifdef GIF_WGRAD_KX3_PROBE  // timing probe (tools/probes/wgrad_kx3_probe.sh; results are WRONG): what a workgroup that forms the THREE dx taps of a
            // kernel row from ONE staged pair of tiles would cost — a third of the operand traffic (19 GB through L2 per 128 -> 128 launch
            // today), 512 channel rows converted per three taps instead of 768 (gy once, the activations at three shifts), 3 x 24 MFMAs
            // per stage.  The host launches a third of the tap groups.
            if (!TAPS) convert(par);
#endif
            __syncthreads();  // operand planes complete; the raw buffer is free again
#ifdef GIF_WGRAD_NO_DMA_PROBE  // timing probe (results wrong): the K loop on stale raw data — what the per-stage DMA issue + landing costs
            if (n0 == n_begin)
#endif
            if (n0 + BKP < n_end) load_global();
            __builtin_amdgcn_sched_barrier(0);
            products(par);
#ifdef GIF_WGRAD_KX3_PROBE
            if (!TAPS) { products(par); products(par); }
#endif
            par ^= 1;
        }
    }
    // final exponents and the guard (a narrow group only costs accuracy where the other side is narrow too: common.h)
    __syncthreads();
    fexp[tid] = h_ex;
    if ((int)(__float_as_uint(h_max) >> 23) - (int)((h_gmin + 1u) >> 23) > gif::kH2Window) atomicOr(&flags[tid < 128 ? 2 : 3], 1);
    __syncthreads();
    if (tid == 0 && p.gate && flags[2] != 0 && flags[3] != 0) atomicMax(p.gate, p.gate_gen);
    int ec[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) ec[j] = fexp[128 + wq0 + j * 32 + li];
    float* out = p.ws + ((size_t)split * p.T + (TAPS ? 0 : t)) * p.RP * p.CP;
    long ocol[NT];  // column offset inside the split's [T][RP][CP] block (TAPS: the column's own tap plane), < 0: padding column
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int c = wq0 + j * 32 + li;
        if (TAPS) {
            const int tl = c / p.Cb, tap = t * p.tpt + tl;
            ocol[j] = (tl < p.tpt && tap < p.T) ? (long)tap * p.RP * p.CP + (c - tl * p.Cb) : -1;
        } else {
            ocol[j] = c0 + c;
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = wp0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int er = fexp[rl];
#pragma unroll
            for (int j = 0; j < NT; ++j)
                if (!TAPS || ocol[j] >= 0) out[(size_t)(r0 + rl) * p.CP + ocol[j]] = ldexpf(acc[i][j][r], -(er + ec[j]));
        }
}

template <typename T, int BP, int BQ, int WP_, int WQ_, bool GLDS, int BKP, bool TAB = false, int X3 = 0>
void wgrad_launch(dim3 grid, int threads, hipStream_t s, const WgradParams& p) {
    static gif::LdsAttr attr;
    const size_t lds = (size_t)2 * BKP * (BP + BQ) * sizeof(T) + (size_t)(TAB ? p.stab_nb * (BP + BQ) : 0) * sizeof(float);
    auto kern = conv_wgrad_mfma<T, BP, BQ, WP_, WQ_, GLDS, BKP, TAB, X3>;
    attr.ensure(reinterpret_cast<const void*>(kern), lds);
    hipLaunchKernelGGL(kern, grid, dim3(threads), lds, s, p);
}

// GIF_H2_WGRAD_V2=0: the per-wave-split f16x2 kernel (conv_wgrad_mfma<..., 2>) instead of the cooperative pre-split one (A/B)
inline bool h2v2_on() {
    static const int on = gif::knob("GIF_H2_WGRAD_V2") ? atoi(gif::knob("GIF_H2_WGRAD_V2")) != 0 : 1;
    return on != 0;
}
// GIF_H2_WGRAD_TAPS=0: the thin-big-side layers on the 128 x 32 per-tap tiles (A/B)
inline bool thin_taps_on() {
    static const int on = gif::knob("GIF_H2_WGRAD_TAPS") ? atoi(gif::knob("GIF_H2_WGRAD_TAPS")) != 0 : 1;
    return on != 0;
}
// taps per 128-column tile and the number of column tiles for a thin big side of Cb (<= 32) channels
inline void thin_tap_tiles(int Cb, int T, int* tpt, int* tgroups) {
    int n = 128 / Cb;
    if (n > T) n = T;
    *tpt = n;
    *tgroups = (T + n - 1) / n;
}
// ------------------------------------------------------------------------------------------------------------------
// f16x2 weight gradient, third form (round 6): NO raw LDS stage.  profiles/r6_wgrad_nodma_probe.txt: 28-30 % of conv_wgrad_h2v2's time
// is its per-stage LDS-DMA — the raw fp32 stage is single-buffered (LDS budget of two workgroups per CU), so the DMA of stage s + 1 can
// only be issued once the conversion has read stage s and has to land before the next conversion starts.  Here the converting threads
// fetch their operands straight into REGISTERS with buffer_load_dwordx4, a whole stage ahead (8 loads = 32 VGPRs per thread and stage,
// double buffered): thread (cg = tid / 4, po = tid % 4) owns the 4 channels 4 cg .. 4 cg + 3 of the stage (waves 0-1: small side, waves
// 2-3: big side — roles are wave-uniform) and the 8 pixels 8 po .. 8 po + 7.  The four threads of a quad hold the 32 pixels of the same
// channels: they agree on the group maxima with two quad-permute DPP steps (16-pixel K group maxima for the guard, the 32-pixel maximum
// for the running exponent) and keep identical private copies of the channel state.  8 pixels of one channel = one 16-byte chunk of
// each f16 term: the operand planes [term][channel][32 pixels] and their swizzle are v2's (its product phase is reused unchanged), as are
// the exponent-change protocol through dexp / flags, the final descale, the guard and the partial-sum workspace.  LDS: 32 KB of planes
// per workgroup.  Addresses: buffer descriptors as in v2 — small side: loop-invariant byte offsets + SGPR stage offset; big side: (ox,
// oy) and a running offset of the thread's FIRST pixel per stage, the other 7 derived with at most one row wrap (host: Ws >= 32).
// Not for the grouped-tap thin layers (conv_wgrad_h2v2<false, true>) and operands beyond 3.5 GiB: those keep v2.
// ------------------------------------------------------------------------------------------------------------------
// The prefetch loads are inline assembly with hand-placed s_waitcnt: as ordinary loads hipcc either sank them next to their use or drained the
// younger stage with a vmcnt(0) ahead of every conversion (its wait insertion merges the two loop halves conservatively) — the prefetch was gone.
// Descriptor words by hand (raw buffer, stride 0): {base lo, base hi, bytes, 0x00020000}.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 rsrc_words(const void* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    return i32x4{(int)(unsigned)a, (int)(unsigned)(a >> 32), (int)bytes, 0x00020000};
}
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void asm_buf_load_x4(f32x4& dst, const i32x4 rsrc, unsigned voff, int soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// wait until at most N vector-memory operations are in flight.  NOT with the fragments as "+v" operands (first version): hipcc then copied
// them into fresh operand registers AHEAD of the wait on one path — a read of registers whose loads had not landed (wrong results whenever
// two workgroups shared a CU).  A scheduling barrier keeps every later instruction behind the statement instead (checked in the ISA).
template <int N>
__device__ __forceinline__ void asm_wait_vm(f32x4 (&)[8]) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
#else
__device__ __forceinline__ void asm_buf_load_x4(f32x4&, const i32x4, unsigned, int) {}
template <int N>
__device__ __forceinline__ void asm_wait_vm(f32x4 (&)[8]) {}
#endif

template <bool TAB>
__global__ void __launch_bounds__(256, 2) conv_wgrad_h2v3(const WgradParams p) {
    constexpr int BP = 128, BQ = 128, BKP = 32, MT = 2, NT = 2;
    extern __shared__ __attribute__((aligned(16))) float wg_smem[];
    unsigned short* Op = reinterpret_cast<unsigned short*>(wg_smem);  // [2 terms][256 channels][32 pixels] f16
    int* dexp = reinterpret_cast<int*>(Op + 2 * 256 * 32);            // [256] exponent change of the stage
    int* fexp = dexp + 256;                                           // [256] final exponents
    int* flags = fexp + 256;                                          // [0], [1]: a channel changed (stage parity); [2], [3]: narrow group on the P / Q side
    float* Stab = reinterpret_cast<float*>(flags + 8);                // TAB: [stab_nb][256]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wp0 = (wave >> 1) * 64, wq0 = (wave & 1) * 64;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = lid % p.tiles_pq;
    const int t = (lid / p.tiles_pq) % p.T;
    const int split = lid / (p.tiles_pq * p.T);
    const int tq = tile % p.tiles_q, tp = tile / p.tiles_q;
    const int r0 = tp * BP, c0 = tq * BQ;
    const bool planes = p.sm_plane != 0;
    const int ky = planes ? 0 : t / p.KW, kx = planes ? 0 : t - ky * p.KW;
    const float* const smb = static_cast<const float*>(p.sm) + (size_t)t * p.sm_plane;
    const float* const bgb = static_cast<const float*>(p.bg) + (size_t)t * p.bg_plane;
    const int n_begin = (int)((long)split * p.chunk);
    int n_end = n_begin + (int)p.chunk;
    if (n_end > (int)p.Ntot) n_end = (int)p.Ntot;
    const unsigned HWs = (unsigned)(p.Hs * p.Ws);

    // ---- conversion role: 4 channels x 8 pixels per thread and stage
    const int cg = tid >> 2, po = tid & 3;
    const bool is_q = __builtin_amdgcn_readfirstlane(wave) >= 2;  // waves 2, 3 convert the big side
    const int ch4 = (is_q ? cg - 32 : cg) * 4;                    // first of the 4 channels inside the side's 128-wide tile
    const int gch = (is_q ? c0 : r0) + ch4;
    const bool ch_ok = gch < (is_q ? p.Cb : p.Cs);                // channel counts are multiples of 4: all four or none
    const int row0 = (is_q ? 128 : 0) + ch4;                      // plane row of channel 0 of this thread
    int tab_row = 0, tab_rem = 0;
    if (tid < 8) flags[tid] = 0;
    if (TAB) {
        const int b_first = n_begin / (int)HWs;
        for (int e = tid; e < p.stab_nb * (BP + BQ); e += 256) {
            int bl = e / (BP + BQ), c = e - bl * (BP + BQ);
            int b = b_first + bl;
            float v = 0.f;
            if (b < p.B) {
                if (c < BP) v = (r0 + c < p.Cs) ? (p.ss ? p.ss[(size_t)b * p.Cs + r0 + c] : 1.f) : 0.f;
                else v = (c0 + c - BP < p.Cb) ? (p.bs ? p.bs[(size_t)b * p.Cb + c0 + c - BP] : 1.f) : 0.f;
            }
            Stab[e] = v;
        }
        tab_rem = n_begin - b_first * (int)HWs;
    }
    const i32x4 rs_p = rsrc_words(smb, (unsigned)n_end * (unsigned)p.Cs * 4u);
    const i32x4 rs_q = rsrc_words(bgb, planes ? (unsigned)n_end * (unsigned)p.Cb * 4u
                                                 : (unsigned)p.B * (unsigned)p.Hb * (unsigned)p.Wb * (unsigned)p.Cb * 4u);
    // small side: byte offset of the thread's first pixel (loop invariant; pixel i adds i rows, the stage travels in the SGPR offset)
    const unsigned p_v0 = ch_ok ? (unsigned)((n_begin + 8 * po) * p.Cs + gch) * 4u : 0xFFFFFFFFu;
    const unsigned p_row = (unsigned)p.Cs * 4u;
    int p_soff = 0;
    // big side: (oy, ox) and running byte offset of the thread's first pixel of the stage
    int q_oy, q_ox;
    unsigned q_bo;
    {
        const unsigned n = (unsigned)(n_begin + 8 * po);
        const unsigned b = n / HWs, r = n - b * HWs;
        const unsigned oy = r / (unsigned)p.Ws;
        q_oy = (int)oy; q_ox = (int)(r - oy * (unsigned)p.Ws);
        q_bo = (unsigned)((((int)b * p.Hb + q_oy * p.stride + ky - p.pad) * p.Wb + q_ox * p.stride + kx - p.pad) * p.Cb + gch) * 4u;
    }
    const unsigned q_px = (unsigned)(p.stride * p.Cb) * 4u;                                 // one pixel along the row
    const unsigned q_drow = (unsigned)((p.Wb - p.Ws) * p.stride * p.Cb) * 4u;               // extra at a row wrap (mod 2^32)
    const unsigned q_dsmp = (unsigned)((p.Hb - p.Hs * p.stride) * p.Wb * p.Cb) * 4u;        // extra at a sample wrap
    const int q_cy = ky - p.pad, q_cx = kx - p.pad;

    f32x4 raw[2][8];
    auto load_stage = [&](auto buf_tag) __attribute__((always_inline)) {
        constexpr int bf = decltype(buf_tag)::value;
        if (!is_q) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm_buf_load_x4(raw[bf][i], rs_p, p_v0 == 0xFFFFFFFFu ? p_v0 : p_v0 + (unsigned)i * p_row, p_soff);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int ox = q_ox + i, oy = q_oy;
                unsigned off = q_bo + (unsigned)i * q_px;
                const bool wx = ox >= p.Ws;           // Ws >= 32: at most one row wrap inside the 8 pixels
                ox -= wx ? p.Ws : 0;
                oy += wx ? 1 : 0;
                off += wx ? q_drow : 0u;
                const bool wy = oy >= p.Hs;
                oy -= wy ? p.Hs : 0;
                off += wy ? q_dsmp : 0u;
                const int iy = oy * p.stride + q_cy, ix = ox * p.stride + q_cx;
                const bool ok = ch_ok && (unsigned)iy < (unsigned)p.Hb && (unsigned)ix < (unsigned)p.Wb;
                asm_buf_load_x4(raw[bf][i], rs_q, ok ? off : 0xFFFFFFFFu, 0);
            }
        }
        // advance to the next stage — in BOTH roles, unconditionally: updates of captured variables inside the role branch made hipcc
        // route them through a selected stack address (a scratch load + s_waitcnt vmcnt(0) per stage: the prefetch was drained)
        p_soff += BKP * p.Cs * 4;
        q_bo += (unsigned)BKP * q_px;
        q_ox += BKP;
        const bool wx = q_ox >= p.Ws;  // (32 pixels: at most one row wrap)
        q_ox -= wx ? p.Ws : 0;
        q_oy += wx ? 1 : 0;
        q_bo += wx ? q_drow : 0u;
        const bool wy = q_oy >= p.Hs;
        q_oy -= wy ? p.Hs : 0;
        q_bo += wy ? q_dsmp : 0u;
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // private channel state (identical in the four threads of a quad)
    int h_ex[4];
    float h_sc[4], h_lim[4], h_max[4];
    unsigned h_gmin[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        h_ex[k] = 126; h_sc[k] = gif::h2_pow2(126); h_lim[k] = gif::kH2Limit * gif::h2_pow2(-126); h_max[k] = 0.f; h_gmin[k] = 0xFFFFFFFFu;
    }
    auto quad_x1 = [](float v) __attribute__((always_inline)) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); };
    auto quad_x2 = [](float v) __attribute__((always_inline)) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true)); };
    auto convert = [&](auto buf_tag, int par) __attribute__((always_inline)) {
        constexpr int bf = decltype(buf_tag)::value;
        float sv[4] = {1.f, 1.f, 1.f, 1.f};
        if (TAB) {
#pragma unroll
            for (int k = 0; k < 4; ++k) sv[k] = Stab[tab_row * (BP + BQ) + row0 + k];
            tab_rem += BKP;
            if (tab_rem >= (int)HWs) { tab_rem -= (int)HWs; ++tab_row; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = TAB ? raw[bf][i][k] * sv[k] : raw[bf][i][k];
            float m8 = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fabsf(v[2]));
            m8 = fmaxf(fmaxf(m8, fabsf(v[3])), fabsf(v[4]));
            m8 = fmaxf(fmaxf(m8, fabsf(v[5])), fabsf(v[6]));
            m8 = fmaxf(m8, fabsf(v[7]));
            const float m16 = fmaxf(m8, quad_x1(m8));   // this thread's 16-pixel K group (pixel octets 2 (po / 2), + 1)
            const float mo16 = quad_x2(m16);            // the stage's other K group
            const float m = fmaxf(m16, mo16);
            h_max[k] = fmaxf(h_max[k], m);
            h_gmin[k] = min(h_gmin[k], min(__float_as_uint(m16) - 1u, __float_as_uint(mo16) - 1u));
            int d = 0;
            if (m > h_lim[k]) {  // the channel outgrew its exponent (all four threads of the quad take the branch together)
                const int ne = gif::h2_exp_for(__float_as_uint(m), gif::kH2Target);
                d = ne - h_ex[k];
                h_ex[k] = ne;
                h_sc[k] = gif::h2_pow2(ne);
                h_lim[k] = ldexpf(gif::kH2Limit, -ne);
                atomicOr(&flags[par], 1);
            }
            const int row = row0 + k;
            if (po == 0) dexp[row] = d;
            gif::u32x4_t hi4, lo4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned h, l;
                gif::split_pair_h2_scalar(v[2 * e], v[2 * e + 1], h_sc[k], h, l);
                hi4[e] = h; lo4[e] = l;
            }
            const int wsw = ((row >> 2) & 1) | ((((row >> 1) ^ (row >> 3)) & 1) << 1);  // v2's plane swizzle f(row)
            unsigned short* const oprow = Op + row * 32 + ((po ^ wsw) << 3);
            *reinterpret_cast<gif::u32x4_t*>(oprow) = hi4;
            *reinterpret_cast<gif::u32x4_t*>(oprow + 256 * 32) = lo4;
        }
    };
    // ---- product role (v2's)
    const int rsw = ((li >> 2) & 1) | ((((li >> 1) ^ (li >> 3)) & 1) << 1);
    auto products = [&](int par) __attribute__((always_inline)) {
        if (flags[par] != 0) {
            int dc[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) dc[j] = dexp[128 + wq0 + j * 32 + li];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = dexp[wp0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j][r] = ldexpf(acc[i][j][r], dr + dc[j]);
                }
        }
        if (tid == 0) flags[par ^ 1] = 0;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            gif::u32x4_t sa[2][MT], sb[2][NT];
            const int ch = ((2 * g + lh) ^ rsw) << 3;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
                for (int i = 0; i < MT; ++i) sa[tt][i] = *reinterpret_cast<const gif::u32x4_t*>(Op + (tt * 256 + wp0 + i * 32 + li) * 32 + ch);
#pragma unroll
                for (int j = 0; j < NT; ++j) sb[tt][j] = *reinterpret_cast<const gif::u32x4_t*>(Op + (tt * 256 + 128 + wq0 + j * 32 + li) * 32 + ch);
            }
            constexpr int TA3[3] = {1, 0, 0}, TB3[3] = {0, 1, 0};
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gif::f16x8_t, sa[TA3[t3]][i]),
                                                                           __builtin_bit_cast(gif::f16x8_t, sb[TB3[t3]][j]), acc[i][j], 0, 0, 0);
        }
    };

    if (n_begin < n_end) {
        __syncthreads();  // flags / Stab initialised
        load_stage(std::integral_constant<int, 0>{});
        int par = 0;
        for (int n0 = n_begin; n0 < n_end; n0 += 2 * BKP) {
            // stage n0 from raw[0]; its successor is requested first and stays in flight behind the conversion, the barrier and the products
            if (n0 + BKP < n_end) {
                load_stage(std::integral_constant<int, 1>{});
#ifdef GIF_V3_DEBUG_WAIT0
                asm_wait_vm<0>(raw[0]);
#else
                asm_wait_vm<8>(raw[0]);
#endif
            } else {
                asm_wait_vm<0>(raw[0]);
            }
            convert(std::integral_constant<int, 0>{}, par);
            __syncthreads();  // operand planes of this stage complete
            products(par);
            par ^= 1;
            __syncthreads();  // every wave has read the planes: the next conversion may overwrite them
            if (n0 + BKP < n_end) {
                if (n0 + 2 * BKP < n_end) {
                    load_stage(std::integral_constant<int, 0>{});
#ifdef GIF_V3_DEBUG_WAIT0
                    asm_wait_vm<0>(raw[1]);
#else
                    asm_wait_vm<8>(raw[1]);
#endif
                } else {
                    asm_wait_vm<0>(raw[1]);
                }
                convert(std::integral_constant<int, 1>{}, par);
                __syncthreads();
                products(par);
                par ^= 1;
                __syncthreads();
            }
        }
    }
    // final exponents and the guard
    __syncthreads();
    if (po == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            fexp[row0 + k] = h_ex[k];
            if ((int)(__float_as_uint(h_max[k]) >> 23) - (int)((h_gmin[k] + 1u) >> 23) > gif::kH2Window) atomicOr(&flags[is_q ? 3 : 2], 1);
        }
    }
    __syncthreads();
    if (tid == 0 && p.gate && flags[2] != 0 && flags[3] != 0) atomicMax(p.gate, p.gate_gen);
    int ec[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) ec[j] = fexp[128 + wq0 + j * 32 + li];
    float* out = p.ws + ((size_t)split * p.T + t) * p.RP * p.CP;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = wp0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int er = fexp[rl];
#pragma unroll
            for (int j = 0; j < NT; ++j) out[(size_t)(r0 + rl) * p.CP + c0 + wq0 + j * 32 + li] = ldexpf(acc[i][j][r], -(er + ec[j]));
        }
}

// buffer-addressed DMA of conv_wgrad_h2v2: both operands (per plane in planes mode) within 3.5 GiB; GIF_H2_WGRAD_BUF=0: the 64-bit form (A/B)
inline bool h2v2_buf_ok(const WgradParams& p) {
    static const int off = gif::knob("GIF_H2_WGRAD_BUF") ? atoi(gif::knob("GIF_H2_WGRAD_BUF")) == 0 : 0;
    // byte offsets are unsigned 32-bit; rows a stage reaches past the end of a tensor (the last split's tail, < 32 pixels) must stay out of
    // range without wrapping: 512 MiB of headroom below 4 GiB (the 64-sample discriminator pass on the blurred 257^2 map is 2.16 GB)
    const long lim = (1L << 32) - (1L << 29);
    return !off && (long)p.B * p.Hs * p.Ws * p.Cs * 4 <= lim && (long)p.B * p.Hb * p.Wb * p.Cb * 4 <= lim;
}
template <bool TAB, bool TAPS>
inline void wgrad_launch_v2_t(dim3 grid, size_t lds, hipStream_t s, const WgradParams& p) {
    static gif::LdsAttr attr[2];
    if (h2v2_buf_ok(p)) {
        attr[1].ensure(reinterpret_cast<const void*>(conv_wgrad_h2v2<TAB, TAPS, true>), lds);
        hipLaunchKernelGGL((conv_wgrad_h2v2<TAB, TAPS, true>), grid, dim3(256), lds, s, p);
    } else {
        attr[0].ensure(reinterpret_cast<const void*>(conv_wgrad_h2v2<TAB, TAPS, false>), lds);
        hipLaunchKernelGGL((conv_wgrad_h2v2<TAB, TAPS, false>), grid, dim3(256), lds, s, p);
    }
}
// conv_wgrad_h2v3 (operands prefetched into registers, no raw LDS stage): everything v2's buffer form takes except the grouped-tap thin
// layers, with Ws >= 32 (one row wrap per 8 pixels / per stage).  OFF by default — GIF_H2_WGRAD_V3=1 (under GIF_EXPERIMENTAL=1) selects it:
// +2-5 % per launch in isolation, nothing measurable in the step (178.7 / 179.1 ms with v2, 180.2 / 177.7 with v3, alternating arms:
// profiles/r6_wgrad_v3_probe.txt), and v2 is the kernel every parity test of rounds 5-6 ran on.
inline bool h2v3_ok(const WgradParams& p, bool taps) {
    static const int on = gif::knob("GIF_H2_WGRAD_V3") ? atoi(gif::knob("GIF_H2_WGRAD_V3")) != 0 : 0;
    return on && !taps && h2v2_buf_ok(p) && p.Ws >= 32;
}
// (measured, profiles/r6_wgrad_v3_probe.txt: +2-5 % on the un-modulated launches and the Winograd plane GEMMs, -2-3 % on the modulated ones,
// whose per-sample scale multiplies sit in the conversion: those keep v2)
inline void wgrad_launch_v2(bool tab, dim3 grid, hipStream_t s, const WgradParams& p, bool taps = false) {
    if (!tab && h2v3_ok(p, taps)) {
        static gif::LdsAttr attr3[2];
        const size_t lds3 = (size_t)2 * 256 * 32 * 2 + (size_t)(512 + 8) * 4 + (size_t)(tab ? p.stab_nb * 256 : 0) * sizeof(float);
        if (tab) {
            attr3[1].ensure(reinterpret_cast<const void*>(conv_wgrad_h2v3<true>), lds3);
            hipLaunchKernelGGL(conv_wgrad_h2v3<true>, grid, dim3(256), lds3, s, p);
        } else {
            attr3[0].ensure(reinterpret_cast<const void*>(conv_wgrad_h2v3<false>), lds3);
            hipLaunchKernelGGL(conv_wgrad_h2v3<false>, grid, dim3(256), lds3, s, p);
        }
        return;
    }
    const size_t lds = (size_t)32 * 256 * 4 + (size_t)2 * 256 * 32 * 2 + (size_t)(512 + 8) * 4 + (size_t)(tab ? p.stab_nb * 256 : 0) * sizeof(float);
#ifdef GIF_WGRAD_KX3_PROBE
    if (!taps && p.T == 9) {
        WgradParams q = p;
        q.T = 3;
        grid = dim3(grid.x / 3);
        if (tab) wgrad_launch_v2_t<true, false>(grid, lds, s, q);
        else wgrad_launch_v2_t<false, false>(grid, lds, s, q);
        return;
    }
#endif
    if (taps) wgrad_launch_v2_t<false, true>(grid, lds, s, p);
    else if (tab) wgrad_launch_v2_t<true, false>(grid, lds, s, p);
    else wgrad_launch_v2_t<false, false>(grid, lds, s, p);
}

// wgrad tiles follow the SAME row/col padding as the forward packing (gif_conv2d_pack_dims(Cs, Cb)):
// rows RP multiple of 32 or 128, cols CP multiple of 8 or 32 — so pad further to the tile here.
inline int tile_of(int c) { return c <= 32 ? 32 : 128; }

// 256x128 tiles (wave tile 128x64: 6 LDS operand reads per 8 MFMAs instead of 4 per 4 — the operand reads, one ds_read_b32 per
// MFMA operand in this [pixel][channel] layout, are what caps the 128x128 kernel) whenever the row count allows it and the
// operands go through the plain LDS-DMA path.  GIF_WGRAD_BIG=0 disables it.
inline bool wgrad_big_tile(int Cs, int Cb, bool scaled, long Ntot) {
    static int off = -1;
    if (off < 0) {
        const char* e = gif::knob("GIF_WGRAD_BIG");
        off = (e && atoi(e) == 0) ? 1 : 0;
    }
    // (below ~16K reduction rows the halved workgroup count costs more than the operand reuse gains: measured)
    return !off && !scaled && Ntot >= 16384 && tile_of(Cs) == 128 && tile_of(Cb) == 128 && ((Cs + 127) / 128 * 128) % 256 == 0;
}
inline int tile_rows(int Cs, int Cb, bool scaled, long Ntot) { return wgrad_big_tile(Cs, Cb, scaled, Ntot) ? 256 : tile_of(Cs); }

template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ wp, int R, int C, int KH,
                                   int KW, int RP, int CP, long sr, long sc, long sky, long skx, float scale) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)KH * KW * RP * CP;
    if (idx >= total) return;
    int c = (int)(idx % CP);
    long rest = idx / CP;
    int r = (int)(rest % RP);
    int t = (int)(rest / RP);
    int ky = t / KW, kx = t - ky * KW;
    float v = 0.f;
    if (r < R && c < C) v = scale * w[r * sr + c * sc + ky * sky + kx * skx];
    wp[idx] = (T)v;
}

// bf16x3 packing: wp3[t][term][r][c] (bf16 bits) = term-th part of the exact three-way split of scale * w
__global__ void pack_weight_x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp3, int R, int C, int KH,
                                      int KW, int RP, int CP, long sr, long sc, long sky, long skx, float scale) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)KH * KW * RP * CP;
    if (idx >= total) return;
    int c = (int)(idx % CP);
    long rest = idx / CP;
    int r = (int)(rest % RP);
    int t = (int)(rest / RP);
    int ky = t / KW, kx = t - ky * KW;
    float v = 0.f;
    if (r < R && c < C) v = scale * w[r * sr + c * sc + ky * sky + kx * skx];
    unsigned h, m, l;
    gif::split_pair(v, 0.f, h, m, l);
    const size_t plane = (size_t)RP * CP;
    unsigned short* o = wp3 + (size_t)t * 3 * plane + (size_t)r * CP + c;
    o[0] = (unsigned short)(h & 0xffffu);
    o[plane] = (unsigned short)(m & 0xffffu);
    o[2 * plane] = (unsigned short)(l & 0xffffu);
}

// tap-dense variant (3x3): wp3[step][term][r][32], K order (tap, channel) without per-tap padding
__global__ void pack_weight_x3_dense_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp3, int R, int C, int cpt,
                                            int steps, int RP, long sr, long sc, long sky, long skx, float scale) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)steps * RP * 32;
    if (idx >= total) return;
    const int k = (int)(idx % 32);
    const long rest = idx / 32;
    const int r = (int)(rest % RP);
    const int st = (int)(rest / RP);
    const int q = st * 8 + k / 4;
    const int t = q / cpt, c = (q - t * cpt) * 4 + k % 4;
    float v = 0.f;
    if (t < 9 && r < R && c < C) v = scale * w[r * sr + c * sc + (t / 3) * sky + (t % 3) * skx];
    unsigned h, m, l;
    gif::split_pair(v, 0.f, h, m, l);
    const size_t plane = (size_t)RP * 32;
    unsigned short* o = wp3 + (size_t)st * 3 * plane + (size_t)r * 32 + k;
    o[0] = (unsigned short)(h & 0xffffu);
    o[plane] = (unsigned short)(m & 0xffffu);
    o[2 * plane] = (unsigned short)(l & 0xffffu);
}

// f16x2 packing (common.h "h2"): one workgroup per packed row.  Header hdr[r] = the row's exponent e (row maximum over all taps and
// channels of scale * w lands in [2^14, 2^15)), hdr[RP + r] = the row's flag; planes wp2[t][2][RP][CP] (f16 bits) = hi / lo of
// scale * w * 2^e.  A 16-channel K group (the kernels' k-groups: 16 consecutive padded channels of one tap) whose maximum lies more
// than 2^kH2WindowW below the row maximum sets the flag: its values no longer carry 22 bits, launches using the row take the bf16x3
// fallback.
// planes3 != NULL: the bf16x3 packing wp3[t][3][RP][CP] of the same weights (the guarded fallback's operand) in the same pass.
// The row's T * CP values are read from HBM ONCE into LDS (the canonical tensor is walked with a stride of 9 or 9 * Cin floats:
// the second read cost as much as the first; 28.6 -> 15.5 us per pack, 86 packs per training iteration: profiles/r5_kernel_stats.md).
__global__ void __launch_bounds__(256) pack_weight_h2_kernel(const float* __restrict__ w, int* __restrict__ hdr, unsigned short* __restrict__ planes,
                                                             unsigned short* __restrict__ planes3, int R, int C, int KH, int KW, int RP, int CP,
                                                             long sr, long sc, long sky, long skx, float scale, int dense_cpt, int dense_steps) {
    // dense_cpt > 0: the tap-dense K order of a 3x3 kernel (gif_pack_weight_f32x3_tapdense): "taps" = dense_steps K steps of CP = 32
    // floats, float k of step st = float k % 4 of 16-byte chunk q = 8 st + k / 4 = tap q / cpt, channel 4 (q % cpt) + k % 4
    extern __shared__ float row_vals[];  // [KH * KW][CP]
    __shared__ float red[256];
    __shared__ int s_flag;
    const int r = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) s_flag = 0;
    const int T = dense_cpt ? dense_steps : KH * KW, n = T * CP;
    float m = 0.f;
    for (int e = tid; e < n; e += 256) {
        // (c fastest over the threads for rows = output channels: consecutive threads walk the input channels, 9 floats apart)
        int t = e / CP, c = e - t * CP;
        bool ok = r < R;
        if (dense_cpt) {
            const int q = t * 8 + c / 4;
            t = q / dense_cpt;
            c = (q - t * dense_cpt) * 4 + c % 4;
            ok = ok && t < 9;
        }
        const int ky = t / KW, kx = t - ky * KW;
        const float v = (ok && c < C) ? scale * w[r * sr + c * sc + ky * sky + kx * skx] : 0.f;
        row_vals[e] = v;
        m = fmaxf(m, fabsf(v));
    }
    red[tid] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
        __syncthreads();
    }
    const float rowmax = red[0];
    const int e2 = gif::h2_exp_for(__float_as_uint(rowmax), gif::kH2TargetW);
    const float sc2 = gif::h2_pow2(e2);
    const size_t plane = (size_t)RP * CP;
    const int gpt = CP / 16, ngroups = T * gpt;  // 16-channel groups per tap / per row
    bool narrow = false;
    for (int g = tid; g < ngroups; g += 256) {
        const int t = g / gpt, c0 = (g - t * gpt) * 16;
        const float* src = row_vals + t * CP + c0;
        unsigned short* o = planes + (size_t)t * 2 * plane + (size_t)r * CP + c0;
        float gm = 0.f;
#pragma unroll 4
        for (int k = 0; k < 16; k += 2) {
            const float v0 = src[k], v1 = src[k + 1];
            gm = fmaxf(gm, fmaxf(fabsf(v0), fabsf(v1)));
            unsigned h, l;
            gif::split_pair_h2(v0, v1, sc2, h, l);
            *reinterpret_cast<unsigned*>(o + k) = h;
            *reinterpret_cast<unsigned*>(o + plane + k) = l;
            if (planes3) {
                unsigned m3;
                gif::split_pair(v0, v1, h, m3, l);
                unsigned short* o3 = planes3 + (size_t)t * 3 * plane + (size_t)r * CP + c0 + k;
                *reinterpret_cast<unsigned*>(o3) = h;
                *reinterpret_cast<unsigned*>(o3 + plane) = m3;
                *reinterpret_cast<unsigned*>(o3 + 2 * plane) = l;
            }
        }
        if (gm > 0.f && (int)(__float_as_uint(rowmax) >> 23) - (int)(__float_as_uint(gm) >> 23) > gif::kH2WindowW) narrow = true;
    }
    if (narrow) atomicOr(&s_flag, 1);
    __syncthreads();
    if (tid == 0) {
        hdr[r] = e2;
        hdr[RP + r] = s_flag;
    }
}

__global__ void unpack_wgrad_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nsplit, int R, int C,
                                    int KH, int KW, int RP, int CP, long sr, long sc, long sky, long skx,
                                    float scale) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)KH * KW * R * C;
    if (idx >= total) return;
    int c = (int)(idx % C);
    long rest = idx / C;
    int r = (int)(rest % R);
    int t = (int)(rest / R);
    int ky = t / KW, kx = t - ky * KW;
    size_t stride = (size_t)KH * KW * RP * CP;
    const float* src = ws + ((size_t)t * RP + r) * CP + c;
    // 8 independent partial sums (fixed assignment split -> lane, fixed final order): the loads of 8 splits are in flight
    // together instead of one dependent ~1 us HBM round trip per split (113 splits on the 128-channel layers)
    float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 8 <= nsplit; s += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) part[k] += src[(size_t)(s + k) * stride];
    }
    for (int k = 0; s < nsplit; ++s, ++k) part[k] += src[(size_t)s * stride];
    const float acc = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
    dw[r * sr + c * sc + ky * sky + kx * skx] = scale * acc;
}

// ------------------------------------------------------------------------------------------------------------------
// Small-channel weight gradient (the condition-noise convs 6->12->24: Cs <= 32, Cb <= 16, 3x3, stride 1, pad 1).
// On the 32x32-tile kernel above these layers run at 5-11 TFLOP/s: nine launches-worth of 32x32x2 MFMAs of which
// (12..24)x(8..12) entries are real.  Here ONE wave owns all 9 taps: per group of 4 consecutive pixels it loads gy[4 px][Cs]
// and the nine shifted x[4 px][Cb] straight from global memory (64-byte segments, L1/L2 resident: no LDS) as the A / B operands
// of v_mfma_f32_16x16x4_f32 (i = cout, j = cin, k = pixel) and accumulates 9 taps x 2 cout tiles = 18 accumulators.
// 1024-thread workgroups: 16 waves share a pixel range; their accumulators are reduced through LDS in a fixed order and
// written as one split of the usual [split][tap][RP][CP] workspace (deterministic, same unpack kernel).
// ------------------------------------------------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));

struct SmallWgradParams {
    const float* sm;  // gy [B,H,W,Cs]
    const float* bg;  // x  [B,H,W,Cb]
    float* ws;        // [nsplit][9][RP][CP]
    int B, H, W, Cs, Cb, RP, CP;
    long Ntot, chunk;  // pixels, pixels per split (multiple of 64)
};

__global__ void __launch_bounds__(1024) conv_wgrad_small_mfma(const SmallWgradParams p) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [8][18][256] floats = 147 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, k = lane >> 4;
    const long n_begin = (long)blockIdx.x * p.chunk;
    long n_end = n_begin + p.chunk;
    if (n_end > p.Ntot) n_end = p.Ntot;
    f32x4v acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[t][ct] = (f32x4v)(0.f);
    const bool cin_ok = c < p.Cb;
    const bool co_ok[2] = {c < p.Cs, 16 + c < p.Cs};
    // this lane's pixel: n = n_begin + (quad index)*4 + k ; waves take quads round-robin (16 waves => +64 pixels per step)
    long n = n_begin + wave * 4 + k;
    int x, y, b;
    {
        long hw = (long)p.H * p.W;
        b = (int)(n / hw);
        long r = n - (long)b * hw;
        y = (int)(r / p.W);
        x = (int)(r - (long)y * p.W);
    }
    float a_cur[2], b_cur[9];
    auto load = [&](float (&av)[2], float (&bv)[9]) __attribute__((always_inline)) {
        const bool ok = n < n_end;
        const float* sp = p.sm + n * p.Cs + c;
        av[0] = (ok && co_ok[0]) ? sp[0] : 0.f;
        av[1] = (ok && co_ok[1]) ? sp[16] : 0.f;
        const float* bp = p.bg + n * p.Cb + c;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const bool in = ok && cin_ok && (unsigned)(y + dy) < (unsigned)p.H && (unsigned)(x + dx) < (unsigned)p.W;
                bv[(dy + 1) * 3 + dx + 1] = in ? bp[((long)dy * p.W + dx) * p.Cb] : 0.f;
            }
    };
    auto advance = [&]() __attribute__((always_inline)) {
        n += 64;
        x += 64;
        while (x >= p.W) { x -= p.W; ++y; }
        while (y >= p.H) { y -= p.H; ++b; }
    };
    const long steps = (n_end - n_begin + 63) / 64;
    load(a_cur, b_cur);
    for (long s = 0; s < steps; ++s) {
        float a_nxt[2], b_nxt[9];
        advance();
        load(a_nxt, b_nxt);  // prefetch of the next group runs under this group's 18 MFMAs
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[ct], b_cur[t], acc[t][ct], 0, 0, 0);
        a_cur[0] = a_nxt[0]; a_cur[1] = a_nxt[1];
#pragma unroll
        for (int t = 0; t < 9; ++t) b_cur[t] = b_nxt[t];
    }
    // fixed-order tree over the 16 waves: waves [h, 2h) hand their accumulators to waves [0, h)
    for (int h = 8; h >= 1; h >>= 1) {
        if (wave >= h && wave < 2 * h) {
            float* dst = red + (size_t)(wave - h) * 18 * 256;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    *reinterpret_cast<f32x4v*>(dst + (t * 2 + ct) * 256 + lane * 4) = acc[t][ct];
        }
        __syncthreads();
        if (wave < h) {
            const float* src = red + (size_t)wave * 18 * 256;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    acc[t][ct] += *reinterpret_cast<const f32x4v*>(src + (t * 2 + ct) * 256 + lane * 4);
        }
        __syncthreads();
    }
    if (wave == 0) {
        // C/D layout of the 16x16 MFMA: column j = lane & 15, rows i = 4*(lane >> 4) + r
        float* out = p.ws + (size_t)blockIdx.x * 9 * p.RP * p.CP;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = ct * 16 + 4 * k + r;
                    out[((size_t)t * p.RP + row) * p.CP + c] = acc[t][ct][r];
                }
    }
}

inline bool small_wgrad_ok(const gif_conv_geom* g, bool scaled) {
    static int off = -1;
    if (off < 0) {
        const char* e = gif::knob("GIF_SMALL_WGRAD");
        off = (e && atoi(e) == 0) ? 1 : 0;
    }
    return !off && !scaled && g->KH == 3 && g->KW == 3 && g->stride == 1 && g->pad == 1 && g->Hs == g->Hb && g->Ws == g->Wb &&
           g->Cs <= 32 && g->Cb <= 16 && (long)g->B * g->Hs * g->Ws >= 65536;
}

// ------------------------------------------------------------------------------------------------------------------
// f16 "halo" weight gradient for the thin high-resolution layers (Cs <= 32, Cb <= 32, stride 1: the 1024^2 block of BASELINE
// configs[4] and the condition-noise / ToRGB layers, model/stg2_generator.py:159-209, stylegan2_common_layers.py:388-431).
// conv_wgrad_mfma above runs one workgroup per (pixel chunk, TAP): nine workgroups stream the same two activation chunks
// through LDS-DMA and each gathers them again — 0.86 ms for 32 x 32 channels at 1024^2, batch 8 (0.18 ms of HBM traffic).
// Here a PERSISTENT workgroup walks 16 x 16-pixel patches of one sample: the gy patch and the x patch + halo are staged in LDS
// once (double buffered: the LDS-DMA of patch k + 1 runs under the MFMAs of patch k; nothing is stored per patch, so the loop
// has no store latency in it), and all taps are formed from LDS: a K step is one patch row (16 pixels), the tap only shifts the
// row / column of the x operand.  K = pixels is the slow axis of both tiles, so an operand fragment (8 consecutive pixels of
// one channel) is gathered with eight 16-bit LDS reads as in the kernel above (lane = channel: a half wave reads 64
// contiguous bytes of one pixel, conflict free); the gy fragment is gathered once per K step for all taps.
// Each of the 4 waves owns every fourth patch row and all taps (9 accumulator tiles of 32 x 32 = 144 VGPRs); at the end the
// waves' accumulators are reduced through LDS in a fixed order and the workgroup writes ONE split of the usual
// [split][tap][RP][CP] workspace (nsplit = workgroups; deterministic, same unpack kernel).  Per-sample scales (modulated
// layers) multiply the fragments: one scalar per lane, a patch belongs to one sample.
// ------------------------------------------------------------------------------------------------------------------
struct HaloWgradParams {
    const void* sm;   // gy [B,H,W,Cs] f16
    const void* bg;   // x  [B,H,W,Cb] f16 (same spatial size: stride 1, pad = (K - 1) / 2 or 0 for 1x1)
    float* ws;        // [nsplit = gridDim.x][T][32][32]
    const float* ss;  // [B,Cs] or null
    const float* bs;  // [B,Cb] or null
    int B, Hs, Ws, Hb, Wb, Cs, Cb, KH, KW, pad;
    int tiles_x, tiles_y, ntiles;
    const void* zero;
};

constexpr int kHwPatchHalfs = 256 * 32;        // gy patch: 16 x 16 pixels x 32 channels
constexpr int kHwHaloHalfs = 5376 * 2;         // x patch + halo: 18 x 18 pixels x 32 channels, rounded up to whole DMA pieces
constexpr int kHwBufHalfs = kHwPatchHalfs + kHwHaloHalfs;

// TR: operand fragments by two transposing LDS reads (ds_read_b64_tr_b16, gfx950) instead of eight 16-bit reads.  Semantics
// (tools/probes/tr16_probe.hip): inside a 16-lane group, lane s supplies the address of an 8-byte chunk and receives element
// s % 4 of the chunks of lanes 4 j + s / 4, j = 0..3.  Lane L of the group that owns channels 16 g .. 16 g + 15 therefore
// supplies pixel k0 + L / 4, channels 16 g + 4 (L % 4) .. + 3, and lane s gets channel 16 g + s at pixels k0 .. k0 + 3.
typedef __fp16 tr_h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ gif::f16x8_t halo_frag_tr(const gif::f16* base_lo) {
    // base_lo: this lane's chunk of the first four pixels; the next four pixels are 4 * 32 halfs further
    typedef __attribute__((address_space(3))) tr_h4* lp_t;
    const tr_h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp_t)(base_lo));
    const tr_h4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp_t)(base_lo + 4 * 32));
    gif::f16x8_t v;
    v[0] = (gif::f16)lo[0]; v[1] = (gif::f16)lo[1]; v[2] = (gif::f16)lo[2]; v[3] = (gif::f16)lo[3];
    v[4] = (gif::f16)hi[0]; v[5] = (gif::f16)hi[1]; v[6] = (gif::f16)hi[2]; v[7] = (gif::f16)hi[3];
    return v;
}

template <bool TR>
__global__ void __launch_bounds__(256, 2) conv_wgrad_halo_f16(const HaloWgradParams p) {
    typedef gif::f16 T;
    extern __shared__ __attribute__((aligned(16))) float hw_smem[];
    T* const lds = reinterpret_cast<T*>(hw_smem);  // [2][kHwBufHalfs]
    const T* const smb = static_cast<const T*>(p.sm);
    const T* const bgb = static_cast<const T*>(p.bg);
    const T* const pzero = static_cast<const T*>(p.zero);
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int tpi = p.tiles_x * p.tiles_y;
    const int HWh = 16 + p.KW - 1;
    const int nchunks_x = (16 + p.KH - 1) * HWh * 4;
    const int T9 = p.KH * p.KW;

    // chunk e = 64 * wave + lane + 256 * pass: channel chunk e % 4 is the same in every pass, the pixel advances by 64
    const int c8 = (lane & 3) * 8;
    const int q_first = (wave * 64 + lane) >> 2;
    const int hy_first = q_first / HWh, hx_first = q_first - hy_first * HWh;
    const int dqy = 64 / HWh, dqx = 64 - dqy * HWh;
    auto issue_patch = [&](int tile, T* buf) __attribute__((always_inline)) {
        const int b = tile / tpi, tr = tile - b * tpi;
        const int ty = tr / p.tiles_x, tx = tr - ty * p.tiles_x;
        const int oy0 = ty * 16, ox0 = tx * 16;
        {   // gy patch: pixel q = 16 * ly + lx
            const T* const gb = smb + (size_t)b * p.Hs * p.Ws * p.Cs;
            int q = q_first;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int gy = oy0 + (q >> 4), gx = ox0 + (q & 15);
                const bool ok = gy < p.Hs && gx < p.Ws && c8 < p.Cs;
                const T* g = ok ? gb + ((gy * p.Ws + gx) * p.Cs + c8) : pzero;
                __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(buf + (it * 256 + wave * 64) * 8), 16, 0, 0);
                q += 64;
            }
        }
        {   // x patch + halo: rows / columns shifted by -pad
            const T* const xb = bgb + (size_t)b * p.Hb * p.Wb * p.Cb;
            const int gy0 = oy0 - p.pad, gx0 = ox0 - p.pad;
            int hy = hy_first, hx = hx_first;
            T* const xbuf = buf + kHwPatchHalfs;
            for (int e0 = wave * 64; e0 < nchunks_x; e0 += 256) {  // wave-uniform
                const int gy = gy0 + hy, gx = gx0 + hx;
                const bool ok = e0 + lane < nchunks_x && (unsigned)gy < (unsigned)p.Hb && (unsigned)gx < (unsigned)p.Wb && c8 < p.Cb;
                const T* g = ok ? xb + ((gy * p.Wb + gx) * p.Cb + c8) : pzero;
                __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(xbuf + e0 * 8), 16, 0, 0);
                hx += dqx; hy += dqy;
                if (hx >= HWh) { hx -= HWh; ++hy; }
            }
        }
    };
    auto load_scales = [&](int tile, float& sa, float& sb) __attribute__((always_inline)) {
        const int b = tile / tpi;
        sa = (p.ss && li < p.Cs) ? p.ss[(size_t)b * p.Cs + li] : 1.f;
        sb = (p.bs && li < p.Cb) ? p.bs[(size_t)b * p.Cb + li] : 1.f;
    };
    const bool scaled = p.ss != nullptr || p.bs != nullptr;
    // transposing reads: lane L = lane % 16 of channel group g = (lane >> 4) & 1 supplies pixel L / 4, channels 16 g + 4 (L % 4)
    const int tr_px = (lane & 15) >> 2, tr_ch = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int stride = (int)gridDim.x;
    int tile = xcd_remap((int)blockIdx.x, stride);
    float sa = 1.f, sb = 1.f, sa_n = 1.f, sb_n = 1.f;
    if (tile < p.ntiles) {
        issue_patch(tile, lds);
        if (scaled) load_scales(tile, sa, sb);
    }
    int cur = 0;
    for (; tile < p.ntiles; tile += stride) {
        __syncthreads();  // patch `tile` landed (vmcnt(0) of every wave); everybody is done reading the other buffer
        const int next = tile + stride;
        if (next < p.ntiles) {
            issue_patch(next, lds + (cur ^ 1) * kHwBufHalfs);
            if (scaled) load_scales(next, sa_n, sb_n);
        }
        const T* const G = lds + cur * kHwBufHalfs;      // [256 px][32]
        const T* const X = G + kHwPatchHalfs;             // [halo px][32]
        const T fa = (T)sa, fb = (T)sb;
#pragma unroll 1
        for (int row = wave; row < 16; row += 4) {  // K step = one patch row: pixels 16 * row + 8 * lh + 0..7
            gif::f16x8_t af;
            if constexpr (TR) {
                af = halo_frag_tr(G + (row * 16 + 8 * lh + tr_px) * 32 + tr_ch);
            } else {
                const T* ga = G + (row * 16 + 8 * lh) * 32 + li;
#pragma unroll
                for (int e = 0; e < 8; ++e) af[e] = ga[e * 32];
            }
            if (scaled) af *= fa;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t < T9) {  // wave-uniform
                    const int ky = t / p.KW, kx = t - ky * p.KW;
                    gif::f16x8_t bf;
                    if constexpr (TR) {
                        bf = halo_frag_tr(X + ((row + ky) * HWh + 8 * lh + kx + tr_px) * 32 + tr_ch);
                    } else {
                        const T* xa = X + ((row + ky) * HWh + 8 * lh + kx) * 32 + li;
#pragma unroll
                        for (int e = 0; e < 8; ++e) bf[e] = xa[e * 32];
                    }
                    if (scaled) bf *= fb;
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[t], 0, 0, 0);
                }
            }
        }
        sa = sa_n; sb = sb_n;
        cur ^= 1;
    }
    // ---- fixed-order reduction of the four waves' accumulators through LDS (the patch buffers are free), then ONE split
    __syncthreads();
    float* const red = hw_smem;  // [2][9][1024] floats = 72 KB <= the two patch buffers (75.8 KB)
    for (int h = 2; h >= 1; h >>= 1) {
        if (wave >= h && wave < 2 * h) {
            float* dst = red + (size_t)(wave - h) * 9 * 1024;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(t * 16 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (wave < h) {
            const float* src = red + (size_t)wave * 9 * 1024;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += src[(t * 16 + r) * 64 + lane];
        }
        __syncthreads();
    }
    if (wave == 0) {
        // C/D layout of the 32x32 MFMA: column (cin) = lane & 31, row (cout) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
        float* out = p.ws + (size_t)blockIdx.x * T9 * 1024;
#pragma unroll
        for (int t = 0; t < 9; ++t)
            if (t < T9) {
#pragma unroll
                for (int r = 0; r < 16; ++r) out[(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[t][r];
            }
    }
}

// GIF_F16_HALO_WGRAD=0: A/B knob (the per-tap kernel).  Workgroups (= splits): two per CU, never more than patches.
inline bool halo_wgrad_ok(const gif_conv_geom* g) {
    static const int off = gif::knob("GIF_F16_HALO_WGRAD") ? atoi(gif::knob("GIF_F16_HALO_WGRAD")) == 0 : 0;
    return !off && g->stride == 1 && g->Cs <= 32 && g->Cb <= 32 && g->KH <= 3 && g->KW <= 3 && g->Hs == g->Hb && g->Ws == g->Wb &&
           g->KH == 2 * g->pad + 1 && g->KW == 2 * g->pad + 1 && g->Hs >= 16 && g->Ws >= 16 &&
           (long)g->B * gif::cdiv(g->Hs, 16) * gif::cdiv(g->Ws, 16) >= 512;
}
inline int halo_wgrad_splits(const gif_conv_geom* g) {
    const long patches = (long)g->B * gif::cdiv(g->Hs, 16) * gif::cdiv(g->Ws, 16);
    return (int)(patches < 512 ? patches : 512);
}

// Winograd F(3x3,2x2) output transform fused with the split reduction: dW = A'^T dU A' with
// A'^T = [1 1 1 0; 0 1 -1 0; 0 1 1 -1] (the F(3,2) matrix with the sign of the input transform's last row folded in,
// because V was produced by the F(2,3) input transform whose last row is the negative of F(3,2)'s).
__global__ void wino_unpack_wgrad_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nsplit, int R, int C,
                                         int RP, int CP, long sr, long sc, long sky, long skx, float scale) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)R * C) return;
    int c = (int)(idx % C), r = (int)(idx / C);
    const size_t plane = (size_t)RP * CP, stride = 16 * plane;
    const float* src = ws + (size_t)r * CP + c;
    float u[4][4];
#pragma unroll
    for (int q = 0; q < 16; ++q) u[q >> 2][q & 3] = 0.f;
    int s = 0;
    for (; s + 2 <= nsplit; s += 2) {  // 32 independent loads in flight; splits summed in a fixed order
        const float* sp = src + (size_t)s * stride;
        float t0[16], t1[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) { t0[q] = sp[q * plane]; t1[q] = sp[stride + q * plane]; }
#pragma unroll
        for (int q = 0; q < 16; ++q) u[q >> 2][q & 3] += t0[q] + t1[q];
    }
    if (s < nsplit) {
        const float* sp = src + (size_t)s * stride;
#pragma unroll
        for (int q = 0; q < 16; ++q) u[q >> 2][q & 3] += sp[q * plane];
    }
    float t[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[0][j] = u[0][j] + u[1][j] + u[2][j];
        t[1][j] = u[1][j] - u[2][j];
        t[2][j] = u[1][j] + u[2][j] - u[3][j];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float* o = dw + r * sr + c * sc + i * sky;
        o[0 * skx] = scale * (t[i][0] + t[i][1] + t[i][2]);
        o[1 * skx] = scale * (t[i][1] - t[i][2]);
        o[2 * skx] = scale * (t[i][1] + t[i][2] - t[i][3]);
    }
}

// wgrad workspace dims: rows/cols padded to the wgrad tile (32 or 128)
inline void wgrad_dims(int Cs, int Cb, int* RP, int* CP) {
    int bp = tile_of(Cs), bq = tile_of(Cb);
    *RP = (Cs + bp - 1) / bp * bp;
    *CP = (Cb + bq - 1) / bq * bq;
}

}  // namespace

extern "C" {

int gif_pack_weight_f32(const float* w, float* wp, int R, int C, int KH, int KW, int RP, int CP, int64_t sr,
                        int64_t sc, int64_t sky, int64_t skx, float scale, gif_stream_t stream) {
    GIF_REQUIRE(w && wp && R > 0 && C > 0 && RP >= R && CP >= C && KH > 0 && KW > 0, "pack_weight: bad arguments");
    long total = (long)KH * KW * RP * CP;
    pack_weight_kernel<float><<<gif::cdiv(total, 256), 256, 0, gif::as_stream(stream)>>>(w, wp, R, C, KH, KW, RP, CP, sr, sc,
                                                                                           sky, skx, scale);
    return gif::check_launch("pack_weight");
}

/* fp32 master weights -> f16 packed operand of the f16 convolution kernels (same [tap][RP][CP] layout) */
int gif_pack_weight_f32x3(const float* w, void* wp3, int R, int C, int KH, int KW, int RP, int CP, int64_t sr, int64_t sc,
                          int64_t sky, int64_t skx, float scale, gif_stream_t stream) {
    GIF_REQUIRE(w && wp3 && R > 0 && C > 0 && RP >= R && CP >= C && KH > 0 && KW > 0, "pack_weight_f32x3: bad arguments");
    long total = (long)KH * KW * RP * CP;
    pack_weight_x3_kernel<<<gif::cdiv(total, 256), 256, 0, gif::as_stream(stream)>>>(w, static_cast<unsigned short*>(wp3), R, C,
                                                                                       KH, KW, RP, CP, sr, sc, sky, skx, scale);
    return gif::check_launch("pack_weight_f32x3");
}

/* f16x2 packing: [RP int32 row exponents][RP int32 row flags][tap][2][RP][CP] f16 — gif_pack_weight_f32h2_bytes of device memory */
int64_t gif_pack_weight_f32h2_bytes(int KH, int KW, int RP, int CP) {
    if (KH <= 0 || KW <= 0 || RP <= 0 || CP <= 0) return 0;
    return (int64_t)gif::h2_header_bytes(RP) + (int64_t)KH * KW * 2 * RP * CP * 2;
}

int gif_pack_weight_f32h2(const float* w, void* wp2, int R, int C, int KH, int KW, int RP, int CP, int64_t sr, int64_t sc,
                          int64_t sky, int64_t skx, float scale, gif_stream_t stream) {
    GIF_REQUIRE(w && wp2 && R > 0 && C > 0 && RP >= R && CP >= C && KH > 0 && KW > 0 && RP % 32 == 0 && CP % 32 == 0,
                "pack_weight_f32h2: bad arguments");
    int* hdr = static_cast<int*>(wp2);
    unsigned short* planes = reinterpret_cast<unsigned short*>(static_cast<char*>(wp2) + gif::h2_header_bytes(RP));
    const size_t lds = (size_t)KH * KW * CP * sizeof(float);
    GIF_REQUIRE(lds <= 64 * 1024, "pack_weight_f32h2: a row of %d x %d values does not fit the staging LDS", KH * KW, CP);
    pack_weight_h2_kernel<<<RP, 256, lds, gif::as_stream(stream)>>>(w, hdr, planes, nullptr, R, C, KH, KW, RP, CP, sr, sc, sky, skx, scale, 0, 0);
    return gif::check_launch("pack_weight_f32h2");
}

/* both packings of the same weights in ONE launch: wp2 (f16x2, as gif_pack_weight_f32h2) and wp3 (bf16x3, as gif_pack_weight_f32x3) */
int gif_pack_weight_f32h2x3(const float* w, void* wp2, void* wp3, int R, int C, int KH, int KW, int RP, int CP, int64_t sr, int64_t sc,
                            int64_t sky, int64_t skx, float scale, gif_stream_t stream) {
    GIF_REQUIRE(w && wp2 && wp3 && R > 0 && C > 0 && RP >= R && CP >= C && KH > 0 && KW > 0 && RP % 32 == 0 && CP % 32 == 0,
                "pack_weight_f32h2x3: bad arguments");
    int* hdr = static_cast<int*>(wp2);
    unsigned short* planes = reinterpret_cast<unsigned short*>(static_cast<char*>(wp2) + gif::h2_header_bytes(RP));
    const size_t lds = (size_t)KH * KW * CP * sizeof(float);
    GIF_REQUIRE(lds <= 64 * 1024, "pack_weight_f32h2x3: a row of %d x %d values does not fit the staging LDS", KH * KW, CP);
    pack_weight_h2_kernel<<<RP, 256, lds, gif::as_stream(stream)>>>(w, hdr, planes, static_cast<unsigned short*>(wp3), R, C, KH, KW, RP, CP, sr,
                                                                    sc, sky, skx, scale, 0, 0);
    return gif::check_launch("pack_weight_f32h2x3");
}

/* tap-dense K order (gif_conv2d_x3_tapdense_steps): wp2 = [2 RP int32 header][steps][2][RP][32] f16, wp3 = [steps][3][RP][32] bf16 (as
 * gif_pack_weight_f32x3_tapdense writes it), one launch */
int64_t gif_pack_weight_f32h2_tapdense_bytes(int cin_act, int KH, int KW, int RP) {
    const int steps = gif_conv2d_x3_tapdense_steps(cin_act, KH, KW);
    if (steps <= 0 || RP <= 0) return 0;
    return (int64_t)gif::h2_header_bytes(RP) + (int64_t)steps * 2 * RP * 32 * 2;
}

int gif_pack_weight_f32h2x3_tapdense(const float* w, void* wp2, void* wp3, int R, int C, int cin_act, int KH, int KW, int RP, int64_t sr,
                                     int64_t sc, int64_t sky, int64_t skx, float scale, gif_stream_t stream) {
    const int steps = gif_conv2d_x3_tapdense_steps(cin_act, KH, KW);
    GIF_REQUIRE(w && wp2 && wp3 && R > 0 && C > 0 && RP >= R && RP % 32 == 0 && cin_act >= C && steps > 0,
                "pack_weight_f32h2x3_tapdense: bad arguments");
    int* hdr = static_cast<int*>(wp2);
    unsigned short* planes = reinterpret_cast<unsigned short*>(static_cast<char*>(wp2) + gif::h2_header_bytes(RP));
    const size_t lds = (size_t)steps * 32 * sizeof(float);
    pack_weight_h2_kernel<<<RP, 256, lds, gif::as_stream(stream)>>>(w, hdr, planes, static_cast<unsigned short*>(wp3), R, C, KH, KW, RP, 32, sr,
                                                                    sc, sky, skx, scale, cin_act / 4, steps);
    return gif::check_launch("pack_weight_f32h2x3_tapdense");
}

/* K steps (32-float chunks) of the tap-dense order, or 0 if the mode does not apply: 3x3 kernels, 8 <= cin_act < 32, cin_act % 4 == 0 */
int gif_conv2d_x3_tapdense_steps(int cin_act, int KH, int KW) {
    if (KH != 3 || KW != 3 || cin_act < 8 || cin_act >= 32 || cin_act % 4 != 0) return 0;
    return (9 * (cin_act / 4) + 7) / 8;
}

/* wp3[step][term][RP][32] (bf16 bits): element k of step s is float k % 4 of 16-byte chunk q = 8 s + k / 4, i.e. tap q / (cin_act/4),
 * channel 4 (q % (cin_act/4)) + k % 4 — the order in which the tap-dense kernels walk K (conv_igemm.hip, GatherParams::dense) */
int gif_pack_weight_f32x3_tapdense(const float* w, void* wp3, int R, int C, int cin_act, int KH, int KW, int RP, int64_t sr, int64_t sc,
                                   int64_t sky, int64_t skx, float scale, gif_stream_t stream) {
    const int steps = gif_conv2d_x3_tapdense_steps(cin_act, KH, KW);
    GIF_REQUIRE(w && wp3 && R > 0 && C > 0 && RP >= R && cin_act >= C && steps > 0, "pack_weight_f32x3_tapdense: bad arguments");
    long total = (long)steps * RP * 32;
    pack_weight_x3_dense_kernel<<<gif::cdiv(total, 256), 256, 0, gif::as_stream(stream)>>>(w, static_cast<unsigned short*>(wp3), R, C,
                                                                                             cin_act / 4, steps, RP, sr, sc, sky, skx, scale);
    return gif::check_launch("pack_weight_f32x3_tapdense");
}

int gif_pack_weight_f16(const float* w, void* wp, int R, int C, int KH, int KW, int RP, int CP, int64_t sr, int64_t sc,
                        int64_t sky, int64_t skx, float scale, gif_stream_t stream) {
    GIF_REQUIRE(w && wp && R > 0 && C > 0 && RP >= R && CP >= C && KH > 0 && KW > 0, "pack_weight_f16: bad arguments");
    long total = (long)KH * KW * RP * CP;
    pack_weight_kernel<gif::f16><<<gif::cdiv(total, 256), 256, 0, gif::as_stream(stream)>>>(w, static_cast<gif::f16*>(wp), R, C, KH,
                                                                                              KW, RP, CP, sr, sc, sky, skx, scale);
    return gif::check_launch("pack_weight_f16");
}

/* ---- f16 operands (BASELINE config 5): always 128x128 tiles on the LDS-DMA kernel (MFMA work on padded channels is cheap
 * at the f16 rate), fp32 partial sums, the same unpack kernel. */
// square tile of the f16 weight-gradient kernel: 32 / 64 for the thin layers of the 512^2 / 1024^2 blocks, else 128
static inline int wgrad_tile_f16(int Cs, int Cb) {
    const int m = Cs > Cb ? Cs : Cb;
    return m <= 32 ? 32 : (m <= 64 ? 64 : 128);
}

// 256 x 256 tiles on 8 waves (wave tile 128 x 64) for the f16 weight gradients with multiples of 256 channels on both sides: half
// the LDS-DMA pieces and 6 instead of 8 operand gathers per MFMA (the 128 x 128 loop is bound by both: DESIGN 3b).  GIF_F16_WGRAD256=0: A/B.
static inline bool wgrad_tile256_f16(const gif_conv_geom* g, bool scaled) {
    static const int off = gif::knob("GIF_F16_WGRAD256") ? atoi(gif::knob("GIF_F16_WGRAD256")) == 0 : 0;
    (void)scaled;  // modulated launches too (the per-sample scale table of a 32-pixel stage is 2 KB per sample)
    return !off && g->Cs % 256 == 0 && g->Cb % 256 == 0 && (long)g->B * g->Hs * g->Ws >= 16384 && ((long)g->Hs * g->Ws) % 32 == 0;
}

int gif_conv2d_wgrad_dims_f16(int Cs, int Cb, int* RP, int* CP) {
    GIF_REQUIRE(Cs > 0 && Cb > 0 && RP && CP, "wgrad_dims_f16: bad arguments");
    const int t = wgrad_tile_f16(Cs, Cb);
    *RP = (Cs + t - 1) / t * t;
    *CP = (Cb + t - 1) / t * t;
    return 0;
}

int gif_conv2d_wgrad_splits_f16(const gif_conv_geom* g) {
    if (!g || g->B <= 0) return 1;
    if (halo_wgrad_ok(g)) return halo_wgrad_splits(g);  // conv_wgrad_halo_f16: one split per persistent workgroup
    int RP, CP;
    gif_conv2d_wgrad_dims_f16(g->Cs, g->Cb, &RP, &CP);
    int t = wgrad_tile_f16(g->Cs, g->Cb);
    if (wgrad_tile256_f16(g, false)) t = 256;
    long Ntot = (long)g->B * g->Hs * g->Ws;
    long tiles = (long)(RP / t) * (CP / t) * g->KH * g->KW;
    long slots = t == 256 ? 512 : t == 128 ? 1024 : 2048;  // resident workgroups: 4 per CU at 32 KB of LDS, more for the small tiles
    long want = tiles >= slots ? 1 : slots / tiles;
    long max_by_work = (Ntot + 4 * BKP_MAX - 1) / (4 * BKP_MAX);
    long max_by_mem = (128L << 20) / ((long)g->KH * g->KW * RP * CP * 4);
    long n = want;
    if (n > max_by_work) n = max_by_work;
    if (n > max_by_mem) n = max_by_mem;
    if (n < 1) n = 1;
    return (int)n;
}

int gif_conv2d_wgrad_f16(const void* small, const void* big, float* ws, const float* small_scale, const float* big_scale,
                         const gif_conv_geom* g, int nsplit, gif_stream_t stream) {
    GIF_REQUIRE(g && small && big && ws && nsplit >= 1, "conv2d_wgrad_f16: bad arguments");
    GIF_REQUIRE(g->Cb % 8 == 0 && g->Cs % 8 == 0, "conv2d_wgrad_f16: channels must be multiples of 8");
    GIF_REQUIRE(g->KH >= 1 && g->KH <= 3 && g->KW >= 1 && g->KW <= 3 && (g->stride == 1 || g->stride == 2),
                "conv2d_wgrad_f16: unsupported kernel/stride");
    GIF_REQUIRE((long)g->B * g->Hs * g->Ws * g->Cs < (1L << 31) && (long)g->B * g->Hb * g->Wb * g->Cb < (1L << 31),
                "conv2d_wgrad_f16: tensors of >= 2^31 elements are not supported (32-bit offsets)");
    WgradParams p{};
    p.sm = small; p.bg = big; p.ws = ws; p.ss = small_scale; p.bs = big_scale;
    p.B = g->B; p.Hs = g->Hs; p.Ws = g->Ws; p.Cs = g->Cs; p.Hb = g->Hb; p.Wb = g->Wb; p.Cb = g->Cb;
    p.KW = g->KW; p.stride = g->stride; p.pad = g->pad; p.T = g->KH * g->KW;
    gif_conv2d_wgrad_dims_f16(g->Cs, g->Cb, &p.RP, &p.CP);
    const int t = wgrad_tile_f16(g->Cs, g->Cb);
    p.Ntot = (long)g->B * g->Hs * g->Ws;
    if (halo_wgrad_ok(g) && nsplit == halo_wgrad_splits(g) && p.RP == 32 && p.CP == 32) {
        HaloWgradParams q{};
        q.sm = small; q.bg = big; q.ws = ws; q.ss = small_scale; q.bs = big_scale;
        q.B = g->B; q.Hs = g->Hs; q.Ws = g->Ws; q.Hb = g->Hb; q.Wb = g->Wb; q.Cs = g->Cs; q.Cb = g->Cb;
        q.KH = g->KH; q.KW = g->KW; q.pad = g->pad;
        q.tiles_x = gif::cdiv(g->Ws, 16); q.tiles_y = gif::cdiv(g->Hs, 16); q.ntiles = g->B * q.tiles_x * q.tiles_y;
        q.zero = gif::zero_page16();
        GIF_REQUIRE(q.zero, "conv2d_wgrad_f16: zero page lookup failed");
        hipStream_t hs = gif::as_stream(stream);
        gif::ProfScope prof(7, 2.0 * p.Ntot * (double)g->Cs * g->Cb * p.T, hs, (int)p.Ntot, g->Cs, g->Cb,
                            p.T * 10 + g->stride + (small_scale || big_scale ? 100 : 0));
        const size_t lds = (size_t)2 * kHwBufHalfs * sizeof(gif::f16);
        static gif::LdsAttr attr;
        static const int tr_off = gif::knob("GIF_F16_HALO_WGRAD_TR") ? atoi(gif::knob("GIF_F16_HALO_WGRAD_TR")) == 0 : 0;  // A/B knob
        if (tr_off) {
            attr.ensure(reinterpret_cast<const void*>(conv_wgrad_halo_f16<false>), lds);
            hipLaunchKernelGGL(conv_wgrad_halo_f16<false>, dim3((unsigned)nsplit), dim3(256), lds, hs, q);
        } else {
            static gif::LdsAttr attr_tr;
            attr_tr.ensure(reinterpret_cast<const void*>(conv_wgrad_halo_f16<true>), lds);
            hipLaunchKernelGGL(conv_wgrad_halo_f16<true>, dim3((unsigned)nsplit), dim3(256), lds, hs, q);
        }
        return gif::check_launch("conv2d_wgrad_f16(halo)");
    }
    long chunk = (p.Ntot + nsplit - 1) / nsplit;
    p.chunk = (chunk + BKP_MAX - 1) / BKP_MAX * BKP_MAX;
    if (p.chunk < BKP_MAX) p.chunk = BKP_MAX;
    const bool scaled = small_scale || big_scale;
    const long HWs = (long)g->Hs * g->Ws;
    const bool st16 = scaled && HWs % 32 != 0;  // modulated layer on 4x4 maps: 16-pixel stages
    // the 64-wide tile has no 16-pixel-stage variant (one DMA pass covers 32 pixel rows): such a launch runs 32x32 tiles over
    // the same 64-padded workspace
    const int tl = (t == 64 && st16) ? 32 : t;
    p.tiles_q = p.CP / tl;
    p.tiles_pq = (p.RP / tl) * p.tiles_q;
    dim3 grid((unsigned)(p.tiles_pq * p.T * nsplit));
    hipStream_t s = gif::as_stream(stream);
    p.zero = gif::zero_page16();
    GIF_REQUIRE(p.zero, "conv2d_wgrad_f16: zero page lookup failed");
    double flops = 2.0 * p.Ntot * (double)g->Cs * g->Cb * p.T;
    gif::ProfScope prof(7, flops, s, (int)p.Ntot, g->Cs, g->Cb, p.T * 10 + g->stride + (small_scale || big_scale ? 100 : 0));
    p.stab_nb = (int)((p.chunk + HWs - 1) / HWs + 1);
    if (p.stab_nb > g->B) p.stab_nb = g->B;
    if (scaled) {
        // the scale table is indexed per stage: a stage must not straddle two samples
        GIF_REQUIRE((size_t)p.stab_nb * 2 * tl * sizeof(float) <= 64 * 1024, "conv2d_wgrad_f16: scale table too large");
        GIF_REQUIRE(HWs % 16 == 0, "conv2d_wgrad_f16: modulated weight gradient needs Hs*Ws %% 16 == 0 (got %ld)", HWs);
    }
    if (tl == 128 && wgrad_tile256_f16(g, scaled) && (!scaled || (size_t)p.stab_nb * 512 * sizeof(float) <= 32 * 1024)) {
        p.tiles_q = p.CP / 256;
        p.tiles_pq = (p.RP / 256) * p.tiles_q;
        const dim3 g256((unsigned)(p.tiles_pq * p.T * nsplit));
        if (!scaled) wgrad_launch<gif::f16, 256, 256, 2, 4, true, 32>(g256, 512, s, p);
        else wgrad_launch<gif::f16, 256, 256, 2, 4, true, 32, true>(g256, 512, s, p);
    } else if (tl == 128) {
        if (!scaled) wgrad_launch<gif::f16, 128, 128, 2, 2, true, 32>(grid, 256, s, p);
        else if (!st16) wgrad_launch<gif::f16, 128, 128, 2, 2, true, 32, true>(grid, 256, s, p);
        else wgrad_launch<gif::f16, 128, 128, 2, 2, true, 16, true>(grid, 256, s, p);
    } else if (tl == 64) {
        if (!scaled) wgrad_launch<gif::f16, 64, 64, 2, 2, true, 32>(grid, 256, s, p);
        else wgrad_launch<gif::f16, 64, 64, 2, 2, true, 32, true>(grid, 256, s, p);
    } else {
        if (!scaled) wgrad_launch<gif::f16, 32, 32, 1, 1, true, 32>(grid, 64, s, p);
        else if (!st16) wgrad_launch<gif::f16, 32, 32, 1, 1, true, 32, true>(grid, 64, s, p);
        else wgrad_launch<gif::f16, 32, 32, 1, 1, true, 16, true>(grid, 64, s, p);
    }
    return gif::check_launch("conv2d_wgrad_f16");
}

int gif_conv2d_wgrad_dims(int Cs, int Cb, int* RP, int* CP) {
    GIF_REQUIRE(Cs > 0 && Cb > 0 && RP && CP, "wgrad_dims: bad arguments");
    wgrad_dims(Cs, Cb, RP, CP);
    return 0;
}

int gif_conv2d_wgrad_splits(const gif_conv_geom* g) {
    if (!g || g->B <= 0) return 1;
    if (small_wgrad_ok(g, false)) return 256;  // one 16-wave workgroup per CU (conv_wgrad_small_mfma); scaled calls never
                                               // reach that kernel and simply use 256 splits of the generic one
    int RP, CP;
    wgrad_dims(g->Cs, g->Cb, &RP, &CP);
    // (scaled launches of the same geometry use 128-row tiles: they simply get half the splits they could use)
    long Ntot = (long)g->B * g->Hs * g->Ws;
    long tiles = (long)(RP / tile_rows(g->Cs, g->Cb, false, Ntot)) * (CP / tile_of(g->Cb)) * g->KH * g->KW;
    if (tile_of(g->Cs) == 128 && tile_of(g->Cb) == 32 && g->KH * g->KW > 1 && thin_taps_on() && h2v2_on() &&
        gif::fp32_mfma_mode() == GIF_FP32_MFMA_F16X2) {
        // thin big side: the f16x2 kernel (conv_wgrad_h2v2<false, true>, the only one that groups taps — the condition mirrors the
        // `h2 && x3_thin && thin_taps_on() && h2v2_on()` of conv2d_wgrad_f32_impl under the process's contraction mode) puts several taps
        // into one column tile: fewer, larger workgroups per split.  The native / bf16x3 kernels run one workgroup per tap and keep
        // their own count (advisor, round 5: they got ~4.5 x the splits, i.e. workspace and unpack traffic, for nothing)
        int tpt, tg;
        thin_tap_tiles(g->Cb, g->KH * g->KW, &tpt, &tg);
        tiles = (long)(RP / 128) * tg;
    }
#ifdef GIF_WGRAD_KX3_PROBE  // (timing probe: a third of the tap groups per split -> three times the splits)
    else if (g->KH * g->KW == 9 && gif::fp32_mfma_mode() == GIF_FP32_MFMA_F16X2) tiles /= 3;
#endif
    // 2 workgroups fit per CU (64 KB LDS each) => 512 concurrent slots on 256 CUs: fill k full rounds of 512
    // and never spill a few blocks into an extra, almost empty round (floor, not ceil)
    long want = tiles >= 1024 ? 1 : 1024 / tiles;
    long max_by_work = (Ntot + 4 * BKP_MAX - 1) / (4 * BKP_MAX);  // >= 4 stages per split
    long bytes_per_split = (long)g->KH * g->KW * RP * CP * 4;
    long max_by_mem = (128L << 20) / bytes_per_split;
    long n = want;
    if (n > max_by_work) n = max_by_work;
    if (n > max_by_mem) n = max_by_mem;
    if (n < 1) n = 1;
    return (int)n;
}

// x3: 0 native fp32 MFMA, 1 bf16x3, 2 f16x2 with the guarded bf16x3 fallback (launch shapes f16x2 is not built for run bf16x3)
static int conv2d_wgrad_f32_impl(const float* small, const float* big, float* ws, const float* small_scale,
                                 const float* big_scale, const gif_conv_geom* g, int nsplit, gif_stream_t stream, int x3_mode) {
    bool x3 = x3_mode != 0;
    GIF_REQUIRE(g && small && big && ws && nsplit >= 1, "conv2d_wgrad: bad arguments");
    GIF_REQUIRE(g->Cb % 4 == 0 && g->Cs % 4 == 0, "conv2d_wgrad: channels must be multiples of 4");
    GIF_REQUIRE(g->KH >= 1 && g->KH <= 3 && g->KW >= 1 && g->KW <= 3 && (g->stride == 1 || g->stride == 2),
                "conv2d_wgrad: unsupported kernel/stride");
    GIF_REQUIRE((long)g->B * g->Hs * g->Ws * g->Cs < (1L << 31) && (long)g->B * g->Hb * g->Wb * g->Cb < (1L << 31),
                "conv2d_wgrad: tensors of >= 2^31 elements are not supported (32-bit offsets)");
    WgradParams p{};
    p.sm = small; p.bg = big; p.ws = ws; p.ss = small_scale; p.bs = big_scale;
    p.B = g->B; p.Hs = g->Hs; p.Ws = g->Ws; p.Cs = g->Cs; p.Hb = g->Hb; p.Wb = g->Wb; p.Cb = g->Cb;
    p.KW = g->KW; p.stride = g->stride; p.pad = g->pad; p.T = g->KH * g->KW;
    wgrad_dims(g->Cs, g->Cb, &p.RP, &p.CP);
    p.Ntot = (long)g->B * g->Hs * g->Ws;
    long chunk = (p.Ntot + nsplit - 1) / nsplit;
    p.chunk = (chunk + BKP_MAX - 1) / BKP_MAX * BKP_MAX;
    if (p.chunk < BKP_MAX) p.chunk = BKP_MAX;
    const int bp = tile_of(g->Cs), bq = tile_of(g->Cb);
    // bf16x3: 128x128 tiles only (the 256x128 tile's 128 accumulator registers leave no room for the split operands); layers
    // with a <= 32-channel side stay on the native kernels (same operands, same workspace)
    const long HWs = (long)g->Hs * g->Ws;
    p.stab_nb = (int)((p.chunk + HWs - 1) / HWs + 1);
    if (p.stab_nb > g->B) p.stab_nb = g->B;
    const bool tab_fits = HWs % 16 == 0 && (size_t)p.stab_nb * 256 * sizeof(float) <= 64 * 1024;
    // bf16x3 tiles: 128x128, and 128x32 for the un-modulated layers with a thin big side (the 24-channel condition-noise maps)
    static const int x3_thin_off = gif::knob("GIF_X3_WGRAD_THIN") ? atoi(gif::knob("GIF_X3_WGRAD_THIN")) == 0 : 0;
    const bool x3_thin = x3 && !x3_thin_off && tile_of(g->Cs) == 128 && tile_of(g->Cb) == 32 && !small_scale && !big_scale;
    x3 = x3 && tile_of(g->Cs) == 128 && (tile_of(g->Cb) == 128 || x3_thin) && (!(small_scale || big_scale) || tab_fits);
    const bool big_tile = !x3 && wgrad_big_tile(g->Cs, g->Cb, small_scale || big_scale, p.Ntot) && !gif::knob("GIF_CONV_VARIANT");
    p.tiles_q = p.CP / bq;
    p.tiles_pq = (p.RP / (big_tile ? 256 : bp)) * p.tiles_q;
    dim3 grid((unsigned)(p.tiles_pq * p.T * nsplit));
    hipStream_t s = gif::as_stream(stream);
    p.zero = gif::zero_page16();
    GIF_REQUIRE(p.zero, "conv2d_wgrad: zero page lookup failed");
    double flops = 2.0 * p.Ntot * (double)g->Cs * g->Cb * p.T;
    if (small_wgrad_ok(g, small_scale || big_scale)) {
        gif::ProfScope prof(1, flops, s, (int)p.Ntot, g->Cs, g->Cb, p.T * 10 + g->stride);
        SmallWgradParams q{};
        q.sm = small; q.bg = big; q.ws = ws;
        q.B = g->B; q.H = g->Hs; q.W = g->Ws; q.Cs = g->Cs; q.Cb = g->Cb; q.RP = p.RP; q.CP = p.CP;
        q.Ntot = p.Ntot;
        long ch = (p.Ntot + nsplit - 1) / nsplit;
        q.chunk = (ch + 63) / 64 * 64;
        const size_t lds = (size_t)8 * 18 * 256 * sizeof(float);
        static gif::LdsAttr attr;
        attr.ensure(reinterpret_cast<const void*>(conv_wgrad_small_mfma), lds);
        hipLaunchKernelGGL(conv_wgrad_small_mfma, dim3((unsigned)nsplit), dim3(1024), lds, s, q);
        return gif::check_launch("conv2d_wgrad(small)");
    }
    {
        static const int x3_simple_p = gif::knob("GIF_X3_WGRAD_SIMPLE") ? atoi(gif::knob("GIF_X3_WGRAD_SIMPLE")) : 0;
        const bool tab_p = (small_scale || big_scale) && tab_fits;
        const bool h2_p = x3 && x3_mode == 2 && !x3_simple_p && (x3_thin || !tab_p || HWs % 32 == 0);  // (the same condition as below)
        gif::ProfScope prof(h2_p ? 15 : x3 ? 9 : 1, flops, s, (int)p.Ntot, g->Cs, g->Cb, p.T * 10 + g->stride + (small_scale || big_scale ? 100 : 0));
        const char* env = gif::knob("GIF_CONV_VARIANT");
        const int variant = env ? atoi(env) : 0;
        const bool glds = !small_scale && !big_scale && variant != 1;
#define GIF_WGRAD_LAUNCH(BP_, BQ_, WP_, WQ_, TH_)                                                                  \
    if (glds) wgrad_launch<float, BP_, BQ_, WP_, WQ_, true, 32>(grid, TH_, s, p);                                        \
    else wgrad_launch<float, BP_, BQ_, WP_, WQ_, false, 32>(grid, TH_, s, p)
        // Un-modulated f16x2 launches run the scale-table instantiation with unit scales (x 1.0f: bit-identical results) wherever it
        // applies: that instantiation's instruction stream is 3-5 % faster than the plain one on every 128 x 128-tile shape — 3.02 -> 2.91 ms
        // at 128@256^2, 2.92 -> 2.77 at 512@64^2, the modulated launches themselves 2.82 / 2.79 — for no reason visible in the source (the
        // extra multiply moves hipcc's interleave of the conversion); -0.6 ms per step, three alternating pairs.  GIF_H2_WGRAD_PLAIN_TAB=0: A/B
        static const int force_tab = gif::knob("GIF_H2_WGRAD_PLAIN_TAB") ? atoi(gif::knob("GIF_H2_WGRAD_PLAIN_TAB")) != 0 : 1;
        const bool tab = ((small_scale || big_scale) || (force_tab && x3 && x3_mode == 2 && HWs % 32 == 0)) && (variant != 1 || x3) && tab_fits;
        static const int x3_simple = gif::knob("GIF_X3_WGRAD_SIMPLE") ? atoi(gif::knob("GIF_X3_WGRAD_SIMPLE")) : 0;  // A/B: 16-pixel stages, no pipeline
        // f16x2: the software-pipelined 32-pixel-stage instantiations; the launch is followed by its guarded bf16x3 twin
        const bool h2 = x3 && x3_mode == 2 && !x3_simple && (x3_thin || !tab || HWs % 32 == 0);
        if (h2) {
            const gif::H2Gate gt = gif::h2_next_gate(s);
            if (gt.err) return gt.err;
            p.gate = gt.word; p.gate_gen = gt.gen; p.h2_stats = gif::h2_stats_words();
            if (x3_thin && thin_taps_on() && p.T > 1 && h2v2_on()) {
                // several taps per 128-column tile (conv_wgrad_h2v2<false, true>); the guarded twin below keeps its per-tap grid
                WgradParams q = p;
                thin_tap_tiles(g->Cb, p.T, &q.tpt, &q.tgroups);
                wgrad_launch_v2(false, dim3((unsigned)(p.tiles_pq * q.tgroups * nsplit)), s, q, true);
            } else if (x3_thin) wgrad_launch<float, 128, 32, 2, 1, true, 32, false, 2>(grid, 128, s, p);
            else if (h2v2_on()) wgrad_launch_v2(tab, grid, s, p);
            else if (tab) wgrad_launch<float, 128, 128, 2, 2, true, 32, true, 2>(grid, 256, s, p);
            else wgrad_launch<float, 128, 128, 2, 2, true, 32, false, 2>(grid, 256, s, p);
        }
        if (h2 && !p.gate) {
            // unguarded (GIF_H2_GUARD=0): done
        } else if (x3_thin) {
            // two waves of 64x32: 3 fragment splits per 12 MFMAs (four waves of 32x32: 2 per 6 — GIF_X3_WGRAD_THIN=4 for the A/B:
            // 128x24 at 256^2 76 -> 80 TFLOP/s, 256x24 at 128^2 70 -> 78, 512x24 at 64^2 80 -> 82)
            static const int thin4 = gif::knob("GIF_X3_WGRAD_THIN") ? atoi(gif::knob("GIF_X3_WGRAD_THIN")) == 4 : 0;
            if (thin4) wgrad_launch<float, 128, 32, 4, 1, true, 32, false, 1>(grid, 256, s, p);
            else wgrad_launch<float, 128, 32, 2, 1, true, 32, false, 1>(grid, 128, s, p);
        } else if (x3 && tab && HWs % 32 == 0 && !x3_simple) {
            wgrad_launch<float, 128, 128, 2, 2, true, 32, true, 1>(grid, 256, s, p);
        } else if (x3 && tab) {
            wgrad_launch<float, 128, 128, 2, 2, true, 16, true, 1>(grid, 256, s, p);
        } else if (x3 && !x3_simple) {
            wgrad_launch<float, 128, 128, 2, 2, true, 32, false, 1>(grid, 256, s, p);
        } else if (x3) {
            wgrad_launch<float, 128, 128, 2, 2, true, 16, false, 1>(grid, 256, s, p);
        } else if (big_tile) {
            wgrad_launch<float, 256, 128, 2, 2, true, 16>(grid, 256, s, p);
        } else if (bp == 128 && bq == 128 && tab) {
            // modulated wgrad (x*s, dy*d): LDS-DMA operands + scale table
            wgrad_launch<float, 128, 128, 2, 2, true, 16, true>(grid, 256, s, p);
        } else if (bp == 128 && bq == 128 && glds && variant != 7) {
            // 16-pixel stages: 32 KB of LDS per workgroup => 4 workgroups (16 waves) per CU; +6 % over 32-pixel stages
            wgrad_launch<float, 128, 128, 2, 2, true, 16>(grid, 256, s, p);
        } else if (bp == 128 && bq == 128) { GIF_WGRAD_LAUNCH(128, 128, 2, 2, 256); }
        else if (bp == 128 && bq == 32) { GIF_WGRAD_LAUNCH(128, 32, 4, 1, 256); }
        else if (bp == 32 && bq == 128) { GIF_WGRAD_LAUNCH(32, 128, 1, 4, 256); }
        else { GIF_WGRAD_LAUNCH(32, 32, 1, 1, 64); }
#undef GIF_WGRAD_LAUNCH
    }
    return gif::check_launch("conv2d_wgrad");
}

int gif_conv2d_wgrad_f32(const float* small, const float* big, float* ws, const float* small_scale,
                         const float* big_scale, const gif_conv_geom* g, int nsplit, gif_stream_t stream) {
    return conv2d_wgrad_f32_impl(small, big, ws, small_scale, big_scale, g, nsplit, stream, 0);
}

int gif_conv2d_wgrad_f32x3(const float* small, const float* big, float* ws, const float* small_scale,
                           const float* big_scale, const gif_conv_geom* g, int nsplit, gif_stream_t stream) {
    return conv2d_wgrad_f32_impl(small, big, ws, small_scale, big_scale, g, nsplit, stream, 1);
}

/* f16x2 (ABI 4): same contract; the f16x2 launch is followed by its guarded bf16x3 twin (a no-op unless the gate was raised) */
int gif_conv2d_wgrad_f32h2(const float* small, const float* big, float* ws, const float* small_scale,
                           const float* big_scale, const gif_conv_geom* g, int nsplit, gif_stream_t stream) {
    return conv2d_wgrad_f32_impl(small, big, ws, small_scale, big_scale, g, nsplit, stream, 2);
}

int gif_unpack_wgrad_f32(const float* ws, float* dw, int nsplit, int R, int C, int KH, int KW, int RP, int CP,
                         int64_t sr, int64_t sc, int64_t sky, int64_t skx, float scale, gif_stream_t stream) {
    GIF_REQUIRE(ws && dw && nsplit >= 1 && R > 0 && C > 0 && RP >= R && CP >= C, "unpack_wgrad: bad arguments");
    long total = (long)KH * KW * R * C;
    unpack_wgrad_kernel<<<gif::cdiv(total, 256), 256, 0, gif::as_stream(stream)>>>(ws, dw, nsplit, R, C, KH, KW, RP,
                                                                                     CP, sr, sc, sky, skx, scale);
    return gif::check_launch("unpack_wgrad");
}

/* ---- Winograd F(3x3,2x2) weight gradient of the stride-1 / pad-1 3x3 convolution -------------------------------------
 * dU_q[o][i] = sum_tiles Mg_q[tile][o] * V_q[tile][i] for the 16 transform positions q (a 1x1 wgrad per plane, run by
 * conv_wgrad_mfma in "planes" mode), then dW = A'^T dU A' in the unpack kernel: 16 instead of 36 multiplies per tile. */
int gif_conv3x3_winograd_wgrad_splits(int B, int H, int W, int Cs, int Cb) {
    if (B <= 0 || H <= 0 || W <= 0 || Cs <= 0 || Cb <= 0) return 1;
    int RP, CP;
    wgrad_dims(Cs, Cb, &RP, &CP);
    long Ntot = (long)B * (H / 2) * (W / 2);
    long tiles = (long)(RP / tile_rows(Cs, Cb, false, Ntot)) * (CP / tile_of(Cb)) * 16;
    long want = tiles >= 1024 ? 1 : 1024 / tiles;
    long max_by_work = (Ntot + 4 * BKP_MAX - 1) / (4 * BKP_MAX);
    long max_by_mem = (128L << 20) / (16L * RP * CP * 4);
    long n = want;
    if (n > max_by_work) n = max_by_work;
    if (n > max_by_mem) n = max_by_mem;
    if (n < 1) n = 1;
    return (int)n;
}

static int conv3x3_winograd_wgrad_impl(const float* x, const float* gy, float* V, float* Mg, float* ws,
                                       const float* small_scale, const float* big_scale, int B, int H, int W, int Cs, int Cb,
                                       int nsplit, gif_stream_t stream, int x3_mode) {
    bool x3 = x3_mode != 0;
    GIF_REQUIRE(gy && V && Mg && ws && nsplit >= 1, "winograd_wgrad: bad arguments");  // x == NULL: V is already filled
    GIF_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "winograd_wgrad: bad dims (H, W must be even)");
    GIF_REQUIRE(Cs > 0 && Cb > 0 && Cs % 4 == 0 && Cb % 4 == 0, "winograd_wgrad: channels must be multiples of 4");
    hipStream_t s = gif::as_stream(stream);
    const long ntiles = (long)B * (H / 2) * (W / 2);
    int CsP = 0, CbP = 0;
    long ntiles_pad = 0;
    gif::winograd_padded_dims(ntiles, Cs, &ntiles_pad, &CsP);
    gif::winograd_padded_dims(ntiles, Cb, &ntiles_pad, &CbP);
    GIF_REQUIRE(ntiles_pad * CsP < (1L << 31) && ntiles_pad * CbP < (1L << 31), "winograd_wgrad: tensor too large for 32-bit offsets");
    double flops = 2.0 * B * H * W * 9.0 * Cs * Cb;  // ALGORITHMIC (direct) FLOPs
    {
        gif::ProfScope prof_t(4, 4.0 * ((double)B * H * W * ((x ? Cb : 0) + Cs) + 16.0 * ntiles_pad * ((x ? CbP : 0) + CsP)), s,
                              (int)((long)B * H * W), Cb, Cs, 2);
        if (x)  // else: the caller kept the forward pass's V (same x, same modulation)
            if (int rc = gif::winograd_input_transform(x, big_scale, V, B, H, W, Cb, s)) return rc;
        if (int rc = gif::winograd_gy_transform(gy, small_scale, Mg, B, H, W, Cs, s)) return rc;
    }
    x3 = x3 && tile_of(CsP) == 128 && tile_of(CbP) == 128;
    static const int x3_simple_p = gif::knob("GIF_X3_WGRAD_SIMPLE") ? atoi(gif::knob("GIF_X3_WGRAD_SIMPLE")) : 0;
    gif::ProfScope prof((x3 && x3_mode == 2 && !x3_simple_p) ? 16 : x3 ? 11 : 3, flops, s, (int)((long)B * H * W), Cs, Cb,
                        1091 + (small_scale || big_scale ? 100 : 0));
    WgradParams p{};
    p.sm = Mg; p.bg = V; p.ws = ws; p.ss = nullptr; p.bs = nullptr;
    // one "image" of 1 x ntiles pixels per plane, 1x1 taps
    p.B = 1; p.Hs = 1; p.Ws = (int)ntiles; p.Cs = CsP; p.Hb = 1; p.Wb = (int)ntiles; p.Cb = CbP;
    p.KW = 1; p.stride = 1; p.pad = 0; p.T = 16;
    p.sm_plane = ntiles_pad * CsP;
    p.bg_plane = ntiles_pad * CbP;
    wgrad_dims(CsP, CbP, &p.RP, &p.CP);
    p.Ntot = ntiles;
    long chunk = (p.Ntot + nsplit - 1) / nsplit;
    p.chunk = (chunk + BKP_MAX - 1) / BKP_MAX * BKP_MAX;
    if (p.chunk < BKP_MAX) p.chunk = BKP_MAX;
    const int bp = tile_of(CsP), bq = tile_of(CbP);
    const bool big = !x3 && wgrad_big_tile(CsP, CbP, false, ntiles);
    p.tiles_q = p.CP / bq;
    p.tiles_pq = (p.RP / (big ? 256 : bp)) * p.tiles_q;
    dim3 grid((unsigned)(p.tiles_pq * p.T * nsplit));
    p.zero = gif::zero_page16();
    GIF_REQUIRE(p.zero, "winograd_wgrad: zero page lookup failed");
    static const int x3_simple = gif::knob("GIF_X3_WGRAD_SIMPLE") ? atoi(gif::knob("GIF_X3_WGRAD_SIMPLE")) : 0;
    const bool h2 = x3 && x3_mode == 2 && !x3_simple;
    if (h2) {
        const gif::H2Gate gt = gif::h2_next_gate(s);
        if (gt.err) return gt.err;
        p.gate = gt.word; p.gate_gen = gt.gen; p.h2_stats = gif::h2_stats_words();
        if (h2v2_on()) {
            // the plane GEMMs too run the scale-table instantiation with a one-row table of ones (see conv2d_wgrad_f32_impl: 3-5 % faster)
            static const int plain_tab = gif::knob("GIF_H2_WGRAD_PLAIN_TAB") ? atoi(gif::knob("GIF_H2_WGRAD_PLAIN_TAB")) != 0 : 1;
            WgradParams q = p;
            q.stab_nb = 1;
            wgrad_launch_v2(plain_tab != 0, grid, s, plain_tab ? q : p);
        }
        else wgrad_launch<float, 128, 128, 2, 2, true, 32, false, 2>(grid, 256, s, p);
    }
    if (h2 && !p.gate) {}
    else if (x3 && !x3_simple) wgrad_launch<float, 128, 128, 2, 2, true, 32, false, 1>(grid, 256, s, p);
    else if (x3) wgrad_launch<float, 128, 128, 2, 2, true, 16, false, 1>(grid, 256, s, p);
    else if (big) wgrad_launch<float, 256, 128, 2, 2, true, 16>(grid, 256, s, p);
    else if (bp == 128 && bq == 128) wgrad_launch<float, 128, 128, 2, 2, true, 16>(grid, 256, s, p);
    else if (bp == 128 && bq == 32) wgrad_launch<float, 128, 32, 4, 1, true, 32>(grid, 256, s, p);
    else if (bp == 32 && bq == 128) wgrad_launch<float, 32, 128, 1, 4, true, 32>(grid, 256, s, p);
    else wgrad_launch<float, 32, 32, 1, 1, true, 32>(grid, 64, s, p);
    return gif::check_launch("conv3x3_winograd_wgrad");
}

int gif_conv3x3_winograd_wgrad_f32(const float* x, const float* gy, float* V, float* Mg, float* ws,
                                   const float* small_scale, const float* big_scale, int B, int H, int W, int Cs, int Cb,
                                   int nsplit, gif_stream_t stream) {
    return conv3x3_winograd_wgrad_impl(x, gy, V, Mg, ws, small_scale, big_scale, B, H, W, Cs, Cb, nsplit, stream, 0);
}

int gif_conv3x3_winograd_wgrad_f32x3(const float* x, const float* gy, float* V, float* Mg, float* ws,
                                     const float* small_scale, const float* big_scale, int B, int H, int W, int Cs, int Cb,
                                     int nsplit, gif_stream_t stream) {
    return conv3x3_winograd_wgrad_impl(x, gy, V, Mg, ws, small_scale, big_scale, B, H, W, Cs, Cb, nsplit, stream, 1);
}

int gif_conv3x3_winograd_wgrad_f32h2(const float* x, const float* gy, float* V, float* Mg, float* ws,
                                     const float* small_scale, const float* big_scale, int B, int H, int W, int Cs, int Cb,
                                     int nsplit, gif_stream_t stream) {
    return conv3x3_winograd_wgrad_impl(x, gy, V, Mg, ws, small_scale, big_scale, B, H, W, Cs, Cb, nsplit, stream, 2);
}

int gif_winograd_unpack_wgrad_f32(const float* ws, float* dw, int nsplit, int R, int C, int RP, int CP, int64_t sr,
                                  int64_t sc, int64_t sky, int64_t skx, float scale, gif_stream_t stream) {
    GIF_REQUIRE(ws && dw && nsplit >= 1 && R > 0 && C > 0 && RP >= R && CP >= C, "winograd_unpack_wgrad: bad arguments");
    long total = (long)R * C;
    wino_unpack_wgrad_kernel<<<gif::cdiv(total, 256), 256, 0, gif::as_stream(stream)>>>(ws, dw, nsplit, R, C, RP, CP, sr,
                                                                                          sc, sky, skx, scale);
    return gif::check_launch("winograd_unpack_wgrad");
}
}
