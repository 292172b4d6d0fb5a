// Internal helpers shared by the gfx950 kernels of libgif_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "gif_hip.h"

// Timing probe only (tools/probes/x3_three_products.sh): -DGIF_X3_FIRST_TERM=3 builds the bf16x3 kernels with the three smallest
// of their six products left out — WRONG numerics (16-bit products), the upper bound of what a two-piece split could buy.
#ifndef GIF_X3_FIRST_TERM
#define GIF_X3_FIRST_TERM 0
#endif

namespace gif {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

#define GIF_REQUIRE(cond, ...)                \
    do {                                      \
        if (!(cond)) {                        \
            ::gif::set_error(__VA_ARGS__);    \
            return GIF_EINVAL;                \
        }                                     \
    } while (0)

inline hipStream_t as_stream(gif_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// A/B, ablation and probe knobs (GIF_* environment variables that select a measured-slower, diagnostic or deliberately wrong
// path) are honoured ONLY in a process that opted in with GIF_EXPERIMENTAL=1: a stray variable in a production environment
// can change neither dispatch nor numerics.  NOT gated (documented interface): GIF_FP32_MFMA (contraction mode, gif_hip.h),
// GIF_PROF_DUMP (profiling output).  The Python side mirrors this in gif_amd._lib.knob.
inline const char* knob(const char* name) {
    static const bool on = [] { const char* e = ::getenv("GIF_EXPERIMENTAL"); return e && atoi(e) != 0; }();
    return on ? ::getenv(name) : nullptr;
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- buffer-addressed LDS-DMA (device pass only; the host pass sees empty stand-ins).  A raw buffer descriptor (stride 0) over
// `bytes` of memory: `buffer_load_dwordx4 v_off, s[rsrc], s_off offen lds` moves 16 bytes per lane straight into LDS (lane-linear
// destination behind the M0 base) and writes ZEROS for a lane whose byte offset (VGPR + SGPR part) is outside [0, bytes) — probed in
// tools/probes/buffer_lds_oob_probe.hip (round 4) — so padding and tails cost one select instead of bounds checks, a 64-bit address
// and a zero-page select per piece.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t make_buf_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void buf_load_lds16(buf_rsrc_t r, __attribute__((address_space(3))) void* lds, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds, 16, (int)voff, soff, 0, 0);
}
#else
struct buf_rsrc_t {};
__device__ __forceinline__ buf_rsrc_t make_buf_rsrc(const void*, unsigned) { return {}; }
__device__ __forceinline__ void buf_load_lds16(buf_rsrc_t, __attribute__((address_space(3))) void*, unsigned, int) {}
#endif

// ---- per-device launch state (runtime.hip).  The library may be driven for several devices of one process (tensors on
// cuda:1, DataParallel-style hosts) and from several threads (forward thread + autograd thread): nothing device-specific is
// cached in a plain static.
constexpr int kMaxDevices = 64;
int current_device();           // hipGetDevice(), validated against kMaxDevices
const float* zero_page16();     // 16 zero bytes in the CURRENT device's memory (source of out-of-range LDS-DMA lanes)
unsigned* f16_sat_flag();       // the CURRENT device's "an f16 store saturated" word (gif_f16_overflow_clear / _or_into)
// "this kernel may use `bytes` of dynamic LDS on the current device": hipFuncSetAttribute once per (kernel, device, size step)
struct LdsAttr {
    unsigned long granted[kMaxDevices] = {};
    void ensure(const void* kernel, size_t bytes);
};

// ---- fp32 contraction mode (runtime.hip; gif_set_fp32_mfma_mode / GIF_FP32_MFMA) ----
int fp32_mfma_mode();

// ---- kernel profiling (runtime.hip): HIP events around launches, grouped in families ----
// 0 direct conv fwd/dgrad on the LDS-DMA kernel (Cin >= 32; flops), 1 direct wgrad (flops), 2 Winograd GEMM fwd/dgrad
// (ALGORITHMIC direct-conv flops; the kernel executes 16/36 of them), 3 Winograd wgrad GEMM (same convention),
// 4 Winograd transforms (HBM bytes), 5 direct conv fwd/dgrad on the register-staged kernel (Cin < 32; flops),
// 6 / 7 f16 conv / wgrad, 8 / 9 bf16x3 conv fwd/dgrad / wgrad (algorithmic flops; the bf16 pipe executes 6x),
// 10 / 11 bf16x3 Winograd GEMM fwd/dgrad / Winograd wgrad plane GEMMs (algorithmic flops; execute 6 * 16/36 of them),
// 12 bf16x3 direct conv in the tap-dense K order, 13 / 14 f16x2 direct conv fwd/dgrad / f16x2 Winograd GEMM (algorithmic flops; the
// f16 pipe executes 3x / 3 * 16/36 of them; the op's time includes its guarded bf16x3 twin launch), 15 / 16 f16x2 weight gradient /
// f16x2 Winograd weight-gradient plane GEMMs (same conventions), 17 f16x2 direct conv in the tap-dense K order
#define GIF_PROF_FAMILIES 18
struct ProfScope {
    int family;
    hipStream_t stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(int family, double flops, hipStream_t s, int d0 = 0, int d1 = 0, int d2 = 0, int d3 = 0);
    ~ProfScope();
};

// ---- fixed-order reduction of per-tile partial sums (elementwise.hip): out[y][c] = sum_k partial[y][k][c], y < Y, k < nblk.
// Long single-group reductions (Y == 1, nblk > 512) go through `tmp` (>= 64 * C floats) in two launches.
int reduce_partials(const float* partial, float* out, int Y, int nblk, int C, float* tmp, hipStream_t s);
// partial rows an epilogue reduction may need for `rows` output rows (conv tiles of >= 64 rows, FIR blocks <= 4096) + tmp
inline int64_t epilogue_ws_rows(int64_t rows) { return 2 * (rows / 64 + 16) + 4096 + 64; }

// ---- Winograd transforms (conv_winograd.hip), shared with the Winograd wgrad in conv_wgrad.hip ----
// padded dims of a transformed operand [16][ntiles_pad][CP]
void winograd_padded_dims(long ntiles, int C, long* ntiles_pad, int* CP);
// V = B^T d B per 4x4 input patch (x [B,H,W,C], optional per-sample channel scale [B,C])
int winograd_input_transform(const float* x, const float* scale, float* V, int B, int H, int W, int C, hipStream_t s);
// Mg = G g G^T per 2x2 tile of gy [B,H,W,C] (F(3x3,2x2) "filter" transform of the output gradient)
int winograd_gy_transform(const float* gy, const float* scale, float* Mg, int B, int H, int W, int C, hipStream_t s);

// ---- activation element types.  fp32 is the reference dtype; f16 is BASELINE config 5 ("fp16 activations with fp32
// demodulation"): activations live in HBM as IEEE half, every kernel computes in fp32 (MFMA accumulators, epilogues,
// reductions, per-sample scales) and converts on load / store.  Stores saturate at the largest finite half so that one
// overflowing activation cannot turn into inf -> NaN downstream.
typedef _Float16 f16;
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// 4 consecutive channels of an NHWC tensor <-> float4 (16-byte access for fp32, 8-byte for f16)
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const f16* p) {
    const f16x4_t h = *reinterpret_cast<const f16x4_t*>(p);
    return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}
__device__ __forceinline__ void store4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float sat_f16(float v) { return fminf(fmaxf(v, -65504.f), 65504.f); }
__device__ __forceinline__ void store4(f16* p, float4 v) {
    f16x4_t h;
    h[0] = (f16)sat_f16(v.x); h[1] = (f16)sat_f16(v.y); h[2] = (f16)sat_f16(v.z); h[3] = (f16)sat_f16(v.w);
    *reinterpret_cast<f16x4_t*>(p) = h;
}
// Stores that may carry GRADIENTS: a saturating store hides an overflow from the loss scaler (the fp32 weight gradients
// computed from clamped activations gradients stay finite), so the clamp — or a non-finite value — raises the device's flag
// word; the trainer clears it before backward() and ORs it into found_inf afterwards (gif_f16_overflow_*).  fp32: plain store.
__device__ __forceinline__ void store4_flag(float* p, float4 v, unsigned*) { store4(p, v); }
__device__ __forceinline__ void store4_flag(f16* p, float4 v, unsigned* flag) {
    if (flag) {
        const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        if (!(m <= 65504.f)) atomicOr(flag, 1u);  // also true for NaN
    }
    store4(p, v);
}

// ---- bf16x3: three-way split of fp32 operands for the bf16 matrix cores -------------------------------------------
// a = hi + mid + lo with hi = bf16(a), mid = bf16(a - hi), lo = bf16(a - hi - mid), every conversion round-to-nearest-even
// (v_cvt_pk_bf16_f32).  Both subtractions are exact in fp32 (a rounding residual always fits), |mid| <= 2^-8 |a|,
// |lo| <= 2^-16 |a|, and hi + mid + lo reproduces a to 2^-24 |a| or better (exactly, whenever the last residual has <= 8
// significant bits).  bf16 has fp32's exponent range, so there is no scaling and no overflow / underflow caveat.
// 8 consecutive-k floats of a lane (two 16-byte LDS fragments) -> one operand of v_mfma_f32_32x32x16_bf16 per term.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// one pair of floats -> the packed {a1, a0} dword of each term (9 VALU: 3 cvt_pk, 2 shl, 2 and, 2 packed subtracts)
__device__ __forceinline__ void split_pair(const float a0, const float a1, unsigned& hi, unsigned& mid, unsigned& lo) {
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a0, a1}, bf16x2_t));
    const float r0 = a0 - __uint_as_float(h << 16), r1 = a1 - __uint_as_float(h & 0xffff0000u);
    const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{r0, r1}, bf16x2_t));
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    hi = h;
    mid = m;
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{s0, s1}, bf16x2_t));
}

// Same split for two floats that do NOT sit in an aligned register pair (operands gathered by scalar LDS reads: the weight
// gradient's eight ds_read_b32 per fragment).  The packed subtract of split_pair wants 64-bit aligned pairs and hipcc pays for
// it with one v_mov per element behind a full lgkmcnt(0) wait; the empty asm statements keep the two subtractions scalar.
__device__ __forceinline__ void split_pair_scalar(const float a0, const float a1, unsigned& hi, unsigned& mid, unsigned& lo) {
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a0, a1}, bf16x2_t));
    float r0 = a0 - __uint_as_float(h << 16);
    asm volatile("" : "+v"(r0));
    float r1 = a1 - __uint_as_float(h & 0xffff0000u);
    const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{r0, r1}, bf16x2_t));
    float s0 = r0 - __uint_as_float(m << 16);
    asm volatile("" : "+v"(s0));
    float s1 = r1 - __uint_as_float(m & 0xffff0000u);
    hi = h;
    mid = m;
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{s0, s1}, bf16x2_t));
}

// ---- f16x2 ("h2"): fp32 contractions as THREE f16 MFMA products under per-row power-of-two scales ---------------
// x * 2^e = hi + lo, hi = rne_f16(x * 2^e), lo = rne_f16(x * 2^e - hi): 11 + 11 significand bits while lo stays in f16's normal
// range, an absolute floor of 2^-25 (half the f16 denormal quantum: v_mfma_f32_32x32x16_f16 keeps denormal inputs on gfx950,
// tools/probes/f16x2_hw_probe.hip) below it.  a * b ~ hi*hi + hi*lo + lo*hi (the dropped lo*lo is <= 2^-22 of the product), fp32
// accumulation, result * 2^-(e_a + e_b).  The exponent e belongs to a ROW of the GEMM (it factors out of the dot product):
//   weights: one exponent per packed row, chosen by the packing kernel (row maximum -> [2^14, 2^15));
//   activations: a RUNNING exponent per row inside the kernel: a new row maximum is scaled into [2^13, 2^14) and the row's
//   accumulators are multiplied by the (exact) power of two; later elements may grow 4x before the next rescale.
// Guard ("window").  An element is represented to max(2^-22 |x|, 2^-25 * 2^-e): 22 bits relative, or an absolute floor of 2^-38 of
// its row maximum (the maximum is scaled to >= 2^13).  For a dot product over K groups of 16 elements let w_a(g) / w_b(g) be the
// "spread" of group g in its row (log2 of row maximum / group maximum).  The floors contribute at most 2^-38 A B 2^-w_b(g) and
// 2^-38 A B 2^-w_a(g) per group; the dot product's largest group product is A B 2^-m with m = min_g (w_a(g) + w_b(g)) — the error
// floor is 2^(m - 38) of it, next to the native fp32 kernel's 2^-24 of the same quantity.  m <= the window of whichever operand
// has ALL its (non-zero) groups inside it, so ONE in-window operand is enough: weights whose rows the packing did not flag (all
// groups within 2^kH2WindowW: m <= 16, floor 2^-22) make every activation harmless — smooth activations, ReLU-sparse gradients
// and the differencing positions of a Winograd transform produce near-zero groups all the time.  Only a launch in which a row
// with a group beyond 2^kH2Window meets a FLAGGED weight row (weight gradient: narrow groups on both sides) raises its gate word
// and is recomputed by the bf16x3 kernel (gated relaunch: the bf16x3 launch that follows returns at once unless the gate is raised).
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
constexpr int kH2Target = 13;        // activations: a new row maximum lands in [2^13, 2^14)
constexpr int kH2TargetW = 14;       // weights (static): the row maximum lands in [2^14, 2^15)
constexpr float kH2Limit = 65472.f;  // rescale before |x| * 2^e reaches the f16 overflow threshold (65520)
constexpr int kH2Window = 14;        // activations: exponent-field distance (row maximum, smallest non-zero group maximum) allowed
constexpr int kH2WindowW = 16;       // weights: one binade more headroom comes from the tighter scaling target
// exponent e with m * 2^e in [2^target, 2^(target+1)), clamped to what a float scale factor can hold
__device__ __host__ __forceinline__ int h2_exp_for(unsigned m_bits, int target) {
    const int e = target + 127 - (int)((m_bits >> 23) & 0xffu);
    return e > 126 ? 126 : (e < -126 ? -126 : e);
}
__device__ __forceinline__ float h2_pow2(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }  // -126 <= e <= 127
// one pair of floats -> the packed {x1, x0} f16 dwords of the two terms (6 VALU: pk_mul, cvt_pk, 2 cvt, pk_fma / 2 sub, cvt_pk)
__device__ __forceinline__ void split_pair_h2(const float a0, const float a1, const float sc, unsigned& hi, unsigned& lo) {
    const float x0 = a0 * sc, x1 = a1 * sc;
    const f16x2_t h = __builtin_convertvector(f32x2_t{x0, x1}, f16x2_t);
    const float r0 = x0 - (float)h[0], r1 = x1 - (float)h[1];  // exact: the rounding residual of an fp32 value fits fp32
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{r0, r1}, f16x2_t));
}
// the same with two scalar multiplies / subtracts (no v_pk_*_f32 with a register-pair element select)
__device__ __forceinline__ void split_pair_h2_scalar(const float a0, const float a1, const float sc, unsigned& hi, unsigned& lo) {
    float x0 = a0 * sc;
    asm volatile("" : "+v"(x0));
    const float x1 = a1 * sc;
    const f16x2_t h = __builtin_convertvector(f32x2_t{x0, x1}, f16x2_t);
    float r0 = x0 - (float)h[0];
    asm volatile("" : "+v"(r0));
    const float r1 = x1 - (float)h[1];
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{r0, r1}, f16x2_t));
}
// header of an f16x2 weight packing: int32 exponents of the RP rows, then RP int32 row flags ("a 16-element K group of this row
// lies outside the window": launches that use the row take the bf16x3 fallback); the f16 planes follow.
inline size_t h2_header_bytes(int RP) { return (size_t)2 * RP * sizeof(int); }
// gate words of the guarded launches (runtime.hip).  Eager launches: a ring of device words and a generation counter — the f16x2
// kernel raises its gate with atomicMax(gate, gen), the fallback launch runs iff *gate >= gen; a word is reused 65536 guarded launches
// later under a larger gen, and generations only grow, so nothing is ever reset and streams do not matter.  (">=", round 6: should a
// LATER user of the word — 65536 guarded launches on, possible only across streams — have raised it first, the earlier launch's twin
// runs as well: a spurious bf16x3 recomputation, never a silently skipped one.)  Launches recorded by a stream CAPTURE (hipGraph)
// cannot bake a generation into their arguments — every replay would see the value of the first — so they get a word of their own from
// a separate pool (never shared with eager launches), generation 1 and a memset node in front of the f16x2 kernel that clears the word
// on every replay.  The pool holds kH2CaptureGates words per device for the life of the process (a graph's words cannot be reclaimed:
// the library does not see graph destruction); when it is exhausted the launch fails with GIF_ENOSUP instead of running unguarded.
struct H2Gate {
    unsigned* word;
    unsigned gen;
    int err;  // != 0: no gate could be provided (set_error has the reason); the caller returns it
};
H2Gate h2_next_gate(hipStream_t s);  // {NULL, 0, 0} when the guard is switched off (GIF_H2_GUARD=0 under GIF_EXPERIMENTAL=1)
unsigned* h2_stats_words();   // [0] guarded ops that took the fallback (one count per op), [1..3] unused

}  // namespace gif
