// Internal helpers shared by the gfx950 kernels of libgif_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "gif_hip.h"

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

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- per-device launch state (runtime.hip).  The library may be driven for several devices of one process (tensors on
// cuda:1, DataParallel-style hosts) and from several threads (forward thread + autograd thread): nothing device-specific is
// cached in a plain static.
constexpr int kMaxDevices = 64;
int current_device();           // hipGetDevice(), validated against kMaxDevices
const float* zero_page16();     // 16 zero bytes in the CURRENT device's memory (source of out-of-range LDS-DMA lanes)
// "this kernel may use `bytes` of dynamic LDS on the current device": hipFuncSetAttribute once per (kernel, device, size step)
struct LdsAttr {
    unsigned long granted[kMaxDevices] = {};
    void ensure(const void* kernel, size_t bytes);
};

// ---- kernel profiling (runtime.hip): HIP events around launches, grouped in families ----
// 0 direct conv fwd/dgrad on the LDS-DMA kernel (Cin >= 32; flops), 1 direct wgrad (flops), 2 Winograd GEMM fwd/dgrad
// (ALGORITHMIC direct-conv flops; the kernel executes 16/36 of them), 3 Winograd wgrad GEMM (same convention),
// 4 Winograd transforms (HBM bytes), 5 direct conv fwd/dgrad on the register-staged kernel (Cin < 32; flops)
#define GIF_PROF_FAMILIES 8
struct ProfScope {
    int family;
    hipStream_t stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(int family, double flops, hipStream_t s, int d0 = 0, int d1 = 0, int d2 = 0, int d3 = 0);
    ~ProfScope();
};

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

}  // namespace gif
