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

// ---- conv profiling (prof.hip) ----
struct ProfScope {
    int family;
    hipStream_t stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(int family, double flops, hipStream_t s, int d0 = 0, int d1 = 0, int d2 = 0, int d3 = 0);
    ~ProfScope();
};

}  // namespace gif
