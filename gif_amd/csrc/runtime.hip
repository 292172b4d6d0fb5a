// Error reporting + event-based kernel timing for bench.py's roofline line.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"

namespace gif {

static thread_local char g_err[512] = "";

__device__ __attribute__((aligned(16))) float g_zero_page[4];
__device__ unsigned g_f16_sat_flag;

__global__ void f16_flag_or_into(const unsigned* flag, float* found_inf) {
    if (*flag) *found_inf = 1.f;
}

int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) return 0;
    return d;
}

const float* zero_page16() {
    static std::mutex mu;
    static const float* page[kMaxDevices] = {};
    const int d = current_device();
    std::lock_guard<std::mutex> lk(mu);
    if (!page[d]) {
        void* zp = nullptr;
        if (hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero_page)) != hipSuccess) return nullptr;
        page[d] = static_cast<const float*>(zp);
    }
    return page[d];
}

// Between gif_f16_overflow_clear() and gif_f16_overflow_or_into() (the trainer brackets backward() with them) the f16 launches
// get the flag pointer; outside that window they get NULL and their stores pay nothing for the check.
static std::atomic<int> g_f16_watch{0};

unsigned* f16_sat_flag_word() {
    static std::mutex mu;
    static unsigned* flag[kMaxDevices] = {};
    const int d = current_device();
    std::lock_guard<std::mutex> lk(mu);
    if (!flag[d]) {
        void* fp = nullptr;
        if (hipGetSymbolAddress(&fp, HIP_SYMBOL(g_f16_sat_flag)) != hipSuccess) return nullptr;
        flag[d] = static_cast<unsigned*>(fp);
    }
    return flag[d];
}

unsigned* f16_sat_flag() { return g_f16_watch.load(std::memory_order_relaxed) ? f16_sat_flag_word() : nullptr; }

void LdsAttr::ensure(const void* kernel, size_t bytes) {
    static std::mutex mu;
    const int d = current_device();
    std::lock_guard<std::mutex> lk(mu);
    if (bytes > granted[d]) {
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        granted[d] = bytes;
    }
}

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct ProfRec {
    hipEvent_t e0, e1;
    double flops;
    int d[4];  // launch shape tag (M, N, K, taps) for the optional per-launch dump (GIF_PROF_DUMP=<file>)
};
static bool g_prof_on = false;
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof[GIF_PROF_FAMILIES];
static std::vector<hipEvent_t> g_pool;

static hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

ProfScope::ProfScope(int fam, double flops, hipStream_t s, int d0, int d1, int d2, int d3) : family(fam), stream(s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    e0 = get_event();
    e1 = get_event();
    if (!e0 || !e1) {
        e0 = e1 = nullptr;
        return;
    }
    g_prof[family].push_back({e0, e1, flops, {d0, d1, d2, d3}});
    (void)hipEventRecord(e0, stream);
}

// -1 = not decided yet: the first query reads GIF_FP32_MFMA
static std::atomic<int> g_fp32_mode{-1};

int fp32_mfma_mode() {
    int m = g_fp32_mode.load(std::memory_order_relaxed);
    if (m >= 0) return m;
    const char* e = getenv("GIF_FP32_MFMA");
    m = GIF_FP32_MFMA_BF16X3;
    if (e && (!strcmp(e, "native") || !strcmp(e, "0"))) m = GIF_FP32_MFMA_NATIVE;
    g_fp32_mode.store(m, std::memory_order_relaxed);
    return m;
}

ProfScope::~ProfScope() {
    if (e1) (void)hipEventRecord(e1, stream);
}

}  // namespace gif

extern "C" {

const char* gif_last_error(void) { return gif::g_err; }

int gif_f16_overflow_clear(gif_stream_t stream) {
    unsigned* f = gif::f16_sat_flag_word();
    GIF_REQUIRE(f, "f16_overflow_clear: no flag word on this device");
    hipError_t e = hipMemsetAsync(f, 0, sizeof(unsigned), gif::as_stream(stream));
    if (e != hipSuccess) { gif::set_error("f16_overflow_clear: %s", hipGetErrorString(e)); return (int)e; }
    gif::g_f16_watch.store(1, std::memory_order_relaxed);
    return 0;
}

int gif_f16_overflow_or_into(float* found_inf, gif_stream_t stream) {
    unsigned* f = gif::f16_sat_flag_word();
    GIF_REQUIRE(f && found_inf, "f16_overflow_or_into: null pointer");
    gif::g_f16_watch.store(0, std::memory_order_relaxed);
    gif::f16_flag_or_into<<<1, 1, 0, gif::as_stream(stream)>>>(f, found_inf);
    return gif::check_launch("f16_overflow_or_into");
}

int gif_f16_overflow_watch(int on) {
    gif::g_f16_watch.store(on ? 1 : 0, std::memory_order_relaxed);
    return 0;
}

// 2: gif_conv_epilogue gradient-producer fusions, rasteriser workspace (B, F, H, W)
// 3: gif_f16_overflow_watch, gif_pack_nhwc / gif_unpack_nhwc, gif_conv2d_f16_halo_eligible (f16 halo kernels), rasteriser clean-workspace per pointer,
//    gif_linear_bank_fwd / _bwd (modulation bank)
int gif_abi_version(void) { return 3; }

int gif_set_fp32_mfma_mode(int mode) {
    if (mode != GIF_FP32_MFMA_NATIVE && mode != GIF_FP32_MFMA_BF16X3) {
        gif::set_error("set_fp32_mfma_mode: unknown mode %d", mode);
        return GIF_EINVAL;
    }
    gif::g_fp32_mode.store(mode, std::memory_order_relaxed);
    return 0;
}

int gif_get_fp32_mfma_mode(void) { return gif::fp32_mfma_mode(); }

int gif_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(gif::g_prof_mu);
    gif::g_prof_on = on != 0;
    return 0;
}

int gif_prof_read(int family, double* ms, double* flops, int64_t* launches) {
    if (family < 0 || family >= GIF_PROF_FAMILIES) return GIF_EINVAL;
    std::lock_guard<std::mutex> lk(gif::g_prof_mu);
    double tms = 0, tf = 0;
    int64_t n = 0;
    const char* dump = getenv("GIF_PROF_DUMP");
    FILE* fh = dump ? fopen(dump, "a") : nullptr;
    for (auto& r : gif::g_prof[family]) {
        (void)hipEventSynchronize(r.e1);
        float t = 0;
        if (hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) {
            tms += t;
            tf += r.flops;
            ++n;
            if (fh) fprintf(fh, "%d,%d,%d,%d,%d,%.6f,%.0f\n", family, r.d[0], r.d[1], r.d[2], r.d[3], t, r.flops);
        }
        gif::g_pool.push_back(r.e0);
        gif::g_pool.push_back(r.e1);
    }
    if (fh) fclose(fh);
    gif::g_prof[family].clear();
    if (ms) *ms = tms;
    if (flops) *flops = tf;
    if (launches) *launches = n;
    return 0;
}
}
