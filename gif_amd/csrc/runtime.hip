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

// f16x2 guard (common.h): gate words of the guarded launches and the fallback statistics
constexpr unsigned kH2GateRing = 65536;
constexpr unsigned kH2CaptureGates = 65536;
__device__ unsigned g_h2_gates[kH2GateRing];
__device__ unsigned g_h2_capture_gates[kH2CaptureGates];
__device__ unsigned g_h2_stats[4];

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

static unsigned* h2_symbol(const void* sym, unsigned* (&cache)[kMaxDevices]) {
    static std::mutex mu;
    const int d = current_device();
    std::lock_guard<std::mutex> lk(mu);
    if (!cache[d]) {
        void* ptr = nullptr;
        if (hipGetSymbolAddress(&ptr, sym) != hipSuccess) return nullptr;
        cache[d] = static_cast<unsigned*>(ptr);
    }
    return cache[d];
}

unsigned* h2_stats_words() {
    static unsigned* cache[kMaxDevices] = {};
    return h2_symbol(HIP_SYMBOL(g_h2_stats), cache);
}

H2Gate h2_next_gate(hipStream_t s) {
    static unsigned* cache[kMaxDevices] = {};
    static unsigned* ccache[kMaxDevices] = {};
    static std::atomic<unsigned long long> next[kMaxDevices];
    static std::atomic<unsigned long long> cnext[kMaxDevices];
    static const bool off = gif::knob("GIF_H2_GUARD") && atoi(gif::knob("GIF_H2_GUARD")) == 0;
    if (off) return {nullptr, 0, 0};
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) {
        (void)hipGetLastError();
        cs = hipStreamCaptureStatusNone;
    }
    if (cs != hipStreamCaptureStatusNone) {
        // recorded launch: a private word, cleared by a memset node on every replay, generation 1 (common.h)
        unsigned* pool = h2_symbol(HIP_SYMBOL(g_h2_capture_gates), ccache);
        const unsigned long long n = cnext[current_device()].fetch_add(1, std::memory_order_relaxed);
        if (!pool || n >= kH2CaptureGates) {
            set_error("f16x2 guard: the %u gate words for stream-captured launches are used up (each captured guarded launch keeps one "
                      "for the life of the process); re-use the captured graphs or run eagerly", kH2CaptureGates);
            return {nullptr, 0, GIF_ENOSUP};
        }
        if (hipMemsetAsync(pool + n, 0, sizeof(unsigned), s) != hipSuccess) {
            set_error("f16x2 guard: hipMemsetAsync of a captured gate word failed");
            return {nullptr, 0, GIF_ENOSUP};
        }
        return {pool + n, 1u, 0};
    }
    unsigned* ring = h2_symbol(HIP_SYMBOL(g_h2_gates), cache);
    if (!ring) {
        set_error("f16x2 guard: gate ring lookup failed");
        return {nullptr, 0, GIF_ENOSUP};
    }
    const unsigned long long n = next[current_device()].fetch_add(1, std::memory_order_relaxed);
    // the words start at zero, generations at 1; a word's generation only grows (atomicMax), so a stale raise of an earlier
    // launch on the same word never reaches a later launch's generation
    return {ring + (n % kH2GateRing), (unsigned)(n / kH2GateRing) + 1u, 0};
}

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
    m = GIF_FP32_MFMA_F16X2;
    if (e && (!strcmp(e, "native") || !strcmp(e, "0"))) m = GIF_FP32_MFMA_NATIVE;
    if (e && (!strcmp(e, "bf16x3") || !strcmp(e, "1"))) m = GIF_FP32_MFMA_BF16X3;
    if (e && (!strcmp(e, "f16x2") || !strcmp(e, "2"))) m = GIF_FP32_MFMA_F16X2;
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
// 4: f16x2 contraction mode (GIF_FP32_MFMA_F16X2, gif_pack_weight_f32h2, gif_conv2d_*_f32h2, gif_h2_fallback_stats)
int gif_abi_version(void) { return 4; }

int gif_set_fp32_mfma_mode(int mode) {
    if (mode != GIF_FP32_MFMA_NATIVE && mode != GIF_FP32_MFMA_BF16X3 && mode != GIF_FP32_MFMA_F16X2) {
        gif::set_error("set_fp32_mfma_mode: unknown mode %d", mode);
        return GIF_EINVAL;
    }
    gif::g_fp32_mode.store(mode, std::memory_order_relaxed);
    return 0;
}

int gif_get_fp32_mfma_mode(void) { return gif::fp32_mfma_mode(); }

/* f16x2 guard statistics of the current device since the last reset: out[0] = guarded ops that took the bf16x3 fallback (the first
 * twin launch of an op counts: phases and remainder launches do not), out[1] = 0.  Synchronises the device. */
int gif_h2_fallback_stats(uint64_t* out2, int reset) {
    unsigned* st = gif::h2_stats_words();
    GIF_REQUIRE(st && out2, "h2_fallback_stats: null pointer");
    unsigned h[4] = {0, 0, 0, 0};
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    if (e == hipSuccess && reset) e = hipMemset(st, 0, sizeof(h));
    if (e != hipSuccess) { gif::set_error("h2_fallback_stats: %s", hipGetErrorString(e)); return (int)e; }
    out2[0] = h[0];
    out2[1] = 0;
    return 0;
}

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
