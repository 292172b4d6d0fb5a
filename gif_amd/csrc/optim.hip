// Fused Adam (+ generator EMA) over the flat gradient bucket, gfx950.
//
// Replaces, per optimiser step, torch.optim.Adam.step() (train.py:173, :243; Adam(lr, betas=(0, 0.99**r)), no weight decay,
// no amsgrad) and generic_utils.accumulate (my_utils/generic_utils.py:63-76, the EMA generator) — ~100 multi-tensor launches —
// by ONE launch.  HBM-bound: per parameter element p r/w, g r, m r/w, v r/w (+ ema r/w) = 28 (36) bytes.
// Gradients, exp_avg and exp_avg_sq live in flat buffers that share one offset table (train_step.FlatGradBucket); the
// parameters (and EMA parameters) stay wherever torch allocated them and are reached through a chunk table, so model.to(),
// load_state_dict() or checkpoint code never see a re-bound storage.
// Arithmetic follows torch's single-tensor Adam: m = lerp(m, g, 1 - b1); v = v * b2 + (1 - b2) * g * g;
// p -= step_size * m / (sqrt(v) / sqrt(bc2) + eps), step_size = lr / bc1 (bias corrections computed on the host in double).
#include "common.h"

namespace {

constexpr int CHUNK = 4096;  // floats per workgroup (256 lanes x 4 float4)

struct AdamArgs {
    const gif_adam_chunk* chunks;
    const float* g;
    float* m;
    float* v;
    float b1, b2, eps, step_size, bc2_sqrt, ema_decay;
    int has_ema;
    const float* inv_scale;  // device scalar: gradients are multiplied by it (loss scaling of the f16 path), or NULL
    const float* found_inf;  // device scalar: non-zero => a gradient overflowed, the whole update is skipped, or NULL
    const float* dev_step;   // device scalar: number of APPLIED steps incl. this one (the bias corrections follow it), or NULL
    float lr;
};

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamArgs& a) {
    const float w = 1.f - a.b1;  // exp_avg.lerp_(grad, 1 - beta1), ATen's two-sided formula (exact at beta1 = 0: m = g)
    m = w < 0.5f ? m + w * (g - m) : g - (g - m) * (1.f - w);
    v = v * a.b2 + (1.f - a.b2) * g * g;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p - a.step_size * (m / denom);
}

__global__ void __launch_bounds__(256) adam_ema_kernel(AdamArgs a) {
    if (a.found_inf && *a.found_inf != 0.f) return;  // uniform over the grid: nobody updates (GradScaler semantics)
    if (a.dev_step) {  // loss-scaled run: skipped steps do not advance the count, so the corrections come from the device
        const double t = (double)*a.dev_step;
        a.step_size = (float)((double)a.lr / (1.0 - pow((double)a.b1, t)));
        a.bc2_sqrt = (float)sqrt(1.0 - pow((double)a.b2, t));
    }
    const float gs = a.inv_scale ? *a.inv_scale : 1.f;
    const gif_adam_chunk c = a.chunks[blockIdx.x];
    float* __restrict__ p = c.param;
    float* __restrict__ e = c.ema;
    const float* __restrict__ g = a.g + c.flat_offset;
    float* __restrict__ m = a.m + c.flat_offset;
    float* __restrict__ v = a.v + c.flat_offset;
    const bool ema = a.has_ema && e != nullptr;
    const float d = a.ema_decay, omd = 1.f - a.ema_decay;
    // flat offsets are multiples of 64 floats; parameter storage from the torch allocator is 16-byte aligned in practice,
    // but a view into a larger tensor need not be: take the scalar path then
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | (ema ? reinterpret_cast<uintptr_t>(e) : 0)) & 15) == 0;
    if (vec) {
        const int n4 = c.n >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) {
            float4 pv = reinterpret_cast<float4*>(p)[i];
            float4 gv = reinterpret_cast<const float4*>(g)[i];
            gv.x *= gs; gv.y *= gs; gv.z *= gs; gv.w *= gs;
            float4 mv = reinterpret_cast<float4*>(m)[i];
            float4 vv = reinterpret_cast<float4*>(v)[i];
            adam_elem(pv.x, gv.x, mv.x, vv.x, a);
            adam_elem(pv.y, gv.y, mv.y, vv.y, a);
            adam_elem(pv.z, gv.z, mv.z, vv.z, a);
            adam_elem(pv.w, gv.w, mv.w, vv.w, a);
            reinterpret_cast<float4*>(p)[i] = pv;
            reinterpret_cast<float4*>(m)[i] = mv;
            reinterpret_cast<float4*>(v)[i] = vv;
            if (ema) {
                float4 ev = reinterpret_cast<float4*>(e)[i];
                ev.x = ev.x * d + omd * pv.x; ev.y = ev.y * d + omd * pv.y;
                ev.z = ev.z * d + omd * pv.z; ev.w = ev.w * d + omd * pv.w;
                reinterpret_cast<float4*>(e)[i] = ev;
            }
        }
        for (int i = (n4 << 2) + threadIdx.x; i < c.n; i += 256) {
            float pv = p[i], mv = m[i], vv = v[i];
            adam_elem(pv, g[i] * gs, mv, vv, a);
            p[i] = pv; m[i] = mv; v[i] = vv;
            if (ema) e[i] = e[i] * d + omd * pv;
        }
    } else {
        for (int i = threadIdx.x; i < c.n; i += 256) {
            float pv = p[i], mv = m[i], vv = v[i];
            adam_elem(pv, g[i] * gs, mv, vv, a);
            p[i] = pv; m[i] = mv; v[i] = vv;
            if (ema) e[i] = e[i] * d + omd * pv;
        }
    }
}

}  // namespace

extern "C" {

int gif_adam_chunk_floats(void) { return CHUNK; }

int gif_adam_ema_step_f32(const gif_adam_chunk* chunks, int nchunks, const float* grad_flat, float* exp_avg_flat,
                          float* exp_avg_sq_flat, float lr, float beta1, float beta2, float eps, double bias_correction1,
                          double bias_correction2, float ema_decay, int has_ema, const float* inv_grad_scale,
                          const float* found_inf, const float* dev_step, gif_stream_t stream) {
    GIF_REQUIRE(chunks && grad_flat && exp_avg_flat && exp_avg_sq_flat && nchunks >= 0, "adam_ema_step: null pointer");
    GIF_REQUIRE(bias_correction1 > 0.0 && bias_correction2 > 0.0, "adam_ema_step: bias corrections must be positive");
    if (nchunks == 0) return 0;
    AdamArgs a;
    a.chunks = chunks; a.g = grad_flat; a.m = exp_avg_flat; a.v = exp_avg_sq_flat;
    a.b1 = beta1; a.b2 = beta2; a.eps = eps;
    a.step_size = (float)((double)lr / bias_correction1);
    a.bc2_sqrt = (float)sqrt(bias_correction2);
    a.ema_decay = ema_decay; a.has_ema = has_ema;
    a.inv_scale = inv_grad_scale; a.found_inf = found_inf;
    a.dev_step = dev_step; a.lr = lr;
    adam_ema_kernel<<<nchunks, 256, 0, gif::as_stream(stream)>>>(a);
    return gif::check_launch("adam_ema_step");
}
}
