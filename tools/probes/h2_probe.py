"""f16x2 direct kernels next to bf16x3 and native fp32 MFMA against an fp64 convolution; quick timing of the big layers.
Run on the GPU box: python tools/probes/h2_probe.py [--time]"""
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from gif_amd import ops  # noqa: E402

ops.WINOGRAD_MIN_C = ops.WINOGRAD_WGRAD_MIN_C = 0  # (round 6: Winograd from 256 channels in f16x2 mode by default; this tool pins the route itself)

CASES = [
    (4, 128, 128, 3, 1, 1, 192),
    (4, 128, 256, 3, 2, 0, 257),
    (4, 256, 256, 3, 1, 1, 64),
    (4, 512, 512, 3, 1, 1, 16),
    (2, 128, 256, 3, 2, 0, 33),
    (4, 128, 24, 3, 1, 1, 64),
    (2, 256, 128, 1, 1, 0, 32),
    (3, 160, 96, 3, 1, 1, 20),
    (32, 512, 512, 3, 1, 1, 32),
]


def err(got, ref):
    return float((got.double() - ref).abs().max() / ref.abs().max())


def main():
    ops.WINOGRAD = False
    dev = "cuda"
    for case in CASES:
        B, ci, co, k, s, p, h = case
        torch.manual_seed(sum(case))
        spec = ops.ConvSpec(k, k, s, p)
        x = torch.randn(B, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(co, ci, k, k, device=dev) / (ci * k * k) ** 0.5
        sc, sd = torch.rand(B, ci, device=dev) + 0.5, torch.rand(B, ops.pad4(co), device=dev) + 0.5
        hs, ws_ = spec.small_hw(h, h)
        gy = torch.randn(B, ops.pad4(co), hs, ws_, device=dev).contiguous(memory_format=torch.channels_last)
        gy[:, co:] = 0
        xd, wd, gyd = x.double(), w.double(), gy[:, :co].double()
        ref_f = F.conv2d(xd * sc.double()[:, :, None, None], wd, stride=s, padding=p)
        op = (h - ((hs - 1) * s + k - 2 * p), h - ((ws_ - 1) * s + k - 2 * p))
        ref_d = F.conv_transpose2d(gyd * sd[:, :co].double()[:, :, None, None], wd, stride=s, padding=p, output_padding=op)
        line = f"{case}:"
        for mode in ("native", "bf16x3", "f16x2"):
            ops.set_fp32_mfma_mode(mode)
            e_f = err(ops.conv_fwd(x, w, spec, in_scale=sc)[:, :co], ref_f)
            e_d = err(ops.conv_bwd_data(gy, w, spec, (h, h), in_scale=sd)[:, :ci], ref_d)
            e_f0 = err(ops.conv_fwd(x, w, spec)[:, :co], F.conv2d(xd, wd, stride=s, padding=p))
            line += f"  {mode} fwd {e_f:.2e} dgrad {e_d:.2e} plain {e_f0:.2e}"
        print(line, " fallbacks", ops.h2_fallback_stats(reset=True), flush=True)

    # rows whose magnitude GROWS along K (every rescale path): channel c scaled by 2^(c / 4), taps in increasing order too
    B, C, H = 2, 128, 32
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, C, H, H, generator=g) * torch.pow(2.0, torch.arange(C) / 4.0)[None, :, None, None]
    x = x.cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, C, 3, 3, generator=g) / 34).cuda()
    ref = F.conv2d(x.double(), w.double(), padding=1)
    spec = ops.ConvSpec(3, 3, 1, 1)
    for mode in ("native", "bf16x3", "f16x2"):
        ops.set_fp32_mfma_mode(mode)
        print("growing rows", mode, f"{err(ops.conv_fwd(x, w, spec), ref):.2e}", "fallbacks", ops.h2_fallback_stats(reset=True))
    # window violation: 16 channels at 2^-24 with weights at 2^24 on them -> the guard must send the op to bf16x3
    xs = torch.randn(B, C, H, H, generator=g)
    xs[:, 32:48] *= 2.0 ** -24
    ws = torch.randn(C, C, 3, 3, generator=g) / 34
    ws[:, 32:48] *= 2.0 ** 24
    xs, ws = xs.cuda().contiguous(memory_format=torch.channels_last), ws.cuda()
    ref = F.conv2d(xs.double(), ws.double(), padding=1)
    for mode in ("native", "bf16x3", "f16x2"):
        ops.set_fp32_mfma_mode(mode)
        print("window violation", mode, f"{err(ops.conv_fwd(xs, ws, spec), ref):.2e}", "fallbacks", ops.h2_fallback_stats(reset=True))
    for mag in (1e-38, 1e-30, 1e-15, 1e15, 1e30):
        xm = (torch.randn(B, C, H, H, generator=g) * mag).cuda().contiguous(memory_format=torch.channels_last)
        ref = F.conv2d(xm.double(), w.double(), padding=1)
        line = f"magnitude {mag:g}:"
        for mode in ("native", "bf16x3", "f16x2"):
            ops.set_fp32_mfma_mode(mode)
            y = ops.conv_fwd(xm, w, spec)
            line += f"  {mode} {err(y, ref):.2e} finite={bool(torch.isfinite(y).all())}"
        print(line, "fallbacks", ops.h2_fallback_stats(reset=True))

    # weight gradients (direct, modulated, thin, Winograd plane GEMMs)
    for (B, ci, co, k, s_, p_, h, wino) in [(4, 128, 128, 3, 1, 1, 64, False), (4, 128, 256, 3, 2, 0, 65, False), (4, 24, 128, 3, 1, 1, 64, False),
                                          (2, 256, 256, 1, 1, 0, 32, False), (4, 128, 128, 3, 1, 1, 64, True), (2, 256, 512, 3, 1, 1, 32, True),
                                          (3, 160, 96, 3, 1, 1, 20, False)]:
        torch.manual_seed(B + ci + co + h)
        ops.WINOGRAD = wino
        ops.WINOGRAD_MIN_TILES, ops.WINOGRAD_WGRAD_MIN_TILES = (1, 1) if wino else (8192, 2048)
        spec = ops.ConvSpec(k, k, s_, p_)
        x = torch.randn(B, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
        hs, ws_ = spec.small_hw(h, h)
        gy = torch.randn(B, co, hs, ws_, device=dev).contiguous(memory_format=torch.channels_last)
        sc, sd = torch.rand(B, ci, device=dev) + 0.5, torch.rand(B, co, device=dev) + 0.5
        wd = torch.zeros(co, ci, k, k, device=dev, dtype=torch.float64, requires_grad=True)
        (ref,) = torch.autograd.grad(F.conv2d(x.double(), wd, stride=s_, padding=p_), wd, gy.double())
        wd2 = torch.zeros(co, ci, k, k, device=dev, dtype=torch.float64, requires_grad=True)
        (ref_s,) = torch.autograd.grad(F.conv2d(x.double() * sc.double()[:, :, None, None], wd2, stride=s_, padding=p_), wd2,
                                       gy.double() * sd.double()[:, :, None, None])
        line = f"wgrad {(B, ci, co, k, s_, h, 'wino' if wino else 'direct')}:"
        for mode in ("native", "bf16x3", "f16x2"):
            ops.set_fp32_mfma_mode(mode)
            n0 = ops.prof_winograd_calls()
            e0 = err(ops.conv_wgrad(gy, x, spec, co, ci), ref)
            e1 = err(ops.conv_wgrad(gy, x, spec, co, ci, small_scale=sd, big_scale=sc), ref_s)
            line += f"  {mode} {e0:.2e} scaled {e1:.2e}" + (" (wino)" if ops.prof_winograd_calls() > n0 else "")
        print(line, " fallbacks", ops.h2_fallback_stats(reset=True), flush=True)
    ops.WINOGRAD = False
    ops.WINOGRAD_MIN_TILES, ops.WINOGRAD_WGRAD_MIN_TILES = 8192, 2048

    if "--time" in sys.argv:
        for (B, ci, co, k, s_, p_, h, wino) in [(32, 128, 256, 3, 2, 0, 257, False), (32, 256, 512, 3, 2, 0, 129, False), (32, 24, 128, 3, 1, 1, 256, False),
                                              (32, 128, 128, 3, 1, 1, 256, True), (32, 256, 256, 3, 1, 1, 128, True), (32, 512, 512, 3, 1, 1, 64, True)]:
            ops.WINOGRAD = wino
            spec = ops.ConvSpec(k, k, s_, p_)
            x = torch.randn(B, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
            hs, ws_ = spec.small_hw(h, h)
            gy = torch.randn(B, co, hs, ws_, device=dev).contiguous(memory_format=torch.channels_last)
            flops = 2.0 * B * hs * ws_ * k * k * ci * co
            line = f"wgrad {(B, ci, co, k, s_, h, 'wino' if wino else 'direct')}:"
            for mode in ("bf16x3", "f16x2"):
                ops.set_fp32_mfma_mode(mode)
                for _ in range(3):
                    ops.conv_wgrad(gy, x, spec, co, ci)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    ops.conv_wgrad(gy, x, spec, co, ci)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 10
                line += f"  {mode} {dt * 1e3:.3f} ms {flops / dt / 1e12:.0f} TF"
            print(line, flush=True)
        ops.WINOGRAD = False

        shapes = [(32, 128, 256, 3, 2, 0, 257, "fwd"), (32, 128, 256, 3, 2, 0, 257, "dgrad"), (32, 128, 128, 3, 1, 1, 256, "fwd"),
                  (32, 256, 256, 3, 1, 1, 128, "fwd"), (32, 512, 512, 3, 1, 1, 16, "fwd"), (32, 128, 24, 3, 1, 1, 256, "fwd"),
                  (32, 256, 512, 3, 2, 0, 129, "dgrad"), (32, 512, 512, 3, 2, 0, 65, "fwd")]
        for (B, ci, co, k, s, p, h, what) in shapes:
            spec = ops.ConvSpec(k, k, s, p)
            x = torch.randn(B, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
            w = torch.randn(co, ci, k, k, device=dev) / (ci * k * k) ** 0.5
            hs, ws_ = spec.small_hw(h, h)
            gy = torch.randn(B, co, hs, ws_, device=dev).contiguous(memory_format=torch.channels_last)
            flops = 2.0 * B * hs * ws_ * k * k * ci * co
            line = f"{(B, ci, co, k, s, h, what)}:"
            for mode in ("bf16x3", "f16x2"):
                ops.set_fp32_mfma_mode(mode)
                fn = (lambda: ops.conv_fwd(x, w, spec)) if what == "fwd" else (lambda: ops.conv_bwd_data(gy, w, spec, (h, h)))
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 10
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n
                line += f"  {mode} {dt * 1e3:.3f} ms {flops / dt / 1e12:.0f} TF"
            print(line, flush=True)


if __name__ == "__main__":
    main()
