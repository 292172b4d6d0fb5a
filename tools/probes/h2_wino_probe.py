"""f16x2 Winograd GEMM next to bf16x3 / native against fp64 (fwd with the fused epilogue, dgrad), and timing of the big layers."""
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from gif_amd import ops  # noqa: E402

ops.WINOGRAD_MIN_C = ops.WINOGRAD_WGRAD_MIN_C = 0  # (round 6: Winograd from 256 channels in f16x2 mode by default; this tool pins the route itself)

dev = "cuda"


def err(got, ref):
    return float((got.double() - ref).abs().max() / ref.abs().max())


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


ops.WINOGRAD = True
ops.WINOGRAD_MIN_TILES = 1
for (B, ci, co, h) in [(4, 128, 128, 64), (2, 256, 512, 32), (3, 512, 256, 16), (2, 128, 128, 34), (32, 128, 128, 64)]:
    torch.manual_seed(B + ci + co + h)
    spec = ops.ConvSpec(3, 3, 1, 1)
    x = cl(torch.randn(B, ci, h, h, device=dev))
    w = torch.randn(co, ci, 3, 3, device=dev) / (ci * 9) ** 0.5
    sc, sd = torch.rand(B, ci, device=dev) + 0.5, torch.rand(B, co, device=dev) + 0.5
    bias = torch.randn(co, device=dev)
    res = cl(torch.randn(B, co, h, h, device=dev))
    gy = cl(torch.randn(B, co, h, h, device=dev))
    z = F.conv2d(x.double() * sc.double()[:, :, None, None], w.double(), padding=1) * sd.double()[:, :, None, None]
    ref_f = 2 ** 0.5 * F.leaky_relu(z + res.double() + bias.double()[None, :, None, None], 0.2)
    ref_d = F.conv_transpose2d(gy.double() * sd.double()[:, :, None, None], w.double(), padding=1) * sc.double()[:, :, None, None]
    line = f"wino {(B, ci, co, h)}:"
    for mode in ("native", "bf16x3", "f16x2"):
        ops.set_fp32_mfma_mode(mode)
        n0 = ops.prof_winograd_calls()
        y = ops.conv_fwd(x, w, spec, in_scale=sc, out_scale=sd, bias=bias, residual=res, act=True, slope=0.2, gain=2 ** 0.5)
        gx = ops.conv_bwd_data(gy, w, spec, (h, h), in_scale=sd, out_scale=sc)
        assert ops.prof_winograd_calls() == n0 + 2
        line += f"  {mode} fwd {err(y, ref_f):.2e} dgrad {err(gx, ref_d):.2e}"
    print(line, " fallbacks", ops.h2_fallback_stats(reset=True), flush=True)

# rows growing along K and across positions; window violation
B, C, H = 2, 128, 32
g = torch.Generator().manual_seed(3)
spec = ops.ConvSpec(3, 3, 1, 1)
w = (torch.randn(C, C, 3, 3, generator=g) / 34).cuda()
xg = cl((torch.randn(B, C, H, H, generator=g) * torch.pow(2.0, torch.arange(C) * (12.0 / C))[None, :, None, None]).cuda())
xs = torch.randn(B, C, H, H, generator=g)
xs[:, 32:48] *= 2.0 ** -24
ws = torch.randn(C, C, 3, 3, generator=g) / 34
ws[:, 32:48] *= 2.0 ** 24
xs, ws = cl(xs.cuda()), ws.cuda()
for name, xx, ww in (("growing rows", xg, w), ("window violation", xs, ws)):
    ref = F.conv2d(xx.double(), ww.double(), padding=1)
    for mode in ("native", "bf16x3", "f16x2"):
        ops.set_fp32_mfma_mode(mode)
        print(name, mode, f"{err(ops.conv_fwd(xx, ww, spec), ref):.2e}", "fallbacks", ops.h2_fallback_stats(reset=True))
for mag in (1e-38, 1e-30, 1e30):
    xm = cl((torch.randn(B, C, H, H, generator=g) * mag).cuda())
    ref = F.conv2d(xm.double(), w.double(), padding=1)
    line = f"magnitude {mag:g}:"
    for mode in ("native", "bf16x3", "f16x2"):
        ops.set_fp32_mfma_mode(mode)
        y = ops.conv_fwd(xm, w, spec)
        line += f"  {mode} {err(y, ref):.2e} finite={bool(torch.isfinite(y).all())}"
    print(line, "fallbacks", ops.h2_fallback_stats(reset=True))

ops.WINOGRAD_MIN_TILES = 8192
for (B, c, h) in [(32, 128, 256), (32, 256, 128), (32, 512, 64), (32, 512, 32)]:
    spec = ops.ConvSpec(3, 3, 1, 1)
    x = cl(torch.randn(B, c, h, h, device=dev))
    w = torch.randn(c, c, 3, 3, device=dev) / (c * 9) ** 0.5
    flops = 2.0 * B * h * h * 9 * c * c
    line = f"wino fwd {(B, c, h)}:"
    for mode in ("bf16x3", "f16x2"):
        ops.set_fp32_mfma_mode(mode)
        for _ in range(3):
            ops.conv_fwd(x, w, spec)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ops.conv_fwd(x, w, spec)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        line += f"  {mode} {dt * 1e3:.3f} ms {flops / dt / 1e12:.0f} TF (incl. input transform)"
    print(line, flush=True)
