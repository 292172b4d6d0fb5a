"""bf16x3 vs native fp32 MFMA on the direct conv kernels: error against an fp64 ATen convolution and time per launch.
Usage (GPU): python tools/probes/x3_probe.py [--batch 32]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gif_amd import ops  # noqa: E402

ops.WINOGRAD_MIN_C = ops.WINOGRAD_WGRAD_MIN_C = 0  # (round 6: Winograd from 256 channels in f16x2 mode by default; this tool pins the route itself)
from tools.kernel_bench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    B = a.batch
    dev = "cuda"
    torch.manual_seed(0)
    ops.WINOGRAD = False  # the direct kernels are what is being compared
    # accuracy: small enough for an fp64 reference
    def err(a, b):
        return float((a.double() - b).abs().max() / b.abs().max())

    print("# error vs fp64 (max |diff| / max |ref|), batch 4")
    for ci, co, k, s, p, h in ((128, 128, 3, 1, 1, 192), (128, 256, 3, 2, 0, 257), (128, 128, 3, 1, 1, 32), (512, 512, 3, 1, 1, 16), (128, 256, 3, 2, 0, 33), (256, 128, 1, 1, 0, 32)):
        spec = ops.ConvSpec(k, k, s, p)
        x = torch.randn(4, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(co, ci, k, k, device=dev) / (ci * k * k) ** 0.5
        sc = torch.rand(4, ci, device=dev) + 0.5
        ref = F.conv2d(x.double() * sc.double()[:, :, None, None], w.double(), stride=s, padding=p)
        hs, ws_ = spec.small_hw(h, h)
        gy = torch.randn(4, co, hs, ws_, device=dev).contiguous(memory_format=torch.channels_last)
        refd = F.conv_transpose2d(gy.double(), w.double(), stride=s, padding=p,
                                  output_padding=(h - ((hs - 1) * s + k - 2 * p), h - ((ws_ - 1) * s + k - 2 * p)))
        wd = w.double().requires_grad_(True)
        sd = torch.rand(4, co, device=dev) + 0.5
        (refw,) = torch.autograd.grad(F.conv2d(x.double() * sc.double()[:, :, None, None], wd, stride=s, padding=p),
                                      wd, gy.double() * sd.double()[:, :, None, None])
        for mode in ("native", "bf16x3"):
            ops.set_fp32_mfma_mode(mode)
            y = ops.conv_fwd(x, w, spec, in_scale=sc)[:, :co]
            gx = ops.conv_bwd_data(gy, w, spec, (h, h))[:, :ci]
            e_f = float((y.double() - ref).abs().max() / ref.abs().max())
            e_d = float((gx.double() - refd).abs().max() / refd.abs().max())
            line = f"{ci:4d}->{co:4d} k{k} s{s} @{h:3d} {mode:7s}: fwd {e_f:.2e}  dgrad {e_d:.2e}"
            for wino in (False, True):
                ops.WINOGRAD = wino
                gw = ops.conv_wgrad(gy, x, spec, co, ci, small_scale=sd, big_scale=sc)
                line += f"  wgrad{'(wino)' if wino else ''} {float((gw.double() - refw).abs().max() / refw.abs().max()):.2e}"
            if (k, s, p) == (3, 1, 1) and h % 2 == 0:
                ops.WINOGRAD, mt = True, ops.WINOGRAD_MIN_TILES
                ops.WINOGRAD_MIN_TILES = 1
                yw = ops.conv_fwd(x, w, spec, in_scale=sc)[:, :co]
                gxw = ops.conv_bwd_data(gy, w, spec, (h, h))[:, :ci]
                ops.WINOGRAD_MIN_TILES = mt
                line += f"  wino fwd {err(yw, ref):.2e} dgrad {err(gxw, refd):.2e}"
            ops.WINOGRAD = False
            print(line)
    print(f"# time per launch, batch {B}")
    shapes = [("128->128 @256", 128, 128, 3, 1, 1, 256), ("256->256 @128", 256, 256, 3, 1, 1, 128),
              ("512->512 @64", 512, 512, 3, 1, 1, 64), ("512->512 @32", 512, 512, 3, 1, 1, 32),
              ("512->512 @16", 512, 512, 3, 1, 1, 16), ("512->512 @8", 512, 512, 3, 1, 1, 8),
              ("128->256 s2 @257", 128, 256, 3, 2, 0, 257), ("256->512 s2 @129", 256, 512, 3, 2, 0, 129),
              ("128->256 1x1 @128", 128, 256, 1, 1, 0, 128), ("128->3 1x1 @256", 128, 3, 1, 1, 0, 256)]
    for name, ci, co, k, s, p, h in shapes:
        spec = ops.ConvSpec(k, k, s, p)
        x = torch.randn(B, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(co, ci, k, k, device=dev)
        sc = torch.rand(B, ci, device=dev) + 0.5
        hs, ws_ = spec.small_hw(h, h)
        gy = torch.randn(B, ops.pad4(co), hs, ws_, device=dev).contiguous(memory_format=torch.channels_last)
        fl = 2.0 * B * hs * ws_ * co * ci * k * k
        line = f"{name:20s}"
        for mode in ("native", "bf16x3"):
            ops.set_fp32_mfma_mode(mode)
            t_f = timeit(lambda: ops.conv_fwd(x, w, spec))
            t_m = timeit(lambda: ops.conv_fwd(x, w, spec, in_scale=sc))
            t_d = timeit(lambda: ops.conv_bwd_data(gy, w, spec, (h, h)))
            t_w = timeit(lambda: ops.conv_wgrad(gy, x, spec, co, ci))
            sd = torch.rand(B, ops.pad4(co), device=dev) + 0.5
            t_wm = timeit(lambda: ops.conv_wgrad(gy, x, spec, co, ci, small_scale=sd, big_scale=sc))
            ops.WINOGRAD = True
            t_ww = timeit(lambda: ops.conv_wgrad(gy, x, spec, co, ci))
            t_wf = timeit(lambda: ops.conv_fwd(x, w, spec, in_scale=sc))
            t_wd = timeit(lambda: ops.conv_bwd_data(gy, w, spec, (h, h)))
            ops.WINOGRAD = False
            line += (f" | {mode}: fwd {fl / t_f / 1e9:6.1f} mod {fl / t_m / 1e9:6.1f} dgrad {fl / t_d / 1e9:6.1f} wgrad {fl / t_w / 1e9:6.1f} "
                     f"wmod {fl / t_wm / 1e9:6.1f} wwino {fl / t_ww / 1e9:6.1f} wino-fwd {fl / t_wf / 1e9:6.1f} wino-dgrad {fl / t_wd / 1e9:6.1f}")
        print(line)


if __name__ == "__main__":
    main()
