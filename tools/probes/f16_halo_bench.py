"""GPU probe: the f16 halo kernel vs the gather kernel (GIF_F16_HALO=0) on the thin layers of BASELINE configs[4] (1024^2, batch 8).
Prints ms per launch and algorithmic TFLOP/s / GB/s per shape.  python tools/probes/f16_halo_bench.py [--batch 8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gif_amd import ops  # noqa: E402


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    B = a.batch
    CL = torch.channels_last
    # (what, direction, Cin_act, Cout, K, stride, pad, H of the op's INPUT, modulated)
    shapes = [("G conv 32->32 @1024 (modulated)", "fwd", 32, 32, 3, 1, 1, 1024, True),
              ("D conv1 32->32 @1024", "fwd", 32, 32, 3, 1, 1, 1024, False),
              ("dgrad 32->32 @1024", "bwd", 32, 32, 3, 1, 1, 1024, False),
              ("noise 24->32 @1024", "fwd", 24, 32, 3, 1, 1, 1024, False),
              ("noise 16->24 @1024", "fwd", 16, 24, 3, 1, 1, 1024, False),
              ("noise 8->16 @1024", "fwd", 8, 16, 3, 1, 1, 1024, False),
              ("ToRGB 32->3 @1024 (modulated)", "fwd", 32, 3, 1, 1, 0, 1024, True),
              ("D from-RGB 16->32 1x1 @1024", "fwd", 16, 32, 1, 1, 0, 1024, False),
              ("G up-conv 64->32 512->1025 (transposed, modulated)", "bwd", 64, 32, 3, 2, 0, 512, True),
              ("D conv2 dgrad 64->32 512->1025 (transposed)", "bwd", 64, 32, 3, 2, 0, 512, False),
              ("G conv 64->64 @512 (modulated)", "fwd", 64, 64, 3, 1, 1, 512, True),
              ("dgrad 64->64 @512", "bwd", 64, 64, 3, 1, 1, 512, False),
              ("G up-conv 128->64 256->513 (transposed, modulated)", "bwd", 128, 64, 3, 2, 0, 256, True)]
    print(f"batch {B}; ms per launch: halo / gather kernel; algorithmic TFLOP/s and GB/s (in + out once) of the halo launch")
    for what, direction, ci, co, k, st, pad, H, mod in shapes:
        spec = ops.ConvSpec(k, k, st, pad)
        x = torch.randn(B, ci, H, H, device="cuda").half().contiguous(memory_format=CL)
        if direction == "fwd":
            w = (torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5).cuda()
            epi = dict(in_scale=torch.rand(B, ci, device="cuda") + 0.5, out_scale=torch.rand(B, ops.cpad(co, torch.float16), device="cuda") + 0.5) if mod else {}
            fn = lambda: ops.conv_fwd(x, w, spec, **epi)  # noqa: E731
            Ho = spec.small_hw(H, H)[0]
            flops = 2.0 * B * Ho * Ho * k * k * ci * co
        else:
            w = (torch.randn(ci, co, k, k) / (ci * k * k) ** 0.5).cuda()  # forward conv co -> ci; its data gradient maps ci -> co
            Ho = spec.big_hw(H, H)[0]
            epi = dict(in_scale=torch.rand(B, ci, device="cuda") + 0.5, out_scale=torch.rand(B, ops.cpad(co, torch.float16), device="cuda") + 0.5) if mod else {}
            fn = lambda: ops.conv_bwd_data(x, w, spec, (Ho, Ho), **epi)  # noqa: E731
            flops = 2.0 * B * H * H * k * k * ci * co
        byts = 2.0 * B * (H * H * ci + Ho * Ho * ops.cpad(co, torch.float16))
        t = {}
        from gif_amd import _lib
        for on in ("1", "0"):
            _lib.load().gif_conv2d_f16_halo_enable(int(on))
            t[on] = timed(fn)
        _lib.load().gif_conv2d_f16_halo_enable(1)
        print(f"{what:58s} {t['1']:8.3f} / {t['0']:8.3f} ms  x{t['0'] / t['1']:5.2f}   {flops / t['1'] / 1e9:7.1f} TFLOP/s  {byts / t['1'] / 1e6:7.0f} GB/s")


def wgrads(B):
    CL = torch.channels_last
    print("weight gradients: ms per launch, halo kernel / per-tap kernel (GIF_F16_HALO_WGRAD is read once per process: run twice)")
    for what, ci, co, k, pad, H in [("wgrad 32x32 3x3 @1024", 32, 32, 3, 1, 1024), ("wgrad 32x24 @1024", 24, 32, 3, 1, 1024),
                                    ("wgrad 24x16 @1024", 16, 24, 3, 1, 1024), ("wgrad 16x8 @1024", 8, 16, 3, 1, 1024),
                                    ("wgrad ToRGB 8x32 1x1 @1024", 32, 8, 1, 0, 1024), ("wgrad 24x16 @512", 16, 24, 3, 1, 512)]:
        x = torch.randn(B, ci, H, H, device="cuda").half().contiguous(memory_format=CL)
        gy = torch.randn(B, co, H, H, device="cuda").half().contiguous(memory_format=CL)
        spec = ops.ConvSpec(k, k, 1, pad)
        t = timed(lambda: ops.conv_wgrad(gy, x, spec, co, ci))
        byts = 2.0 * B * H * H * (ci + co)
        print(f"{what:40s} {t:8.3f} ms   {2.0 * B * H * H * k * k * ci * co / t / 1e9:7.1f} TFLOP/s  {byts / t / 1e6:7.0f} GB/s")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--wgrad":
        wgrads(8)
        sys.exit(0)
    main()
