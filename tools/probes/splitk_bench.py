"""GPU probe: the low-resolution 512-channel bf16x3 launches with and without split-K (GIF_SPLITK is read once per process: run twice,
or use --all to spawn both).  python tools/probes/splitk_bench.py"""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def timed(fn, n=20):
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
    from gif_amd import ops
    CL = torch.channels_last
    print(f"GIF_SPLITK={os.environ.get('GIF_SPLITK', '1')} GIF_SPLITK_MAX_TILES={os.environ.get('GIF_SPLITK_MAX_TILES', '384')}")
    for B in (32, 64):
        for H in (4, 8, 16):
            x = torch.randn(B, 512, H, H, device="cuda").contiguous(memory_format=CL)
            w = (torch.randn(512, 512, 3, 3) / 68).cuda()
            s = torch.rand(B, 512, device="cuda") + 0.5
            spec = ops.ConvSpec(3, 3, 1, 1)
            fl = 2.0 * B * H * H * 9 * 512 * 512
            t_f = timed(lambda: ops.conv_fwd(x, w, spec))
            t_m = timed(lambda: ops.conv_fwd(x, w, spec, in_scale=s, out_scale=s))
            t_d = timed(lambda: ops.conv_bwd_data(x, w, spec, (H, H)))
            xs = torch.randn(B, 512, 2 * H + 1, 2 * H + 1, device="cuda").contiguous(memory_format=CL)
            t_s2 = timed(lambda: ops.conv_fwd(xs, w, ops.ConvSpec(3, 3, 2, 0)))
            t_t = timed(lambda: ops.conv_bwd_data(x, w, ops.ConvSpec(3, 3, 2, 0), (2 * H + 1, 2 * H + 1)))
            print(f"B {B:2d} {H:2d}x{H:<2d} M {B * H * H:6d}: fwd {t_f * 1e3:6.1f} us {fl / t_f / 1e9:6.1f} TF | modulated {t_m * 1e3:6.1f} us | dgrad {t_d * 1e3:6.1f} us"
                  f" | stride-2 fwd {t_s2 * 1e3:6.1f} us {fl / t_s2 / 1e9:6.1f} TF | transposed {t_t * 1e3:6.1f} us {fl / t_t / 1e9:6.1f} TF")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--all":
        for env in ({"GIF_SPLITK": "0"}, {}, {"GIF_SPLITK_MAX_TILES": "768"}):
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, **env))
    else:
        main()
