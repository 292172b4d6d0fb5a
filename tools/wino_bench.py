"""Direct implicit-GEMM vs Winograd F(2x2,3x3) on the model's stride-1 3x3 layer shapes (algorithmic TFLOP/s).
Usage: python tools/wino_bench.py [--batch 32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gif_amd import ops  # noqa: E402

ops.WINOGRAD_MIN_C = ops.WINOGRAD_WGRAD_MIN_C = 0  # (round 6: Winograd from 256 channels in f16x2 mode by default; this tool pins the route itself)
from tools.kernel_bench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    B = a.batch
    spec = ops.ConvSpec(3, 3, 1, 1)
    for ci, co, h in [(128, 128, 256), (256, 256, 128), (512, 512, 64), (512, 512, 32), (512, 512, 16), (516, 512, 4)]:
        x = torch.randn(B, ci, h, h, device="cuda").contiguous(memory_format=torch.channels_last)
        w = torch.randn(co, ci, 3, 3, device="cuda") / 50
        s = torch.rand(B, ci, device="cuda") + 0.5
        d = torch.rand(B, co, device="cuda") + 0.5
        fl = 2.0 * B * h * h * co * ci * 9
        ops.WINOGRAD = False
        t_dir = timeit(lambda: ops.conv_fwd(x, w, spec, in_scale=s, out_scale=d, act=True))
        y0 = ops.conv_fwd(x, w, spec, in_scale=s, out_scale=d, act=True)
        ops.WINOGRAD, ops.WINOGRAD_MIN_TILES = True, 0
        t_win = timeit(lambda: ops.conv_fwd(x, w, spec, in_scale=s, out_scale=d, act=True))
        y1 = ops.conv_fwd(x, w, spec, in_scale=s, out_scale=d, act=True)
        err = ((y1 - y0).abs().max() / y0.abs().max()).item()
        print(f"{ci:4d}->{co:4d} @{h:3d}^2 x{B}: direct {t_dir:7.3f} ms {fl / t_dir / 1e9:6.1f} TF | winograd {t_win:7.3f} ms "
              f"{fl / t_win / 1e9:6.1f} TF-equiv | x{t_dir / t_win:4.2f} | rel diff {err:.1e}", flush=True)
        if h < 16:
            continue
        gy = torch.randn(B, co, h, h, device="cuda").contiguous(memory_format=torch.channels_last)
        for mod in (False, True):
            kw = dict(small_scale=d, big_scale=s) if mod else {}
            ops.WINOGRAD_WGRAD = False
            t_dir = timeit(lambda: ops.conv_wgrad(gy, x, spec, co, ci, **kw))
            w0 = ops.conv_wgrad(gy, x, spec, co, ci, **kw)
            ops.WINOGRAD_WGRAD = True
            t_win = timeit(lambda: ops.conv_wgrad(gy, x, spec, co, ci, **kw))
            w1 = ops.conv_wgrad(gy, x, spec, co, ci, **kw)
            err = ((w1 - w0).abs().max() / w0.abs().max()).item()
            print(f"      wgrad{' (modulated)' if mod else '':12s}: direct {t_dir:7.3f} ms {fl / t_dir / 1e9:6.1f} TF | winograd {t_win:7.3f} ms "
                  f"{fl / t_win / 1e9:6.1f} TF-equiv | x{t_dir / t_win:4.2f} | rel diff {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
