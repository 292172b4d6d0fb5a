"""bf16x3 weight gradient of the thin (24-channel side) layers, tile variants via GIF_X3_WGRAD_THIN (GPU probe; one child per knob)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child():
    import torch
    from gif_amd import ops
    from tools.kernel_bench import timeit
    spec = ops.ConvSpec(3, 3, 1, 1)
    for B, Cs, Cb, H in ((32, 128, 24, 256), (32, 256, 24, 128), (32, 512, 24, 64), (32, 24, 12, 256)):
        x = torch.randn(B, Cb, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
        gy = torch.randn(B, Cs, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
        ref = None
        t = timeit(lambda: ops.conv_wgrad(gy, x, spec, Cs, Cb), iters=10)
        out = ops.conv_wgrad(gy, x, spec, Cs, Cb)
        print(f"wgrad {Cs}x{Cb} @{H}: {t:7.3f} ms {2.0 * B * H * H * Cs * Cb * 9 / t / 1e9:6.1f} TF  checksum {out.double().abs().sum().item():.6e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        for knob in ("1", "2", "0"):
            print("=== GIF_X3_WGRAD_THIN=" + knob, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, GIF_X3_WGRAD_THIN=knob))
