"""conv3x3_rows_thin_h2 (round 6) vs the gather kernel on the C -> 24 data gradients of the headline step, batch 32.
GIF_H2_ROWS_THIN is read once per process: one child per arm (GIF_EXPERIMENTAL=1 set here).  python tools/probes/rows_thin_probe.py"""
import os
import subprocess
import sys

SHAPES = [(32, 128, 24, 256), (32, 256, 24, 128), (32, 512, 24, 64), (32, 512, 24, 32), (64, 128, 24, 256)]


def child():
    import torch
    sys.path.insert(0, ".")
    from gif_amd import ops
    ops.set_fp32_mfma_mode("f16x2")
    spec = ops.ConvSpec(3, 3, 1, 1)
    for B, C, n, H in SHAPES:
        torch.manual_seed(0)
        w = torch.randn(C, n, 3, 3, device="cuda") / (n * 9) ** 0.5
        gy = torch.randn(B, C, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
        for _ in range(3):
            y = ops.conv_bwd_data(gy, w, spec, (H, H))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            y = ops.conv_bwd_data(gy, w, spec, (H, H))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{str((B, C, n, H)):24s} {ms:7.3f} ms  {2.0 * B * H * H * 9 * C * n / ms / 1e9:6.1f} TF   checksum {float(y.double().abs().sum()):.9e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        for arm, env in (("row kernel (default)", {}), ("gather kernel (GIF_H2_ROWS_THIN=0)", {"GIF_H2_ROWS_THIN": "0"}), ("row kernel (default), again", {})):
            print("== " + arm, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, GIF_EXPERIMENTAL="1", **env))
