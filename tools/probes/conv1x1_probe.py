"""1x1 f16x2 convolutions of the discriminator's skip path (short K loops): 256 x 128 8-wave tiles (default) vs 128 x 128 4-wave tiles
(GIF_X3_BIG=0: two workgroups per CU).  One child per arm.  python tools/probes/conv1x1_probe.py"""
import os
import subprocess
import sys

SHAPES = [(64, 128, 256, 128, "fwd"), (32, 128, 256, 128, "fwd"), (64, 256, 512, 64, "fwd"), (64, 512, 512, 32, "fwd"), (64, 128, 256, 128, "dgrad"),
          (64, 256, 512, 64, "dgrad"), (32, 128, 128, 256, "fwd3x3")]


def child():
    import torch
    sys.path.insert(0, ".")
    from gif_amd import ops
    ops.set_fp32_mfma_mode("f16x2")
    ops.WINOGRAD = False
    for B, ci, co, H, what in SHAPES:
        k = 3 if what == "fwd3x3" else 1
        spec = ops.ConvSpec(k, k, 1, k // 2)
        w = torch.randn(co, ci, k, k, device="cuda") / (ci * k * k) ** 0.5
        x = torch.randn(B, ci, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
        gy = torch.randn(B, co, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
        run = (lambda: ops.conv_bwd_data(gy, w, spec, (H, H))) if what == "dgrad" else (lambda: ops.conv_fwd(x, w, spec))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        gb = 4.0 * B * H * H * (ci + co) / 1e9
        print(f"{str((B, ci, co, H, what)):34s} {ms:7.3f} ms  {2.0 * B * H * H * k * k * ci * co / ms / 1e9:6.1f} TF   {gb / ms:5.2f} TB/s of in + out", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for arm, env in (("default (256 x 128 tiles, 8 waves)", {}), ("GIF_X3_BIG=0 (128 x 128 tiles, 4 waves, two workgroups per CU)", {"GIF_X3_BIG": "0"}), ("default, again", {})):
            print("== " + arm, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, GIF_EXPERIMENTAL="1", **env))
