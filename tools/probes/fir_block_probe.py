"""4x4 FIR resampling by 2 (up = 2 / down = 2): the block kernels (round 6) against the one-pixel-per-lane kernels (GIF_FIR_BLOCK=0) on the
shapes of the headline step — ms per launch, in + out bytes per second, and a SHA-1 of every output (the two forms must give the same bits).
GIF_FIR_BLOCK is read once per process: one child per arm (GIF_EXPERIMENTAL=1 set here).  python tools/probes/fir_block_probe.py"""
import hashlib
import os
import subprocess
import sys

# B, C, H (input), up, down, pad0, out
SHAPES = [(64, 128, 128, 2, 1, 2, 256), (64, 128, 128, 2, 1, 1, 255), (64, 256, 64, 2, 1, 2, 128), (32, 4, 128, 2, 1, 2, 256), (64, 128, 256, 1, 2, 1, 128),
          (64, 128, 257, 1, 2, 0, 127), (64, 256, 128, 1, 2, 1, 64), (64, 512, 32, 1, 2, 1, 16), (3, 8, 37, 2, 1, 2, 74), (3, 8, 37, 1, 2, 2, 19), (3, 8, 37, 2, 1, 1, 73)]


def child():
    import torch
    sys.path.insert(0, ".")
    from gif_amd import ops
    k = torch.tensor([1.0, 3.0, 3.0, 1.0], device="cuda")
    k = (k[:, None] * k[None, :] / 64.0).contiguous()
    for dt in (torch.float32, torch.float16):
        for B, C, H, up, down, pad0, out in SHAPES:
            torch.manual_seed(B + C + H)
            x = torch.randn(B, C, H, H, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
            kk = k * (up * up)
            for flip in (True, False):
                y = ops.upfirdn2d(x, kk, up, down, pad0, (out, out), flip)
            for _ in range(2):
                ops.upfirdn2d(x, kk, up, down, pad0, (out, out), True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                y = ops.upfirdn2d(x, kk, up, down, pad0, (out, out), True)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            gb = (x.numel() + y.numel()) * x.element_size() / 1e9
            sha = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12]
            print(f"{str(dt)[6:]:8s} {str((B, C, H, up, down, pad0, out)):34s} {ms:7.3f} ms  {gb / ms:5.2f} TB/s of in + out   sha1 {sha}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for arm, env in (("block kernels (default)", {}), ("one pixel per lane (GIF_FIR_BLOCK=0)", {"GIF_FIR_BLOCK": "0"})):
            print("== " + arm, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, GIF_EXPERIMENTAL="1", **env))
