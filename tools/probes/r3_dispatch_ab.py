"""Round-3 dispatch experiments on one MI355X (the env knobs are read once per process => one child per configuration).
First series (profiles/r3_dispatch_ab.txt; knobs removed again): short-K transposed phases on 128x128 tiles (no gain, modulated
layers slower), 128x64 tiles of 4x1 waves for the 4^2..16^2 bf16x3 layers (+15 %: adopted), 32-row blur windows (slower).
Second series:
  GIF_X3_MULTI_BIG=0 : the four phases of a big bf16x3 transposed conv as separate (bulk + remainder) launches instead of one grid
Usage (GPU): python tools/probes/r3_dispatch_ab.py            (parent: runs every configuration)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CONFIGS = [("default (merged big transposed phases)", {}), ("per-phase launches", {"GIF_X3_MULTI_BIG": "0"})]


def child():
    import torch
    from gif_amd import ops
    from tools.kernel_bench import timeit
    B, dev = 32, "cuda"
    ops.WINOGRAD = False
    rows = []
    # (name, Cin(big side), Cout(small side), K, stride, pad, Hbig)
    shapes = [("s2 128->256 @257", 128, 256, 3, 2, 0, 257), ("s2 256->512 @129", 256, 512, 3, 2, 0, 129),
              ("s2 512->512 @65", 512, 512, 3, 2, 0, 65), ("s2 512->512 @33", 512, 512, 3, 2, 0, 33),
              ("s1 512->512 @16", 512, 512, 3, 1, 1, 16), ("s1 512->512 @8", 512, 512, 3, 1, 1, 8),
              ("s1 512->512 @4", 512, 512, 3, 1, 1, 4), ("1x1 128->256 @128", 128, 256, 1, 1, 0, 128)]
    for name, ci, co, k, s, p, h in shapes:
        spec = ops.ConvSpec(k, k, s, p)
        x = torch.randn(B, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(co, ci, k, k, device=dev)
        sc = torch.rand(B, ops.pad4(co), device=dev) + 0.5
        hs, ws_ = spec.small_hw(h, h)
        gy = torch.randn(B, ops.pad4(co), hs, ws_, device=dev).contiguous(memory_format=torch.channels_last)
        fl = 2.0 * B * hs * ws_ * co * ci * k * k
        t_f = timeit(lambda: ops.conv_fwd(x, w, spec), iters=10)
        t_d = timeit(lambda: ops.conv_bwd_data(gy, w, spec, (h, h)), iters=10)
        t_dm = timeit(lambda: ops.conv_bwd_data(gy, w, spec, (h, h), in_scale=sc), iters=10)
        rows.append(f"{name:20s} fwd {t_f:7.3f} ms {fl / t_f / 1e9:6.1f} TF | dgrad/transposed {t_d:7.3f} ms {fl / t_d / 1e9:6.1f} TF | "
                    f"modulated {t_dm:7.3f} ms {fl / t_dm / 1e9:6.1f} TF")
    k4 = torch.tensor([1., 3., 3., 1.], device=dev)
    k4 = (k4[:, None] * k4[None, :] / 16).contiguous()
    for c, h in ((128, 257), (256, 129), (512, 65), (128, 256)):
        x = torch.randn(B, c, h, h, device=dev).contiguous(memory_format=torch.channels_last)
        t = timeit(lambda: ops.upfirdn2d(x, k4, 1, 1, 1, (h - 1, h - 1)), iters=10)
        rows.append(f"blur pad(1,1) C={c} {h}->{h - 1}: {t:7.3f} ms {(x.numel() + B * c * (h - 1) ** 2) * 4 / t / 1e6:8.1f} GB/s")
    print("\n".join(rows), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        return child()
    for name, env in CONFIGS:
        e = dict(os.environ)
        e.update(env)
        print(f"=== {name} {env}", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e, check=False)


if __name__ == "__main__":
    main()
