"""f16 weight-gradient kernel on the benchmark's shapes (GPU probe): python tools/probes/f16_wgrad_bench.py [tree-root]"""
import os
import sys

import torch

ROOT = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gif_amd import ops  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402

print("# tree", ROOT)
#          B, Cs(gy), Cb(x), K, stride, H
SHAPES = [(32, 512, 512, 3, 1, 64), (32, 256, 256, 3, 1, 128), (32, 128, 128, 3, 1, 256), (32, 128, 24, 3, 1, 256),
          (32, 256, 128, 3, 2, 257), (8, 32, 32, 3, 1, 1024), (8, 64, 64, 3, 1, 512), (8, 32, 24, 3, 1, 1024), (32, 128, 128, 1, 1, 256)]
for B, Cs, Cb, K, st, H in SHAPES:
    spec = ops.ConvSpec(K, K, st, 1 if (K == 3 and st == 1) else 0)
    x = torch.randn(B, Cb, H, H, device="cuda").half().contiguous(memory_format=torch.channels_last)
    hs, ws = spec.small_hw(H, H)
    gy = torch.randn(B, Cs, hs, ws, device="cuda").half().contiguous(memory_format=torch.channels_last)
    t = timeit(lambda: ops.conv_wgrad(gy, x, spec, Cs, Cb), iters=10)
    fl = 2.0 * B * hs * ws * Cs * Cb * K * K
    print(f"wgrad f16 B{B} {Cs}x{Cb} k{K} s{st} @{H}: {t:8.3f} ms {fl / t / 1e9:7.1f} TF", flush=True)
