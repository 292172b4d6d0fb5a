"""Times wino_gemm_x3 alone (the input transform is timed separately and subtracted) on the benchmark's Winograd layers.
GIF_WINO_DBG selects an ablation build of the kernel (wrong results, see conv_winograd.hip)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gif_amd import ops  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402

ops.set_fp32_mfma_mode("bf16x3")
spec = ops.ConvSpec(3, 3, 1, 1)
out = []
for C, H in ((128, 256), (256, 128), (512, 64), (512, 32)):
    x = torch.randn(32, C, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, C, 3, 3, device="cuda")
    fl = 2.0 * 32 * H * H * C * C * 9
    t = timeit(lambda: ops.conv_fwd(x, w, spec), iters=10)
    out.append(f"{C}@{H}: {t:.3f} ms {fl / t / 1e9:.0f} TF")
print(os.environ.get("GIF_WINO_DBG", "0"), " | ".join(out))
