"""Weight gradient of the 24-channel condition-noise layers: bf16x3 128x32 tile vs the native kernel (GIF_X3_WGRAD_THIN=0)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gif_amd import ops  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402

ops.set_fp32_mfma_mode("bf16x3")
spec = ops.ConvSpec(3, 3, 1, 1)
x = torch.randn(4, 24, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
gy = torch.randn(4, 128, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
wd = torch.zeros(128, 24, 3, 3, device="cuda", dtype=torch.float64, requires_grad=True)
(ref,) = torch.autograd.grad(F.conv2d(x.double(), wd, padding=1), wd, gy.double())
gw = ops.conv_wgrad(gy, x, spec, 128, 24)
print("error vs fp64", float((gw.double() - ref).abs().max() / ref.abs().max()))
for C, H in ((128, 256), (256, 128), (512, 64)):
    x = torch.randn(32, 24, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    gy = torch.randn(32, C, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    t = timeit(lambda: ops.conv_wgrad(gy, x, spec, C, 24), iters=10)
    print(f"{C}x24 @{H}: {t:.3f} ms {2.0 * 32 * H * H * C * 24 * 9 / t / 1e9:.1f} TF")
