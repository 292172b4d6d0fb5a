"""Is the bf16x3 conv kernel limited by its instruction stream or by the chip's power budget?  The same launch on random data
and on all-zero data (identical instruction stream and memory traffic; no toggling in the MFMA datapath)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gif_amd import ops  # noqa: E402

ops.WINOGRAD_MIN_C = ops.WINOGRAD_WGRAD_MIN_C = 0  # (round 6: Winograd from 256 channels in f16x2 mode by default; this tool pins the route itself)
from tools.kernel_bench import timeit  # noqa: E402

ops.set_fp32_mfma_mode("bf16x3")
spec = ops.ConvSpec(3, 3, 1, 1)
for C, H in ((128, 256), (512, 64)):
    fl = 2.0 * 32 * H * H * C * C * 9
    for kind in ("random", "zeros"):
        mk = torch.randn if kind == "random" else torch.zeros
        x = mk(32, C, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
        w = mk(C, C, 3, 3, device="cuda")
        ops.WINOGRAD = False
        t_d = timeit(lambda: ops.conv_fwd(x, w, spec), iters=10)
        t_w = timeit(lambda: ops.conv_wgrad(x, x, spec, C, C), iters=10)
        ops.WINOGRAD = True
        t_g = timeit(lambda: ops.conv_fwd(x, w, spec), iters=10)
        print(f"{C}@{H} {kind:6s}: direct conv {fl / t_d / 1e9:6.1f} TF  direct wgrad {fl / t_w / 1e9:6.1f} TF  winograd conv (incl. transform) {fl / t_g / 1e9:6.1f} TF")
