"""Launches the Winograd kernels on the three big layer shapes so that `rocprofv3 --pmc ...` can attribute counters."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gif_amd import ops  # noqa: E402

ops.WINOGRAD_MIN_C = ops.WINOGRAD_WGRAD_MIN_C = 0  # (round 6: Winograd from 256 channels in f16x2 mode by default; this tool pins the route itself)

B = 32
ops.WINOGRAD_MIN_TILES = 0
spec = ops.ConvSpec(3, 3, 1, 1)
for c, h in ((128, 256), (512, 64)):
    x = torch.randn(B, c, h, h, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(c, c, 3, 3, device="cuda")
    for _ in range(2):
        ops.conv_fwd(x, w, spec)
torch.cuda.synchronize()
