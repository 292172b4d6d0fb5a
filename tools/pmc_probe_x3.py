"""Launches the bf16x3 (or, GIF_PROBE_MODE=f16x2, the f16x2) kernels once (after a warm-up) on the benchmark's big layer shapes so that `rocprofv3 --pmc ...` can
attribute counters to them.  Usage (on the GPU box):
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES ... --output-format csv -d out -- python tools/pmc_probe_x3.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gif_amd import ops  # noqa: E402

B = int(os.environ.get("PROBE_BATCH", "32"))
ops.set_fp32_mfma_mode(os.environ.get("GIF_PROBE_MODE", "bf16x3"))
ops.WINOGRAD_MIN_C = ops.WINOGRAD_WGRAD_MIN_C = 0  # (round 6: the probe pins the route per launch itself)
spec = ops.ConvSpec(3, 3, 1, 1)
for C, H in ((128, 256), (512, 64)):
    x = torch.randn(B, C, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, C, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, C, 3, 3, device="cuda")
    for _ in range(2):
        for wino in (False, True):
            ops.WINOGRAD = wino
            ops.conv_fwd(x, w, spec)
            ops.conv_wgrad(gy, x, spec, C, C)
# thin big side (24-channel condition-noise map): the 5-taps-per-tile weight gradient and the C -> 24 data gradient
ops.WINOGRAD = False
x = torch.randn(B, 24, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last)
gy = torch.randn(B, 128, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last)
w = torch.randn(128, 24, 3, 3, device="cuda")
for _ in range(2):
    ops.conv_wgrad(gy, x, spec, 128, 24)
    ops.conv_bwd_data(gy, w, spec, (256, 256))
torch.cuda.synchronize()
