"""Launches the dominant conv kernels once each on the benchmark's biggest layer shape (128->128 @256x256, batch 32)
so that `rocprofv3 --pmc ...` can attribute counters to them.  Usage (on the GPU box):
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES ... --output-format csv -d out -- python tools/pmc_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gif_amd import ops  # noqa: E402

B = int(os.environ.get("PROBE_BATCH", "32"))
spec = ops.ConvSpec(3, 3, 1, 1)
x = torch.randn(B, 128, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last)
gy = torch.randn(B, 128, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last)
w = torch.randn(128, 128, 3, 3, device="cuda")
k = torch.tensor([1., 3., 3., 1.], device="cuda")
k = (k[:, None] * k[None, :] / 64).contiguous()
for _ in range(2):
    ops.conv_fwd(x, w, spec)
    ops.conv_bwd_data(gy, w, spec, (256, 256))
    ops.conv_wgrad(gy, x, spec, 128, 128)
    ops.upfirdn2d(x, k, 1, 1, 2, (257, 257))
    ops.bias_act(x, None, None)
torch.cuda.synchronize()
