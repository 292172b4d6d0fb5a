"""Does an HBM-bound kernel run slower right after a power-hungry MFMA kernel?  Times the 4x4 blur (257^2 <- 256^2, 128 channels,
batch 32) alone and interleaved with a bf16x3 convolution, with events around the blur only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gif_amd import ops  # noqa: E402

x = torch.randn(32, 128, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last)
x2 = torch.randn(32, 128, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last)
w = torch.randn(128, 128, 3, 3, device="cuda")
k = torch.tensor([1., 3., 3., 1.], device="cuda")
k = (k[:, None] * k[None, :] / 64).contiguous()
spec = ops.ConvSpec(3, 3, 1, 1)
ops.WINOGRAD = False


def blur_ms(before, n=20):
    tot = 0.0
    for _ in range(n):
        before()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.upfirdn2d(x, k, 1, 1, 2, (257, 257))
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n


for name, fn in (("alone", lambda: None), ("after a bf16x3 conv on another tensor", lambda: ops.conv_fwd(x2, w, spec)),
                 ("after bias_act on another tensor (HBM-bound)", lambda: ops.bias_act(x2, None, None)),
                 ("after the conv that PRODUCED its input", None)):
    if fn is None:
        def fn():
            global x
            x = ops.conv_fwd(x2, w, spec)
    blur_ms(fn, 3)
    print(f"blur {name}: {blur_ms(fn):.3f} ms")
