"""ms per launch of the bf16x3 direct kernel on the benchmark's big layer shapes (forward, data gradient, stride 2, transposed, 1x1)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gif_amd import ops  # noqa: E402

ops.set_fp32_mfma_mode("bf16x3")
ops.WINOGRAD = False
B = 32
for name, cin, cout, H, k, stride in (("128->128 3x3 @256", 128, 128, 256, 3, 1), ("256->256 3x3 @128", 256, 256, 128, 3, 1),
                                      ("512->512 3x3 @64", 512, 512, 64, 3, 1), ("512->512 3x3 @16", 512, 512, 16, 3, 1),
                                      ("128->256 3x3 s2 @257", 128, 256, 257, 3, 2), ("256->128 1x1 @256", 256, 128, 256, 1, 1),
                                      ("24->128 3x3 @256 (tap-dense)", 24, 128, 256, 3, 1)):
    x = torch.randn(B, cin, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, k, k, device="cuda")
    spec = ops.ConvSpec(k, k, stride, 1 if (k == 3 and stride == 1) else 0)
    for _ in range(2):
        y = ops.conv_fwd(x, w, spec)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(5):
        ops.conv_fwd(x, w, spec)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 5
    fl = 2.0 * y.shape[0] * y.shape[2] * y.shape[3] * cout * cin * k * k
    print(f"{name:32s} {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TFLOP/s")
