"""4x4 blur (up = down = 1) bytes per second by channel count and dtype: the lane mapping (channels fastest, then 4-pixel strips) leaves thin
tensors with 64-byte runs.  python tools/probes/blur_bw_probe.py"""
import sys

import torch

sys.path.insert(0, ".")
from gif_amd import ops  # noqa: E402

k = torch.tensor([1.0, 3.0, 3.0, 1.0], device="cuda")
k = (k[:, None] * k[None, :] / 64.0).contiguous()
for dt in (torch.float16, torch.float32):
    for B, C, H in [(8, 32, 1024), (8, 64, 512), (8, 128, 256), (8, 16, 1024), (32, 128, 256), (32, 32, 256), (8, 64, 1024)]:
        x = torch.randn(B, C, H, H, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
        for _ in range(3):
            y = ops.upfirdn2d(x, k, 1, 1, 2, (H + 1, H + 1), True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y = ops.upfirdn2d(x, k, 1, 1, 2, (H + 1, H + 1), True)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        gb = (x.numel() + y.numel()) * x.element_size() / 1e9
        print(f"{str(dt)[6:]:8s} {str((B, C, H)):18s} {ms:7.3f} ms  {gb / ms:5.2f} TB/s of in + out", flush=True)
