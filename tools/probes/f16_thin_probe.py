"""The thin high-resolution f16 layers of BASELINE configs[4] (1024^2 / 512^2, batch 8) in isolation: ms per launch and the in + out bytes per second
(they are HBM-shaped: 18 KFLOP per pixel at 32 -> 32 channels).  python tools/probes/f16_thin_probe.py"""
import sys

import torch

sys.path.insert(0, ".")
from gif_amd import ops  # noqa: E402

B = 8
spec = ops.ConvSpec(3, 3, 1, 1)
for ci, co, H, what in [(32, 32, 1024, "fwd"), (32, 32, 1024, "fwd mod"), (32, 32, 1024, "dgrad"), (64, 32, 1024, "fwd"), (64, 64, 512, "fwd"), (64, 64, 512, "fwd mod"),
                        (64, 64, 512, "dgrad"), (24, 32, 1024, "fwd"), (32, 24, 1024, "dgrad"), (128, 128, 256, "fwd")]:
    x = torch.randn(B, ci, H, H, device="cuda", dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, co, H, H, device="cuda", dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 3, 3, device="cuda") / (ci * 9) ** 0.5)
    si, so = torch.rand(B, ci, device="cuda") + 0.5, torch.rand(B, co, device="cuda") + 0.5
    bias = torch.randn(co, device="cuda")
    if what == "fwd":
        run = lambda: ops.conv_fwd(x, w, spec, bias=bias, act=True)  # noqa: E731
    elif what == "fwd mod":
        run = lambda: ops.conv_fwd(x, w, spec, in_scale=si, out_scale=so, bias=bias, act=True)  # noqa: E731
    else:
        run = lambda: ops.conv_bwd_data(gy, w, spec, (H, H))  # noqa: E731
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gb = 2.0 * B * H * H * (ops.cpad(ci, torch.float16) + ops.cpad(co, torch.float16)) / 1e9
    print(f"{str((ci, co, H, what)):28s} {ms:7.3f} ms  {2.0 * B * H * H * 9 * ci * co / ms / 1e9:6.1f} TF   {gb / ms:5.2f} TB/s of in + out", flush=True)
