"""ms per launch of the stride-1 3x3 direct f16x2 launches of the headline step (see kxshare_probe.sh)."""
import sys

import torch

sys.path.insert(0, ".")
from gif_amd import ops  # noqa: E402

ops.set_fp32_mfma_mode("f16x2")
ops.WINOGRAD = False
SHAPES = [(32, 24, 128, 256, "fwd"), (32, 24, 256, 128, "fwd"), (32, 24, 512, 64, "fwd"), (32, 128, 128, 256, "fwd"), (32, 128, 128, 256, "fwd mod"), (32, 128, 128, 256, "dgrad"), (64, 128, 128, 256, "fwd"),
          (32, 24, 128, 256, "dgrad"), (32, 24, 256, 128, "dgrad"), (32, 24, 512, 64, "dgrad"), (32, 256, 256, 128, "fwd"), (32, 512, 512, 16, "fwd")]
spec = ops.ConvSpec(3, 3, 1, 1)
for B, ci, co, h, what in SHAPES:
    w = torch.randn(co, ci, 3, 3, device="cuda") / (ci * 9) ** 0.5
    x = torch.randn(B, ops.pad4(ci), h, h, device="cuda").contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, co, h, h, device="cuda").contiguous(memory_format=torch.channels_last)
    sc = torch.rand(B, ops.pad4(ci), device="cuda") + 0.5
    if what == "fwd":
        run = lambda: ops.conv_fwd(x, w, spec)  # noqa: E731
    elif what == "fwd mod":
        run = lambda: ops.conv_fwd(x, w, spec, in_scale=sc)  # noqa: E731
    else:
        run = lambda: ops.conv_bwd_data(gy, w, spec, (h, h))  # noqa: E731
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{str((B, ci, co, h, what)):36s} {ms:7.3f} ms  {2.0 * B * h * h * 9 * ci * co / ms / 1e9:6.1f} TF", flush=True)
