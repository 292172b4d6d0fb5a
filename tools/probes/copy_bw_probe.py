"""What a plain device copy reaches on this box (the ceiling of the step's HBM-bound passes: blur, FIR resampling, Winograd transforms):
torch's copy kernel and hipMemcpyAsync D2D on 256 MB .. 2 GB, read + write bytes per second.  python tools/probes/copy_bw_probe.py"""
import torch

for mb in (256, 1024, 2048):
    n = mb * (1 << 20) // 4
    x = torch.empty(n, device="cuda", dtype=torch.float32).normal_()
    y = torch.empty_like(x)
    for name, fn in (("copy_ kernel", lambda: y.copy_(x)), ("x * 1.0 (elementwise)", lambda: torch.mul(x, 1.0, out=y)),
                     ("read only (sum)", lambda: x.sum())):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        moved = (1 if name.startswith("read") else 2) * n * 4
        print(f"{mb:5d} MB  {name:24s} {ms:7.3f} ms  {moved / ms / 1e9:6.2f} TB/s")
