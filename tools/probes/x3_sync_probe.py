"""Where the waves of the bf16x3 direct kernel wait: cycles at the mid-stage sync (own DMA wait / barrier) against cycles in the K loop.
Needs the probe build of the library (tools/probes/x3_sync_probe.sh build): conv_igemm.hip with -DGIF_X3_TIMING_PROBE."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gif_amd import _lib, ops  # noqa: E402

lib = _lib.load()
read = lib.gif_debug_x3_probe_read
read.restype, read.argtypes = ctypes.c_int, [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
MODE = os.environ.get("GIF_PROBE_MODE", "bf16x3")
ops.set_fp32_mfma_mode(MODE)
ops.WINOGRAD = False
B = 32
out = (ctypes.c_ulonglong * 4)()
print(MODE + " direct kernel, per wave and K loop: share of the loop's cycles spent waiting for the wave's own DMA (s_waitcnt vmcnt(0)) and at the barrier")
for name, cin, cout, H, k, stride in (("128->128 3x3 @256", 128, 128, 256, 3, 1), ("256->256 3x3 @128", 256, 256, 128, 3, 1),
                                      ("512->512 3x3 @64", 512, 512, 64, 3, 1), ("512->512 3x3 @16", 512, 512, 16, 3, 1),
                                      ("128->256 3x3 s2 @256", 128, 256, 257, 3, 2), ("256->128 1x1 @256", 256, 128, 256, 1, 1)):
    x = torch.randn(B, cin, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, k, k, device="cuda")
    spec = ops.ConvSpec(k, k, stride, 1 if (k == 3 and stride == 1) else 0)
    ops.conv_fwd(x, w, spec)
    assert read(out, 1) == 0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(3):
        ops.conv_fwd(x, w, spec)
    ev1.record()
    assert read(out, 1) == 0
    wait, sync, loop, waves = [int(v) for v in out]
    ms = ev0.elapsed_time(ev1) / 3
    print(f"{name:24s} {ms:7.3f} ms/launch  waves {waves // 3:7d}  loop {loop / max(waves, 1):10.0f} cycles/wave  own DMA wait {100.0 * wait / max(loop, 1):5.1f} %  barrier {100.0 * sync / max(loop, 1):5.1f} %")
