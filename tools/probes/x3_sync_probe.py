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
out = (ctypes.c_ulonglong * 12)()
print(MODE + " direct kernel, per wave and K loop: share of the loop's cycles spent waiting for the wave's own DMA (s_waitcnt vmcnt(0)) and at the barrier")
for name, cin, cout, H, k, stride, epi in (("128->128 3x3 @256", 128, 128, 256, 3, 1, ""), ("128->128 3x3 @256, modulated + bias + lrelu (G)", 128, 128, 256, 3, 1, "mod"),
                                           ("128->128 3x3 @256, bias + lrelu (D)", 128, 128, 256, 3, 1, "act"),
                                           ("256->256 3x3 @128", 256, 256, 128, 3, 1, ""),
                                           ("512->512 3x3 @64", 512, 512, 64, 3, 1, ""), ("512->512 3x3 @16", 512, 512, 16, 3, 1, ""),
                                           ("128->256 3x3 s2 @256", 128, 256, 257, 3, 2, ""), ("256->128 1x1 @256", 256, 128, 256, 1, 1, ""),
                                           ("128->256 1x1 @128", 128, 256, 128, 1, 1, "")):
    x = torch.randn(B, cin, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    spec = ops.ConvSpec(k, k, stride, 1 if (k == 3 and stride == 1) else 0)
    kw = {}
    if epi == "mod":
        kw = dict(in_scale=torch.rand(B, cin, device="cuda") + 0.5, out_scale=torch.rand(B, cout, device="cuda") + 0.5, bias=torch.randn(cout, device="cuda"), act=True)
    elif epi == "act":
        kw = dict(bias=torch.randn(cout, device="cuda"), act=True)
    _conv = ops.conv_fwd
    ops.conv_fwd = lambda x_, w_, s_: _conv(x_, w_, s_, **kw)
    ops.conv_fwd(x, w, spec)
    assert read(out, 1) == 0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(3):
        ops.conv_fwd(x, w, spec)
    ev1.record()
    assert read(out, 1) == 0
    wait, sync, loop, waves, pro, epi, tail, stg, idx, fill = [int(v) for v in out][:10]
    ms = ev0.elapsed_time(ev1) / 3
    w = max(waves, 1)
    ops.conv_fwd = _conv
    print(f"{name:48s} {ms:7.3f} ms/launch  waves {waves // 3:7d}  loop {loop / w:8.0f} cycles/wave  own DMA wait {100.0 * wait / max(loop, 1):5.1f} %  barrier {100.0 * sync / max(loop, 1):5.1f} %"
          f"  | before the loop {pro / w:7.0f}  after it (epilogue) {epi / w:7.0f} cycles/wave = {100.0 * pro / max(pro + loop + epi, 1):4.1f} % / {100.0 * epi / max(pro + loop + epi, 1):4.1f} % of the wave's life; of the epilogue: scale-back + guard {tail / w:6.0f}, accumulators -> LDS incl. both barriers {stg / w:6.0f}, row loop {(epi - tail - stg) / w:6.0f}; before the loop: index tables {idx / w:6.0f}, first DMA issue -> ring filled {fill / w:6.0f}")
