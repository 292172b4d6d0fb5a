"""Timing probe for the GEMM of an UNFUSED Winograd F(4x4,3x3) (round 6; review item 1 of round 5: "measure both cost models first with a
timing-only probe").  Needs the probe build of the library (tools/probes/wino_f4_probe.sh build: conv_winograd.hip with -DGIF_WINO_F4_PROBE,
in which wino_gemm_h2 runs 36 position GEMMs and stores every position's accumulator as its own M plane — no fold, no epilogue; RESULTS ARE
MEANINGLESS).  The kernel is driven through gif_conv3x3_winograd_f32h2 on a geometry of half the height and width (a quarter of the F(2x2)
tiles = the F(4x4) tile count of the real layer), with V / U2 buffers extended to 36 planes by hand.  Reported: ms per GEMM launch (the entry's
own F(2x2) input transform of the small geometry is subtracted by timing it alone), and the HBM-bound passes an F(4x4) route adds or changes,
priced at the measured rate of today's transform kernels."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from gif_amd import _lib, ops  # noqa: E402

lib = _lib.load()
ops.set_fp32_mfma_mode("f16x2")
ops.WINOGRAD_MIN_C = ops.WINOGRAD_WGRAD_MIN_C = 0
F4 = "--f4" in sys.argv
NPOS = 36 if F4 else 16
st = torch.cuda.current_stream().cuda_stream


def ev_time(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B, C, H in ((32, 256, 128), (64, 256, 128), (32, 512, 64), (32, 512, 32)):
    Hg = H // 2 if F4 else H  # geometry handed to the entry point
    x = torch.randn(B, C, Hg, Hg, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, C, 3, 3, device="cuda") / (C * 9) ** 0.5
    RP, CP = ctypes.c_int(), ctypes.c_int()
    _lib.check(lib.gif_winograd_pack_dims_x3(C, C, ctypes.byref(RP), ctypes.byref(CP)), "dims")
    nb = lib.gif_winograd_weight_f32h2_bytes(RP.value, CP.value)
    U2 = torch.empty((nb,), device="cuda", dtype=torch.uint8)
    so, si, sky, skx = w.stride()
    _lib.check(lib.gif_winograd_weight_f32h2(w.data_ptr(), U2.data_ptr(), None, C, C, RP.value, CP.value, so, si, sky, skx, 0, 1.0, st), "wt")
    hdr = 2 * RP.value * 4
    planes = U2[hdr:]
    U2x = torch.cat((U2[:hdr], planes, planes, planes[: (NPOS - 32) * planes.numel() // 16] if NPOS > 32 else planes[:0]))
    nv = lib.gif_winograd_workspace_floats(B, Hg, Hg, C)
    V = torch.randn(nv * NPOS // 16 + 1024, device="cuda")
    y = torch.empty(B, C, Hg, Hg, device="cuda").contiguous(memory_format=torch.channels_last)
    ntiles_pad = nv // (16 * CP.value)
    M = torch.empty(NPOS * ntiles_pad * RP.value + 1024, device="cuda")
    e = _lib.ConvEpilogue(None, None, None, M.data_ptr() if F4 else None, 0, 0.2, 1.0)
    run = lambda: _lib.check(lib.gif_conv3x3_winograd_f32h2(x.data_ptr(), U2x.data_ptr(), None, y.data_ptr(), V.data_ptr(), B, Hg, Hg, C, C,  # noqa: E731
                                                           ctypes.byref(e), st), "wino")
    t_all = ev_time(run)
    # the entry's own input transform, alone (same kernel, same geometry): via the weight-gradient entry's transform? simplest: time conv3x3 on a
    # 1-output-channel-tile problem is not the same -> measure the transform by HBM bytes at the rate of r6_bench_default.json's transforms family
    bytes_t = 4.0 * (B * Hg * Hg * C + 16.0 * ntiles_pad * CP.value)
    t_tr = bytes_t / 5.17e12 * 1e3
    unit_gb = 4.0 * B * H * H * C / 1e9
    print(f"{'F(4x4) probe' if F4 else 'F(2x2) today'} B={B} C={C} H={H}: entry {t_all:.3f} ms, its F(2x2) input transform ~{t_tr:.3f} ms (at 5.17 TB/s) -> GEMM ~{t_all - t_tr:.3f} ms"
          f"   [1 activation unit = {unit_gb:.2f} GB = {unit_gb / 5.17:.3f} ms at 5.17 TB/s]", flush=True)
