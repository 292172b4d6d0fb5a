"""f16x2 weight gradient (conv_wgrad_h2v2): buffer-addressed DMA (round 6) vs the 64-bit form, batch-32 headline shapes.
GIF_H2_WGRAD_BUF is read once per process: the parent runs one child per arm (needs GIF_EXPERIMENTAL=1, set here).
Run on the GPU box: python tools/probes/wgrad_buf_probe.py"""
import os
import subprocess
import sys

SHAPES = [  # B, ci (big side), co (small side), k, stride, pad, H (big), winograd planes
    (32, 128, 128, 3, 1, 1, 256, False),
    (64, 128, 128, 3, 1, 1, 256, False),
    (32, 128, 256, 3, 2, 0, 257, False),
    (32, 256, 256, 3, 1, 1, 128, False),
    (32, 256, 512, 3, 2, 0, 129, False),
    (32, 512, 512, 3, 1, 1, 64, False),
    (32, 24, 128, 3, 1, 1, 256, False),
    (32, 24, 256, 3, 1, 1, 128, False),
    (32, 256, 256, 3, 1, 1, 128, True),
    (32, 512, 512, 3, 1, 1, 64, True),
    (32, 512, 512, 3, 1, 1, 32, True),
]


def child():
    import torch
    sys.path.insert(0, ".")
    from gif_amd import ops
    ops.set_fp32_mfma_mode("f16x2")
    for B, ci, co, k, s, p, h, wino in SHAPES:
        ops.WINOGRAD, ops.WINOGRAD_MIN_C, ops.WINOGRAD_WGRAD_MIN_C = wino, 0, 0
        spec = ops.ConvSpec(k, k, s, p)
        x = torch.randn(B, ci, h, h, device="cuda").contiguous(memory_format=torch.channels_last)
        hs, ws_ = spec.small_hw(h, h)
        gy = torch.randn(B, co, hs, ws_, device="cuda").contiguous(memory_format=torch.channels_last)
        sc, sd = torch.rand(B, ci, device="cuda") + 0.5, torch.rand(B, co, device="cuda") + 0.5
        out = []
        for scaled in (False, True):
            kw = dict(small_scale=sd, big_scale=sc) if scaled else {}
            for _ in range(3):
                r = ops.conv_wgrad(gy, x, spec, co, ci, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            e0.record()
            for _ in range(n):
                r = ops.conv_wgrad(gy, x, spec, co, ci, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            out.append(f"{ms:7.3f} ms {2.0 * B * hs * ws_ * k * k * ci * co / ms / 1e9:6.1f} TF")
        print(f"{str((B, ci, co, k, s, h)) + (' wino planes' if wino else ''):44s} plain {out[0]}   modulated {out[1]}   checksum {float(r.double().abs().sum()):.6e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        arms = (("v3: operands prefetched into registers (GIF_H2_WGRAD_V3=1)", {"GIF_H2_WGRAD_V3": "1"}), ("default: v2, buffer-addressed DMA", {}),
                ("v2, 64-bit addresses (GIF_H2_WGRAD_BUF=0)", {"GIF_H2_WGRAD_BUF": "0"}), ("v3, again", {"GIF_H2_WGRAD_V3": "1"}))
        if "--tab" in sys.argv:
            arms = (("default (un-modulated launches on the scale-table kernel with unit scales)", {}), ("plain instantiation (GIF_H2_WGRAD_PLAIN_TAB=0)", {"GIF_H2_WGRAD_PLAIN_TAB": "0"}), ("default, again", {}))
        if "--v3" in sys.argv:
            arms = arms[:2] + arms[3:]
        for arm, env in arms:
            print("== " + arm, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, GIF_EXPERIMENTAL="1", **env))
