"""Per-kernel micro-benchmarks on one MI355X: TFLOP/s of the MFMA conv kernels on the model's layer shapes and
GB/s of the HBM-bound kernels.  Usage: python tools/kernel_bench.py [--batch 32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gif_amd import ops  # noqa: E402


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    B = a.batch
    dev = "cuda"
    print(f"# batch {B}")
    # (name, Cin, Cout, K, stride, pad, Hin)
    shapes = [("g/d 128->128 @256", 128, 128, 3, 1, 1, 256), ("256->256 @128", 256, 256, 3, 1, 1, 128),
              ("512->512 @64", 512, 512, 3, 1, 1, 64), ("512->512 @32", 512, 512, 3, 1, 1, 32),
              ("512->512 @8", 512, 512, 3, 1, 1, 8), ("noise 24->128 @256", 24, 128, 3, 1, 1, 256),
              ("noise 6->12 @256", 8, 12, 3, 1, 1, 256), ("d first 9->128 1x1 @256", 12, 128, 1, 1, 0, 256),
              ("torgb 128->3 @256", 128, 3, 1, 1, 0, 256), ("d conv2 128->256 s2 @257", 128, 256, 3, 2, 0, 257),
              ("d skip 128->256 1x1 s2 @255", 128, 256, 1, 2, 0, 255)]
    for name, ci, co, k, s, p, h in shapes:
        spec = ops.ConvSpec(k, k, s, p)
        x = torch.randn(B, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(co, ci, k, k, device=dev)
        hs, ws_ = spec.small_hw(h, h)
        gy = torch.randn(B, ops.pad4(co), hs, ws_, device=dev).contiguous(memory_format=torch.channels_last)
        fl = 2.0 * B * hs * ws_ * co * ci * k * k
        t_f = timeit(lambda: ops.conv_fwd(x, w, spec))
        t_d = timeit(lambda: ops.conv_bwd_data(gy, w, spec, (h, h)))
        t_w = timeit(lambda: ops.conv_wgrad(gy, x, spec, co, ci))
        print(f"{name:32s} fwd {t_f:8.3f} ms {fl / t_f / 1e9:7.1f} TF | dgrad {t_d:8.3f} ms {fl / t_d / 1e9:7.1f} TF | "
              f"wgrad {t_w:8.3f} ms {fl / t_w / 1e9:7.1f} TF")
    # rasteriser: FLAME-sized synthetic mesh (V=5023-ish, F=9976) at 256x256, batch B (atomic / latency bound)
    import numpy as np
    from gif_amd import standard_rasterize as sr
    rng = np.random.RandomState(0)
    nlat, nlon = 72, 70  # 5040 vertices, 9936 faces: a UV sphere of FLAME size
    th, ph = np.meshgrid(np.linspace(0.05, np.pi - 0.05, nlat), np.linspace(0, 2 * np.pi, nlon, endpoint=False), indexing="ij")
    vs = np.stack([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)], -1).reshape(-1, 3).astype(np.float32) * 0.8
    idx = np.arange(nlat * nlon).reshape(nlat, nlon)
    a, b_, c, d = idx[:-1], np.roll(idx, -1, 1)[:-1], idx[1:], np.roll(idx, -1, 1)[1:]
    fs = np.concatenate([np.stack([a, c, b_], -1).reshape(-1, 3), np.stack([b_, c, d], -1).reshape(-1, 3)]).astype(np.int32)
    v_t = torch.from_numpy(np.repeat(vs[None], B, 0)).cuda()
    f_t = torch.from_numpy(np.repeat(fs[None], B, 0)).cuda()
    fv = sr.face_vertices(sr.to_image_space(v_t, 256, 256), f_t)

    def rast():
        d_, t_, b2 = sr.new_buffers(B, 256, 256, "cuda")
        sr.standard_rasterize(fv, d_, t_, b2, 256, 256)
        return t_

    t_r = timeit(rast)
    cov = int((rast() >= 0).sum())
    print(f"rasterize V={vs.shape[0]} F={fs.shape[0]} @256x256 x{B}: {t_r:7.3f} ms  {B * fs.shape[0] / t_r / 1e6:8.1f} Gtri/s-e-3 "
          f"({B * fs.shape[0] / t_r / 1e3:.0f} Mtri/s)  {B * 65536 / t_r / 1e3:.0f} Mpix/s  covered {cov / (B * 65536):.2f}")
    # HBM-bound kernels at the top resolution
    for c, h in ((128, 256), (256, 128), (512, 64)):
        x = torch.randn(B, c, h, h, device=dev).contiguous(memory_format=torch.channels_last)
        k = torch.tensor([1., 3., 3., 1.], device=dev)
        k = (k[:, None] * k[None, :] / 64).contiguous()
        nbytes = x.numel() * 4
        t = timeit(lambda: ops.bias_act(x, None, None))
        print(f"bias_act      C={c} H={h}: {t:7.3f} ms {2 * nbytes / t / 1e6:8.1f} GB/s")
        t = timeit(lambda: ops.bias_act_bwd(x, x, True))
        print(f"bias_act_bwd  C={c} H={h}: {t:7.3f} ms {3 * nbytes / t / 1e6:8.1f} GB/s")
        t = timeit(lambda: ops.upfirdn2d(x, k, 1, 1, 2, (h + 1, h + 1)))
        print(f"blur pad(2,2) C={c} H={h}: {t:7.3f} ms {2 * nbytes / t / 1e6:8.1f} GB/s")
        t = timeit(lambda: ops.mul_reduce(x, x))
        print(f"mul_reduce    C={c} H={h}: {t:7.3f} ms {2 * nbytes / t / 1e6:8.1f} GB/s")


if __name__ == "__main__":
    main()
