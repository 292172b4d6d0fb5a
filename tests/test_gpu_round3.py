"""-m gpu, round 3: the LDS-tile rasteriser (lane / wave work classes), the gradient-producer epilogue
fusions (leaky-ReLU backward, bias gradient and modulation gradient inside the kernel that produces the gradient), and the
adversarial accuracy cases of the bf16x3 contraction mode."""
import os
import sys

import numpy as np
import pytest
import torch

from gpu_util import dev  # noqa: F401

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"
    from gif_amd import _lib
    _lib.load()


# ------------------------------------------------------------------------------------------------ rasteriser work classes
def _mixed_mesh(batch, seed):
    """body.obj + screen-filling back-drop faces (every tile, wave class) + random medium faces (wave class) per image."""
    from tools.raster_bench import body_mesh, random_medium, with_backdrop
    v, f = with_backdrop(*body_mesh(batch, seed), n_big=4)
    mv, mf = random_medium(batch, nfaces=400, size=0.2, seed=seed + 1)
    mv[..., 2] = mv[..., 2] * 0.5 + v[..., 2].mean()  # interleave in depth with the body
    return np.concatenate([v, mv], 1), np.concatenate([f, mf + v.shape[1]], 1)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_rasterize_all_face_classes_bit_exact(dtype):
    from gif_amd import standard_rasterize as sr
    from oracle import rasterize_oracle as ro
    B, H, W = 3, 256, 256
    v, f = _mixed_mesh(B, 5)
    fv = ro.face_vertices(ro.to_image_space(v, H, W), f).astype(dtype)
    bb = (np.floor(fv[..., 0].max(-1)).clip(0, W - 1) - np.ceil(fv[..., 0].min(-1)).clip(0, W - 1) + 1).clip(0) * \
         (np.floor(fv[..., 1].max(-1)).clip(0, H - 1) - np.ceil(fv[..., 1].min(-1)).clip(0, H - 1) + 1).clip(0)
    assert (bb <= 16).sum() > 1000 and ((bb > 16) & (bb <= 4096)).sum() > 300 and (bb > 4096).sum() >= 4 * B, "small, medium and screen-filling boxes present"
    d0 = np.zeros((B, H, W), dtype) + dtype(1e6)
    t0 = np.zeros((B, H, W), np.int32) - 1
    b0 = np.zeros((B, H, W, 3), dtype)
    ro.standard_rasterize(fv, d0, t0, b0, H, W)
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    d1 = torch.zeros(B, H, W, device="cuda", dtype=tdt) + 1e6
    t1 = torch.zeros(B, H, W, device="cuda", dtype=torch.int32) - 1
    b1 = torch.zeros(B, H, W, 3, device="cuda", dtype=tdt)
    fvt = torch.from_numpy(fv).cuda()
    it = np.int64 if dtype == np.float64 else np.int32
    for _ in range(2):  # second call on the filled buffers: idempotent
        sr.standard_rasterize(fvt, d1, t1, b1, H, W)
        assert np.array_equal(t1.cpu().numpy(), t0), "face indices"
        assert np.array_equal(d1.cpu().numpy().view(it), d0.view(it)), "depth bits"
        assert np.array_equal(b1.cpu().numpy().view(it), b0.view(it)), "barycentric bits"
    assert (t0 >= 0).mean() > 0.9  # the back-drop covers the image


def test_rasterize_many_large_faces():
    """1500 overlapping faces that each cover most of the image: every tile's candidate list is (almost) the whole chunk and every
    face is walked by a wave — same pixels, same bits, exact-depth ties to the lowest index."""
    from gif_amd import standard_rasterize as sr
    from oracle import rasterize_oracle as ro
    B, H, W, F = 1, 128, 128, 1500
    rng = np.random.RandomState(3)
    c = rng.uniform(40, 88, (B, F, 1, 2)).astype(np.float32)
    ang = rng.uniform(0, 2 * np.pi, (B, F, 1)).astype(np.float32) + rng.choice([-1.0, 1.0], (B, F, 1)).astype(np.float32) * np.array([0, 2.1, 4.2], np.float32)  # both windings
    xy = c + 60 * np.stack([np.cos(ang), np.sin(ang)], -1).astype(np.float32)
    z = rng.uniform(1, 3, (B, F, 3, 1)).astype(np.float32)
    fv = np.ascontiguousarray(np.concatenate([xy, z], -1))
    d0, t0, b0 = ro.new_buffers(B, H, W)
    ro.standard_rasterize(fv, d0, t0, b0, H, W)
    assert (t0 >= 0).mean() > 0.5
    d1, t1, b1 = sr.new_buffers(B, H, W, "cuda")
    sr.standard_rasterize(torch.from_numpy(fv).cuda(), d1, t1, b1, H, W)
    assert np.array_equal(t1.cpu().numpy(), t0)
    assert np.array_equal(d1.cpu().numpy().view(np.int32), d0.view(np.int32))
    assert np.array_equal(b1.cpu().numpy().view(np.int32), b0.view(np.int32))


# ------------------------------------------------------------------------------------------------ gradient-producer epilogue fusions
def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _check_fused(out_f, fuse, plain, x_dot, s_out, residual, mask_src, slope, gain, what, tol=1.0):
    """out_f / fuse.dot / fuse.colsum of a fused launch against float64 math on the PLAIN launch's output (same kernels, no
    fusion): v = plain; dot = sum_hw v * x; v = s * v + residual; v *= gain * (mask > 0 ? 1 : slope); colsum = sum v.
    tol: 1 for fp32 tensors; f16 tensors: the plain output and the fused output are each rounded to 11 bits (2^-11 = 4.9e-4)."""
    v = plain.double()
    if fuse.dot_src is not None:
        dot = (v * x_dot.double()).sum(dim=(2, 3))
        e = ((fuse.dot.double() - dot).abs().max() / dot.abs().max()).item()
        assert e < 2e-5 * tol, f"{what}: dot rel err {e:.2e}"
    if s_out is not None:
        v = v * s_out.double()[:, :, None, None]
    if residual is not None:
        v = v + residual.double()
    if mask_src is not None:
        v = v * gain * torch.where(mask_src > 0, 1.0, slope).double()
    e = ((out_f.double() - v).abs().max() / v.abs().max()).item()
    assert e < (2e-6 if tol == 1.0 else 1.5e-3), f"{what}: output rel err {e:.2e}"
    if fuse.want_colsum:
        cs = v.sum(dim=(0, 2, 3))
        e = ((fuse.colsum.double() - cs).abs().max() / cs.abs().max()).item()
        assert e < 2e-5 * tol, f"{what}: colsum rel err {e:.2e}"


# (B, C_small, C_big, K, stride, pad, H_big, op, winograd, mode)   op: "dgrad" = conv_bwd_data small -> big, "fwd" = conv_fwd big -> small
FUSE_CASES = [
    (8, 128, 128, 3, 1, 1, 64, "dgrad", True, "bf16x3"),    # Winograd bf16x3 GEMM epilogue (8192 tiles)
    (8, 128, 128, 3, 1, 1, 64, "dgrad", True, "native"),    # Winograd native GEMM epilogue
    (8, 64, 64, 3, 1, 1, 64, "dgrad", True, "bf16x3"),      # Winograd, 64 output channels: native GEMM, narrow N tile
    (8, 128, 128, 3, 1, 1, 128, "dgrad", False, "bf16x3"),  # direct 256x128 tiles (8 waves)
    (4, 256, 256, 3, 1, 1, 32, "dgrad", False, "bf16x3"),   # direct, 64x64 tiles
    (32, 512, 512, 3, 1, 1, 32, "dgrad", False, "bf16x3"),  # direct, 128x64 tiles of the low-resolution layers
    (8, 128, 128, 3, 1, 1, 64, "dgrad", False, "native"),   # direct native fp32 MFMA, 128x128 tiles
    (4, 128, 24, 3, 1, 1, 64, "dgrad", False, "bf16x3"),    # 256x32 tiles (<= 32 output channels)
    (4, 4, 128, 1, 1, 0, 64, "dgrad", False, "bf16x3"),     # ToRGB's data gradient: 4 contraction channels, register-staged kernel
    (4, 128, 64, 3, 2, 0, 65, "fwd", False, "bf16x3"),      # stride-2 forward conv = data gradient of the up-sampling conv_transpose
    (4, 128, 64, 3, 2, 0, 65, "dgrad", False, "bf16x3"),    # transposed stride 2: four phases (mask + colsum only)
    (4, 512, 512, 3, 2, 0, 9, "dgrad", False, "bf16x3"),    # transposed stride 2, phases merged into one launch
    (32, 256, 128, 3, 2, 0, 129, "dgrad", False, "bf16x3"), # transposed stride 2, bulk + remainder launches per phase
    # the same direct launches on the f16x2 kernels (round 5: the descaled accumulators go through the shared epilogue)
    (8, 128, 128, 3, 1, 1, 128, "dgrad", False, "f16x2"),   # 256x128 tiles (8 waves)
    (8, 128, 128, 3, 1, 1, 96, "dgrad", False, "f16x2"),    # 128x128 tiles (4 waves, two workgroups per CU)
    (4, 256, 256, 3, 1, 1, 32, "dgrad", False, "f16x2"),    # 64x64 tiles
    (32, 512, 512, 3, 1, 1, 32, "dgrad", False, "f16x2"),   # 128x64 tiles
    (4, 128, 24, 3, 1, 1, 64, "dgrad", False, "f16x2"),     # 256x32 tiles
    (4, 128, 64, 3, 2, 0, 65, "fwd", False, "f16x2"),       # stride-2 forward
    (4, 128, 64, 3, 2, 0, 65, "dgrad", False, "f16x2"),     # transposed stride 2: four phases
    (4, 512, 512, 3, 2, 0, 9, "dgrad", False, "f16x2"),     # phases merged into one launch
    (32, 256, 128, 3, 2, 0, 129, "dgrad", False, "f16x2"),  # big phases merged / bulk + remainder
]


@pytest.mark.parametrize("case", FUSE_CASES)
def test_grad_fuse_conv_epilogues(case, monkeypatch):
    from gif_amd import ops
    B, Cs, Cb, K, st, pad, Hb, op, wino, mode = case
    monkeypatch.setattr(ops, "WINOGRAD", wino)
    monkeypatch.setattr(ops, "WINOGRAD_MIN_C", 0)  # (the cases pin the route themselves; default dispatch: Winograd from 256 channels)
    monkeypatch.setattr(ops, "WINOGRAD_WGRAD_MIN_C", 0)
    prev = ops.get_fp32_mfma_mode()
    ops.set_fp32_mfma_mode(mode)
    try:
        g = torch.Generator().manual_seed(Hb + Cs)
        spec = ops.ConvSpec(K, K, st, pad)
        Hs = spec.small_hw(Hb, Hb)[0]
        w = (torch.randn(Cs, Cb, K, K, generator=g) / (Cb * K * K) ** 0.5).cuda()
        if op == "dgrad":
            Cin, Cout, Hin, Hout = Cs, Cb, Hs, Hb
            run = lambda src, **epi: ops.conv_bwd_data(src, w, spec, (Hb, Hb), **epi)  # noqa: E731
        else:
            Cin, Cout, Hin, Hout = Cb, Cs, Hb, Hs
            run = lambda src, **epi: ops.conv_fwd(src, w, spec, **epi)  # noqa: E731
        src = _cl(torch.randn(B, Cin, Hin, Hin, generator=g).cuda())
        x = _cl(torch.randn(B, Cout, Hout, Hout, generator=g).cuda())  # the tensor the gradient belongs to (activation output)
        res = _cl(torch.randn(B, Cout, Hout, Hout, generator=g).cuda())
        d_in = (torch.rand(B, Cin, generator=g) + 0.5).cuda()
        s_out = (torch.rand(B, Cout, generator=g) + 0.5).cuda()
        single = not (op == "dgrad" and st == 2)
        can_dot = single and (Hout * Hout) % 1024 == 0
        plain = run(src, in_scale=d_in)
        if wino and op == "dgrad":
            n0 = ops.prof_winograd_calls()
        # (a) everything at once: modulation output scale, dot product, mask, column sums, residual
        fuse = ops.GradFuse(mask_src=x, mask_slope=0.2, mask_gain=2 ** 0.5, want_colsum=True, dot_src=x if can_dot else None)
        out = run(src, in_scale=d_in, out_scale=s_out, residual=res, fuse=fuse)
        if wino and op == "dgrad":
            assert ops.prof_winograd_calls() > n0, "the Winograd path was expected to run"
        _check_fused(out, fuse, plain, x, s_out, res, x, 0.2, 2 ** 0.5, f"{case} all")
        # (b) mask only, no sums (the discriminator in the G step: no bias gradient wanted); ReLU-style slope 0
        fuse = ops.GradFuse(mask_src=x, mask_slope=0.0, mask_gain=1.0)
        out = run(src, in_scale=d_in, fuse=fuse)
        _check_fused(out, fuse, plain, None, None, None, x, 0.0, 1.0, f"{case} mask")
        # (c) dot only, different tensor than the mask
        if can_dot:
            other = _cl(torch.randn(B, Cout, Hout, Hout, generator=g).cuda())
            fuse = ops.GradFuse(dot_src=other)
            out = run(src, in_scale=d_in, out_scale=s_out, fuse=fuse)
            _check_fused(out, fuse, plain, other, s_out, None, None, 1.0, 1.0, f"{case} dot")
            # determinism of the partial-sum reduction
            fuse2 = ops.GradFuse(dot_src=other)
            run(src, in_scale=d_in, out_scale=s_out, fuse=fuse2)
            assert torch.equal(fuse.dot, fuse2.dot)
        else:
            with pytest.raises(Exception, match="dot fusion"):
                run(src, in_scale=d_in, fuse=ops.GradFuse(dot_src=x))
    finally:
        ops.set_fp32_mfma_mode(prev)


# f16 activations: the same epilogues read the mask / dot source as halfs; sums are fp32 and taken before the store rounds
F16_FUSE_CASES = [
    (8, 128, 128, 3, 1, 1, 64, "dgrad"),   # 128x128 tiles
    (4, 256, 256, 3, 1, 1, 16, "dgrad"),   # 64x64 tiles
    (4, 128, 24, 3, 1, 1, 64, "dgrad"),    # 256x32 tiles
    (4, 64, 64, 3, 1, 1, 64, "dgrad"),     # 128x64 tiles
    (4, 8, 128, 1, 1, 0, 64, "dgrad"),     # ToRGB's data gradient (3 colour channels padded to 8)
    (4, 128, 64, 3, 2, 0, 65, "fwd"),      # stride-2 forward conv = data gradient of the up-sampling conv_transpose
    (4, 128, 64, 3, 2, 0, 65, "dgrad"),    # transposed stride 2: four phases (mask + colsum only)
    (8, 64, 256, 3, 1, 1, 128, "dgrad"),   # 256x256 tiles (8 waves), modulated, every fusion
]


@pytest.mark.parametrize("case", F16_FUSE_CASES)
def test_grad_fuse_conv_epilogues_f16(case):
    from gif_amd import ops
    B, Cs, Cb, K, st, pad, Hb, op = case
    g = torch.Generator().manual_seed(Hb + Cs)
    spec = ops.ConvSpec(K, K, st, pad)
    Hs = spec.small_hw(Hb, Hb)[0]
    w = (torch.randn(Cs, Cb, K, K, generator=g) / (Cb * K * K) ** 0.5).cuda()
    if op == "dgrad":
        Cin, Cout, Hin, Hout = Cs, Cb, Hs, Hb
        run = lambda src, **epi: ops.conv_bwd_data(src, w, spec, (Hb, Hb), **epi)  # noqa: E731
    else:
        Cin, Cout, Hin, Hout = Cb, Cs, Hb, Hs
        run = lambda src, **epi: ops.conv_fwd(src, w, spec, **epi)  # noqa: E731
    h = lambda *shape: _cl(torch.randn(*shape, generator=g).cuda().half())  # noqa: E731
    src, x, res = h(B, Cin, Hin, Hin), h(B, Cout, Hout, Hout), h(B, Cout, Hout, Hout)
    d_in = (torch.rand(B, Cin, generator=g) + 0.5).cuda()
    s_out = (torch.rand(B, Cout, generator=g) + 0.5).cuda()
    can_dot = not (op == "dgrad" and st == 2) and (Hout * Hout) % 1024 == 0
    # reference: the same launch with an fp32 result is not available for every op, so the plain f16 launch stands in (its own
    # rounding is inside the tolerance); the fp32 sums of the fused launch are held to the sums of those rounded values
    plain = run(src, in_scale=d_in)
    fuse = ops.GradFuse(mask_src=x, mask_slope=0.2, mask_gain=2 ** 0.5, want_colsum=True, dot_src=x if can_dot else None)
    out = run(src, in_scale=d_in, out_scale=s_out, residual=res, fuse=fuse)
    assert out.dtype == torch.float16
    _check_fused(out, fuse, plain, x, s_out, res, x, 0.2, 2 ** 0.5, f"f16 {case} all", tol=50.0)
    fuse = ops.GradFuse(mask_src=x, mask_slope=0.0, mask_gain=1.0)
    out = run(src, in_scale=d_in, fuse=fuse)
    _check_fused(out, fuse, plain, None, None, None, x, 0.0, 1.0, f"f16 {case} mask", tol=50.0)
    fuse2 = ops.GradFuse(mask_src=x, mask_slope=0.0, mask_gain=1.0)
    assert torch.equal(run(src, in_scale=d_in, fuse=fuse2), out)


@pytest.mark.parametrize("shape", [(16, 128, 129), (4, 256, 33), (2, 512, 9), (3, 64, 40)])
def test_grad_fuse_blur_adjoint(shape):
    """The blur's adjoint (the gradient w.r.t. ConvLayer conv1's activated output inside a ResBlock) with the leaky-ReLU mask and
    the bias-gradient column sums in the FIR kernels' epilogue (sliding-window kernel at the first shape, tiled kernel below)."""
    from gif_amd import ops
    B, C, H = shape
    g = torch.Generator().manual_seed(H)
    k1 = torch.tensor([1., 3., 3., 1.])
    k = (k1[:, None] * k1[None, :] / 64).cuda()
    gy = _cl(torch.randn(B, C, H, H, generator=g).cuda())      # gradient w.r.t. the blurred map (pad (2,2): H = Hin + 1)
    y = _cl(torch.randn(B, C, H - 1, H - 1, generator=g).cuda())  # conv1's activated output
    plain = ops.upfirdn2d(gy, k, 1, 1, 1, (H - 1, H - 1), False)
    fuse = ops.GradFuse(mask_src=y, mask_slope=0.2, mask_gain=2 ** 0.5, want_colsum=True)
    out = ops.upfirdn2d(gy, k, 1, 1, 1, (H - 1, H - 1), False, fuse=fuse)
    _check_fused(out, fuse, plain, None, None, None, y, 0.2, 2 ** 0.5, f"blur adjoint {shape}")
    fuse = ops.GradFuse(mask_src=y, mask_slope=0.2, mask_gain=2 ** 0.5)
    out = ops.upfirdn2d(gy, k, 1, 1, 1, (H - 1, H - 1), False, fuse=fuse)
    _check_fused(out, fuse, plain, None, None, None, y, 0.2, 2 ** 0.5, f"blur adjoint {shape} mask only")
    # f16 activations through the same kernels
    gy, y = gy.half(), y.half()
    plain = ops.upfirdn2d(gy, k, 1, 1, 1, (H - 1, H - 1), False)
    fuse = ops.GradFuse(mask_src=y, mask_slope=0.2, mask_gain=2 ** 0.5, want_colsum=True)
    out = ops.upfirdn2d(gy, k, 1, 1, 1, (H - 1, H - 1), False, fuse=fuse)
    _check_fused(out, fuse, plain, None, None, None, y, 0.2, 2 ** 0.5, f"blur adjoint f16 {shape}", tol=50.0)


@pytest.mark.parametrize("res,step,f16", [(32, 3, False), (64, 4, False), (64, 4, True)])
def test_model_gradients_fused_equal_standalone(res, step, f16, monkeypatch):
    """Whole G-through-D and D gradients with the activation ports on (default) against GIF_FUSE_GRAD off — the same kernels
    minus the stand-alone passes: agreement at accumulation-order level, and the stand-alone passes really disappear."""
    import contextlib
    import io
    from gif_amd import ops
    from gif_amd.discriminator import Discriminator
    from gif_amd.generator import StyledGenerator
    import torch.nn.functional as F
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        G = StyledGenerator(embedding_vocab_size=16, rendered_flame_ascondition=True, normal_maps_as_cond=True).cuda()
        D = Discriminator(size=res, num_color_chnls=9).cuda()
    B = 8
    cond = (torch.rand(B, 6, res, res) * 2 - 1).cuda()
    real = (torch.rand(B, 3, res, res) * 2 - 1).cuda()
    idx = torch.randint(0, 16, (B,)).cuda()
    gp = [p for n, p in G.named_parameters() if not any(f"progression.{i}." in n or f"to_rgb.{i}." in n for i in range(step + 1, 9))]
    dp = list(D.parameters())
    calls = {"bias_act_bwd": 0, "mul_reduce": 0}
    for name in calls:
        fn = getattr(ops, name)
        monkeypatch.setattr(ops, name, (lambda f, n: (lambda *a, **k: (calls.__setitem__(n, calls[n] + 1), f(*a, **k))[1]))(fn, name))

    def run(fused):
        monkeypatch.setattr(ops, "FUSE_GRAD", fused)
        for n in calls:
            calls[n] = 0
        fake = G(cond, None, step=step, alpha=1, input_indices=idx)
        lg = F.softplus(-D(fake, condition=cond, step=step, alpha=1)[0]).mean()
        gg = torch.autograd.grad(lg, gp, allow_unused=True)
        ld = F.softplus(-D([real], condition=cond, step=step, alpha=1)[0]).mean() + F.softplus(D([fake[0].detach()], condition=cond, step=step, alpha=1)[0]).mean()
        gd = torch.autograd.grad(ld, dp)
        return gg, gd, dict(calls)

    if f16:
        # f16 activations: same ports, masks read as halfs.  The two runs round at different points, so they differ by f16 rounding
        # noise — up to 9 % of the max for the 6->12-channel condition-noise weights, whose f16 gradient is 13 % off the fp32 one
        # with or without the fusions (tools/probes/f16_fuse_diag.py).  Bound: per tensor, no further from the fp32 gradients than
        # twice the stand-alone f16 passes are.
        ggr, gdr, _ = run(False)
        G.set_activation_dtype(torch.float16), D.set_activation_dtype(torch.float16)
    gg0, gd0, c0 = run(False)
    gg1, gd1, c1 = run(True)
    assert c1["bias_act_bwd"] < c0["bias_act_bwd"] and c1["mul_reduce"] < c0["mul_reduce"], (c0, c1)
    if f16:
        for a, b, r in list(zip(gg1, gg0, ggr)) + list(zip(gd1, gd0, gdr)):
            assert (a is None) == (r is None)
            if r is not None:
                m = r.abs().max() + 1e-20
                e_fused, e_plain = ((a - r).abs().max() / m).item(), ((b - r).abs().max() / m).item()
                assert e_fused <= 2 * e_plain + 1e-2, (e_fused, e_plain, tuple(r.shape))
        return
    worst = 0.0
    for a, b in list(zip(gg1, gg0)) + list(zip(gd1, gd0)):
        assert (a is None) == (b is None)
        if b is not None:
            worst = max(worst, ((a - b).abs().max() / (b.abs().max() + 1e-20)).item())
    assert worst < 5e-5, worst


# ------------------------------------------------------------------------------------------------ RCCL with more than one rank
def _run_bench(nproc, extra, port, steps=3, r1_every=2, env_extra=None):
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    # the driver's command form: no launcher around it, bench.py spawns its own ranks (bench.self_launch); `port` is unused
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(nproc), "--steps", str(steps), "--warmup", "1", "--res", "64",
           "--batch", "8", "--vocab", "64", "--r1-every", str(r1_every), "--no-cpu-baseline", "--no-prof", "--check-replicas"] + extra
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


def _check_comm_fields(line, ranks):
    import math
    assert line["n_gpus"] == ranks and line["replicas"]["ranks"] == ranks and line["replicas"]["backend"] == "nccl"
    assert line["replicas"]["replicas_identical"]
    assert "comm_exposed_ms" in line and math.isfinite(line["comm_exposed_ms"]) and line["comm_exposed_ms"] >= 0.0
    assert math.isfinite(line["value"]) and line["value"] > 0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs of one node (RCCL over xGMI); the round's boxes have one")
def test_bench_two_ranks_on_rccl_replicas_identical_and_overlap_equals_in_place():
    """`python bench.py --gpus 2` (self-launching two ranks on RCCL), 17 timed steps at R1 every 16th (one R1 iteration inside), for
    BOTH D-step schedules (fused default, --two-call-d): the replicas stay bit-identical, RCCL reports two ranks, `comm_exposed_ms`
    is present and finite, and the deferred (overlapped) exchange + optimiser schedule gives exactly the weights of the in-place
    schedule."""
    for extra in ([], ["--two-call-d"]):
        a = _run_bench(2, extra, 29631, steps=17, r1_every=16)
        _check_comm_fields(a, 2)
        assert a["config"]["overlap_comm"] is True and a["config"]["r1_iterations_timed"] == 1
        assert ("two calls" in a["config"]["d_step_discriminator_passes"]) == bool(extra)
        b = _run_bench(2, extra + ["--no-overlap-comm"], 29633, steps=17, r1_every=16)
        _check_comm_fields(b, 2)
        assert b["config"]["overlap_comm"] is False
        assert a["replicas"]["param_digest"] == b["replicas"]["param_digest"], "overlapped and in-place schedules must give identical weights"


def test_bench_forced_collectives_take_the_single_gpu_launch_sequence():
    """SCALE's N = 1 line must equal BENCH: a run with the collective code paths forced on (GIF_FORCE_DIST=1, one rank — what an
    N > 1 rank executes apart from the size of the group) takes the SAME D-step path as the plain single-process run (the fused pass
    over [real; fake]) and ends with bit-identical weights; --two-call-d is reported as such.  17 steps, one R1 iteration inside."""
    plain = _run_bench(1, [], 0, steps=17, r1_every=16)
    forced = _run_bench(1, [], 0, steps=17, r1_every=16, env_extra={"GIF_FORCE_DIST": "1"})
    assert plain["config"]["d_step_discriminator_passes"] == forced["config"]["d_step_discriminator_passes"]
    assert "one pass" in forced["config"]["d_step_discriminator_passes"]
    _check_comm_fields(forced, 1)
    assert forced["config"]["overlap_comm"] is True and plain["config"]["overlap_comm"] is False
    assert forced["replicas"]["param_digest"] == plain["replicas"]["param_digest"], "deferred exchanges must not change the weights"
    two = _run_bench(1, ["--two-call-d"], 0, steps=17, r1_every=16, env_extra={"GIF_FORCE_DIST": "1"})
    _check_comm_fields(two, 1)
    assert "two calls" in two["config"]["d_step_discriminator_passes"]


def test_bench_single_rank_collective_code_path_and_line_contract():
    """One rank with GIF_FORCE_DIST=1: process-group initialisation on RCCL, the construction-time broadcast, the asynchronous AVG
    all-reduce and its waits all execute; the JSON line carries the round-3 fields (named roofline ceiling, comm accounting)."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GIF_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--res", "64", "--batch", "8",
                          "--vocab", "64", "--no-cpu-baseline", "--check-replicas"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["replicas"]["backend"] == "nccl" and line["replicas"]["replicas_identical"]
    assert line["config"]["overlap_comm"] is True and line["comm_exposed_ms"] >= 0.0
    r = line["roofline"]
    assert r["unit"] == "TFLOP/s" and "ceiling" in r and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["executed_frac"] - r["executed_achieved"] / r["executed_peak"]) < 1e-9
    rr = line["roofline_rasterize"]
    assert rr.get("error") is None and rr["unit"] == "GB/s" and rr["Mtri_per_s"] > 0 and rr["cpu_baseline"]["value"] > 0


# ------------------------------------------------------------------------------------------------ advisor findings of round 2
def test_f16_gradient_store_saturation_raises_the_overflow_flag():
    """f16 activation stores clamp at +-65504: an overflowing activation GRADIENT would never reach the loss scaler as inf.  The
    gradient-carrying stores raise a device flag instead (gif_f16_overflow_clear / _or_into), which DeviceLossScaler ORs into
    found_inf: an overflow now halves the scale and skips the step."""
    from gif_amd import _lib, ops
    from gif_amd.train_step import DeviceLossScaler
    lib = _lib.load()
    stream = torch.cuda.current_stream().cuda_stream
    B, C, H = 2, 64, 16
    w = (torch.randn(C, C, 3, 3) / 24).cuda()
    spec = ops.ConvSpec(3, 3, 1, 1)
    found = torch.zeros((), device="cuda")

    def flag_after(fn):
        assert lib.gif_f16_overflow_clear(stream) == 0
        fn()
        found.zero_()
        assert lib.gif_f16_overflow_or_into(found.data_ptr(), stream) == 0
        return found.item()

    small = _cl(torch.randn(B, C, H, H).cuda().half())
    huge = _cl((torch.randn(B, C, H, H) * 3e4).cuda().half())
    assert flag_after(lambda: ops.conv_bwd_data(small, w, spec, (H, H))) == 0.0
    out = ops.conv_bwd_data(huge, w, spec, (H, H))
    assert out.abs().max().item() == 65504.0, "the store saturates"
    assert flag_after(lambda: ops.conv_bwd_data(huge, w, spec, (H, H))) == 1.0, "data-gradient epilogue"
    y = _cl(torch.randn(B, C, H, H).cuda().half())
    assert flag_after(lambda: ops.bias_act_bwd(small, y, True)) == 0.0
    assert flag_after(lambda: ops.bias_act_bwd(huge * 2, y, True)) == 1.0, "leaky-ReLU backward"
    k1 = torch.tensor([1., 3., 3., 1.])
    k = (k1[:, None] * k1[None, :] / 4).cuda()  # gain 16: drives the FIR output past the f16 range
    assert flag_after(lambda: ops.upfirdn2d(huge, k, 1, 1, 1, (H - 1, H - 1), False)) == 1.0, "FIR adjoint"
    s = torch.full((B, C), 4.0, device="cuda")
    assert flag_after(lambda: ops.mul_reduce(huge, small, scale=s, want_scaled=True)) == 1.0, "modulation-gradient pass"
    # the scaler: finite fp32 bucket + raised flag => found_inf, scale halves
    sc = DeviceLossScaler(torch.device("cuda"), init_scale=1024.0)
    sc.begin_backward()
    ops.conv_bwd_data(huge, w, spec, (H, H))
    sc.update(torch.ones(16, device="cuda"))
    assert sc.found_inf.item() == 1.0 and sc.scale.item() == 512.0 and sc.skipped.item() == 1.0
    sc.begin_backward()
    ops.conv_bwd_data(small, w, spec, (H, H))
    sc.update(torch.ones(16, device="cuda"))
    assert sc.found_inf.item() == 0.0 and sc.scale.item() == 512.0


def test_flat_adam_under_loss_scaling_counts_applied_steps_only():
    """A skipped (overflowed) step must not advance Adam's bias corrections, and the EMA of parameters outside the gradient bucket
    still runs on applied steps (advisor finding, round 2).  Reference: torch.optim.Adam stepping only on the applied steps."""
    from gif_amd.optim import FlatAdam
    from gif_amd.train_step import FlatGradBucket
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Linear(32, 8)).cuda()
    ema = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Linear(32, 8)).cuda()
    ema.load_state_dict(net.state_dict())
    ref = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Linear(32, 8)).cuda()
    ref.load_state_dict(net.state_dict())
    ref_ema = [p.detach().clone() for p in ref.parameters()]
    frozen = net[1].bias  # outside the bucket: no gradient, still part of the EMA
    bucket = FlatGradBucket(net.parameters(), active=lambda p: p is not frozen)
    opt = FlatAdam(net.parameters(), lr=0.01, betas=(0.5, 0.9), bucket=bucket, ema_params=list(ema.parameters()))
    ropt = torch.optim.Adam([p for p in ref.parameters()], lr=0.01, betas=(0.5, 0.9))
    scale = 8.0
    inv = torch.tensor(1.0 / scale, device="cuda")
    decay = 0.9
    x = torch.randn(5, 4, 16, device="cuda")
    for it, overflow in enumerate([True, False, False, True, False]):
        with torch.no_grad():  # an externally changed frozen parameter must reach the EMA on applied steps
            frozen.add_(0.01)
            list(ref.parameters())[3].add_(0.01)
        bucket.zero()
        (net(x[it]).pow(2).mean() * scale).backward()
        found = torch.tensor(1.0 if overflow else 0.0, device="cuda")
        opt.step(ema_decay=decay, inv_grad_scale=inv, found_inf=found)
        if not overflow:
            ropt.zero_grad()
            ref(x[it]).pow(2).mean().backward()
            list(ref.parameters())[3].grad = None
            ropt.step()
            with torch.no_grad():
                for e, p in zip(ref_ema, ref.parameters()):
                    e.mul_(decay).add_(p, alpha=1 - decay)
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    for a, b in zip(ema.parameters(), ref_ema):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    assert float(opt.state_dict()["state"][0]["step"]) == 3.0, "three applied steps"


# ------------------------------------------------------------------------------------------------ fused style path (demodulation)
@pytest.mark.parametrize("shape", [(32, 512, 512, 3), (8, 128, 256, 3), (5, 64, 32, 3), (4, 256, 3, 1)])
def test_demodulation_kernels_and_function(shape):
    """d = rsqrt(scale^2 * s^2 @ wsq^T + eps) as one skinny GEMM (squares in the operand load, rsqrt in the epilogue), its
    backward as two GEMMs + one pass over the weight, against float64 math; DemodFn against the any-order reference
    composition to first and second order (stylegan2_common_layers.py:311-320)."""
    from gif_amd import functional as GF
    from gif_amd import ops
    B, cin, cout, k = shape
    g = torch.Generator().manual_seed(cin + cout)
    cin_pad, cout_pad = cin + 4, ops.pad4(cout) + 4
    s = torch.zeros(B, cin_pad)
    s[:, :cin] = torch.randn(B, cin, generator=g) + 1.0
    w = torch.randn(cout, cin, k, k, generator=g)
    gd = torch.randn(B, cout_pad, generator=g)
    gs_in = torch.randn(B, cin_pad, generator=g)
    scale2, eps = 1.0 / (cin * k * k), 1e-8
    sd, wd = s.double(), w.double()
    wsq_ref = wd.pow(2).sum(dim=(2, 3))
    d_ref = torch.rsqrt(scale2 * (sd[:, :cin].pow(2) @ wsq_ref.t()) + eps)
    sc, wc, gdc = s.cuda(), w.cuda(), gd.cuda()
    wsq = ops.weight_sq_sum(wc)
    assert ((wsq.double().cpu() - wsq_ref).abs().max() / wsq_ref.abs().max()).item() < 1e-6
    d = ops.style_demod(sc, wsq, scale2, eps, cout_pad)
    assert d.shape == (B, cout_pad) and (d[:, cout:] == 1).all()
    assert ((d[:, :cout].double().cpu() - d_ref).abs().max() / d_ref.abs().max()).item() < 2e-6
    g_acc = gd.double()[:, :cout] * (-0.5 * scale2) * d_ref.pow(3)
    gs_ref = torch.zeros(B, cin_pad, dtype=torch.float64)  # padding columns are written as zero
    gs_ref[:, :cin] = gs_in.double()[:, :cin] + 2 * sd[:, :cin] * (g_acc @ wsq_ref)
    gs = ops.style_demod_bwd_s(gdc, d, wsq, sc, gs_in.cuda(), scale2)
    assert ((gs.double().cpu() - gs_ref).abs().max() / gs_ref.abs().max()).item() < 5e-6
    gs0 = ops.style_demod_bwd_s(gdc, d, wsq, sc, None, scale2)
    assert (gs0[:, cin:] == 0).all()
    gs0_ref = gs_ref.clone()
    gs0_ref[:, :cin] -= gs_in.double()[:, :cin]
    assert ((gs0.double().cpu() - gs0_ref).abs().max() / gs_ref.abs().max()).item() < 5e-6
    gwsq_ref = g_acc.t() @ sd[:, :cin].pow(2)
    gwsq = ops.style_demod_bwd_w(gdc, d, sc, cout, cin, scale2)
    assert ((gwsq.double().cpu() - gwsq_ref).abs().max() / gwsq_ref.abs().max()).item() < 5e-6
    gw = ops.demod_wgrad(wc, gwsq)
    gw_ref = 2 * wd * gwsq_ref[:, :, None, None]
    assert ((gw.double().cpu() - gw_ref).abs().max() / gw_ref.abs().max()).item() < 5e-6
    # the Function: first order (fused kernels) and second order (recorded composition) vs autograd through the reference
    s1 = sc.clone().requires_grad_(True)
    w1 = wc.clone().requires_grad_(True)
    s2 = sc.clone().requires_grad_(True)
    w2 = wc.clone().requires_grad_(True)
    d1 = GF.demodulation(s1, w1, scale2 ** 0.5, eps, cout_pad)
    d2 = GF._demod_reference(s2, w2, scale2, eps, cout_pad)
    assert ((d1 - d2).abs().max() / d2.abs().max()).item() < 2e-6
    a1 = torch.autograd.grad(d1, [s1, w1], gdc)
    a2 = torch.autograd.grad(d2, [s2, w2], gdc)
    for x1, x2 in zip(a1, a2):
        assert ((x1 - x2).abs().max() / x2.abs().max()).item() < 1e-5
    d1 = GF.demodulation(s1, w1, scale2 ** 0.5, eps, cout_pad)
    d2 = GF._demod_reference(s2, w2, scale2, eps, cout_pad)
    (g1,) = torch.autograd.grad(d1, s1, gdc, create_graph=True)
    (g2,) = torch.autograd.grad(d2, s2, gdc, create_graph=True)
    h1 = torch.autograd.grad(g1.pow(2).sum(), [s1, w1])
    h2 = torch.autograd.grad(g2.pow(2).sum(), [s2, w2])
    for x1, x2 in zip(h1, h2):
        assert ((x1 - x2).abs().max() / x2.abs().max()).item() < 1e-4
