"""-m gpu: every HIP kernel (through the C ABI, via gif_amd.ops / gif_amd.functional) against the CPU oracle on
the same seeded inputs.  Integer/index results bit-exact; fp32 results within the stated tolerance (summation
order differs: MFMA k-ordered fmaf chains vs MKL-DNN)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import assert_close, dev, host, pad4, rel_err

pytestmark = pytest.mark.gpu

TOL = 2e-5  # fp32 accumulation-order tolerance, relative to the tensor's max magnitude


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"
    from gif_amd import _lib
    _lib.load()


# ------------------------------------------------------------------------------------------------ rasteriser
def _body():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "body_mesh.npz"))
    return g, (g["vertices"] * np.float32(0.8))[None], g["faces"][None]


def test_rasterize_golden_visibility():
    from gif_amd import standard_rasterize as sr
    g, v, f = _body()
    vt, ft = torch.from_numpy(v).cuda(), torch.from_numpy(f).cuda()
    vis = sr.get_visibility(vt, ft, 512, 512)
    assert np.array_equal(vis[0].cpu().numpy().astype(np.uint8), g["vis"])
    visz = sr.get_visibility_z(vt, ft, 512, 512)
    assert np.array_equal(visz[0].cpu().numpy().astype(np.uint8), g["vis_z"])


@pytest.mark.parametrize("hw", [(512, 512), (256, 256), (64, 96)])
def test_rasterize_bit_exact_vs_oracle(hw):
    from gif_amd import standard_rasterize as sr
    from oracle import rasterize_oracle as ro
    h, w = hw
    g, v, f = _body()
    # batch of 3: rotated / scaled copies so that depth order and coverage differ per image
    rng = np.random.RandomState(0)
    vs = []
    for i in range(3):
        a = rng.uniform(-0.6, 0.6)
        Rm = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
        vs.append((v[0] @ Rm.T * np.float32(1.0 + 0.2 * i)).astype(np.float32))
    v3 = np.stack(vs)
    f3 = np.repeat(f, 3, 0)
    vi = ro.to_image_space(v3, h, w)
    fv = ro.face_vertices(vi, f3)
    d0, t0, b0 = ro.new_buffers(3, h, w)
    ro.standard_rasterize(fv, d0, t0, b0, h, w)
    fvt = torch.from_numpy(fv).cuda()
    d1, t1, b1 = sr.new_buffers(3, h, w, "cuda")
    out = sr.standard_rasterize(fvt, d1, t1, b1, h, w)
    assert out[0] is d1 and out[1] is t1 and out[2] is b1  # in place + returned, like the reference
    assert np.array_equal(t1.cpu().numpy(), t0), "face index buffer"
    assert np.array_equal(d1.cpu().numpy().view(np.uint32), d0.view(np.uint32)), "depth bits"
    assert np.array_equal(b1.cpu().numpy().view(np.uint32), b0.view(np.uint32)), "barycentric bits"
    # colours variant: vertex normals-like attribute
    col = np.ascontiguousarray(rng.standard_normal(fv.shape).astype(np.float32))
    d2, t2, i2 = ro.new_buffers(3, h, w)
    ro.standard_rasterize_colors(fv, col, d2, t2, i2, h, w)
    d3, t3, i3 = sr.new_buffers(3, h, w, "cuda")
    sr.standard_rasterize_colors(fvt, torch.from_numpy(col).cuda(), d3, t3, i3, h, w)
    assert np.array_equal(t3.cpu().numpy(), t2)
    assert np.array_equal(i3.cpu().numpy().view(np.uint32), i2.view(np.uint32)), "interpolated attribute bits"
    # a second call on the already-filled buffers is idempotent (the reference launches its kernel twice)
    sr.standard_rasterize(fvt, d1, t1, b1, h, w)
    assert np.array_equal(t1.cpu().numpy(), t0) and np.array_equal(d1.cpu().numpy(), d0)


def test_rasterize_edge_cases():
    from gif_amd import standard_rasterize as sr
    d, t, b = sr.new_buffers(2, 8, 8, "cuda")
    sr.standard_rasterize(torch.zeros(2, 0, 3, 3, device="cuda"), d, t, b, 8, 8)  # no faces
    assert (t == -1).all() and (d == 1e6).all()
    # exact depth tie between two identical faces -> lowest face index, deterministically
    tri = torch.tensor([[[1, 1, 2], [1, 6, 2], [6, 1, 2]]], dtype=torch.float32)
    fv = torch.stack([tri, tri], 1).reshape(1, 2, 3, 3).cuda().contiguous()
    for _ in range(5):
        d, t, b = sr.new_buffers(1, 8, 8, "cuda")
        sr.standard_rasterize(fv, d, t, b, 8, 8)
        assert set(t.unique().tolist()) == {-1, 0}
    with pytest.raises(RuntimeError, match="contiguous"):
        sr.standard_rasterize(fv.transpose(2, 3), d, t, b, 8, 8)


# ------------------------------------------------------------------------------------------------ convolutions
CONV_CASES = [
    # (B, Cin, Cout, K, stride, pad, H)            which layer of the model it stands for
    (2, 128, 128, 3, 1, 1, 32),   # G/D 3x3 same
    (3, 512, 512, 3, 1, 1, 8),    # 512-ch layers, batch not a tile multiple
    (2, 6, 12, 3, 1, 1, 16),      # noise conv 1 (6 -> 8 padded in, 12 out)
    (2, 12, 24, 3, 1, 1, 16),     # noise conv 2
    (2, 24, 256, 3, 1, 1, 16),    # noise conv 3
    (2, 9, 128, 1, 1, 0, 32),     # D first layer (9 -> 12 padded)
    (2, 128, 3, 1, 1, 0, 32),     # ToRGB-shaped 1x1
    (2, 128, 256, 3, 2, 0, 33),   # D conv2: stride 2 on the blurred (H+1) map
    (2, 128, 256, 1, 2, 0, 31),   # D skip: 1x1 stride 2 on the (H-1) map
    (1, 513, 512, 3, 1, 1, 4),    # final_conv (513 -> 516 padded)
    (5, 64, 160, 3, 1, 1, 7),     # ragged everything
]


def _conv_case(case, seed=0):
    B, Ci, Co, K, s, p, H = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Ci, H, H, generator=g)
    w = torch.randn(Co, Ci, K, K, generator=g) / (Ci * K * K) ** 0.5
    return x, w


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd(case):
    from gif_amd import ops
    B, Ci, Co, K, s, p, H = case
    x, w = _conv_case(case)
    ref = F.conv2d(x, w, stride=s, padding=p)
    got = ops.conv_fwd(dev(x), w.cuda(), ops.ConvSpec(K, K, s, p))
    assert got.shape[1] == pad4(Co)
    assert_close(host(got, Co), ref, TOL, f"conv_fwd {case}")
    if pad4(Co) != Co:
        assert (host(got)[:, Co:] == 0).all(), "padded output channels must be zero"


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_bwd_data(case):
    from gif_amd import ops
    B, Ci, Co, K, s, p, H = case
    x, w = _conv_case(case)
    Hs = (H + 2 * p - K) // s + 1
    gy = torch.randn(B, Co, Hs, Hs, generator=torch.Generator().manual_seed(1))
    ref = F.conv_transpose2d(gy, w, stride=s, padding=p, output_padding=H - ((Hs - 1) * s + K - 2 * p))
    got = ops.conv_bwd_data(dev(gy), w.cuda(), ops.ConvSpec(K, K, s, p), (H, H))
    assert_close(host(got, Ci), ref, TOL, f"conv_bwd_data {case}")


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_wgrad(case):
    from gif_amd import ops
    B, Ci, Co, K, s, p, H = case
    x, w = _conv_case(case)
    w = w.requires_grad_(True)
    y = F.conv2d(x, w, stride=s, padding=p)
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(2))
    (ref,) = torch.autograd.grad(y, w, gy)
    got = ops.conv_wgrad(dev(gy), dev(x), ops.ConvSpec(K, K, s, p), Co, Ci)
    assert_close(got, ref, 5e-5, f"conv_wgrad {case}")


def test_conv_scales_and_epilogue():
    """modulation (in_scale), demodulation (out_scale), residual, bias, leaky-ReLU fused in the conv kernel."""
    from gif_amd import ops
    g = torch.Generator().manual_seed(3)
    B, Ci, Co, H = 3, 128, 256, 16
    x, w = torch.randn(B, Ci, H, H, generator=g), torch.randn(Co, Ci, 3, 3, generator=g) / 34
    s, d = torch.rand(B, Ci, generator=g) + 0.5, torch.rand(B, Co, generator=g) + 0.5
    res, bias = torch.randn(B, Co, H, H, generator=g), torch.randn(Co, generator=g)
    ref = F.conv2d(x * s[:, :, None, None], w, padding=1) * d[:, :, None, None]
    got = ops.conv_fwd(dev(x), w.cuda(), ops.ConvSpec(3, 3, 1, 1), in_scale=s.cuda(), out_scale=d.cuda())
    assert_close(host(got), ref, TOL, "scaled conv")
    ref2 = 2 ** 0.5 * F.leaky_relu(ref + res + bias[None, :, None, None], 0.2)
    got2 = ops.conv_fwd(dev(x), w.cuda(), ops.ConvSpec(3, 3, 1, 1), in_scale=s.cuda(), out_scale=d.cuda(),
                        bias=bias.cuda(), residual=dev(res), act=True)
    assert_close(host(got2), ref2, TOL, "fused epilogue")
    # transposed stride-2 with scales: the generator's up-sampling branch
    wt = torch.randn(Ci, Co, 3, 3, generator=g) / 34  # canonical [O=Ci(small side), I=Co]
    ref3 = F.conv_transpose2d(x * s[:, :, None, None], wt, stride=2) * d[:, :, None, None]
    got3 = ops.conv_bwd_data(dev(x), wt.cuda(), ops.ConvSpec(3, 3, 2, 0), (2 * H + 1, 2 * H + 1), in_scale=s.cuda(),
                             out_scale=d.cuda())
    assert_close(host(got3), ref3, TOL, "scaled transposed conv")


WINO_CASES = [
    # (B, Cin, Cout, H, W)
    (2, 128, 128, 32, 32),   # full tiles
    (3, 512, 512, 8, 8),     # 48 2x2 tiles: a partial 128-tile block
    (1, 513, 512, 4, 4),     # final_conv: 516 padded input channels (K not a multiple of 32)
    (5, 64, 36, 6, 10),      # non-square, Cout < one 64-wide block, ragged tile count
    (2, 36, 160, 12, 4),     # Cin barely above one K block
]


@pytest.mark.parametrize("case", WINO_CASES)
def test_winograd_conv_fwd_and_bwd_data(case):
    """Winograd F(2x2,3x3) kernels vs the direct convolution (ATen on the host) — forward and data gradient."""
    from gif_amd import ops
    B, Ci, Co, H, W = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)
    ref = F.conv2d(x, w, padding=1)
    got = ops.conv3x3_winograd(dev(x), w.cuda(), True, pad4(Co))
    assert_close(host(got, Co), ref, 3e-5, f"winograd fwd {case}")
    if pad4(Co) != Co:
        assert (host(got)[:, Co:] == 0).all()
    gy = torch.randn(B, Co, H, W, generator=g)
    ref_b = F.conv_transpose2d(gy, w, padding=1)
    got_b = ops.conv3x3_winograd(dev(gy), w.cuda(), False, pad4(Ci))
    assert_close(host(got_b, Ci), ref_b, 3e-5, f"winograd bwd_data {case}")
    # non-contiguous canonical view (the transposed-weight view the modulated conv hands over)
    wt = w.transpose(0, 1).contiguous().transpose(0, 1)
    got_s = ops.conv3x3_winograd(dev(x), wt.cuda(), True, pad4(Co))
    assert torch.equal(got_s, got)


@pytest.mark.parametrize("case", WINO_CASES + [(4, 128, 256, 16, 16)])
def test_winograd_wgrad(case):
    """Winograd F(3x3,2x2) weight gradient vs autograd of the direct convolution, with and without modulation scales."""
    from gif_amd import ops
    B, Ci, Co, H, W = case
    g = torch.Generator().manual_seed(13)
    x, gy = torch.randn(B, Ci, H, W, generator=g), torch.randn(B, Co, H, W, generator=g)
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(x, w, padding=1), w, gy)
    got = ops.conv3x3_winograd_wgrad(dev(gy), dev(x), Co, Ci)
    assert got.shape == (Co, Ci, 3, 3)
    assert_close(got, ref, 5e-5, f"winograd wgrad {case}")
    s, d = torch.rand(B, Ci, generator=g) + 0.5, torch.rand(B, Co, generator=g) + 0.5
    (ref2,) = torch.autograd.grad(F.conv2d(x * s[:, :, None, None], w, padding=1) * d[:, :, None, None], w, gy)
    sp = F.pad(s, (0, pad4(Ci) - Ci)).cuda()
    dp = F.pad(d, (0, pad4(Co) - Co)).cuda()
    got2 = ops.conv3x3_winograd_wgrad(dev(gy), dev(x), Co, Ci, 0.25, small_scale=dp, big_scale=sp)
    assert_close(got2, 0.25 * ref2, 5e-5, f"winograd modulated wgrad {case}")
    again = ops.conv3x3_winograd_wgrad(dev(gy), dev(x), Co, Ci, 0.25, small_scale=dp, big_scale=sp)
    assert torch.equal(again, got2), "split-K reduction must be deterministic"
    # the forward pass's transformed input (same x, same modulation) can stand in for the input transform
    _, v = ops.conv3x3_winograd(dev(x), w.detach().cuda(), True, pad4(Co), keep_v=True, in_scale=sp)
    reuse = ops.conv3x3_winograd_wgrad(dev(gy), dev(x), Co, Ci, 0.25, small_scale=dp, big_scale=sp, big_v=v)
    assert torch.equal(reuse, got2), "wgrad from the kept V must equal the recomputed one bit for bit"


def test_winograd_scales_and_epilogue(monkeypatch):
    from gif_amd import ops
    g = torch.Generator().manual_seed(12)
    B, Ci, Co, H = 3, 128, 256, 16
    x, w = torch.randn(B, Ci, H, H, generator=g), torch.randn(Co, Ci, 3, 3, generator=g) / 34
    s, d = torch.rand(B, Ci, generator=g) + 0.5, torch.rand(B, Co, generator=g) + 0.5
    res, bias = torch.randn(B, Co, H, H, generator=g), torch.randn(Co, generator=g)
    ref = 2 ** 0.5 * F.leaky_relu(F.conv2d(x * s[:, :, None, None], w * 0.7, padding=1) * d[:, :, None, None] + res
                                  + bias[None, :, None, None], 0.2)
    got = ops.conv3x3_winograd(dev(x), w.cuda(), True, Co, 0.7, in_scale=s.cuda(), out_scale=d.cuda(), bias=bias.cuda(),
                               residual=dev(res), act=True)
    assert_close(host(got), ref, 3e-5, "winograd fused epilogue")
    # dispatch: conv_fwd / conv_bwd_data route eligible shapes to Winograd once the tile threshold allows it
    monkeypatch.setattr(ops, "WINOGRAD_MIN_TILES", 0)
    monkeypatch.setattr(ops, "WINOGRAD_MIN_C", 0)
    monkeypatch.setattr(ops, "WINOGRAD_WGRAD_MIN_C", 0)
    assert ops.winograd_eligible(ops.ConvSpec(3, 3, 1, 1), B, H, H, Ci)
    assert not ops.winograd_eligible(ops.ConvSpec(3, 3, 2, 0), B, H, H, Ci)
    assert not ops.winograd_eligible(ops.ConvSpec(3, 3, 1, 1), B, 7, 7, Ci)
    assert not ops.winograd_eligible(ops.ConvSpec(3, 3, 1, 1), B, H, H, 24)
    got2 = ops.conv_fwd(dev(x), w.cuda(), ops.ConvSpec(3, 3, 1, 1), 0.7, in_scale=s.cuda(), out_scale=d.cuda(),
                        bias=bias.cuda(), residual=dev(res), act=True)
    assert torch.equal(got2, got)
    gy = torch.randn(B, Co, H, H, generator=g)
    ref3 = F.conv_transpose2d(gy * d[:, :, None, None], w, padding=1) * s[:, :, None, None]
    got3 = ops.conv_bwd_data(dev(gy), w.cuda(), ops.ConvSpec(3, 3, 1, 1), (H, H), in_scale=d.cuda(), out_scale=s.cuda())
    assert_close(host(got3), ref3, 3e-5, "winograd dgrad with scales")


def test_conv_wgrad_with_scales():
    from gif_amd import ops
    g = torch.Generator().manual_seed(4)
    B, Ci, Co, H = 4, 128, 128, 16
    x, gy = torch.randn(B, Ci, H, H, generator=g), torch.randn(B, Co, H, H, generator=g)
    s, d = torch.rand(B, Ci, generator=g) + 0.5, torch.rand(B, Co, generator=g) + 0.5
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    y = F.conv2d(x * s[:, :, None, None], w, padding=1) * d[:, :, None, None]
    (ref,) = torch.autograd.grad(y, w, gy)
    got = ops.conv_wgrad(dev(gy), dev(x), ops.ConvSpec(3, 3, 1, 1), Co, Ci, 0.5, small_scale=d.cuda(), big_scale=s.cuda())
    assert_close(got, 0.5 * ref, 5e-5, "scaled wgrad")


# ------------------------------------------------------------------------------------------------ FIR / pointwise
@pytest.mark.parametrize("cfg", [(1, 1, (2, 2), 16), (1, 1, (1, 1), 17), (2, 1, (2, 1), 8), (1, 2, (1, 1), 16),
                                 (1, 1, (-1, 2), 9), (2, 2, (3, 0), 8), (1, 1, (1, 1), 33), (1, 2, (1, 1), 17), (2, 1, (1, 1), 9),
                                 (1, 2, (2, 2), 31), (2, 1, (3, 1), 12)])
def test_upfirdn2d_fwd_bwd(cfg):
    from gif_amd import functional as GF
    from oracle import stylegan2_ref as R
    up, down, pad, H = cfg
    g = torch.Generator().manual_seed(5)
    k = R.make_kernel([1, 3, 3, 1]) * up ** 2
    x = torch.randn(2, 12, H, H + 3, generator=g).requires_grad_(True)
    ref = R.upfirdn2d(x, k, up, down, pad)
    xd = dev(x, True)
    got = GF.upfirdn2d(xd, k.cuda(), up, down, pad)
    assert_close(host(got), ref, 1e-6, f"upfirdn2d {cfg}")
    gy = torch.randn(ref.shape, generator=g)
    (gref,) = torch.autograd.grad(ref, x, gy)
    (ggot,) = torch.autograd.grad(got, xd, dev(gy))
    assert_close(host(ggot), gref, 1e-6, f"upfirdn2d backward {cfg}")


def test_bias_act_and_reductions():
    from gif_amd import functional as GF, ops
    g = torch.Generator().manual_seed(6)
    x = torch.randn(3, 36, 9, 11, generator=g).requires_grad_(True)
    b = torch.randn(36, generator=g).requires_grad_(True)
    r = torch.randn(3, 36, 9, 11, generator=g).requires_grad_(True)
    ref = 2 ** 0.5 * F.leaky_relu(x + r + b[None, :, None, None], 0.2)
    xd, bd, rd = dev(x, True), b.detach().cuda().requires_grad_(True), dev(r, True)
    got = GF.bias_act(xd, bd, rd)
    assert_close(host(got), ref, 1e-6, "bias_act")
    gy = torch.randn(ref.shape, generator=g)
    gref = torch.autograd.grad(ref, (x, b, r), gy)
    ggot = torch.autograd.grad(got, (xd, bd, rd), dev(gy))
    for a, e, n in zip(ggot, gref, "x b r".split()):
        assert_close(a.detach().cpu(), e, 2e-6, f"bias_act grad {n}")
    assert_close(ops.colsum(dev(x)).cpu(), x.detach().sum(dim=(0, 2, 3)), 2e-6, "colsum")
    out, scaled = ops.mul_reduce(dev(x), dev(r), scale=torch.ones(3, 36).cuda() * 2, want_scaled=True)
    assert_close(out.cpu(), (x * r).detach().sum(dim=(2, 3)), 5e-6, "mul_reduce")
    assert_close(host(scaled), 2 * x.detach(), 1e-7, "mul_reduce scaled")
    big = torch.randn(2, 128, 64, 64, generator=g)
    assert_close(ops.colsum(dev(big)).cpu(), big.sum(dim=(0, 2, 3)), 1e-5, "colsum big")
    assert_close(ops.sqnorm_per_sample(dev(big)).cpu(), big.pow(2).sum(dim=(1, 2, 3)), 1e-5, "sqnorm")


@pytest.mark.parametrize("B", [4, 8, 2, 32])
def test_minibatch_stddev(B):
    from gif_amd import functional as GF
    from oracle import stylegan2_ref as R
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, 512, 4, 4, generator=g).requires_grad_(True)
    ref = R.minibatch_stddev(x)
    xd = dev(x, True)
    got = GF.minibatch_stddev(xd, min(B, 4), 516)
    assert_close(host(got, 513), ref, 1e-6, "mbstd fwd")
    assert (host(got)[:, 513:] == 0).all()
    gy = torch.randn(B, 513, 4, 4, generator=g)
    (gref,) = torch.autograd.grad(ref, x, gy, create_graph=True)
    (ggot,) = torch.autograd.grad(got, xd, dev(gy), create_graph=True)
    assert_close(host(ggot), gref, 1e-5, "mbstd bwd")
    # second order (R1 path): d/dx of <grad, v>
    v = torch.randn(B, 512, 4, 4, generator=g)
    (g2ref,) = torch.autograd.grad((gref * v).sum(), x)
    (g2got,) = torch.autograd.grad((ggot * dev(v)).sum(), xd)
    assert_close(host(g2got), g2ref, 1e-4, "mbstd double backward")


# ------------------------------------------------------------------------------------------------ autograd wiring
def test_conv_autograd_first_and_second_order():
    """GF.conv2d vs torch autograd: grads w.r.t. x and w, and the R1-style double backward."""
    from gif_amd import functional as GF
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 16, 12, 12, generator=g).requires_grad_(True)
    w = (torch.randn(32, 16, 3, 3, generator=g) / 12).requires_grad_(True)
    xd, wd = dev(x, True), w.detach().cuda().requires_grad_(True)
    for stride, pad in ((1, 1), (2, 0)):
        ref = F.conv2d(x, w * 0.7, stride=stride, padding=pad)
        got = GF.conv2d(xd, wd, stride, pad, wscale=0.7)
        assert_close(host(got), ref, TOL, "conv2d")
        # R1-style: penalty = |d sum(y^2)/dx|^2, then gradient of the penalty w.r.t. w and x
        (gx_ref,) = torch.autograd.grad((ref ** 2).sum(), x, create_graph=True)
        (gx_got,) = torch.autograd.grad((got ** 2).sum(), xd, create_graph=True)
        assert_close(host(gx_got), gx_ref, 5e-5, "first-order grad")
        pr, pg = gx_ref.pow(2).sum(), gx_got.pow(2).sum()
        r2 = torch.autograd.grad(pr, (x, w))
        g2 = torch.autograd.grad(pg, (xd, wd))
        assert_close(host(g2[0]), r2[0], 2e-4, "second-order grad x")
        assert_close(g2[1].cpu(), r2[1], 2e-4, "second-order grad w")


def test_modulated_conv_autograd():
    from gif_amd import layers as L
    from oracle import stylegan2_ref as R
    torch.manual_seed(9)
    for upsample in (False, True):
        m = L.ModulatedConv2d(64, 128, 3, 512, upsample=upsample).cuda()
        sd = {k: v.detach().cpu().clone().requires_grad_(not k.endswith("kernel")) for k, v in m.state_dict().items()}
        x = torch.randn(3, 64, 8, 8).requires_grad_(True)
        st = torch.randn(3, 512).requires_grad_(True)
        ref = R.modulated_conv2d(x, sd["weight"], sd["modulation.weight"], sd["modulation.bias"], st, True, upsample,
                                 sd.get("blur.kernel"))
        xd, std = dev(x, True), st.detach().cuda().requires_grad_(True)
        got = m(xd, std)
        assert_close(host(got), ref, 3e-5, f"modconv up={upsample}")
        gy = torch.randn(ref.shape)
        gref = torch.autograd.grad(ref, (x, st, sd["weight"], sd["modulation.weight"], sd["modulation.bias"]), gy)
        ggot = torch.autograd.grad(got, (xd, std, m.weight, m.modulation.weight, m.modulation.bias), dev(gy))
        for a, e, n in zip(ggot, gref, ["x", "style", "weight", "mod.w", "mod.b"]):
            a = host(a) if a.dim() == 4 else a.detach().cpu()
            assert_close(a, e, 1e-4, f"modconv up={upsample} grad {n}")


# ------------------------------------------------------------------------------------------------ fused epilogues
def test_conv_bias_act_fused_any_order():
    """ConvLayer = conv + bias + lrelu in ONE kernel; gradients up to second order (R1 path) vs torch autograd."""
    from gif_amd import functional as GF
    g = torch.Generator().manual_seed(10)
    x = torch.randn(2, 16, 12, 12, generator=g).requires_grad_(True)
    w = (torch.randn(32, 16, 3, 3, generator=g) / 12).requires_grad_(True)
    b = (0.3 * torch.randn(32, generator=g)).requires_grad_(True)
    xd, wd, bd = dev(x, True), w.detach().cuda().requires_grad_(True), b.detach().cuda().requires_grad_(True)
    for stride, pad in ((1, 1), (2, 0)):
        ref = 2 ** 0.5 * F.leaky_relu(F.conv2d(x, w * 0.7, stride=stride, padding=pad) + b[None, :, None, None], 0.2)
        got = GF.conv2d_bias_act(xd, wd, bd, stride, pad, wscale=0.7)
        assert_close(host(got), ref, TOL, "fused conv+bias+act")
        gref = torch.autograd.grad((ref ** 2).sum(), (x, w, b), create_graph=True)
        ggot = torch.autograd.grad((got ** 2).sum(), (xd, wd, bd), create_graph=True)
        for a, e, n in zip(ggot, gref, "x w b".split()):
            assert_close(host(a) if a.dim() == 4 and n == "x" else a.detach().cpu(), e, 1e-4, f"fused conv grad {n}")
        r2 = torch.autograd.grad(gref[0].pow(2).sum(), (x, w, b))
        g2 = torch.autograd.grad(ggot[0].pow(2).sum(), (xd, wd, bd))
        assert_close(host(g2[0]), r2[0], 3e-4, "fused conv second-order x")
        assert_close(g2[1].cpu(), r2[1], 3e-4, "fused conv second-order w")
        assert_close(g2[2].cpu(), r2[2], 3e-4, "fused conv second-order b")


@pytest.mark.parametrize("upsample", [False, True])
def test_styled_conv_fused_vs_oracle(upsample):
    """StyledConv (modconv + condition-noise + bias + lrelu, fused epilogues) forward and all gradients vs the oracle."""
    from gif_amd import layers as L
    from oracle import stylegan2_ref as R
    torch.manual_seed(11)
    m = L.StyledConv(64, 128, 3, noise_in_dims=6, upsample=upsample).cuda()
    with torch.no_grad():
        m.activate.bias.normal_(0, 0.2)
        for i in (0, 2, 4):
            m.noise.noise_conv[i].weight.mul_(20)
    keys = list(m.state_dict().keys())
    sd = {k: v.detach().cpu().clone().requires_grad_(not k.endswith("kernel")) for k, v in m.state_dict().items()}
    H = 8
    Ho = 2 * H if upsample else H
    x = torch.randn(3, 64, H, H).requires_grad_(True)
    st = torch.randn(3, 512).requires_grad_(True)
    cond = torch.rand(3, 6, Ho, Ho) * 2 - 1
    ref = R.styled_conv(sd, "", x, st, cond, upsample)
    xd, std = dev(x, True), st.detach().cuda().requires_grad_(True)
    got = m(xd, std, dev(cond))
    assert_close(host(got), ref, 3e-5, f"StyledConv up={upsample}")
    gy = torch.randn(ref.shape)
    names = [k for k in keys if not k.endswith("kernel")]
    gref = torch.autograd.grad(ref, [x, st] + [sd[k] for k in names], gy)
    params = dict(m.named_parameters())
    ggot = torch.autograd.grad(got, [xd, std] + [params[k] for k in names], dev(gy))
    for a, e, n in zip(ggot, gref, ["x", "style"] + names):
        a = host(a) if (a.dim() == 4 and n == "x") else a.detach().cpu()
        assert_close(a, e, 2e-4, f"StyledConv up={upsample} grad {n}")


# ------------------------------------------------------------------------------------------------ condition render
def test_vertex_normals_orth_proj_and_condition_render():
    from gif_amd import render
    from oracle import mesh_ref as MR
    from oracle import rasterize_oracle as ro
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    g = np.load(os.path.join(gdir, "mesh_golden.npz"))
    faces = np.load(os.path.join(gdir, "body_mesh.npz"))["faces"]
    v, f = torch.from_numpy(g["vertices"]).cuda(), torch.from_numpy(faces).cuda()
    n = render.vertex_normals(v, f)
    assert np.abs(n.cpu().numpy() - g["normals"]).max() < 5e-6, "vertex normals vs reference golden"
    n2 = render.vertex_normals(v, f)
    assert torch.equal(n, n2), "gather formulation is deterministic"
    p = render.batch_orth_proj(v, torch.from_numpy(g["cam"]).cuda())
    assert np.array_equal(p.cpu().numpy(), g["proj"])
    # full condition render at 256x256 vs the oracle pipeline (numpy normals + C rasteriser)
    tex = (v - v.amin(dim=1, keepdim=True)) / (v.amax(dim=1, keepdim=True) - v.amin(dim=1, keepdim=True))
    cond = render.render_condition(p * 0.9, f, tex, 256, 256)
    assert cond.shape == (2, 6, 256, 256) and cond.min() >= -1 and cond.max() <= 1
    pn = (p * 0.9).cpu().numpy()
    nn = MR.vertex_normals(pn, faces) * np.float32(0.5) + np.float32(0.5)
    vi = ro.to_image_space(pn, 256, 256)
    f2 = np.repeat(faces[None], 2, 0)
    fv = ro.face_vertices(vi, f2)
    d0, t0, img = ro.new_buffers(2, 256, 256)
    ro.standard_rasterize_colors(fv, ro.face_vertices(nn.astype(np.float32), f2), d0, t0, img, 256, 256)
    ref_n = np.floor(np.clip(img, 0, 1) * 255) / 255.0 * 2 - 1
    got_n = cond[:, 3:].permute(0, 2, 3, 1).cpu().numpy()
    # 8-bit quantisation: identical up to the rare pixel whose value sits on a quantisation edge
    assert (np.abs(got_n - ref_n) > 1e-6).mean() < 1e-3
    assert np.abs(got_n - ref_n).max() <= 2 / 255 + 1e-6


@pytest.mark.parametrize("R,S", [(256, 256), (256, 128), (256, 64), (256, 4), (64, 8), (32, 16)])
def test_condition_pyramid_level(R, S):
    """HIP pyramid level == F.interpolate(bilinear, align_corners=False) for the model's integer ratios, fwd + bwd."""
    from gif_amd import functional as GF
    g = torch.Generator().manual_seed(12)
    B = 2 if R == 256 else 3
    x = (torch.rand(B, 8, R, R, generator=g) * 2 - 1).requires_grad_(True)
    ref = F.interpolate(x, size=(S, S), mode="bilinear", align_corners=False)
    xd = dev(x, True)
    got = GF.bilinear_down(xd, S)
    assert_close(host(got), ref, 1e-6, f"pyramid {R}->{S}")
    gy = torch.randn(ref.shape, generator=g)
    (gref,) = torch.autograd.grad(ref, x, gy)
    (ggot,) = torch.autograd.grad(got, xd, dev(gy))
    assert_close(host(ggot), gref, 1e-6, f"pyramid backward {R}->{S}")


def test_texture_map_vs_reference_golden():
    """FlameTextureSpace.compute_texture_map (HIP) vs the real reference method's output on the synthetic UV fixture."""
    from gif_amd.texture_space import FlameTextureSpace
    from test_oracle_texture import load_fixture
    g, td, verts, normals = load_fixture()
    fts = FlameTextureSpace(td, None).cuda()
    img = torch.from_numpy(g["img"]).cuda().requires_grad_(True)
    tex, mask = fts.compute_texture_map(img, verts.cuda(), normals.cuda(), camera_params=torch.from_numpy(g["cam"]).cuda())
    assert tex.shape == (2, 3, 256, 256) and mask.shape == (2, 1, 256, 256)
    assert np.abs(tex.detach().cpu().numpy() - g["tex"]).max() < 2e-6, "texture image"
    assert mask.dtype == torch.bool, "the reference returns a bool visibility mask (stg2_generator.py:415)"
    assert np.array_equal(mask.cpu().numpy(), g["mask"]), "visibility mask"
    (tex * torch.linspace(-1, 1, tex.numel(), device="cuda").view_as(tex)).sum().backward()
    gi = img.grad.cpu().numpy()
    assert np.abs(gi - g["grad_img"]).max() <= 2e-4 * np.abs(g["grad_img"]).max(), "gradient w.r.t. the source image"
    with pytest.raises(Exception, match="FLAME"):
        fts(img, torch.zeros(2, 159, device="cuda"))


# ------------------------------------------------------------------------------------------------ input pipeline (8(f).3)
def test_fast_image_reshape_vs_reference_goldens_and_oracle():
    """HIP resize behind the reference's fast_image_reshape signature: goldens of the real reference, the oracle on larger
    seeded inputs, first- and second-order gradients vs ATen on the host."""
    from golden.make_resize_golden import CASES
    from gif_amd.data import fast_image_reshape
    from oracle import resize_ref as R
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "resize_golden.npz"))
    for name, B, C, H, W, ho, wo, mode, clamp in CASES:
        got = fast_image_reshape(torch.from_numpy(g[name + "_x"]).cuda(), ho, wo, non_diff_allowed=clamp, mode=mode)
        assert got.shape == (B, C, wo, ho)
        assert_close(got, torch.from_numpy(g[name + "_y"]), 2e-6, f"resize golden {name}")
    x = torch.rand(4, 3, 512, 512, generator=torch.Generator().manual_seed(7)) * 2 - 1
    for mode, size in (("bicubic", 256), ("bilinear", 256), ("bicubic", 300)):
        ref = R.fast_image_reshape(x, size, size, mode=mode)
        assert_close(fast_image_reshape(x.cuda(), size, size, mode=mode), ref, 2e-6, f"resize {mode} 512->{size}")
    xs = torch.rand(2, 3, 12, 9, generator=torch.Generator().manual_seed(8))
    wgt = torch.randn(2, 3, 20, 15, generator=torch.Generator().manual_seed(9))
    for mode in ("bicubic", "bilinear"):
        xr = xs.clone().requires_grad_(True)
        (F.interpolate(xr, size=(20, 15), mode=mode) * wgt).sum().backward()
        xg = xs.cuda().requires_grad_(True)
        y = fast_image_reshape(xg, 15, 20, mode=mode)  # (height_out, width_out) -> rows = width_out, like the reference
        (gx,) = torch.autograd.grad((y * wgt.cuda()).sum(), xg, create_graph=True)
        assert_close(gx, xr.grad, 5e-6, f"resize backward {mode}")
        # the map is linear: d/d(wgt-like cotangent) of <gx, v> is resize(v)
        v = torch.randn_like(xs).cuda()
        yv = fast_image_reshape(v, 15, 20, mode=mode)
        gy = torch.ones_like(y, requires_grad=True)
        (gx2,) = torch.autograd.grad(fast_image_reshape(xg, 15, 20, mode=mode), xg, gy, create_graph=True)
        (ggy,) = torch.autograd.grad((gx2 * v).sum(), gy)
        assert_close(ggy, yv, 5e-6, f"resize double backward {mode}")
    with pytest.raises(Exception):
        fast_image_reshape(xs, 4, 4)  # CPU tensor: no fallback


def test_synthetic_batches_are_device_resident_and_rank_distinct():
    from gif_amd.data import SyntheticBatches
    a = SyntheticBatches(4, 64, 1000, torch.device("cuda"), rank=0)
    b = SyntheticBatches(4, 64, 1000, torch.device("cuda"), rank=1)
    real, cond, idx = next(a)
    assert real.shape == (4, 3, 64, 64) and cond.shape == (4, 6, 64, 64) and idx.shape == (4,) and idx.dtype == torch.int64
    assert real.is_cuda and cond.is_cuda and idx.is_cuda
    assert real.min() >= -1 and real.max() <= 1 and int(idx.max()) < 1000
    assert not torch.equal(real, next(b)[0]), "ranks must draw different data"
    a2 = SyntheticBatches(4, 64, 1000, torch.device("cuda"), rank=0)
    assert torch.equal(real, next(a2)[0]), "same seed and rank: reproducible"


def test_texture_interpolation_loss_vs_reference_goldens_and_oracle_grads():
    """InterpolatedTextureLoss core on the fused HIP kernel: values produced by the REAL reference methods (pairwise loss and
    the whole tex_sp_intrp_loss loop, same-size and resized face mask) and gradients vs the oracle's autograd."""
    from gif_amd.losses import InterpolatedTextureLoss
    from oracle import texture_loss_ref as R
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "texture_loss_golden.npz"))
    tex_h, msk_h = torch.from_numpy(g["textures"]), torch.from_numpy(g["tx_masks"])
    for tag in ("same", "big"):
        face_h = torch.from_numpy(g[f"face_{tag}"])
        crit = InterpolatedTextureLoss(6, face_h.cuda())
        assert len(crit.pairs) == 10 and crit.max_num == 5
        tex = tex_h.cuda().requires_grad_(True)
        pw = crit.pairwise_texture_loss(tex[0], tex[1])
        assert abs(pw.item() - float(g[f"pair01_{tag}"])) < 2e-6 * abs(float(g[f"pair01_{tag}"])) + 1e-7
        pairs = g[f"loop_pairs_{tag}"]
        loss = crit.texture_pairs_loss(tex, msk_h.cuda(), pairs)
        assert abs(loss.item() - float(g[f"loop_{tag}"])) < 3e-6 * abs(float(g[f"loop_{tag}"]))
        loss.backward()
        tex_r = tex_h.clone().requires_grad_(True)
        R.texture_pairs_loss(face_h, tex_r, msk_h, pairs).backward()
        assert_close(tex.grad, tex_r.grad, 5e-6, f"texture loss grad ({tag})")
        np.random.seed(3)
        drawn = crit.texture_pairs_loss(tex.detach(), msk_h.cuda())  # pairs drawn like the reference
        np.random.seed(3)
        ref_pairs = crit.pairs[np.random.choice(len(crit.pairs), crit.max_num, replace=False)]
        assert abs(drawn.item() - R.texture_pairs_loss(face_h, tex_h, msk_h, ref_pairs).item()) < 1e-5
    with pytest.raises(Exception, match="FLAME"):
        crit.tex_sp_intrp_loss(torch.zeros(6, 236, device="cuda"), None, 6, 1.0, 10)


def test_conv_launch_split_on_tile_quantisation():
    """Launches whose 128-row tile count is a little above a multiple of the 512 resident workgroups are split into a bulk
    launch (128x128 tiles) and a tail launch (64x64 tiles) over disjoint row ranges: forward conv, the four phases of a
    transposed stride-2 conv, and a modulated conv with a fused epilogue must all still match ATen."""
    from gif_amd import ops
    g = torch.Generator().manual_seed(21)
    # forward: M = 257*259 = 66563 rows = 520.02 tiles of 128
    x = torch.randn(1, 128, 257, 259, generator=g)
    w = torch.randn(128, 128, 3, 3, generator=g) / 34
    s_, d_ = torch.rand(1, 128, generator=g) + 0.5, torch.rand(1, 128, generator=g) + 0.5
    bias = torch.randn(128, generator=g)
    ops_w = ops.WINOGRAD
    ops.WINOGRAD = False  # this test is about the direct kernels
    try:
        ref = F.conv2d(x, w, padding=1)
        assert_close(host(ops.conv_fwd(dev(x), w.cuda(), ops.ConvSpec(3, 3, 1, 1))), ref, TOL, "split fwd")
        ref2 = 2 ** 0.5 * F.leaky_relu(F.conv2d(x * s_[:, :, None, None], w, padding=1) * d_[:, :, None, None]
                                       + bias[None, :, None, None], 0.2)
        got2 = ops.conv_fwd(dev(x), w.cuda(), ops.ConvSpec(3, 3, 1, 1), in_scale=s_.cuda(), out_scale=d_.cuda(),
                            bias=bias.cuda(), act=True)
        assert_close(host(got2), ref2, TOL, "split modulated fwd + epilogue")
        # transposed stride 2: phases of 182x182 / 182x181 / ... pixels x batch 2 = ~518 tiles each
        xs = torch.randn(2, 128, 181, 181, generator=g)
        wt = torch.randn(128, 128, 3, 3, generator=g) / 34
        ref3 = F.conv_transpose2d(xs, wt, stride=2)
        got3 = ops.conv_bwd_data(dev(xs), wt.cuda(), ops.ConvSpec(3, 3, 2, 0), (363, 363))
        assert_close(host(got3), ref3, TOL, "split transposed conv")
    finally:
        ops.WINOGRAD = ops_w


def test_conv_bias_act_passthrough_accumulates_in_the_dgrad_kernel():
    """ConvBiasActFn(passthrough=True) hands back an alias of x for a second consumer; that consumer's gradient must come out
    ADDED to the conv's data gradient (fused as the dgrad kernel's residual), to first and second order."""
    from gif_amd import functional as GF
    g = torch.Generator().manual_seed(31)
    x0 = torch.randn(2, 64, 12, 12, generator=g)
    w0 = torch.randn(64, 64, 3, 3, generator=g) / 24
    b0 = torch.randn(64, generator=g) * 0.1
    m = torch.randn(2, 64, 12, 12, generator=g)      # weights of the second consumer: z = sum(m * x^2)
    t = torch.randn(2, 64, 12, 12, generator=g)

    def run(passthrough):
        x = dev(x0).requires_grad_(True)
        w, b = w0.cuda().requires_grad_(True), b0.cuda().requires_grad_(True)
        if passthrough:
            y, xa = GF.conv2d_bias_act(x, w, b, 1, 1, 0.5, passthrough=True)
        else:
            y, xa = GF.conv2d_bias_act(x, w, b, 1, 1, 0.5), x
        loss = (y * dev(t)).sum() + (dev(m) * xa * xa).sum()
        (gx,) = torch.autograd.grad(loss, x, create_graph=True)
        gw, gb, ggx = torch.autograd.grad((gx * gx).sum(), [w, b, x], allow_unused=True)
        return gx.detach(), gw, gb, ggx

    a, bb = run(True), run(False)
    for got, ref, what in zip(a, bb, ("gx", "d|gx|^2/dw", "d|gx|^2/db", "d|gx|^2/dx")):
        if ref is None:
            assert got is None or float(got.abs().max()) == 0.0, what
        else:
            assert_close(got, ref, 1e-5, f"passthrough {what}")
    # reference value of gx on the host
    xr = x0.clone().requires_grad_(True)
    yr = 2 ** 0.5 * F.leaky_relu(F.conv2d(xr, w0 * 0.5, padding=1) + b0[None, :, None, None], 0.2)
    ((yr * t).sum() + (m * xr * xr).sum()).backward()
    assert_close(host(a[0]), xr.grad, TOL, "passthrough gx vs ATen")
    # the alias alone (y unused) and y alone (alias unused) still differentiate
    x = dev(x0).requires_grad_(True)
    y, xa = GF.conv2d_bias_act(x, w0.cuda(), b0.cuda(), 1, 1, 0.5, passthrough=True)
    (g1,) = torch.autograd.grad((xa * dev(m)).sum(), x, retain_graph=True)
    assert_close(host(g1), m, 1e-6, "alias-only gradient")
    (g2,) = torch.autograd.grad((y * dev(t)).sum(), x)
    assert float((g2 - (a[0] - 2 * dev(m) * dev(x0))).abs().max()) < 1e-4


@pytest.mark.parametrize("case", [(2, 6, 12, 192, 192), (1, 12, 24, 260, 254), (3, 16, 32, 150, 150), (2, 4, 8, 181, 183)])
def test_small_channel_wgrad_kernel(case):
    """conv_wgrad_small_mfma (all 9 taps in one wave on 16x16x4 MFMAs, no LDS staging) — the condition-noise convs' weight
    gradients — vs autograd of the direct convolution; ragged widths, channel counts below the MFMA tile, determinism."""
    from gif_amd import ops
    B, Ci, Co, H, W = case
    g = torch.Generator().manual_seed(41)
    x = torch.randn(B, Ci, H, W, generator=g)
    gy = torch.randn(B, Co, H, W, generator=g)
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(x, w, padding=1), w, gy)
    got = ops.conv_wgrad(dev(gy), dev(x), ops.ConvSpec(3, 3, 1, 1), Co, Ci, 0.5)
    assert_close(got, 0.5 * ref, 5e-5, f"small-channel wgrad {case}")
    assert torch.equal(got, ops.conv_wgrad(dev(gy), dev(x), ops.ConvSpec(3, 3, 1, 1), Co, Ci, 0.5))


@pytest.mark.parametrize("case", [
    ("winograd 128->128 @256^2", 128, 128, 3, 1, 1, 256),
    ("winograd 512->512 @64^2", 512, 512, 3, 1, 1, 64),
    ("stride-2 128->256 @257^2", 128, 256, 3, 2, 0, 257),
    ("1x1 12->128 @256^2", 12, 128, 1, 1, 0, 256),
])
def test_full_size_conv_linearity_and_adjointness(case):
    """BASELINE sizes (batch 32, the benchmark's layer shapes) have no CPU reference that finishes in seconds; what must hold
    at any size is checked instead: linearity of the forward op, and that forward, data gradient and weight gradient are
    adjoints of ONE bilinear map:  <conv(x, w), g> == <x, dgrad(g, w)> == <w, wgrad(g, x)>  (dot products in float64)."""
    from gif_amd import ops
    name, Ci, Co, K, s, p, H = case
    B = 32
    spec = ops.ConvSpec(K, K, s, p)
    gen = torch.Generator(device="cuda").manual_seed(51)
    x1 = torch.randn(B, Ci, H, H, device="cuda", generator=gen).contiguous(memory_format=torch.channels_last)
    x2 = torch.randn(B, Ci, H, H, device="cuda", generator=gen).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Co, Ci, K, K, device="cuda", generator=gen) / (K * Ci ** 0.5)
    y1 = ops.conv_fwd(x1, w, spec)
    y2 = ops.conv_fwd(x2, w, spec)
    y12 = ops.conv_fwd(2.0 * x1 - 3.0 * x2, w, spec)
    lin = 2.0 * y1 - 3.0 * y2
    assert ((y12 - lin).abs().max() / lin.abs().max()).item() < 2e-5, f"{name}: linearity"
    del y2, y12, lin, x2
    g = torch.randn(y1.shape, device="cuda", generator=gen).contiguous(memory_format=torch.channels_last)

    def dot(a, b):
        return (a.double() * b.double()).sum().item()

    lhs = dot(y1, g)
    gx = ops.conv_bwd_data(g, w, spec, (H, H))
    mid = dot(x1, gx)
    gw = ops.conv_wgrad(g, x1, spec, Co, Ci)
    rhs = dot(w, gw)
    scale = (y1.double().norm() * g.double().norm()).item()
    assert abs(lhs - mid) < 2e-6 * scale, f"{name}: <conv(x), g> vs <x, dgrad(g)>: {lhs} {mid}"
    assert abs(lhs - rhs) < 2e-6 * scale, f"{name}: <conv(x), g> vs <w, wgrad(g, x)>: {lhs} {rhs}"
    assert gx.shape == x1.shape and gw.shape == w.shape
    assert torch.equal(gw, ops.conv_wgrad(g, x1, spec, Co, Ci)), f"{name}: wgrad must be deterministic"


def test_rasterize_bit_exact_at_benchmark_batch():
    """Config 3 size: 32 poses of the bundled body mesh (V=6890, F=13776) at 256x256 — depth bits, face indices and
    barycentric bits of the HIP rasteriser against the C oracle, plus idempotence of a second launch."""
    from gif_amd import standard_rasterize as sr
    from oracle import rasterize_oracle as ro
    g, v, f = _body()
    rng = np.random.RandomState(7)
    vs = []
    for i in range(32):
        a, b = rng.uniform(-0.8, 0.8), rng.uniform(-0.3, 0.3)
        Ry = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
        Rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]], np.float32)
        vs.append((v[0] @ (Ry @ Rx).T * np.float32(rng.uniform(0.8, 1.2))).astype(np.float32))
    v32 = np.stack(vs)
    f32 = np.repeat(f, 32, 0)
    h = w = 256
    fv = ro.face_vertices(ro.to_image_space(v32, h, w), f32)
    d0, t0, b0 = ro.new_buffers(32, h, w)
    ro.standard_rasterize(fv, d0, t0, b0, h, w)
    fvt = torch.from_numpy(fv).cuda()
    d1, t1, b1 = sr.new_buffers(32, h, w, "cuda")
    sr.standard_rasterize(fvt, d1, t1, b1, h, w)
    assert np.array_equal(t1.cpu().numpy(), t0), "face indices"
    assert np.array_equal(d1.cpu().numpy().view(np.uint32), d0.view(np.uint32)), "depth bits"
    assert np.array_equal(b1.cpu().numpy().view(np.uint32), b0.view(np.uint32)), "barycentric bits"
    assert (t0 >= 0).mean() > 0.05
    sr.standard_rasterize(fvt, d1, t1, b1, h, w)
    assert np.array_equal(t1.cpu().numpy(), t0) and np.array_equal(d1.cpu().numpy().view(np.uint32), d0.view(np.uint32))
