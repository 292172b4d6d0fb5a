"""-m gpu: round-4 kernels and fixes.

* f16 halo convolution (conv_igemm.hip: conv_halo_f16) — the thin high-resolution layers of BASELINE configs[4]
  (stg2_generator.py:159-209 at step 7 / 8, stylegan2_common_layers.py:343-347): against an fp32 ATen-CPU computation on the
  same f16-rounded operands, against the gather kernel it replaces (GIF_F16_HALO=0), with every epilogue form.
* loss-scaler window semantics, FlatAdam checkpoint load after loss-scaled steps (advisor findings of round 3)."""
import os

import pytest
import torch
import torch.nn.functional as F

from gpu_util import assert_close, rel_err

pytestmark = pytest.mark.gpu

H16 = torch.float16
TOL = 2e-3


def pad8(c):
    return (c + 7) // 8 * 8


def dev16(x):
    c = x.shape[1]
    if pad8(c) != c:
        x = F.pad(x, (0, 0, 0, 0, 0, pad8(c) - c))
    return x.detach().cuda().to(H16).contiguous(memory_format=torch.channels_last)


def r16(x):
    return x.to(H16).float()


def host(y, c=None):
    y = y.detach().float()
    if c is not None:
        y = y[:, :c]
    return y.cpu().contiguous()


class halo:
    """gif_conv2d_f16_halo_enable(0) keeps the gather kernel (the round-3 path); the switch is restored to on."""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        from gif_amd import _lib
        _lib.load().gif_conv2d_f16_halo_enable(1 if self.on else 0)

    def __exit__(self, *a):
        from gif_amd import _lib
        _lib.load().gif_conv2d_f16_halo_enable(1)


HALO_CASES = [
    # (B, Cin, Cout, K, stride, pad, H): forward AND data gradient are taken, whichever of them is halo-eligible runs it
    (2, 32, 32, 3, 1, 1, 64),    # the 1024^2 block's layers (BN 32, CP 32)
    (2, 64, 64, 3, 1, 1, 48),    # the 512^2 block's layers (BN 64, CP 64), 3 x 3 patches
    (3, 64, 32, 3, 1, 1, 40),    # BN 32 / CP 64 forward, BN 64 / CP 32 data gradient; patches overhang (40 = 2.5 x 16)
    (2, 24, 32, 3, 1, 1, 32),    # condition-noise conv 3: 24 channels = 3 real chunks + 1 zero chunk per pixel
    (2, 12, 24, 3, 1, 1, 32),    # condition-noise conv 2 (16 / 24 padded channels)
    (2, 6, 12, 3, 1, 1, 32),     # condition-noise conv 1 (8 / 16 padded channels)
    (2, 32, 3, 1, 1, 0, 32),     # ToRGB at 32 channels: single tap, no halo; its data gradient has 8 contraction channels
    (2, 9, 32, 1, 1, 0, 48),     # D's first layer at 1024^2: 1x1, 16 padded input channels
    (2, 64, 32, 3, 2, 0, 33),    # data gradient = transposed stride 2: four output-parity phases of 4 / 2 / 2 / 1 taps, 17 / 16-pixel sub-grids
    (2, 32, 64, 3, 2, 0, 65),    # (its forward is strided: gather kernel) data gradient 64 -> 32 on 33 / 32-pixel sub-grids
    (1, 48, 40, 3, 1, 1, 16),    # ragged channel counts, exactly one patch
    (5, 32, 32, 3, 1, 1, 17),    # one pixel more than a patch in both directions
]


@pytest.mark.parametrize("case", HALO_CASES)
def test_f16_halo_conv_fwd_and_data_gradient(case):
    from gif_amd import ops
    B, Ci, Co, K, s, p, H = case
    g = torch.Generator().manual_seed(11)
    x = r16(torch.randn(B, Ci, H, H, generator=g))
    w = torch.randn(Co, Ci, K, K, generator=g) / (Ci * K * K) ** 0.5
    wq = r16(w)
    spec = ops.ConvSpec(K, K, s, p)
    ref = F.conv2d(x, wq, stride=s, padding=p)
    Hs = ref.shape[2]
    gy = r16(torch.randn(B, Co, Hs, Hs, generator=g))
    refd = F.conv_transpose2d(gy, wq, stride=s, padding=p, output_padding=H - ((Hs - 1) * s + K - 2 * p))
    out = {}
    for on in (True, False):
        with halo(on):
            out[on] = (ops.conv_fwd(dev16(x), w.cuda(), spec), ops.conv_bwd_data(dev16(gy), w.cuda(), spec, (H, H)))
    assert_close(host(out[True][0], Co), ref, TOL, f"halo conv_fwd {case}")
    assert_close(host(out[True][1], Ci), refd, TOL, f"halo conv_bwd_data {case}")
    if pad8(Co) != Co:
        assert (host(out[True][0])[:, Co:] == 0).all(), "padded output channels must be zero"
    # un-modulated launches: both kernels form exactly the same f16 products and fp32 sums up to the summation order
    assert_close(out[True][0], out[False][0].float(), 1e-3, f"halo vs gather kernel, forward {case}")
    assert_close(out[True][1], out[False][1].float(), 1e-3, f"halo vs gather kernel, data gradient {case}")


def test_f16_halo_conv_modulation_and_epilogues():
    """in_scale (weight-side in the halo kernel: w * s rounded to half, where the gather kernel rounds x * s), out_scale, bias,
    residual, activation, fp32 output — on a ModulatedConv2d-shaped 32 -> 32 layer, a 64 -> 64 one, and the transposed
    (up-sampling) form 64 -> 32."""
    from gif_amd import ops
    g = torch.Generator().manual_seed(12)
    for (Ci, Co, H) in ((32, 32, 48), (64, 64, 32), (40, 56, 32)):
        B = 3
        x, w = r16(torch.randn(B, Ci, H, H, generator=g)), torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5
        s, d = torch.rand(B, Ci, generator=g) + 0.5, torch.rand(B, Co, generator=g) + 0.5
        res, bias = r16(torch.randn(B, Co, H, H, generator=g)), torch.randn(Co, generator=g)
        # reference in fp32 on the exact operands: the only extra rounding of the halo path is half(w * s) per (sample, tap)
        ref = F.conv2d(x * s[:, :, None, None], r16(w), padding=1) * d[:, :, None, None]
        spec = ops.ConvSpec(3, 3, 1, 1)
        with halo(True):
            got = ops.conv_fwd(dev16(x), w.cuda(), spec, in_scale=F.pad(s, (0, pad8(Ci) - Ci)).cuda(),
                               out_scale=F.pad(d, (0, pad8(Co) - Co)).cuda())
        assert_close(host(got, Co), ref, TOL, f"halo modulated conv {Ci}->{Co}")
        ref2 = 2 ** 0.5 * F.leaky_relu(ref + res + bias[None, :, None, None], 0.2)
        with halo(True):
            got2 = ops.conv_fwd(dev16(x), w.cuda(), spec, in_scale=F.pad(s, (0, pad8(Ci) - Ci)).cuda(),
                                out_scale=F.pad(d, (0, pad8(Co) - Co)).cuda(), bias=F.pad(bias, (0, pad8(Co) - Co)).cuda(),
                                residual=dev16(res), act=True)
            got3 = ops.conv_fwd(dev16(x), w.cuda(), spec, bias=F.pad(bias, (0, pad8(Co) - Co)).cuda(), out_f32=True)
        assert_close(host(got2, Co), ref2, TOL, f"halo fused epilogue {Ci}->{Co}")
        assert got3.dtype == torch.float32
        assert_close(host(got3, Co), F.conv2d(x, r16(w), padding=1) + bias[None, :, None, None], 5e-4, "halo fp32 output")
    # transposed stride 2 with scales: the generator's up-sampling conv of the 1024^2 block (64 -> 32)
    B, Ci, Co, H = 2, 64, 32, 32
    x = r16(torch.randn(B, Ci, H, H, generator=g))
    wt = torch.randn(Ci, Co, 3, 3, generator=g) / 24
    s, d = torch.rand(B, Ci, generator=g) + 0.5, torch.rand(B, Co, generator=g) + 0.5
    ref = F.conv_transpose2d(x * s[:, :, None, None], r16(wt), stride=2) * d[:, :, None, None]
    with halo(True):
        got = ops.conv_bwd_data(dev16(x), wt.cuda(), ops.ConvSpec(3, 3, 2, 0), (2 * H + 1, 2 * H + 1), in_scale=s.cuda(), out_scale=d.cuda())
    assert_close(host(got), ref, TOL, "halo modulated transposed conv")


def test_f16_halo_conv_gradient_producer_fusions():
    """The gradient-producer epilogue (leaky-ReLU-backward mask, bias-gradient column sums, modulation-gradient dot product:
    gif_conv_epilogue ABI 2) rides on the halo kernel's 16 x 16-pixel patches: one partial-sum row per patch."""
    from gif_amd import ops
    g = torch.Generator().manual_seed(13)
    B, C, H = 3, 32, 64
    w = torch.randn(C, C, 3, 3, generator=g) / 17
    gy = r16(torch.randn(B, C, H, H, generator=g))
    xprev = r16(torch.randn(B, C, H, H, generator=g))  # the consumer's saved input: mask and dot source
    sc = torch.rand(B, C, generator=g) + 0.5
    spec = ops.ConvSpec(3, 3, 1, 1)
    contraction = F.conv_transpose2d(gy, r16(w), padding=1)
    ref_dot = (contraction * xprev).sum(dim=(2, 3))
    mask = torch.where(xprev > 0, torch.tensor(1.0), torch.tensor(0.2)) * 2 ** 0.5
    ref_v = contraction * sc[:, :, None, None] * mask
    res = {}
    for on in (True, False):
        with halo(on):
            fuse = ops.GradFuse(mask_src=dev16(xprev), mask_slope=0.2, mask_gain=2 ** 0.5, want_colsum=True, dot_src=dev16(xprev))
            v = ops.conv_bwd_data(dev16(gy), w.cuda(), spec, (H, H), out_scale=sc.cuda(), fuse=fuse)
            res[on] = (v, fuse.colsum, fuse.dot)
    v, cs, dot = res[True]
    assert_close(host(v), ref_v, TOL, "halo fused data gradient")
    assert_close(cs, ref_v.sum(dim=(0, 2, 3)), 1e-3, "halo fused column sums (fp32, before the store rounds)")
    assert_close(dot, ref_dot, 1e-3, "halo fused modulation-gradient dot product")
    assert_close(v, res[False][0].float(), 1e-3, "halo vs gather kernel")
    assert_close(cs, res[False][1], 1e-4, "column sums: halo vs gather kernel")
    assert_close(dot, res[False][2], 1e-4, "dot products: halo vs gather kernel")
    # determinism: fixed-order reductions, no atomics
    with halo(True):
        fuse = ops.GradFuse(mask_src=dev16(xprev), mask_slope=0.2, mask_gain=2 ** 0.5, want_colsum=True, dot_src=dev16(xprev))
        v2 = ops.conv_bwd_data(dev16(gy), w.cuda(), spec, (H, H), out_scale=sc.cuda(), fuse=fuse)
    assert torch.equal(v2, v) and torch.equal(fuse.colsum, cs) and torch.equal(fuse.dot, dot)


def test_f16_halo_kernel_is_the_one_that_runs():
    """The halo path must not silently fall back: the profiled launch family of an eligible shape reports halo launches."""
    from gif_amd import _lib
    lib = _lib.load()
    assert lib.gif_conv2d_f16_halo_eligible(32, 32, 3, 3, 1, 64, 64) == 1
    assert lib.gif_conv2d_f16_halo_eligible(128, 128, 3, 3, 1, 64, 64) == 0, "> 64 channels stay on the gather kernel"
    assert lib.gif_conv2d_f16_halo_eligible(32, 64, 3, 3, 2, 64, 64) == 0, "a strided FORWARD conv is a strided gather"
    assert lib.gif_conv2d_f16_halo_eligible(32, 32, 3, 3, 1, 8, 8) == 0, "sub-grids smaller than a patch"
    with halo(False):
        assert lib.gif_conv2d_f16_halo_eligible(32, 32, 3, 3, 1, 64, 64) == 0


# ------------------------------------------------------------------------------------------------ advisor findings of round 3
def test_flat_adam_load_state_dict_after_loss_scaled_steps_resets_the_device_step():
    from gif_amd.optim import FlatAdam
    from gif_amd.train_step import FlatGradBucket
    torch.manual_seed(0)
    net = torch.nn.Linear(8, 4).cuda()
    bucket = FlatGradBucket(net.parameters())
    opt = FlatAdam(net.parameters(), lr=0.01, betas=(0.5, 0.9), bucket=bucket)
    inv, found = torch.tensor(1.0, device="cuda"), torch.zeros((), device="cuda")
    x = torch.randn(4, 8, device="cuda")
    fresh = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in opt.state_dict().items() if k != "state"}
    snap = None
    for it in range(5):
        bucket.zero()
        net(x).pow(2).mean().backward()
        opt.step(inv_grad_scale=inv, found_inf=found)
        if it == 1:
            import copy
            snap = copy.deepcopy(opt.state_dict())
    assert opt._step_dev is not None and opt._step_dev.item() == 5.0
    opt.load_state_dict(snap)
    assert opt._step_dev is None, "the stale device-side count must not survive a checkpoint load"
    bucket.zero()
    net(x).pow(2).mean().backward()
    opt.step(inv_grad_scale=inv, found_inf=found)
    assert opt._step_dev.item() == 3.0, "continues from the LOADED step (2), not from the stale device counter (5)"
    assert float(next(iter(opt.state_dict()["state"].values()))["step"]) == 3.0


def test_loss_scaler_window_covers_inner_gradients_and_ignores_forward_stores():
    """begin_step() clears; only launches inside `watching()` check their f16 stores; end_backward() snapshots into THIS scaler, so
    a saturating launch issued afterwards (the other network's pass before a deferred update) cannot leak into it."""
    from gif_amd import ops
    from gif_amd.train_step import DeviceLossScaler, FlatGradBucket
    B, C, H = 2, 64, 16
    w = (torch.randn(C, C, 3, 3) / 24).cuda()
    spec = ops.ConvSpec(3, 3, 1, 1)
    huge = (torch.randn(B, C, H, H) * 3e4).cuda().half().contiguous(memory_format=torch.channels_last)
    small = torch.randn(B, C, H, H).cuda().half().contiguous(memory_format=torch.channels_last)
    net = torch.nn.Linear(4, 4).cuda()
    bucket = FlatGradBucket(net.parameters())

    def run(inner_saturates, after_saturates, outside_saturates):
        sc = DeviceLossScaler(torch.device("cuda"), init_scale=1024.0)
        bucket.zero()
        sc.begin_step()
        if outside_saturates:  # a forward pass between the gradient passes: not watched
            ops.conv_bwd_data(huge, w, spec, (H, H))
        with sc.watching():    # the regulariser's inner gradient
            ops.conv_bwd_data(huge if inner_saturates else small, w, spec, (H, H))
        with sc.watching():    # backward()
            ops.conv_bwd_data(small, w, spec, (H, H))
        sc.end_backward(bucket)
        if after_saturates:    # the OTHER network's launches before this network's deferred update
            other = DeviceLossScaler(torch.device("cuda"), init_scale=1024.0)
            other.begin_step()
            with other.watching():
                ops.conv_bwd_data(huge, w, spec, (H, H))
            other.end_backward(None)
            assert other.sat.item() == 1.0
        sc.update(bucket.flat)
        return sc.found_inf.item(), sc.scale.item()

    assert run(False, False, False) == (0.0, 1024.0)
    assert run(True, False, False) == (1.0, 512.0), "a clamped store in the inner gradient pass skips the step"
    assert run(False, True, False) == (0.0, 1024.0), "launches after end_backward() belong to somebody else"
    assert run(False, False, True) == (0.0, 1024.0), "stores outside the watch window are not gradient stores"
    assert not torch.isfinite(bucket.flat).all() or bucket.flag_slot.item() == 0.0


# ------------------------------------------------------------------------------------------------ input assembly
@pytest.mark.parametrize("dtype", [torch.float32, H16])
def test_pack_nhwc_equals_cat_pad_convert_and_is_twice_differentiable(dtype):
    """gif_pack_nhwc_*: torch.cat((image, condition), 1) + channel padding + conversion + NHWC in one pass
    (stg2_discriminator.py:48-53), sources with NCHW, channels-last and sliced strides; gradients and the gradient of the
    gradient (R1) against autograd through the ATen composition."""
    from gif_amd import functional as GF
    from gif_amd import ops
    g = torch.Generator().manual_seed(21)
    B, H, W = 3, 20, 24
    img = torch.randn(B, 3, H, W, generator=g).cuda()
    cond_full = torch.randn(B, 8, H, W, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    cond = cond_full[:, 1:7]  # a sliced channels-last view: 6 channels with foreign strides
    cp = ops.cpad(9, dtype)
    ref = F.pad(torch.cat((img, cond), 1), (0, 0, 0, 0, 0, cp - 9)).to(dtype)
    got = ops.pack_nhwc(img, 0, cond, 3, cp, dtype)
    assert got.dtype == dtype and got.is_contiguous(memory_format=torch.channels_last) and torch.equal(got, ref)
    assert torch.equal(ops.pack_nhwc(img, 0, None, 0, ops.cpad(3, dtype), dtype)[:, :3], img.to(dtype))
    assert torch.equal(ops.unpack_nhwc(got, 3, 6), ref[:, 3:9].float())
    # autograd: first and second order through a nonlinearity downstream
    a = img.clone().requires_grad_(True)
    c = cond.clone().requires_grad_(True)
    w = torch.randn(cp, H, W, generator=g).cuda()

    def loss_of(y):
        return (y.float() * w).pow(2).sum()

    res = []
    for fn in (lambda: GF.pack_nhwc(a, c, cp, dtype), lambda: F.pad(torch.cat((a, c), 1), (0, 0, 0, 0, 0, cp - 9)).to(dtype)):
        ga, gc = torch.autograd.grad(loss_of(fn()), (a, c), create_graph=True)
        (gga,) = torch.autograd.grad(ga.pow(2).sum() + gc.sum(), a)
        res.append((ga.detach(), gc.detach(), gga))
    tol = 1e-6 if dtype == torch.float32 else 2e-3
    for x, y, what in zip(res[0], res[1], ("d/d image", "d/d condition", "second order d/d image")):
        assert_close(x, y, tol, f"pack_nhwc {what} ({dtype})")


# ------------------------------------------------------------------------------------------------ f16 halo weight gradient
HALO_WGRAD_CASES = [
    # (B, Cin, Cout, K, pad, H): >= 512 patches of 16 x 16 pixels, <= 32 channels on both sides
    (2, 32, 32, 3, 1, 256),    # the 1024^2 block's layers
    (4, 16, 24, 3, 1, 200),    # condition-noise conv 2, patches overhang (200 = 12.5 x 16)
    (2, 6, 12, 3, 1, 256),     # condition-noise conv 1 (8 / 16 padded channels)
    (2, 32, 3, 1, 0, 256),     # ToRGB-shaped 1x1: single tap, no halo
    (9, 24, 32, 3, 1, 128),    # more workgroups' worth of samples than a workgroup walks in one stride
]


@pytest.mark.parametrize("case", HALO_WGRAD_CASES)
def test_f16_halo_weight_gradient(case):
    """conv_wgrad_halo_f16 (persistent workgroups over 16 x 16-pixel patches, all taps from one LDS patch) against autograd's
    weight gradient on the same f16-rounded operands; fp32 result; deterministic."""
    from gif_amd import _lib, ops
    B, Ci, Co, K, pad, H = case
    g = torch.Generator().manual_seed(31)
    x = r16(torch.randn(B, Ci, H, H, generator=g))
    gy = r16(torch.randn(B, Co, H, H, generator=g))
    spec = ops.ConvSpec(K, K, 1, pad)
    geom = _lib.ConvGeom(B, H, H, pad8(Ci), H, H, pad8(Co), K, K, 1, pad)
    import ctypes
    nsplit = _lib.load().gif_conv2d_wgrad_splits_f16(ctypes.byref(geom))
    assert nsplit == min(512, B * ((H + 15) // 16) ** 2), "the halo kernel's split count = persistent workgroups"
    wl = torch.zeros(Co, Ci, K, K, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(x, wl, padding=pad), wl, gy)
    got = ops.conv_wgrad(dev16(gy), dev16(x), spec, Co, Ci)
    assert got.dtype == torch.float32
    assert_close(got, ref, 5e-4, f"f16 halo wgrad {case}")
    assert torch.equal(got, ops.conv_wgrad(dev16(gy), dev16(x), spec, Co, Ci)), "fixed-order reductions"


def test_f16_halo_weight_gradient_modulated():
    """dW = sum (gy * d) (x) (x * s): per-sample scales multiply the f16 fragments (one scalar per lane; a patch is one sample)."""
    from gif_amd import ops
    g = torch.Generator().manual_seed(32)
    B, C, H = 3, 32, 224
    x, gy = r16(torch.randn(B, C, H, H, generator=g)), r16(torch.randn(B, C, H, H, generator=g))
    s, d = torch.rand(B, C, generator=g) + 0.5, torch.rand(B, C, generator=g) + 0.5
    xs, gyd = r16(x * r16(s)[:, :, None, None]), r16(gy * r16(d)[:, :, None, None])
    wl = torch.zeros(C, C, 3, 3, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(xs, wl, padding=1), wl, gyd)
    got = ops.conv_wgrad(dev16(gy), dev16(x), ops.ConvSpec(3, 3, 1, 1), C, C, small_scale=d.cuda(), big_scale=s.cuda())
    assert_close(got, ref, 1e-3, "f16 modulated halo wgrad")


def test_f16_weight_gradient_256_tiles_plain_and_modulated():
    """conv_wgrad_mfma<f16, 256, 256> (8 waves, wave tile 128 x 64): layers with multiples of 256 channels on both sides and >= 16384
    reduction rows; plain and with per-sample scales (the scale table of a 256 x 256 tile)."""
    from gif_amd import ops
    g = torch.Generator().manual_seed(51)
    for (B, Cs, Cb, K, s, p, H) in ((4, 256, 256, 3, 1, 1, 64), (2, 512, 256, 3, 2, 0, 257), (5, 256, 512, 1, 1, 0, 64)):
        spec = ops.ConvSpec(K, K, s, p)
        x = r16(torch.randn(B, Cb, H, H, generator=g))
        hs = spec.small_hw(H, H)[0]
        gy = r16(torch.randn(B, Cs, hs, hs, generator=g))
        wl = torch.zeros(Cs, Cb, K, K, requires_grad=True)
        (ref,) = torch.autograd.grad(F.conv2d(x, wl, stride=s, padding=p), wl, gy)
        got = ops.conv_wgrad(dev16(gy), dev16(x), spec, Cs, Cb)
        assert_close(got, ref, 5e-4, f"f16 wgrad on 256x256 tiles {(B, Cs, Cb, K, s, H)}")
        if s == 1 and (hs * hs) % 32 == 0:
            sc, d = torch.rand(B, Cb, generator=g) + 0.5, torch.rand(B, Cs, generator=g) + 0.5
            xs, gyd = r16(x * r16(sc)[:, :, None, None]), r16(gy * r16(d)[:, :, None, None])
            (refm,) = torch.autograd.grad(F.conv2d(xs, wl, stride=s, padding=p), wl, gyd)
            gotm = ops.conv_wgrad(dev16(gy), dev16(x), spec, Cs, Cb, small_scale=d.cuda(), big_scale=sc.cuda())
            assert_close(gotm, refm, 1e-3, f"f16 modulated wgrad on 256x256 tiles {(B, Cs, Cb, K, H)}")


# ---------------------------------------------------------------------------------------------------------------------
# modulation bank (csrc/linear.hip: linear_bank_*): every ModulatedConv2d's EqualLinear of a generator pass in one launch
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,widths", [(32, [512, 512, 256, 128, 64]), (5, [16, 24, 32, 8, 512]), (1, [64]), (40, [128] * 27),
                                      (32, [512] * 13 + [256, 256, 128, 128, 64, 64, 32, 32, 16, 16] + [512] * 4 + [256, 128, 64, 32, 16])])
def test_linear_bank_kernels_vs_fp64(M, widths):
    """forward (one launch), weight + bias gradients (one launch), input gradient over the concatenated reduction axis (one
    launch) against float64 on the host; bit-identical across two runs (fixed summation order)."""
    from gif_amd import ops
    g = torch.Generator().manual_seed(M * 100 + len(widths))
    K, scale = 512, 1.0 / 512 ** 0.5
    x = torch.randn(M, K, generator=g)
    ws = [torch.randn(n, K, generator=g) for n in widths]
    bs = [torch.randn(n, generator=g) if i % 3 else None for i, n in enumerate(widths)]
    gs = [torch.randn(M, n, generator=g) for n in widths]
    xd, wd, bd, gd = x.cuda(), [w.cuda() for w in ws], [None if b is None else b.cuda() for b in bs], [t.cuda() for t in gs]
    assert ops.linear_bank_ok(xd, wd)
    outs = ops.linear_bank_fwd(xd, wd, bd, scale)
    outs2 = ops.linear_bank_fwd(xd, wd, bd, scale)
    for i, (o, o2, w, b) in enumerate(zip(outs, outs2, ws, bs)):
        ref = scale * (x.double() @ w.double().t()) + (0 if b is None else b.double())
        assert_close(o, ref.float(), 2e-6, f"bank forward, layer {i} (n={widths[i]})")
        assert torch.equal(o, o2)
        assert_close(o, ops.linear_nt(xd, wd[i], bd[i], scale), 1e-6, f"bank vs per-layer kernel, layer {i}")
    gx, gws, gbs = ops.linear_bank_bwd(xd, wd, gd, scale, True, True, True)
    gx2, gws2, gbs2 = ops.linear_bank_bwd(xd, wd, gd, scale, True, True, True)
    ref_gx = scale * sum(t.double() @ w.double() for t, w in zip(gs, ws))
    assert_close(gx, ref_gx.float(), 3e-6, "bank input gradient")
    assert torch.equal(gx, gx2)
    for i, (t, w) in enumerate(zip(gs, ws)):
        assert_close(gws[i], (scale * (t.double().t() @ x.double())).float(), 2e-6, f"bank weight gradient {i}")
        assert_close(gbs[i], t.double().sum(0).float(), 2e-6, f"bank bias gradient {i}")
        assert torch.equal(gws[i], gws2[i]) and torch.equal(gbs[i], gbs2[i])
    # subsets of the outputs: gx only / weights only
    gx3, w3, b3 = ops.linear_bank_bwd(xd, wd, gd, scale, True, False, False)
    assert w3 is None and b3 is None and torch.equal(gx3, gx)
    gx4, w4, b4 = ops.linear_bank_bwd(xd, wd, gd, scale, False, True, False)
    assert gx4 is None and b4 is None and all(torch.equal(a, b) for a, b in zip(w4, gws))


def test_linear_bank_rejects_what_it_cannot_take():
    from gif_amd import ops, _lib
    x = torch.randn(4, 512, device="cuda")
    assert not ops.linear_bank_ok(x, [torch.randn(12, 512, device="cuda")])          # n % 8
    assert not ops.linear_bank_ok(x, [torch.randn(16, 512, device="cuda")] * 41)     # table size
    assert not ops.linear_bank_ok(x.half(), [torch.randn(16, 512, device="cuda")])
    with pytest.raises(_lib.GifHipError):
        ops.linear_bank_fwd(x, [torch.randn(12, 512, device="cuda")], [None], 1.0)


def test_generator_modulation_bank_vs_per_layer_launches(monkeypatch):
    """The whole generator with the bank on / off (GIF_STYLE_BANK): same image and gradients to fp32 rounding, recorded
    (create_graph) backward included; 15 modulation launches become 1 forward (+2 backward)."""
    import contextlib, io
    from gif_amd import layers, ops
    from gif_amd.generator import StyledGenerator
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        G = StyledGenerator(embedding_vocab_size=16, rendered_flame_ascondition=True, normal_maps_as_cond=True).cuda()
    gen = torch.Generator().manual_seed(1)
    cond = (torch.rand(4, 6, 64, 64, generator=gen) * 2 - 1).cuda()
    idx = torch.randint(0, 16, (4,), generator=gen).cuda()
    out = {}
    for bank in (True, False):
        monkeypatch.setattr(layers, "_STYLE_BANK", bank)
        G.zero_grad(set_to_none=True)
        img = G(cond, None, step=4, alpha=1, input_indices=idx)[-1]
        img.pow(2).mean().backward()
        grads = {k: p.grad.clone() for k, p in G.named_parameters() if p.grad is not None}
        # recorded backward (path-length form): d/dparams of |d img / d w|^2
        w = torch.randn(4, 512, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2)).requires_grad_(True)
        img2 = G.generator([w], None, G._condition_pyramid(cond, 4), step=4, alpha=1)[-1]
        (gw,) = torch.autograd.grad(img2.sum(), w, create_graph=True)
        pen = gw.pow(2).sum()
        pgr = torch.autograd.grad(pen, [p for p in G.generator.parameters()], allow_unused=True)
        out[bank] = (img.detach(), grads, gw.detach(), pgr)
    assert_close(out[True][0], out[False][0], 1e-5, "image")
    for k in out[False][1]:
        assert_close(out[True][1][k], out[False][1][k], 2e-4, f"grad {k}")
    assert_close(out[True][2], out[False][2], 1e-4, "d img / d w")
    for a, b in zip(out[True][3], out[False][3]):
        assert (a is None) == (b is None)
        if a is not None:
            assert_close(a, b, 5e-4, "second-order parameter gradient")


# ---------------------------------------------------------------------------------------------------------------------
# buffer-addressed LDS-DMA of the bf16x3 / f16 direct kernels (conv_igemm.hip): padding = an out-of-range lane offset
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("cin,cout,H,W,k,stride,pad", [(32, 40, 19, 23, 3, 1, 1), (64, 32, 17, 17, 3, 2, 0), (32, 32, 9, 31, 1, 1, 0),
                                                        (40, 64, 13, 13, 3, 2, 1), (32, 32, 12, 12, 3, 1, 0), (128, 128, 33, 33, 3, 2, 1)])
def test_buffer_addressed_dma_borders_and_ragged_tiles(dtype, cin, cout, H, W, k, stride, pad):
    """Odd image sizes, every padding / stride form, tiles whose last rows lie past M: forward and data gradient of the direct
    kernels (Winograd off) against ATen-CPU; every border pixel is a lane that sends an out-of-range offset."""
    from gif_amd import ops
    from gpu_util import dev
    g = torch.Generator().manual_seed(cin * 1000 + H * 10 + k)
    B = 3
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    spec = ops.ConvSpec(k, k, stride, pad)
    old = ops.WINOGRAD
    ops.WINOGRAD = False
    try:
        if dtype == "f16":
            xd, wr, tol = dev16(x), r16(w), TOL
            ref = F.conv2d(r16(x), wr, stride=stride, padding=pad)
            y = ops.conv_fwd(xd, w.cuda(), spec)
            assert_close(host(y, cout), ref, tol, "forward")
            gy = torch.randn(ref.shape, generator=g)
            refg = F.conv_transpose2d(r16(gy), wr, stride=stride, padding=pad, output_padding=(H + 2 * pad - k) % stride)
            gx = ops.conv_bwd_data(dev16(gy), w.cuda(), spec, (H, W))
            assert_close(host(gx, cin), refg[:, :, :H, :W], tol, "data gradient")
        else:
            ops.set_fp32_mfma_mode("bf16x3")
            ref = F.conv2d(x.double(), w.double(), stride=stride, padding=pad).float()
            y = ops.conv_fwd(dev(x), w.cuda(), spec)
            assert_close(host(y, cout), ref, 2e-5, "forward")
            gy = torch.randn(ref.shape, generator=g)
            refg = F.conv_transpose2d(gy.double(), w.double(), stride=stride, padding=pad,
                                      output_padding=(H + 2 * pad - k) % stride).float()
            gx = ops.conv_bwd_data(dev(gy), w.cuda(), spec, (H, W))
            assert_close(host(gx, cin), refg[:, :, :H, :W], 2e-5, "data gradient")
    finally:
        ops.WINOGRAD = old
