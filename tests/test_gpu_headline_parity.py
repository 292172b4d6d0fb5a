"""-m gpu: ORACLE-checked parity at the BASELINE sizes (round-1 verdict item 1: the full-size checks used to compare the HIP
kernels with each other).

(a) per-layer: the benchmark's layer shapes at batch 32 through the DEFAULT dispatch (Winograd / direct / transposed phases /
    small-channel kernels exactly as bench.py takes them) — forward, data gradient and weight gradient against ONE ATen-CPU
    fp32 convolution each (the GPU box's host has enough cores for a 0.3 TFLOP conv in seconds).
(b) 256x256, batch 4, full backward: every G and D parameter gradient, the R1 penalty and the gradients through its double
    backward against oracle/stylegan2_ref.py (the restatement that is pinned to the real reference).
(c) BASELINE config 3 at its stated size: mesh -> HIP vertex normals + rasteriser -> 6-channel condition -> GifTrainer.step
    at 256x256, batch 32, R1 iteration; per-sample images / scores / R1 penalties of one minibatch-stddev group against the
    oracle, and the trainer's first-step losses against the values assembled from those per-sample quantities.
Tolerances are the small-size tests' (CONV_CASES): 2e-5 of the tensor max for forward / dgrad, 5e-5 for wgrad (fp32 sums of
2 M products), 3e-4 for whole-model gradients."""
import contextlib
import io
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import assert_close, assert_grads_close, dev, host, pad4, rel_err

pytestmark = pytest.mark.gpu

B_HEAD = 32  # BASELINE batch per GPU

# (name, Cin, Cout, K, stride, pad, Hin, transposed)
LAYER_SHAPES = [
    ("3x3 128->128 @256^2 (G st_cv2 / D conv1)", 128, 128, 3, 1, 1, 256, False),
    ("3x3 256->256 @128^2", 256, 256, 3, 1, 1, 128, False),
    ("3x3 512->512 @64^2", 512, 512, 3, 1, 1, 64, False),
    ("3x3 stride-2 128->256 on the blurred 257^2 map (D conv2)", 128, 256, 3, 2, 0, 257, False),
    ("transposed 3x3 stride-2 256->128, 128^2 -> 257^2 (G up-sampling)", 256, 128, 3, 2, 0, 128, True),
    ("3x3 24->128 @256^2 (condition-noise conv 3)", 24, 128, 3, 1, 1, 256, False),
]


@pytest.mark.parametrize("shape", LAYER_SHAPES, ids=[s[0].split(" (")[0].replace(" ", "_") for s in LAYER_SHAPES])
def test_benchmark_layer_vs_aten_cpu(shape):
    """fwd / dgrad / wgrad of one benchmark layer, batch 32, default dispatch, vs ATen CPU fp32."""
    from gif_amd import functional as GF
    name, Ci, Co, K, st, pd, H, transposed = shape
    g = torch.Generator().manual_seed(100 + LAYER_SHAPES.index(shape))
    B = B_HEAD
    x = torch.randn(B, Ci, H, H, generator=g)
    if not transposed:
        w = torch.randn(Co, Ci, K, K, generator=g)
        wscale = 1 / math.sqrt(Ci * K * K)
        Ho = (H + 2 * pd - K) // st + 1
    else:
        # conv_transpose2d(x, W^T, stride 2): the underlying forward conv maps Cout -> Cin, canonical weight [Ci, Co, K, K]
        w = torch.randn(Ci, Co, K, K, generator=g)
        wscale = 1 / math.sqrt(Ci * K * K)
        Ho = (H - 1) * st + K - 2 * pd
    gy = torch.randn(B, Co, Ho, Ho, generator=g)
    # ---- ATen CPU reference (one conv call + its autograd)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    if not transposed:
        yr = F.conv2d(xr, wr * wscale, stride=st, padding=pd)
    else:
        yr = F.conv_transpose2d(xr, wr * wscale, stride=st, padding=pd)
    gxr, gwr = torch.autograd.grad(yr, (xr, wr), gy)
    # ---- HIP path through the autograd Functions the models use
    xd = dev(x, requires_grad=True)
    wd = w.cuda().requires_grad_(True)
    if not transposed:
        yd = GF.conv2d(xd, wd, st, pd, wscale=wscale)
    else:
        yd = GF.conv_transpose2d(xd, wd, st, pd, (Ho, Ho), wscale=wscale)
    gxd, gwd = torch.autograd.grad(yd, (xd, wd), dev(gy))
    assert_close(host(yd, Co), yr, 2e-5, f"{name}: forward")
    assert_close(host(gxd, Ci), gxr, 2e-5, f"{name}: data gradient")
    assert_close(gwd, gwr, 5e-5, f"{name}: weight gradient")


def test_benchmark_modulated_layers_vs_aten_cpu():
    """The generator's two modulated forms at the benchmark size, batch 32, default dispatch: the fused StyledConv
    (128->128 @256^2: modulation, demodulation, condition-noise residual, bias, leaky ReLU in one launch) and the up-sampling
    modulated transposed conv (256->128, 128^2 -> 257^2) — forward and every gradient vs an ATen-CPU composition."""
    from gif_amd import functional as GF
    g = torch.Generator().manual_seed(77)
    B = B_HEAD
    # ---- fused same-resolution StyledConv
    Ci = Co = 128
    H = 256
    x = torch.randn(B, Ci, H, H, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g)
    s, d = torch.rand(B, Ci, generator=g) + 0.5, torch.rand(B, Co, generator=g) + 0.5
    res, bias = torch.randn(B, Co, H, H, generator=g), torch.randn(Co, generator=g) * 0.1
    gy = torch.randn(B, Co, H, H, generator=g)
    wscale = 1 / math.sqrt(Ci * 9)
    xd, rd = dev(x, True), dev(res, True)
    wd, sd_, dd, bd = (t.cuda().requires_grad_(True) for t in (w, s, d, bias))
    yd = GF.modulated_conv2d_act(xd, wd, sd_, dd, rd, bd, 1, wscale)
    gots = torch.autograd.grad(yd, (xd, wd, sd_, dd, rd, bd), dev(gy))
    y_hip = host(yd)
    # ATen-CPU reference.  The leaky-ReLU backward only depends on the SIGN of the output; of 268 M pre-activations a few
    # hundred lie within fp32 rounding of zero, where the CPU and the HIP forward may land on different sides — each such
    # element would move ~1000 entries of the (otherwise linear) gradients by up to 1e-1 of their max and say nothing about
    # the kernels.  So the reference backward uses the mask of the HIP forward (whose values are checked right here), which
    # makes every gradient below an exactly linear function of gy that must agree to accumulation-order rounding.
    leaves = [t.clone().requires_grad_(True) for t in (x, w, s, d, res, bias)]
    xr, wr, sr, dr, rr, br = leaves
    zr = F.conv2d(xr * sr[:, :, None, None], wr * wscale, padding=1) * dr[:, :, None, None] + rr + br[None, :, None, None]
    yr = 2 ** 0.5 * F.leaky_relu(zr, 0.2)
    assert_close(y_hip, yr.detach(), 2e-5, "fused StyledConv forward")
    flips = ((y_hip > 0) != (yr.detach() > 0)).float().mean().item()
    assert flips < 1e-5, f"sign of the activation differs on a fraction {flips:.2e} of the outputs"
    mask = torch.where(y_hip > 0, torch.tensor(1.0), torch.tensor(0.2)) * 2 ** 0.5
    refs = torch.autograd.grad(zr * mask, leaves, gy)
    for nm, got, ref, tol in zip(("x", "w", "s", "d", "residual", "bias"), gots, refs, (2e-5, 5e-5, 5e-5, 5e-5, 2e-5, 5e-5)):
        assert_close(got, ref, tol, f"fused StyledConv grad {nm}")
    del leaves, refs, gots, yr, yd, xd, rd
    # ---- up-sampling branch: modulated conv_transpose2d, stride 2 (the blur that follows is covered by the FIR tests)
    Ci, Co, H = 256, 128, 128
    x = torch.randn(B, Ci, H, H, generator=g)
    wt = torch.randn(Ci, Co, 3, 3, generator=g)  # canonical forward-conv weight of the underlying conv (Co -> Ci)
    s, d = torch.rand(B, Ci, generator=g) + 0.5, torch.rand(B, Co, generator=g) + 0.5
    Ho = 2 * H + 1
    gy = torch.randn(B, Co, Ho, Ho, generator=g)
    wscale = 1 / math.sqrt(Ci * 9)
    leaves = [t.clone().requires_grad_(True) for t in (x, wt, s, d)]
    xr, wr, sr, dr = leaves
    yr = F.conv_transpose2d(xr * sr[:, :, None, None], wr * wscale, stride=2) * dr[:, :, None, None]
    refs = torch.autograd.grad(yr, leaves, gy)
    xd = dev(x, True)
    wd, sd_, dd = (t.cuda().requires_grad_(True) for t in (wt, s, d))
    yd = GF.modulated_conv2d(xd, wd, sd_, dd, stride=2, pad=0, transposed=True, out_hw=(Ho, Ho), wscale=wscale)
    gots = torch.autograd.grad(yd, (xd, wd, sd_, dd), dev(gy))
    assert_close(host(yd), yr, 2e-5, "modulated transposed conv forward")
    for nm, got, ref, tol in zip(("x", "w", "s", "d"), gots, refs, (2e-5, 5e-5, 5e-5, 5e-5)):
        assert_close(got, ref, tol, f"modulated transposed conv grad {nm}")


def _build(res, vocab=64):
    from gif_amd.discriminator import Discriminator
    from gif_amd.generator import StyledGenerator
    with contextlib.redirect_stdout(io.StringIO()):
        g = StyledGenerator(embedding_vocab_size=vocab, rendered_flame_ascondition=True, normal_maps_as_cond=True)
        d = Discriminator(size=res, num_color_chnls=9)
    return g, d


def _leaves(sd):
    return {k: (v.clone().requires_grad_(True) if (not k.endswith('kernel') and 'embd_weight' not in k) else v.clone())
            for k, v in sd.items()}


def test_full_backward_256_batch4_vs_oracle():
    """(b) one D loss (with R1 on the real images) and one G loss at 256x256, batch 4: EVERY parameter gradient of both
    networks against the CPU oracle's autograd, plus the R1 penalty values."""
    from gif_amd import losses
    from oracle import stylegan2_ref as R
    torch.manual_seed(0)
    G, D = _build(256, vocab=16)
    g_sd = R.seeded_state_dict(G.state_dict(), 41)
    d_sd = R.seeded_state_dict(D.state_dict(), 42)
    G.load_state_dict(g_sd, strict=True)
    D.load_state_dict(d_sd, strict=True)
    gen = torch.Generator().manual_seed(43)
    B = 4
    real = torch.rand(B, 3, 256, 256, generator=gen) * 2 - 1
    cond = torch.rand(B, 6, 256, 256, generator=gen) * 2 - 1
    idx = torch.randint(0, 16, (B,), generator=gen)
    # ---- oracle
    gl, dl = _leaves(g_sd), _leaves(d_sd)
    real_r = real.clone().requires_grad_(True)
    fake_r = R.generator_forward(gl, cond, 6, idx)
    rs = R.discriminator_forward(dl, real_r, cond, 256)
    r1_r = R.grad_penalty_loss([real_r], rs)
    d_loss_r = F.softplus(-rs).mean() + r1_r.mean() + F.softplus(R.discriminator_forward(dl, fake_r.detach(), cond, 256)).mean()
    d_keys = [k for k, v in dl.items() if v.requires_grad]
    d_grads_r = dict(zip(d_keys, torch.autograd.grad(d_loss_r, [dl[k] for k in d_keys])))
    g_loss_r = F.softplus(-R.discriminator_forward(dl, fake_r, cond, 256)).mean()
    g_keys = [k for k, v in gl.items() if v.requires_grad]
    g_grads_r = dict(zip(g_keys, torch.autograd.grad(g_loss_r, [gl[k] for k in g_keys], allow_unused=True)))
    # ---- HIP
    G, D = G.cuda(), D.cuda()
    condd, idxd = cond.cuda(), idx.cuda()
    real_d = real.cuda().requires_grad_(True)
    fake_d = G(condd, None, step=6, alpha=1, input_indices=idxd)
    assert (fake_d[0].detach().cpu() - fake_r.detach()).abs().max().item() < 1e-3, "north-star: G output L_inf < 1e-3"
    rs_d, _ = D([real_d], condition=condd)
    r1_d = losses.grad_penalty_loss([real_d], rs_d, step=None)
    assert_close(r1_d, r1_r, 3e-4, "R1 penalties at 256x256")
    d_loss_d = F.softplus(-rs_d).mean() + r1_d.mean() + F.softplus(D([fake_d[0].detach()], condition=condd)[0]).mean()
    assert abs(d_loss_d.item() - d_loss_r.item()) < 1e-4 * max(1.0, abs(d_loss_r.item()))
    d_named = dict(D.named_parameters())
    d_grads_d = dict(zip(d_keys, torch.autograd.grad(d_loss_d, [d_named[k] for k in d_keys])))
    g_loss_d = F.softplus(-D(fake_d, condition=condd)[0]).mean()
    assert abs(g_loss_d.item() - g_loss_r.item()) < 1e-4 * max(1.0, abs(g_loss_r.item()))
    g_named = dict(G.named_parameters())
    g_grads_d = dict(zip(g_keys, torch.autograd.grad(g_loss_d, [g_named[k] for k in g_keys], allow_unused=True)))
    # every parameter gradient of both networks; tolerance policy: gpu_util.assert_grads_close (activation sign flips)
    # (review item 7: the cap on tensors between 3e-4 and 2e-3 is the observed count + 2 — 3 of D's 38 and 6 of G's 223 — not 10 %)
    for keys, got, ref, what, cap in ((d_keys, d_grads_d, d_grads_r, "D", 5), (g_keys, g_grads_d, g_grads_r, "G", 8)):
        worst, n_out, l2 = assert_grads_close([got[k] for k in keys], [ref[k] for k in keys], keys, tight=3e-4, loose=2e-3,
                                              max_outliers=cap, what=f"{what} parameter gradients at 256x256, batch 4")
        print(f"{what}: worst tensor {worst:.2e}, {n_out} of {len(keys)} tensors above 3e-4, relative L2 over all parameters {l2:.2e}")



def test_f16_full_backward_256_vs_oracle():
    """Row N1 (BASELINE configs[4]: f16 activations, fp32 demodulation) as a TRAINING configuration: the D loss, the R1 penalty and
    the G loss at 256x256, batch 2 — every parameter gradient of both networks and the R1 penalties of the f16-activation HIP path
    against the fp32 CPU oracle's autograd (review item 4b: the f16 gradients used to be compared with the HIP fp32 path only, at
    32x32, with 5e-2 / 0.3 / 25 % tolerances).  The three gradient sets are checked separately because they differ in kind:
      * G loss and plain D loss (first-order): the signal crosses ~27 + 27 f16-rounded tensors (u = 2^-11 each, forward and
        backward) and a leaky ReLU per layer whose branch flips for ~1 pre-activation in 2000 at f16 resolution; the 4x4 .. 16x16
        layers see few positions, so single flips move their gradients by percents;
      * the R1 term differentiates D twice: its parameter gradients carry the rounding of the first backward's f16 activation
        gradients through a second backward.
    What bounds them (measured, scale-independent: 2^12, 2^18 give the same figures, i.e. no underflow): NOT the accumulation of
    unit round-offs (sqrt(54) u = 3.6e-3) but the leaky-ReLU branch flips — a pre-activation whose f16-path value differs from the
    fp32 one by eps (1-2.4e-3 of the activation scale: the forward analysis of tests/test_gpu_f16.py) lands on the other side of
    zero with probability ~0.8 eps, and a flipped element's backward factor jumps between 1 and 0.2: a relative L2 perturbation of
    0.8 sqrt(0.8 eps) = 2.3-3.5e-2 of the gradient signal crossing that layer.  Measured: D 1.8e-2, G 1.5e-2 relative L2 over all
    parameters, R1 term 0.9e-2; worst tensors at the 4x4 .. 16x16 layers (few positions per sum): 0.14.  The bounds asserted are
    2 x those measurements, per set: relative L2 over all parameters, the worst tensor, and the number of tensors above 0.1.
    (R1 set: weights only — d R1 / d bias is identically zero for a piecewise-linear D, both sides return rounding noise there.)"""
    import statistics
    from gif_amd import losses
    from oracle import stylegan2_ref as R
    torch.manual_seed(0)
    G, D = _build(256, vocab=16)
    g_sd = R.seeded_state_dict(G.state_dict(), 51)
    d_sd = R.seeded_state_dict(D.state_dict(), 52)
    G.load_state_dict(g_sd, strict=True)
    D.load_state_dict(d_sd, strict=True)
    gen = torch.Generator().manual_seed(53)
    B = 2
    real = torch.rand(B, 3, 256, 256, generator=gen) * 2 - 1
    cond = torch.rand(B, 6, 256, 256, generator=gen) * 2 - 1
    idx = torch.randint(0, 16, (B,), generator=gen)
    # ---- fp32 oracle
    gl, dl = _leaves(g_sd), _leaves(d_sd)
    real_r = real.clone().requires_grad_(True)
    fake_r = R.generator_forward(gl, cond, 6, idx)
    rs = R.discriminator_forward(dl, real_r, cond, 256)
    r1_r = R.grad_penalty_loss([real_r], rs)
    d_plain_r = F.softplus(-rs).mean() + F.softplus(R.discriminator_forward(dl, fake_r.detach(), cond, 256)).mean()
    d_keys = [k for k, v in dl.items() if v.requires_grad]
    ref_sets = {"D plain": dict(zip(d_keys, torch.autograd.grad(d_plain_r, [dl[k] for k in d_keys], retain_graph=True))),
                "D R1": dict(zip(d_keys, torch.autograd.grad(r1_r.mean(), [dl[k] for k in d_keys], allow_unused=True)))}
    g_loss_r = F.softplus(-R.discriminator_forward(dl, fake_r, cond, 256)).mean()
    g_keys = [k for k, v in gl.items() if v.requires_grad]
    ref_sets["G"] = dict(zip(g_keys, torch.autograd.grad(g_loss_r, [gl[k] for k in g_keys], allow_unused=True)))
    # ---- HIP, f16 activations, static loss scale 2^12 (R1's inner gradient on scores pre-multiplied by 2^10, as the trainer does)
    H16 = torch.float16
    G, D = G.cuda().set_activation_dtype(H16), D.cuda().set_activation_dtype(H16)
    condd, idxd = cond.cuda(), idx.cuda()
    real_d = real.cuda().requires_grad_(True)
    S = 2.0 ** int(os.environ.get("GIF_TEST_F16_LOG2_SCALE", "12"))
    fake_d = G(condd, None, step=6, alpha=1, input_indices=idxd)
    rs_d, _ = D([real_d], condition=condd)
    r1_d = losses.grad_penalty_loss([real_d], rs_d, step=None, grad_scale=2.0 ** 10)
    e_r1 = rel_err(r1_d, r1_r)
    d_plain_d = F.softplus(-rs_d).mean() + F.softplus(D([fake_d[0].detach()], condition=condd)[0]).mean()
    d_named, g_named = dict(D.named_parameters()), dict(G.named_parameters())

    def grads(loss, named, keys, **kw):
        return dict(zip(keys, ((None if g is None else g / S) for g in
                               torch.autograd.grad(loss * S, [named[k] for k in keys], allow_unused=True, **kw))))
    got_sets = {"D plain": grads(d_plain_d, d_named, d_keys, retain_graph=True), "D R1": grads(r1_d.mean(), d_named, d_keys)}
    g_loss_d = F.softplus(-D(fake_d, condition=condd)[0]).mean()
    got_sets["G"] = grads(g_loss_d, g_named, g_keys)
    u = 2.0 ** -11
    print(f"\nf16 @256: losses D {d_plain_d.item():.5f} vs {d_plain_r.item():.5f}, G {g_loss_d.item():.5f} vs {g_loss_r.item():.5f}; "
          f"R1 penalties rel err {e_r1:.2e} ({e_r1 / u:.1f} u)")
    assert abs(d_plain_d.item() - d_plain_r.item()) < 4 * u * max(1.0, abs(d_plain_r.item()))   # measured 0.3 u
    assert abs(g_loss_d.item() - g_loss_r.item()) < 4 * u * max(1.0, abs(g_loss_r.item()))       # measured 0.5 u
    assert e_r1 < 12 * u, e_r1                                                                   # measured 5.7 u
    # (relative L2 over all parameters, worst tensor, tensors above 32 u) allowed = 2 x measured, see the printed line
    BOUNDS = {"D plain": (3.7e-2, 0.29, 4), "D R1": (1.8e-2, 0.3, 4), "G": (3.0e-2, 0.12, 2)}
    for what in ("D plain", "D R1", "G"):
        ref, got = ref_sets[what], got_sets[what]
        keys = [k for k in ref if ref[k] is not None and ref[k].abs().max().item() > 0 and not (what == "D R1" and k.endswith("bias"))]
        errs = sorted(((rel_err(got[k], ref[k]), k) for k in keys), reverse=True)
        num = sum((got[k].detach().float().cpu() - ref[k]).pow(2).sum().item() for k in keys)
        den = sum(ref[k].pow(2).sum().item() for k in keys)
        l2 = (num / den) ** 0.5
        n32 = sum(1 for e, _ in errs if e > 0.1)
        print(f"{what}: relative L2 over all parameters {l2:.2e} ({l2 / u:.1f} u); worst tensors "
              + ", ".join(f"{k} {e:.2e}" for e, k in errs[:3]) + f"; median {statistics.median(e for e, _ in errs):.2e}; "
              f"{n32} of {len(errs)} tensors above 0.1")
        l2_b, worst_b, n_b = BOUNDS[what]
        if not os.environ.get("GIF_TEST_F16_EXPLORE"):
            assert l2 <= l2_b and errs[0][0] <= worst_b and n32 <= n_b, (what, l2, errs[0], n32, BOUNDS[what])


def test_config3_at_stated_size_vs_oracle():
    """(c) BASELINE config 3: rendered condition -> full G+D training iteration at 256x256, batch 32, R1 iteration."""
    from gif_amd import losses, render
    from gif_amd.train_step import GifTrainer
    from oracle import stylegan2_ref as R
    mesh = np.load(os.path.join(os.path.dirname(__file__), "golden", "body_mesh.npz"))
    B, RES = 32, 256
    rng = np.random.RandomState(0)
    verts = []
    for i in range(B):  # small random rotations about y, as a stand-in for FLAME pose variation
        a = rng.uniform(-0.4, 0.4)
        Rm = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
        verts.append((mesh["vertices"] @ Rm.T).astype(np.float32))
    v = torch.from_numpy(np.stack(verts)).cuda()
    f = torch.from_numpy(mesh["faces"]).cuda()
    cam = torch.tensor([[0.95, 0.0, 0.35]] * B, device="cuda")
    v_ndc = render.batch_orth_proj(v, cam)
    tex = (v - v.amin(dim=1, keepdim=True)) / (v.amax(dim=1, keepdim=True) - v.amin(dim=1, keepdim=True))
    cond = render.render_condition(v_ndc, f, tex, RES, RES)
    assert cond.shape == (B, 6, RES, RES) and (cond[:, 3:].abs().sum(dim=(1, 2, 3)) > 0).all()
    torch.manual_seed(0)
    G, D = _build(RES, vocab=64)
    G_ema, _ = _build(RES, vocab=64)
    g_sd = R.seeded_state_dict(G.state_dict(), 51)
    d_sd = R.seeded_state_dict(D.state_dict(), 52)
    G.load_state_dict(g_sd, strict=True)
    G_ema.load_state_dict(g_sd, strict=True)
    D.load_state_dict(d_sd, strict=True)
    G, G_ema, D = G.cuda(), G_ema.cuda(), D.cuda()
    gen = torch.Generator(device="cuda").manual_seed(53)
    real = torch.rand(B, 3, RES, RES, device="cuda", generator=gen) * 2 - 1
    idx = torch.randint(0, 64, (B,), device="cuda", generator=gen)
    # per-sample quantities of the first D step, HIP
    real_d = real.clone().requires_grad_(True)
    with torch.no_grad():
        fake = G(cond, None, step=6, alpha=1, input_indices=idx)[0]
        fs = D([fake], condition=cond)[0]
    rs = D([real_d], condition=cond)[0]
    r1 = losses.grad_penalty_loss([real_d], rs, step=None).detach()
    rs = rs.detach()
    # the oracle on ONE minibatch-stddev group: view(4, B/4, ...) groups samples {b, b+8, b+16, b+24}
    sel = torch.tensor([0, 8, 16, 24])
    c4, i4, r4 = cond[sel.cuda()].cpu(), idx[sel.cuda()].cpu(), real[sel.cuda()].cpu().requires_grad_(True)
    with torch.no_grad():
        fake_o = R.generator_forward(g_sd, c4, 6, i4)
        fs_o = R.discriminator_forward(d_sd, fake_o, c4, RES)
    rs_o = R.discriminator_forward(d_sd, r4, c4, RES)
    r1_o = R.grad_penalty_loss([r4], rs_o).detach()
    assert (fake[sel.cuda()].cpu() - fake_o).abs().max().item() < 1e-3, "G images of the group vs oracle (north-star bound)"
    assert_close(fake[sel.cuda()], fake_o, 1e-4, "G images of the group")
    assert_close(fs[sel.cuda()], fs_o, 3e-4, "D(fake) scores of the group")
    assert_close(rs[sel.cuda()], rs_o.detach(), 3e-4, "D(real) scores of the group")
    # R1 = 5 * ||d sum(scores) / d image||^2 passes through every leaky-ReLU mask of D.  Round 3 held it to 2e-2 here on the
    # strength of a sign-flip argument; measured in round 4 on this batch: 7.5e-6 on the Winograd path, 2.0e-5 on the direct
    # kernels, 137 of 1.2e9 leaky-ReLU outputs on different sides between the two HIP forwards — so it is held to the batch-4
    # test's 3e-4 on both paths.
    assert_close(r1[sel.cuda()], r1_o, 3e-4, "R1 penalties of the group")
    # The sign-flip explanation, demonstrated at THIS size (review item, round 3): the same batch with the Winograd kernels off
    # (direct kernels only, the path of the batch-4 test) holds the penalties to the batch-4 bound, and the leaky-ReLU outputs of
    # the two HIP forwards differ in sign at a counted handful of positions — the masks the penalty's gradient passes through.
    from gif_amd import ops

    def d_forward_recording_signs():
        signs, hooks = [], []
        for m in D.modules():
            if type(m).__name__ == "ConvLayer":
                hooks.append(m.register_forward_hook(
                    lambda _m, _i, out: signs.append(torch.signbit((out[0] if isinstance(out, (tuple, list)) else out).detach()))))
        x = real.clone().requires_grad_(True)
        sc = D([x], condition=cond)[0]
        pen = losses.grad_penalty_loss([x], sc, step=None).detach()
        for h in hooks:
            h.remove()
        return pen, signs

    r1_w, signs_w = d_forward_recording_signs()
    assert torch.equal(r1_w, r1), "the recorded pass repeats the one above (deterministic kernels)"
    wino_was = ops.WINOGRAD
    ops.WINOGRAD = False
    try:
        r1_d, signs_d = d_forward_recording_signs()
    finally:
        ops.WINOGRAD = wino_was
    flips = sum(int((a != b).sum().item()) for a, b in zip(signs_w, signs_d))
    total = sum(a.numel() for a in signs_w)
    err_w, err_d = rel_err(r1[sel.cuda()], r1_o), rel_err(r1_d[sel.cuda()], r1_o)
    print(f"R1 at batch 32 vs oracle: Winograd path {err_w:.2e}, direct kernels {err_d:.2e}; "
          f"{flips} of {total} leaky-ReLU outputs of D differ in sign between the two HIP forwards")
    assert_close(r1_d[sel.cuda()], r1_o, 3e-4, "R1 penalties of the group, Winograd kernels off")
    assert len(signs_w) == len(signs_d) and total > 0
    assert flips <= 1e-5 * total, f"{flips} sign flips of {total}: more than rounding of near-zero pre-activations explains"
    assert err_w <= 3e-4 or flips > 0, "a Winograd-path deviation above the direct bound must come with flipped activations"
    d_expected = (F.softplus(-rs).mean() + r1.mean() + F.softplus(fs).mean()).item()
    # the training iteration itself (R1 iteration: i + 1 divisible by 16)
    tr = GifTrainer(G, D, G_ema, step=6, r1_every=16)
    w0 = D.convs[1].conv1[0].weight.detach().clone()
    d_loss, g_loss = tr.step(15, real, cond, idx)
    torch.cuda.synchronize()
    assert abs(d_loss.item() - d_expected) < 1e-4 * max(1.0, abs(d_expected)), (d_loss.item(), d_expected)
    assert torch.isfinite(g_loss).item()
    assert (D.convs[1].conv1[0].weight - w0).abs().max().item() > 0, "discriminator parameters must move"
